#!/usr/bin/env python3
"""Compile the reference's MJCF morphologies into sgrl_amd/assets/models/*.json (build container only).

The XML files themselves stay in /root/reference; what is committed is the numeric output of
sgrl_amd.mjcf.compile_mjcf (flat arrays as hex floats), i.e. derived input data for the engine.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from sgrl_amd import mjcf  # noqa: E402


def main():
    base = "/root/reference/src/environments"
    os.makedirs(mjcf.ASSET_DIR, exist_ok=True)
    n = 0
    for sub in ["3d_hoppers", "3d_walkers", "3d_humanoids", "3d_cheetahs", "zero_shot"]:
        d = os.path.join(base, sub)
        for f in sorted(os.listdir(d)):
            if not f.endswith(".xml"):
                continue
            m = mjcf.compile_mjcf(os.path.join(d, f))
            mjcf.save_model(m, os.path.join(mjcf.ASSET_DIR, m.name + ".json"))
            print("%-45s L=%2d nq=%2d nv=%2d nu=%2d ngeom=%2d npair=%2d mass=%.3f int=%d dt=%g" % (
                m.name, m.num_limbs, m.nq, m.nv, m.nu, m.ngeom, m.npair, m.body_mass.sum(), m.integrator, m.timestep))
            n += 1
    print(n, "models")


if __name__ == "__main__":
    main()
