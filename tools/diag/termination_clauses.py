#!/usr/bin/env python3
"""Which clause of the done rule ends episodes, per morphology family (CPU only, on the oracle).

For one morphology per family and three action sources -- zero, U(-1, 1) random, a proportional joint-space controller holding the reset pose ("pd") -- runs
episodes from the reference's reset distribution (reference src/environments/<name>.py:150-164: qpos0 + U(+-noise) on EVERY
coordinate, the root quaternion included) and records the episode length and which clause of `done` (reference <name>.py:29-37)
was violated at the terminal step; plus the stand test: from qpos0 exactly (no reset noise) with zero action, does the model
settle upright?  Usage: termination_clauses.py [episodes] [out.json]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import physics_ref                                # noqa: E402
from sgrl_amd import mjcf, model_pack                         # noqa: E402
from sgrl_amd.env_spec import env_spec_for                    # noqa: E402

FAMILIES = {"hopper": "3d_hopper_5_full", "walker": "3d_walker_7_full", "humanoid": "3d_humanoid_9_full", "cheetah": "3d_cheetah_14_full"}
H_DONE_RULE, H_NHB, H_HB0 = 10, 13, 14
F_LO, F_HI, F_AL = 4, 5, 6


def angles(q):
    w, x, y, z = q
    r00, r10 = 1 - 2 * y * y - 2 * z * z, 2 * x * y + 2 * z * w
    r20, r21, r22 = 2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y
    return np.arctan2(-r20, np.hypot(r21, r22)), np.arctan2(r21, r22)      # pitch, roll


def clauses(ib, fb, quat_before, qpos, qvel, obs):
    """names of the violated clauses of the family's done rule (engine: csrc/step_body.h reward_done; oracle/physics.c)."""
    rule, lo, hi, al = int(ib[H_DONE_RULE]), fb[F_LO], fb[F_HI], fb[F_AL]
    pitch, roll = angles(quat_before)
    height = qpos[2]
    out = []
    if rule == 0:
        if not height > lo: out.append("height<=lo")
        if not height < hi: out.append("height>=hi")
        if not abs(pitch) < al: out.append("pitch")
        if not abs(roll) < al: out.append("roll")
    elif rule == 1:
        qq = qpos[3:7]
        ang = 2 * np.arctan2(np.hypot(qq[1], qq[2]), np.hypot(qq[0], qq[3]))
        if not (np.isfinite(qpos).all() and np.isfinite(qvel).all()): out.append("non-finite")
        if not ((np.abs(qpos[3:]) < 100).all() and (np.abs(qvel) < 100).all()): out.append("|state|>=100")
        if not height > lo: out.append("height<=lo")
        if not abs(ang) < al: out.append("tilt")
    else:
        for i in range(int(ib[H_NHB])):
            height = min(height, obs[41 * (int(ib[H_HB0 + i]) - 1) + 40])      # z of the front thighs (obs element 40 of a limb = its z)
        if not height > lo: out.append("height<=lo")
        if not abs(pitch) < al: out.append("pitch")
        if not abs(roll) < al: out.append("roll")
        if not float(np.square(qvel).sum()) > 1: out.append("|qvel|^2<=1 (standing still)")
    return out


def run(name, episodes, max_steps=300):
    m = mjcf.load_asset(name)
    ib, fb = model_pack.pack_model(m, spec=env_spec_for(name), max_rows=256)
    om = physics_ref.OracleModel(ib, fb)
    L = om.L
    res = {"morphology": name, "done_rule": int(ib[H_DONE_RULE]), "height_lo": fb[F_LO], "height_hi": fb[F_HI], "angle_limit": fb[F_AL],
           "reset_pos_noise": fb[10], "reset_vel_noise": fb[11]}
    # stand test: qpos0 exactly, zero action, 300 steps; done clauses ignored (the cheetah rule ends a STILL robot)
    env = physics_ref.OracleEnv(om, seed=0, env_id=0)
    env.reset()
    env.qpos[:] = fb[16:16 + om.nq]
    env.qvel[:] = 0
    env.refresh()
    z0 = float(env.qpos[2])
    stand = {"z_start": z0}
    for t in range(300):
        env.step(np.zeros(3 * L), auto_reset=False)
        if t + 1 in (150, 300):
            pitch, roll = angles(env.qpos[3:7])
            stand["after_%d_steps" % (t + 1)] = {"z": float(env.qpos[2]), "pitch": float(pitch), "roll": float(roll),
                                                 "qvel_sq": float(np.square(env.qvel).sum())}
    res["stand_from_qpos0_zero_action"] = stand
    for policy in ("zero", "random", "pd"):
        lengths, ended_by = [], {}
        rng = np.random.RandomState(1)
        for ep in range(episodes):
            env = physics_ref.OracleEnv(om, seed=11, env_id=ep)
            obs = env.reset()
            q_ref = obs.reshape(L, 41)[:, 24:27].copy()
            for t in range(max_steps):
                qb = env.qpos[3:7].copy()
                if policy == "zero":
                    a = np.zeros(3 * L)
                elif policy == "random":
                    a = rng.uniform(-1, 1, size=3 * L)
                else:
                    o = obs.reshape(L, 41)
                    a = np.clip(4.0 * (q_ref - o[:, 24:27]), -1, 1).ravel()      # proportional hold of the reset pose: joint k of limb l is driven by action 3 l + k
                obs, r, d, info = env.step(a, auto_reset=False)
                if d:
                    if info["TimeLimit.truncated"]:
                        cl = ["time limit"]
                    else:
                        cl = clauses(ib, fb, qb, env.qpos, env.qvel, obs) or ["(none reproduced)"]
                    for c in cl:
                        ended_by[c] = ended_by.get(c, 0) + 1
                    break
            else:
                ended_by["survived %d steps" % max_steps] = ended_by.get("survived %d steps" % max_steps, 0) + 1
            lengths.append(t + 1)
        res[policy] = {"episodes": episodes, "length_min_median_max": [int(np.min(lengths)), float(np.median(lengths)), int(np.max(lengths))],
                       "length_histogram_10_20_50_100_300": [int(np.sum(np.array(lengths) <= b)) for b in (10, 20, 50, 100, 300)],
                       "ended_by": ended_by}
    return res


if __name__ == "__main__":
    episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    out = {fam: run(name, episodes) for fam, name in FAMILIES.items()}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
