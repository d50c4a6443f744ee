"""Two environments per wavefront (sgrl_amd/csrc/wave_half.h: the half-wave instance of the step kernel for the light morphologies)
on the CPU: the engine source + the half-wave algorithms run on a 64-fiber SIMT emulator (tests/emu/emu_pair.cpp) whose cross-lane
primitives move data exactly as the gfx950 encodings do, on ONE shared buffer with the pair layout.  Checked against the oracle and
against the one-environment lane emulator; the halves are put into DIFFERENT situations (one lying on the floor with many contact
rows, the other in the air) so that their data-dependent branches diverge."""
import numpy as np
import pytest

import emu_ref
from helpers import packed, oracle_model
from oracle import physics_ref

LIGHT = ["3d_walker_2_right_leg_left_knee", "3d_walker_3_left_knee_right_knee", "3d_hopper_3_shin", "3d_walker_4_right_knee_left_foot",
         "3d_hopper_4_lower_shin"]


@pytest.mark.parametrize("name", LIGHT)
def test_pair_layout_fits_eight_workgroups_per_cu(name):
    m, ib, fb = packed(name)
    pe = emu_ref.PairEmu(ib, fb)
    b = pe.layout_bytes()
    assert 0 < b <= 20480 and (160 * 1024) // (((b + 1279) // 1280) * 1280) >= 8


@pytest.mark.parametrize("name", LIGHT)
def test_paired_episodes_match_oracle_and_the_single_lane_emulator(name):
    m, ib, fb = packed(name)
    _, om = oracle_model(name)
    ids = (3, 4)
    pe = emu_ref.PairEmu(ib, fb, seed=7, env_ids=ids, max_episode_steps=45)
    singles = [emu_ref.EmuEnv(ib, fb, seed=7, env_id=e, max_episode_steps=45) for e in ids]
    oracles = [physics_ref.OracleEnv(om, seed=7, env_id=e, max_episode_steps=45) for e in ids]
    o_pair = pe.reset()
    for h in range(2):
        assert np.abs(o_pair[h] - oracles[h].reset()).max() < 1e-13
        assert np.abs(o_pair[h] - singles[h].reset()).max() < 1e-13
    rng = np.random.RandomState(1)
    ndone = 0
    for t in range(100):
        acts = [rng.uniform(-1, 1, size=3 * om.L).astype(np.float32) for _ in range(2)]
        res = pe.step(acts)
        for h in range(2):
            o1, r1, d1, i1 = oracles[h].step(acts[h].astype(np.float64))
            o2, r2, d2, i2 = singles[h].step(acts[h])
            o3, r3, d3, i3 = res[h]
            assert d1 == d3 == d2
            ndone += d1
            assert np.abs(o1 - o3).max() < 1e-9 and abs(r1 - r3) < 1e-8
            assert np.abs(o2 - o3).max() < 1e-9
            assert np.array_equal(i3["obs32"], o3.astype(np.float32))
            assert np.array_equal(pe.envs[h].cnt[:2], singles[h].cnt[:2])
    assert ndone > 0      # auto-reset (falls: ONE half resets while the other keeps stepping; the time limit: both) was exercised


@pytest.mark.parametrize("name", LIGHT)
def test_halves_in_different_contact_situations(name):
    """A on the floor on its side (many contact rows: the evaluations that leave the LDS row arrays for the HBM slab), B dropped from
    the air (no rows at first): row counts, free sets and pivoting rounds differ between the halves of the wavefront."""
    m, ib, fb = packed(name)
    _, om = oracle_model(name)
    pe = emu_ref.PairEmu(ib, fb, seed=2, env_ids=(0, 1))
    oracles = [physics_ref.OracleEnv(om, seed=2, env_id=e) for e in (0, 1)]
    pe.reset()
    for oe in oracles:
        oe.reset()
    nq = om.nq
    lying = np.array(pe.envs[0].qpos)             # pressed flat into the floor with every hinge beyond its limit: 18 .. 30 rows, more than
    lying[2] = 0.03                               # the LDS row arrays of walker_3 / hopper_3 hold (22 / 23) -> the HBM-slab row path
    lying[3:7] = [0.70710678, 0, 0.70710678, 0] if "hopper" in name else [1, 0, 0, 0]
    lying[7:] = 0.8
    air = np.array(pe.envs[1].qpos)
    air[2] += 1.0
    for h, q in enumerate((lying, air)):
        pe.envs[h].rec[:nq] = q
        pe.envs[h].rec[nq:nq + om.nv] *= 0.0
        oracles[h].qpos[:] = q
        oracles[h].qvel[:] = 0.0
    o = pe.refresh()
    for h in range(2):
        assert np.abs(oracles[h].refresh() - o[h]).max() < 1e-12
    rng = np.random.RandomState(5)
    rows_seen = [0, 0]
    for t in range(25):
        for h in range(2):
            rows_seen[h] = max(rows_seen[h], om.forward(oracles[h].qpos, oracles[h].qvel, np.zeros(om.nu))[2]["nrow"])
        acts = [rng.uniform(-1, 1, size=3 * om.L).astype(np.float32) for _ in range(2)]
        res = pe.step(acts, auto_reset=False)
        for h in range(2):
            o1, r1, d1, _ = oracles[h].step(acts[h].astype(np.float64), auto_reset=False)
            o3, r3, d3, _ = res[h]
            assert d1 == d3
            assert np.abs(o1 - o3).max() < 1e-7, (t, h)
    assert rows_seen[0] > rows_seen[1] and rows_seen[0] >= 18      # the halves did see different constraint problems
    if "walker_2" not in name:
        assert rows_seen[0] > 23                  # ... and half A left the LDS row arrays (20 .. 23 rows) for the slab


def test_gauss_seidel_fallback_of_the_half_wave():
    """The half wave's projected Gauss-Seidel (wave_half.h pgs: the fallback when block pivoting gives up, and the whole solver of a
    model packed with solver = 0) never runs in the product's pair instances on sane states -- here it is the only solver."""
    name = "3d_walker_3_left_knee_right_knee"
    m, ib, fb = packed(name, solver=0)
    _, om = oracle_model(name, solver=0)
    pe = emu_ref.PairEmu(ib, fb, seed=5, env_ids=(0, 1))
    oracles = [physics_ref.OracleEnv(om, seed=5, env_id=e) for e in (0, 1)]
    o = pe.reset()
    for h in range(2):
        assert np.abs(o[h] - oracles[h].reset()).max() < 1e-13
    rng = np.random.RandomState(2)
    for t in range(30):
        acts = [rng.uniform(-1, 1, size=3 * om.L).astype(np.float32) for _ in range(2)]
        res = pe.step(acts)
        for h in range(2):
            o1, r1, d1, _ = oracles[h].step(acts[h].astype(np.float64))
            assert d1 == res[h][2] and np.abs(o1 - res[h][0]).max() < 1e-6      # an iteration stopped by tolerance: 1e-6, not rounding
