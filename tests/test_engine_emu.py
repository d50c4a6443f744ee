"""CPU-side checks of the ENGINE SOURCE (sgrl_amd/csrc/step_body.h) compiled with a serial lane emulator
(tests/emu/emu_step.cpp) against the oracle.  This is a unit test of the kernel logic in the GPU-less build
container; the GPU parity tests proper are in test_engine_gpu.py and go through the C ABI."""
import numpy as np
import pytest

import emu_ref
from helpers import packed, oracle_model
from oracle import physics_ref

NAMES = ["3d_hopper_3_shin", "3d_hopper_5_full", "3d_walker_2_right_leg_left_knee", "3d_walker_7_full",
         "3d_walker_v2_5_foot", "3d_humanoid_9_full", "3d_cheetah_14_full"]


@pytest.mark.parametrize("name", NAMES)
def test_forward_dynamics_match_oracle(name):
    m, ib, fb = packed(name)
    _, om = oracle_model(name)
    env = physics_ref.OracleEnv(om, seed=1)
    env.reset()
    rng = np.random.RandomState(0)
    for t in range(60):
        a = rng.uniform(-1, 1, size=3 * om.L)
        if t % 5 == 0:
            q1, _, d1 = om.forward(env.qpos, env.qvel, a[3:])
            q2, d2 = emu_ref.forward(ib, fb, env.qpos, env.qvel, a[3:])
            assert d1["nrow"] == d2["nrow"] and d1["ncon"] == d2["ncon"]
            assert np.abs(q1 - q2).max() <= 1e-9 * (1 + np.abs(q1).max())
        env.step(a)


@pytest.mark.parametrize("name", NAMES)
def test_free_running_episodes_match_oracle_and_lane_order_is_irrelevant(name):
    m, ib, fb = packed(name)
    _, om = oracle_model(name)
    trajs = []
    for rev in (False, True):
        emu_ref.set_reverse(rev)
        try:
            e1 = physics_ref.OracleEnv(om, seed=7, env_id=3)
            e2 = emu_ref.EmuEnv(ib, fb, seed=7, env_id=3)
            o1, o2 = e1.reset(), e2.reset()
            assert np.abs(o1 - o2).max() < 1e-13
            rng = np.random.RandomState(1)
            traj = []
            ndone = 0
            for t in range(80):
                a = rng.uniform(-1, 1, size=3 * om.L).astype(np.float32)
                o1, r1, d1, i1 = e1.step(a.astype(np.float64))
                o2, r2, d2, i2 = e2.step(a)
                assert d1 == d2
                ndone += d1
                tol = 1e-7 if "cheetah" in name else 1e-9
                assert np.abs(o1 - o2).max() < tol and abs(r1 - r2) < tol * 10
                assert abs(i1["dist"] - i2["dist"]) < 1e-2  # float32 output
                assert np.array_equal(i2["obs32"], o2.astype(np.float32))
                traj.append(o2.copy())
            trajs.append(np.array(traj))
        finally:
            emu_ref.set_reverse(False)
    # ascending vs descending lane execution must agree bit for bit (no intra-phase cross-lane dependency)
    assert np.array_equal(trajs[0], trajs[1])


def test_observation_padding_and_time_limit():
    m, ib, fb = packed("3d_walker_2_right_leg_left_knee")
    e = emu_ref.EmuEnv(ib, fb, seed=1, max_episode_steps=2, obs_max_len=287)
    o = e.reset()
    assert o.shape == (287,) and (o[82:] == 0).all() and np.abs(o[:82]).max() > 0
    z = np.zeros(21, dtype=np.float32)
    _, _, d1, i1 = e.step(z)
    _, _, d2, i2 = e.step(z)
    assert (d1, d2) == (False, True) and i2["TimeLimit.truncated"]
    assert e.cnt[0] == 0 and e.cnt[1] == 1   # auto-reset started episode 1


def test_rng_matches_oracle_bit_for_bit():
    m, ib, fb = packed("3d_cheetah_14_full")
    _, om = oracle_model("3d_cheetah_14_full")
    for env_id in (0, 5, 8191):
        e1 = physics_ref.OracleEnv(om, seed=123456789012345, env_id=env_id)
        e2 = emu_ref.EmuEnv(ib, fb, seed=123456789012345, env_id=env_id)
        e1.reset()
        e2.reset()
        assert np.array_equal(e1.target, e2.target)
        # normal draws go through libm log/cos in both builds here
        np.testing.assert_allclose(e1.qvel, e2.qvel, rtol=0, atol=1e-15)
        np.testing.assert_allclose(e1.qpos, e2.qpos, rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", ["3d_walker_7_full", "3d_hopper_4_lower_shin"])
def test_explicit_inverse_path_and_factor_path_agree(name):
    """nv <= 24: the engine source has two formulations of the mass-matrix solves (explicit L^-1 products, which the
    HIP wave policy selects, and in-place L substitutions).  Both must follow the oracle, and each other to rounding."""
    m, ib, fb = packed(name)
    _, om = oracle_model(name)
    outs = []
    for linv in (True, False):
        emu_ref.set_linv(linv)
        try:
            e1 = physics_ref.OracleEnv(om, seed=11, env_id=2)
            e2 = emu_ref.EmuEnv(ib, fb, seed=11, env_id=2)
            e1.reset(); e2.reset()
            rng = np.random.RandomState(4)
            tr = []
            for t in range(60):
                a = rng.uniform(-1, 1, size=3 * om.L).astype(np.float32)
                o1, r1, d1, _ = e1.step(a.astype(np.float64))
                o2, r2, d2, _ = e2.step(a)
                assert d1 == d2 and np.abs(o1 - o2).max() < 1e-9
                tr.append(o2.copy())
            outs.append(np.array(tr))
        finally:
            emu_ref.set_linv(True)
    assert np.abs(outs[0] - outs[1]).max() < 1e-9
