"""ModularEnvWrapper semantics (reference src/wrappers.py:39-65) against tests/golden/wrapper_pad.npz, which holds the
reference wrapper's own outputs: the un-padded / re-ordered action handed to the env and the zero-padded observation."""
import os

import numpy as np
import pytest

from sgrl_amd import mjcf

NAMES = ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full", "3d_cheetah_14_full",
         "3d_walker_2_right_leg_left_knee"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "wrapper_pad.npz"))


def test_action_scatter_matches_reference_wrapper(gold):
    for name in NAMES:
        m = mjcf.load_asset(name)
        a_in = gold[name + "/action_in"]            # padded policy-order action (length 3 * 14)
        env_a = gold[name + "/env_action"]          # what the reference wrapper passed to env.step
        assert env_a.shape == (m.nu,)
        # the engine's table: actuator u is fed by policy slot act_slot[u] (csrc/step_body.h env_step)
        mine = np.array([a_in[s] for s in m.act_slot])
        assert np.array_equal(mine, env_a), name
        assert (m.act_slot >= 3).all() and m.act_slot.max() < 3 * m.num_limbs


def test_observation_zero_padding_convention(gold):
    for name in NAMES:
        m = mjcf.load_asset(name)
        L = m.num_limbs
        for key in ("obs_step", "obs_reset"):
            ob = gold["%s/%s" % (name, key)]
            assert ob.shape == (41 * 14,)
            assert (ob[41 * L:] == 0).all() and (ob[:41 * L] != 0).all()
