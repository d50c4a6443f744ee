"""oracle/gym_seeding.py: the reference's per-worker reset stream (gym 0.17.2 seeding + reset_model's draw order).
The draw order / distributions are pinned to the draws recorded while the reference's own reset_model ran on a seeded
RandomState (tests/golden/env_arith.npz, tools/capture_golden.py:360-382); gym's seed hashing itself is restated from
the published algorithm and is unpinned (gym is absent)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import gym_seeding as gs
from tests.helpers import oracle_model


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "env_arith_meta.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(golden_dir, "env_arith.npz")), meta


def test_reset_draw_order_matches_the_executed_reference(gold):
    z, meta = gold
    assert len(meta) == 19
    for envname in meta:
        m, om = oracle_model(envname)
        draws = gs.reset_draws(np.random.RandomState(4242), envname, m.nq, m.nv)      # the seed the capture used
        np.testing.assert_array_equal(draws, z[envname + "/reset/draws"])
        init_qpos = np.zeros(m.nq)
        init_qpos[2], init_qpos[3] = 1.3, 1.0
        q, v, tgt = gs.state_from_draws(draws, envname, init_qpos, m.nq, m.nv)
        np.testing.assert_allclose(q, z[envname + "/reset/qpos"], atol=1e-15)
        np.testing.assert_allclose(v, z[envname + "/reset/qvel"], atol=1e-15)
        np.testing.assert_allclose(tgt, z[envname + "/reset/target"], atol=1e-9)


def test_seed_hashing_structure():
    # create_seed: mod 2^64, rejects negatives
    assert gs.create_seed(5) == 5 and gs.create_seed(2 ** 64 + 7) == 7
    with pytest.raises(ValueError):
        gs.create_seed(-1)
    # hash_seed: little-endian value of the first eight digest bytes (the zero pad word adds nothing)
    for seed in (0, 1, 12345, 2 ** 40 + 3):
        d = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
        assert gs.hash_seed(seed) == int.from_bytes(d, "little")
        limbs = gs._int_list_from_bigint(gs.hash_seed(seed))
        assert all(0 <= x < 2 ** 32 for x in limbs) and sum(x << (32 * i) for i, x in enumerate(limbs)) == gs.hash_seed(seed)
    assert gs._int_list_from_bigint(0) == [0]
    # streams: reproducible, distinct across seeds, and NOT RandomState(seed) itself
    a, s = gs.np_random(3)
    b, _ = gs.np_random(3)
    c, _ = gs.np_random(4)
    xa, xb, xc = a.uniform(size=5), b.uniform(size=5), c.uniform(size=5)
    assert s == 3 and np.array_equal(xa, xb) and not np.array_equal(xa, xc)
    assert not np.array_equal(xa, np.random.RandomState(3).uniform(size=5))


def test_all_reference_workers_start_identically():
    """reference utils.py:19 seeds EVERY worker with the same seed: replicas of one morphology would be identical -- the reason
    the engine keys its counter RNG by (seed, env id, episode) instead."""
    m, om = oracle_model("3d_walker_7_full")
    q1, v1, t1 = gs.first_reset_state(0, "3d_walker_7_full", m.qpos0, m.nq, m.nv)
    q2, v2, t2 = gs.first_reset_state(0, "3d_walker_7_full", m.qpos0, m.nq, m.nv)
    assert np.array_equal(q1, q2) and np.array_equal(v1, v2) and np.array_equal(t1, t2)
    assert abs(np.linalg.norm(q1[3:7] - np.r_[0, q1[4:6], 0]) - 1) < 0.02 and np.abs(v1).max() <= 0.005
    assert abs(np.hypot(*t1) - 10000.0) < 1e-6
