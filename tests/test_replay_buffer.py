"""DeviceReplayBuffer against tests/golden/replay_buffer.npz -- the arrays and pointers of the reference's
own ReplayBuffer (src/common/buffer.py:35-84) after a scripted add_transition sequence that wraps the ring
(tools/capture_golden_buffer.py)."""
import os

import numpy as np
import pytest
import torch

from sgrl_amd.replay import DeviceReplayBuffer


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "replay_buffer.npz"))


def _check(rb, g, i):
    assert rb.curr == int(g["curr_%d" % i]) and rb.max_sample_size == int(g["mss_%d" % i])
    st = rb.state_arrays()
    for k, gk in (("obs_buffer", "obs"), ("action_buffer", "act"), ("next_obs_buffer", "nxt"), ("reward_buffer", "rew"),
                  ("done_buffer", "done")):
        assert np.array_equal(st[k], g["%s_%d" % (gk, i)]), (k, i)


def test_row_by_row_matches_reference(gold):
    g = gold
    rb = DeviceReplayBuffer(g["obs"].shape[1], g["act"].shape[1], int(g["cap"]))
    for i in range(g["obs"].shape[0]):
        rb.add_transition(g["obs"][i], g["act"][i], g["nxt"][i], g["rew"][i], g["done"][i])
        if i in (3, 9, 10, 26):
            _check(rb, g, i)


def test_batched_masked_append_is_equivalent(gold):
    g = gold
    t = lambda k: torch.from_numpy(g[k])
    rb = DeviceReplayBuffer(g["obs"].shape[1], g["act"].shape[1], int(g["cap"]))
    # feed rows 0..26 in three batches, each interleaved with rows that the mask drops
    for lo, hi in ((0, 4), (4, 11), (11, 27)):
        n = hi - lo
        obs = torch.zeros(2 * n, g["obs"].shape[1]); act = torch.zeros(2 * n, g["act"].shape[1]); nxt = torch.zeros_like(obs)
        rew = torch.zeros(2 * n); done = torch.zeros(2 * n); mask = torch.zeros(2 * n, dtype=torch.bool)
        obs[0::2], act[0::2], nxt[0::2], rew[0::2], done[0::2] = t("obs")[lo:hi], t("act")[lo:hi], t("nxt")[lo:hi], t("rew")[lo:hi], t("done")[lo:hi]
        mask[0::2] = True
        rb.add_transitions(obs, act, nxt, rew, done, mask)
        _check(rb, g, hi - 1)


def test_sample_and_snapshot_round_trip(gold):
    g = gold
    rb = DeviceReplayBuffer(g["obs"].shape[1], g["act"].shape[1], int(g["cap"]))
    rb.add_transitions(*(torch.from_numpy(g[k][:7]) for k in ("obs", "act", "nxt", "rew", "done")))
    b = rb.sample(256)
    assert b["obs"].shape == (7, g["obs"].shape[1]) and b["reward"].shape == (7, 1) and b["done"].shape == (7, 1)
    assert sorted(b["reward"].flatten().tolist()) == sorted(g["rew"][:7].tolist())    # without replacement
    rb2 = DeviceReplayBuffer(g["obs"].shape[1], g["act"].shape[1], int(g["cap"]))
    rb2.load_state_arrays(rb.state_arrays())
    assert rb2.curr == 7 and torch.equal(rb2.obs_buffer, rb.obs_buffer)
