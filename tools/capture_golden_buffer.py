#!/usr/bin/env python3
"""Golden vectors for the replay ring buffer: executes the reference's ReplayBuffer (reference
src/common/buffer.py:35-126, constructed as in src/main.py:141-155 with modular=True) on a scripted sequence of
add_transition calls that wraps around, and stores the resulting arrays + pointers.  Build container only."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import numpy as np
import refstub
refstub.install()
from common.buffer import ReplayBuffer
from gym.spaces import Box
L = 3
obs_space = Box(-np.ones(41 * L), np.ones(41 * L))
act_space = Box(-np.ones(3 * (L - 1)), np.ones(3 * (L - 1)))
cap = 10
rb = ReplayBuffer(obs_space, act_space, max_buffer_size=cap, modular=True)
rng = np.random.RandomState(3)
n = 27
obs = rng.rand(n, 41 * L).astype(np.float32); act = rng.rand(n, 3 * L).astype(np.float32)
nxt = rng.rand(n, 41 * L).astype(np.float32); rew = rng.rand(n).astype(np.float32); done = (rng.rand(n) > 0.7).astype(np.float32)
snaps = {}
for i in range(n):
    rb.add_transition(obs[i], act[i], nxt[i], rew[i], done[i])
    if i in (3, 9, 10, 26):
        snaps["curr_%d" % i] = np.int64(rb.curr); snaps["mss_%d" % i] = np.int64(rb.max_sample_size)
        snaps["obs_%d" % i] = rb.obs_buffer.copy(); snaps["act_%d" % i] = rb.action_buffer.copy()
        snaps["nxt_%d" % i] = rb.next_obs_buffer.copy(); snaps["rew_%d" % i] = rb.reward_buffer.copy(); snaps["done_%d" % i] = rb.done_buffer.copy()
assert rb.action_dim == 3 * L
np.savez_compressed(os.path.join(REPO, "tests", "golden", "replay_buffer.npz"), obs=obs, act=act, nxt=nxt, rew=rew, done=done,
                    cap=np.int64(cap), **snaps)
print("replay_buffer.npz written; action_dim", rb.action_dim, "curr", rb.curr, "max_sample_size", rb.max_sample_size)
