#!/usr/bin/env python3
"""Diagnostic: free-running engine vs oracle, report where the deviation first jumps and the solver diagnostics there."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from tests.test_engine_gpu import _make, _oracle_envs
names = sys.argv[1:] or ["3d_cheetah_14_full", "3d_cheetah_11_leftfleg"]
env = _make(names, 2); env.reset_device()
oes = _oracle_envs(env, names, 5)
for oe in oes: oe.reset()
rng = np.random.RandomState(1)
prev = np.zeros(env.num_envs)
for t in range(1000):
    a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
    env.step_device(torch.from_numpy(a).cuda()); torch.cuda.synchronize()
    rec, cnt = env.get_records()
    for i, oe in enumerate(oes):
        o, r, d, info = oe.step(a[i].astype(np.float64))
        q, v, xy, tg = env.state_of(rec, i)
        e = max(np.abs(q - oe.qpos).max() / (1 + np.abs(oe.qpos).max()), np.abs(v - oe.qvel).max() / (1 + np.abs(oe.qvel).max()))
        if e > 1e-9 and prev[i] <= 1e-9:
            print("step %d env %d: deviation jumps %.2e -> %.2e  cnt %s oracle counters %s done %s" % (t, i, prev[i], e, cnt[i].tolist(), list(oe.counters), d))
        prev[i] = e
print("final deviations", prev)
