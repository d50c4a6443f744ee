#!/usr/bin/env python3
"""Diagnostic: per-step teacher-forced deviation engine vs oracle for the cheetah family (ill-conditioned contact sets)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from tests.test_engine_gpu import _make, _oracle_envs
names = sys.argv[1:] or ["3d_cheetah_14_full"]
env = _make(names, 2); env.reset_device()
oes = _oracle_envs(env, names, 5)
for oe in oes: oe.reset()
rng = np.random.RandomState(0)
errs = []
for t in range(120):
    rec, cnt = env.get_records()
    for i, oe in enumerate(oes):
        m = env.models[env.env_morph[i]]
        rec[i, :m.nq] = oe.qpos; rec[i, m.nq:m.nq + m.nv] = oe.qvel
        rec[i, m.nq + m.nv:m.nq + m.nv + 2] = oe.torso_xy_stale; rec[i, m.nq + m.nv + 2:m.nq + m.nv + 4] = oe.target
        cnt[i, 0], cnt[i, 1] = oe.counters[0], oe.counters[1]
    cnt[:, 3] = 0
    env.set_records(rec, cnt)
    a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
    env.step_device(torch.from_numpy(a).cuda(), auto_reset=False); torch.cuda.synchronize()
    rec2, cnt2 = env.get_records()
    for i, oe in enumerate(oes):
        o, r, d, info = oe.step(a[i].astype(np.float64), auto_reset=False)
        q, v, xy, tg = env.state_of(rec2, i)
        e = max(np.abs(q - oe.qpos).max() / (1 + np.abs(oe.qpos).max()), np.abs(v - oe.qvel).max() / (1 + np.abs(oe.qvel).max()))
        errs.append((e, t, i, int(cnt2[i, 3])))
        if d: oe.counters[1] += 1; oe.reset()
errs.sort(reverse=True)
print("worst 8 (err, step, env, diag):", [(float("%.2e" % e), t, i, hex(dg)) for e, t, i, dg in errs[:8]])
print("median %.2e  p90 %.2e" % (np.median([e for e, *_ in errs]), np.percentile([e for e, *_ in errs], 90)))
