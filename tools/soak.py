#!/usr/bin/env python3
"""Soak: thousands of random-action steps of a big mixed batch; every output must stay finite, no constraint row may be
dropped, episodes must keep turning over."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgrl_amd import mjcf
from sgrl_amd.rollout import Rollout
which = sys.argv[1] if len(sys.argv) > 1 else "walker"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
names = sorted(n for n in mjcf.list_assets() if which == "all" or which in n)
per = max(1, 8192 // len(names))
ro = Rollout(names, per, seed=11, device="cuda:0")
ro.reset()
t0 = time.time()
bad = 0
for t in range(steps):
    obs, rew, done, dist = ro.step(ro.random_actions())
    if t % 250 == 249:
        bad += int((~torch.isfinite(obs)).sum()) + int((~torch.isfinite(rew)).sum()) + int((~torch.isfinite(dist)).sum())
rec, cnt = ro.env.get_records()
import numpy as np
print("%s: %d morphologies x %d envs, %d steps in %.1f s | non-finite outputs %d | non-finite state %d | envs with dropped rows %d | "
      "episodes per env min %d mean %.1f" % (which, len(names), per, steps, time.time() - t0, bad, int((~np.isfinite(rec)).sum()),
                                           int((cnt[:, 2] > 0).sum()), int(cnt[:, 1].min()), float(cnt[:, 1].mean())))
