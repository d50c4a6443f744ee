#!/usr/bin/env python3
"""Config 5's collection step on one box, A/B: the round bookkeeping as one launch (sgrl_round_record) or as tensor operations, and
the round flag read one step late or immediately -- alternating, same process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd import mjcf, rollout
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD)
per = 8192 // len(names)
for rep in range(2):
    for fused, lag in ((True, True), (False, True), (False, False), (True, False)):
        rollout.FUSED_RECORD = fused
        tr = DeviceTrainer(names, per, args=default_train_args(), seed=1, device="cuda:0", max_buffer_size=20000, graph_updates=False, lag_flag=lag)
        tr.warmup(60)
        for _ in range(8):          # untimed policy steps: the first one's one-time costs (code objects, weight pack, stream probe) stay out of the window
            if tr.collect_step():
                tr.begin_round()
        torch.cuda.synchronize()
        t0 = time.time(); K = 60
        for _ in range(K):
            if tr.collect_step():
                tr.begin_round()
        torch.cuda.synchronize()
        print("rep %d fused_record %-5s lag_flag %-5s: %.3f ms per collection step" % (rep, fused, lag, (time.time() - t0) / K * 1e3), flush=True)
        tr.ro.env.close()
        del tr
        torch.cuda.empty_cache()
rollout.FUSED_RECORD = True
