#!/usr/bin/env python3
"""Config 5's collection step, one arm per process (for rocprofv3 --kernel-trace --stats): collect_one.py <fused_record 0|1> [steps]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd import mjcf, rollout
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
rollout.FUSED_RECORD = sys.argv[1] == "1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
names = sorted(n for n in mjcf.list_assets() if n not in HELD)
tr = DeviceTrainer(names, 8192 // len(names), args=default_train_args(), seed=1, device="cuda:0", max_buffer_size=200000, graph_updates=False)
tr.warmup(60)
for _ in range(8):          # untimed policy steps (one-time costs of the first one: code objects, weight pack, stream probe)
    if tr.collect_step():
        tr.begin_round()
torch.cuda.synchronize()
t0 = time.time()
host = 0.0
for _ in range(K):
    h0 = time.time()
    fin = tr.collect_step()
    host += time.time() - h0
    if fin:
        tr.begin_round()
t_host_enqueue = host / K
torch.cuda.synchronize()
print("fused_record %s: %.3f ms per collection step (host time inside collect_step %.3f ms)" % (rollout.FUSED_RECORD, (time.time() - t0) / K * 1e3, t_host_enqueue * 1e3), flush=True)
