#!/usr/bin/env python3
"""BASELINE.json config 5 on one GPU: the 23 cwhh training morphologies x 356 envs, DeviceTrainer (collection rounds +
TD3 update schedule).  Reports collection env-steps/s (policy forward + exploration noise + engine step + replay ingest)
and TD3 updates/s (batch = args.agent_batch_size = 256, the reference's configs/default.py:61; one morphology per update, PyTorch-ROCm autograd with HIP no-grad targets).
Writes gpurun_out/train_bench.json."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer
HELD_OUT = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
            "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD_OUT)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 // len(names)
args = default_train_args()
GRAPH = os.environ.get("SGRL_GRAPH_UPDATES", "1") != "0"
TUNE = os.environ.get("SGRL_TUNE_GEMMS", "1") != "0"
tr = DeviceTrainer(names, per, args=args, seed=1, device="cuda:0", max_buffer_size=200000, graph_updates=GRAPH, tune_gemms=TUNE)
n = tr.ro.env.num_envs
tr.warmup(60)                      # fills the buffers (random actions), several rounds
for _ in range(8):                 # untimed POLICY steps: the first one loads code objects (PyTorch's noise / clamp kernels, the SET forward's
    if tr.collect_step():          # stream probe) and packs the weights -- 16 to 64 ms once, which a 40-step window would book as 0.4 to 1.6 ms
        tr.begin_round()           # per step (round 6: that is what made the one-launch round bookkeeping look 1 ms slower than it is faster)
torch.cuda.synchronize()
t0 = time.time(); K = 40
for _ in range(K):
    if tr.collect_step():
        tr.begin_round()
torch.cuda.synchronize()
t_collect = (time.time() - t0) / K
# updates through the trainer's own schedule: a first pass warms / tunes / captures, the second is timed
tr.update_after_round(max_iters=4)
torch.cuda.synchronize()
t0 = time.time()
iters = tr.update_after_round(max_iters=10)
torch.cuda.synchronize()
U = 10 * len(names)
t_update = (time.time() - t0) / U
out = {"morphologies": len(names), "envs": n, "ms_per_collection_step": round(t_collect * 1e3, 3),
       "collection_env_steps_per_s": round(n / t_collect, 1), "ms_per_td3_update": round(t_update * 1e3, 3),
       "td3_updates_per_s": round(1.0 / t_update, 2), "batch_size": tr.batch_size, "hipgraph_updates": GRAPH, "tunableop": TUNE,
       "row_overflow_envs": tr.ro.env.row_overflow_envs(), "buffer_rows": [b.max_sample_size for b in tr.buffers]}
print(json.dumps(out))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "train_bench.json"), "w"), indent=1)
