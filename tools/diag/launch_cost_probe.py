#!/usr/bin/env python3
"""Diagnostic: host time of one SET forward call (enqueue only) against its GPU time -- is the forward launch-bound?"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd.set_policy import make_policy
from sgrl_amd.set_hip import HipSetActor
from sgrl_amd import graph as G, mjcf
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
per = 1024
dev = torch.device("cuda:0")
pol = make_policy(device="cuda:0").eval()
gds = [G.getGraphDict(mjcf.load_asset(n).parents, ["pre", "inlcrs", "postlcrs"], [], device=dev) for n in names]
act = HipSetActor(pol)
act.configure(gds, [per] * len(names))
obs = torch.randn((per * len(names), 287), device="cuda") * 0.5
out = torch.zeros((per * len(names), 21), device="cuda")
for _ in range(5): act.forward_batch(obs, out=out, act_ld=21)
torch.cuda.synchronize()
res = {}
# (a) host time per call while the queue is EMPTY at the start (sync before each call)
hs = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); act.forward_batch(obs, out=out, act_ld=21); hs.append(time.perf_counter() - t0)
res["host_ms_per_call_after_sync"] = round(float(np.median(hs)) * 1e3, 3)
# (b) back-to-back calls: host enqueue time and total wall
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): act.forward_batch(obs, out=out, act_ld=21)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
res["back_to_back_host_ms_per_call"] = round((t1 - t0) / 30 * 1e3, 3)
res["back_to_back_wall_ms_per_call"] = round((t2 - t0) / 30 * 1e3, 3)
res["gpu_ms_per_forward_hip_events"] = round(act.time_forward(obs, out, 10), 3)
# (c) the same forward captured into a hipGraph (single-stream inside a capture) and replayed
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
import gc; gc.collect(); gc.disable()
with torch.cuda.stream(s):
    act.forward_batch(obs, out=out, act_ld=21)
    with torch.cuda.graph(g, stream=s):
        act.forward_batch(obs, out=out, act_ld=21)
gc.enable()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
res["graph_replay_host_ms"] = round((t1 - t0) / 30 * 1e3, 3)
res["graph_replay_wall_ms (single-stream capture)"] = round((t2 - t0) / 30 * 1e3, 3)
print(json.dumps(res, indent=1))
