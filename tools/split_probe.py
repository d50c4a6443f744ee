#!/usr/bin/env python3
"""Probe: the walker mix split into a part whose node count fills whole rounds of workgroups (32 768 nodes = 512 blocks of 64
rows) and the rest (one three-limb morphology, 3 072 nodes) on a second handle and stream, against the one-handle forward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgrl_amd.set_policy import make_policy
from sgrl_amd.set_hip import HipSetActor
from sgrl_amd import graph as G, mjcf
dev = torch.device("cuda:0")
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
per = 1024
L = {n: len(mjcf.load_asset(n).parents) for n in names}
print({n: L[n] for n in names})
three = [n for n in names if L[n] == 3][-1]
order = [n for n in names if n != three] + [three]
pol = make_policy(device="cuda:0").eval()
gd = lambda ns: [G.getGraphDict(mjcf.load_asset(n).parents, ["pre", "inlcrs", "postlcrs"], [], device=dev) for n in ns]
obs = torch.randn((per * len(names), 287), device="cuda") * 0.5
out = torch.zeros((per * len(names), 21), device="cuda")

def timed(fn, reps=5, inner=20):
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / inner)
    return float(np.median(best)), float(min(best))

one = HipSetActor(pol); one.configure(gd(order), [per] * 8); one.hold_weights(True)
f_one = lambda: one.forward_batch(obs, out=out, act_ld=21)
f_one(); torch.cuda.synchronize()
ref = out.clone()
print("one handle, %d nodes: median %.3f ms (min %.3f)" % ((one.num_nodes,) + timed(f_one)))

A = HipSetActor(pol); A.configure(gd(order[:7]), [per] * 7); A.hold_weights(True)
B = HipSetActor(pol); B.configure(gd(order[7:]), [per]); B.hold_weights(True)
nA = per * 7
sB = torch.cuda.Stream()
oA, oB, uA, uB = obs[:nA], obs[nA:], out[:nA], out[nA:]
def f_split():
    cur = torch.cuda.current_stream()
    sB.wait_stream(cur)
    with torch.cuda.stream(sB):
        B.forward_batch(oB, out=uB, act_ld=21)
    A.forward_batch(oA, out=uA, act_ld=21)
    cur.wait_stream(sB)
out.zero_(); f_split(); torch.cuda.synchronize()
print("split result equals the one-handle result:", bool(torch.equal(out, ref)), float((out - ref).abs().max()))
print("split %d + %d nodes on two streams: median %.3f ms (min %.3f)" % ((A.num_nodes, B.num_nodes) + timed(f_split)))
fa = lambda: A.forward_batch(oA, out=uA, act_ld=21)
fb = lambda: B.forward_batch(oB, out=uB, act_ld=21)
print("part A alone: median %.3f ms (min %.3f)" % timed(fa))
print("part B alone: median %.3f ms (min %.3f)" % timed(fb))
