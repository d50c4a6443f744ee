#!/usr/bin/env python3
"""Diagnostic: the walker-mix SET forward as ONE batch vs as TWO half batches (morphologies 0-3 / 4-7) on two handles and two
streams (kernels of the two forwards interleave: bandwidth-bound and matrix-bound kernels overlap)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd.set_policy import make_policy
from sgrl_amd.set_hip import HipSetActor
from sgrl_amd import graph as G, mjcf
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
pol = make_policy(device="cuda:0").eval()
gds = [G.getGraphDict(mjcf.load_asset(n).parents, ["pre", "inlcrs", "postlcrs"], [], device=dev) for n in names]
Ls = [len(mjcf.load_asset(n).parents) for n in names]
n_env = per * len(names)
obs = torch.randn((n_env, 287), device="cuda") * 0.5
out1 = torch.zeros((n_env, 21), device="cuda")
one = HipSetActor(pol)
one.configure(gds, [per] * len(names))

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3

ms_one = timeit(lambda: one.forward_batch(obs, out=out1, act_ld=21))
ref = out1.clone()
res = {"one_batch_ms": round(ms_one, 4)}
for split in ([4], [3], [2, 4, 6]):
    cuts = [0] + split + [len(names)]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        h = HipSetActor(pol)
        h.configure(gds[a:b], [per] * (b - a))
        parts.append((h, a * per, b * per, torch.cuda.Stream(device=dev)))
    out2 = torch.zeros_like(out1)
    def run():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        for h, r0, r1, st in parts:
            st.wait_event(ev)
            with torch.cuda.stream(st):
                h.forward_batch(obs[r0:r1], out=out2[r0:r1], act_ld=21)
        for h, r0, r1, st in parts:
            e2 = torch.cuda.Event(); e2.record(st); cur.wait_event(e2)
    ms = timeit(run)
    torch.cuda.synchronize()
    res["split_at_%s_ms" % "_".join(map(str, split))] = round(ms, 4)
    res["split_at_%s_maxdiff" % "_".join(map(str, split))] = float((out2 - ref).abs().max())
    del parts
print(json.dumps(res))
