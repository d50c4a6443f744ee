#!/usr/bin/env python3
"""Diagnostic: sgrl_linear_wgrad_group (12 weight gradients per launch) in isolation: time per launch for groups of 1 / 4 / 12
problems of some shapes of the TD3 update (M = 700 rows)."""
import sys, os, time, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import train_ops as T
L = T._L()
dev = torch.device("cuda:0")
ws = T._scratch(dev)
st = torch.cuda.current_stream(dev).cuda_stream

def run(shapes, reps=50, relu=False, bias=True):
    recs = []
    for (M, N, K) in shapes:
        dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); y = torch.rand(M, N, device=dev)
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        recs.append((dy, y, x, dw, db, M, N, K))
    d = np.zeros(len(recs), dtype=T._DESC)
    for i, (dy, y, x, dw, db, M, N, K) in enumerate(recs):
        d[i] = (dy.data_ptr(), y.data_ptr() if relu else 0, 0, x.data_ptr(), dw.data_ptr(), db.data_ptr() if bias else 0, N, N, K, K, M, N, K, 1 if relu else 0)
    def go():
        T._check(L, L.sgrl_linear_wgrad_group(len(recs), ctypes.c_void_p(d.ctypes.data), T._p(ws), ctypes.c_void_p(st)), "wgroup")
    for _ in range(5): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    # check one result
    dy, y, x, dw, db, M, N, K = recs[0]
    g = dy * (y > 0) if relu else dy
    err = float((dw - g.t() @ x).abs().max())
    return e0.elapsed_time(e1) / reps * 1e3, err

res = {}
for name, shp in (("256x256", (700, 256, 256)), ("1024x256", (700, 1024, 256)), ("256x576", (700, 256, 576)), ("128x256", (700, 128, 256)),
                  ("64x128_M2100", (2100, 64, 128)), ("768x256", (700, 768, 256))):
    for n in (1, 4, 12):
        us, err = run([shp] * n)
        res["%s x%d" % (name, n)] = (round(us, 1), "%.1e" % err)
# one critic layer's set (attention + feed-forward block), roughly
layer = [(2100, 256, 128), (2100, 64, 128), (700, 256, 576), (700, 128, 256), (700, 768, 256), (2100, 64, 128), (700, 256, 576), (700, 128, 256),
         (700, 256, 256), (700, 256, 256), (700, 128, 256), (700, 1024, 256)]
us, err = run(layer)
res["one layer's 12 products"] = (round(us, 1), "%.1e" % err)
for i, s in enumerate(layer):
    res["  alone %s" % (s,)] = run([s])[0].__round__(1)
print(json.dumps(res, indent=1))
