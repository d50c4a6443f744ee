#!/usr/bin/env python3
"""Diagnostic: the TD3 update's dense products (csrc/train_gemm.hip) one shape at a time at the reference's update batch (256
transitions x 7 limbs = 1 792 rows, 5 376 for the three-vector channels): forward (x . w^T), input gradient (g . w), both twin
forms and one grouped weight-gradient launch, HIP-event time per launch over back-to-back launches on one stream, results checked
against float64.  SGRL_TRAIN_RM=0 selects the transposing staging of rounds 2-4."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import train_ops as T
L = T._L()
dev = torch.device("cuda:0")
ws = T._scratch(dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = T._p
REPS = int(os.environ.get("REPS", "200"))
RELU = int(os.environ.get("DGRAD_RELU", "1"))      # 0: the input gradients without the ReLU mask of the forward output

def timed(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3

SHAPES = [(1792, 256, 256), (1792, 128, 256), (1792, 256, 1024), (1792, 1024, 256), (1792, 768, 256), (5376, 30, 128), (5376, 128, 256),
          (5376, 128, 32), (5376, 252, 128), (5376, 256, 128)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
res = {}
for (M, N, K) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); g = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    x1 = torch.randn(M, K, device=dev); w1 = torch.randn(N, K, device=dev) / K ** 0.5; y1 = torch.empty(M, N, device=dev)
    g1 = torch.randn(M, N, device=dev); dx1 = torch.empty(M, K, device=dev)
    def fwd(): T._check(L, L.sgrl_linear_forward(P(x), K, P(w), K, P(b), None, P(y), N, M, N, K, 1, st), "fwd")
    def dgrad(): T._check(L, L.sgrl_linear_backward(P(g), N, P(y), N, RELU, None, None, 0, P(w), K, P(dx), K, None, 0, None, None, M, N, K, P(ws), st), "dgrad")
    def fwd2(): T._check(L, L.sgrl_linear_forward_twin(P(x), P(x1), K, P(w), P(w1), K, P(b), P(b), None, None, P(y), P(y1), N, M, N, K, 1, st), "fwd2")
    def dgrad2(): T._check(L, L.sgrl_linear_dgrad_twin(P(g), P(g1), N, P(y), P(y1), N, RELU, None, None, P(w), P(w1), K, P(dx), P(dx1), K, None, None, M, N, K, st), "dgrad2")
    r = {}
    r["fwd_us"] = round(timed(fwd), 2)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    r["fwd_err"] = "%.1e" % float((y.double() - ref).abs().max())
    r["dgrad_us"] = round(timed(dgrad), 2)
    refd = (g.double() * (y > 0) if RELU else g.double()) @ w.double()
    r["dgrad_err"] = "%.1e" % float((dx.double() - refd).abs().max())
    r["fwd_twin_us"] = round(timed(fwd2), 2)
    ref1 = torch.relu(x1.double() @ w1.double().t() + b.double())
    r["fwd_twin_err"] = "%.1e" % float((y1.double() - ref1).abs().max())
    r["dgrad_twin_us"] = round(timed(dgrad2), 2)
    refd1 = (g1.double() * (y1 > 0) if RELU else g1.double()) @ w1.double()
    r["dgrad_twin_err"] = "%.1e" % float((dx1.double() - refd1).abs().max())
    # one grouped weight-gradient launch of 12 such layers
    recs = []
    for _ in range(12):
        recs.append((torch.randn(M, N, device=dev), torch.rand(M, N, device=dev), torch.randn(M, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
    d = np.zeros(12, dtype=T._DESC)
    for i, (dy_, y_, x_, dw_, db_) in enumerate(recs):
        d[i] = (dy_.data_ptr(), 0, 0, x_.data_ptr(), dw_.data_ptr(), db_.data_ptr(), N, N, K, K, M, N, K, 0)
    def wg(): T._check(L, L.sgrl_linear_wgrad_group(12, ctypes.c_void_p(d.ctypes.data), P(ws), st), "wgroup")
    r["wgrad_group12_us"] = round(timed(wg), 2)
    dy_, y_, x_, dw_, db_ = recs[5]
    r["wgrad_err"] = "%.1e" % float((dw_.double() - dy_.double().t() @ x_.double()).abs().max())
    gf = 2.0 * M * N * K / 1e9
    r["gflop"] = round(gf, 3)
    r["fwd_tflops"] = round(gf / r["fwd_us"] * 1e3, 1)
    res["%dx%dx%d" % (M, N, K)] = r
    print("%dx%dx%d" % (M, N, K), json.dumps(r), flush=True)
