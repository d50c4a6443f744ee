"""Grouped weight gradients (csrc/train_gemm.hip k_sgemm_wgroup: 32 x 32 and 64 x 64 tiles, contractions split over workgroups whose
partial tiles travel as agent-scope stores / loads) under the conditions of a training loop: many launches back to back on one
stream WITHOUT synchronisation, every launch twelve different products of the update's shapes with fresh data, all of them reusing
the same scratch slots and counters -- every result against float64, and the whole sequence replayed from a hipGraph."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS = [1792, 2304, 3328, 3584, 5376, 700]
OUTS = [(256, 256), (1024, 256), (256, 528), (768, 256), (128, 256), (30, 128), (252, 128), (128, 32), (256, 136), (1, 256), (64, 64)]


def _launches(n_launch, seed):
    from sgrl_amd import train_ops as T
    rng = np.random.RandomState(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = []
    for _ in range(n_launch):
        recs = []
        for _ in range(12):
            M = ROWS[rng.randint(len(ROWS))]
            N, K = OUTS[rng.randint(len(OUTS))]
            dy = torch.randn(M, N, device="cuda", generator=g)
            x = torch.randn(M, K, device="cuda", generator=g)
            rd = (torch.rand(M, device="cuda", generator=g) + 0.5) if rng.rand() < 0.3 else None
            recs.append((dy, x, rd, torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")))
        d = np.zeros(12, dtype=T._DESC)
        for i, (dy, x, rd, dw, db) in enumerate(recs):
            M, N = dy.shape
            K = x.shape[1]
            d[i] = (dy.data_ptr(), 0, 0 if rd is None else rd.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), N, N, K, K, M, N, K, 0)
        out.append((recs, d))
    return out


def _run(launches):
    from sgrl_amd import train_ops as T
    L = T._L()
    dev = torch.device("cuda:0")
    ws = T._scratch(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for recs, d in launches:
        T._check(L, L.sgrl_linear_wgrad_group(12, ctypes.c_void_p(d.ctypes.data), T._p(ws), st), "wgroup")


def _verify(launches):
    worst = 0.0
    for recs, _ in launches:
        for dy, x, rd, dw, db in recs:
            g = dy.double() if rd is None else dy.double() / rd.double()[:, None]
            ref, refb = g.t() @ x.double(), g.sum(0)
            tol = 3e-6 * np.sqrt(dy.shape[0] / 64 + 1)
            e = float((dw.double() - ref).abs().max()) / (float(ref.abs().max()) + 1.0)
            eb = float((db.double() - refb).abs().max()) / (float(refb.abs().max()) + 1.0)
            worst = max(worst, e, eb)
            assert e < tol and eb < tol, (tuple(dy.shape), tuple(x.shape), e, eb)
    return worst


def test_back_to_back_groups_of_different_products_match_float64():
    launches = _launches(40, seed=1)
    _run(launches)              # 40 launches in flight, the scratch slots reused by every one of them
    torch.cuda.synchronize()
    _verify(launches)
    # again with other data in the SAME output tensors' neighbours (fresh tensors, same scratch)
    launches2 = _launches(40, seed=2)
    _run(launches2)
    torch.cuda.synchronize()
    _verify(launches2)


def test_the_same_sequence_replayed_from_a_graph():
    launches = _launches(24, seed=3)
    _run(launches)              # warm-up on the current stream (scratch of this stream exists before the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _run(launches)
    for rep in range(5):
        for recs, _ in launches:
            for dy, x, rd, dw, db in recs:
                dw.fill_(float("nan")); db.fill_(float("nan"))
                dy.normal_(); x.normal_()          # new data at the baked addresses
        g.replay()
        torch.cuda.synchronize()
        _verify(launches)
