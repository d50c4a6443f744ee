#!/usr/bin/env python3
"""Diagnostic: what the vendor f32 GEMM reaches on the SET actor's shapes (ceiling estimate for set_actor.hip's k_gemm)."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
N0 = 35840
for name, M, N, K in [("qkv", N0, 768, 256), ("l4", N0, 1024, 256), ("lg1", N0, 256, 544), ("l3", N0, 256, 256),
                      ("vg", 3 * N0, 256, 128), ("gout", 3 * N0, 128, 256), ("lg2", N0, 128, 256)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    for _ in range(3): C = A @ W.t()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): C = A @ W.t()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%-5s M %6d N %4d K %3d: %7.1f us %6.1f TFLOP/s" % (name, M, N, K, dt * 1e6, 2.0 * M * N * K / dt / 1e12))
