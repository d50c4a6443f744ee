#!/usr/bin/env python3
"""Diagnostic build with in-kernel s_memtime stamps per phase of the dynamics evaluation (never the shipped .so:
builds sgrl_amd/libsgrl_hip_prof.so with -DSGRL_PHASE_PROF and loads it in place of the product library for this
process only).  Prints the share of wave cycles per phase for the walker mix."""
import ctypes, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from sgrl_amd import _lib
prof_so = os.path.join(REPO, "sgrl_amd", "libsgrl_hip_prof.so")
srcs = [os.path.join(_lib.CSRC, f) for f in ("engine.hip", "set_actor.hip", "train_gemm.hip", "render.hip")]
if not os.path.exists(prof_so) or os.path.getmtime(prof_so) < max(os.path.getmtime(os.path.join(_lib.CSRC, f)) for f in os.listdir(_lib.CSRC)):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-DSGRL_PHASE_PROF",
                           "-DSGRL_NO_SPECS", "-o", prof_so] + srcs)      # generic kernel only (build it before going to the GPU box)
_lib.LIB_PATH = prof_so
from sgrl_amd.vec_env import BatchedModularVecEnv
names = sorted(n for n in __import__("sgrl_amd.mjcf", fromlist=["x"]).list_assets() if "walker" in n)
env = BatchedModularVecEnv(names, 1024, seed=1, device="cuda:0")
L = _lib.lib()
env.reset_device()
a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
for _ in range(30): env.step_device(a)
torch.cuda.synchronize()
buf = np.zeros((env.num_envs, 16), dtype=np.uint64)
L.sgrl_phase_prof(None, env.num_envs, 1)
for _ in range(5): env.step_device(a)
L.sgrl_phase_prof(ctypes.c_void_p(buf.ctypes.data), env.num_envs, 0)
names_p = ["kinematics", "com/cinert/cdof", "crba", "collide", "rne bias", "enumerate rows", "build rows", "lcp rest", "cholesky", "halfsolve", "diag/b", "backsolve",
           "lcp:flist+A_FF", "lcp:factor+solve", "lcp:u+test"]
for k, sl in enumerate(env.morph_slices):
    b = buf[sl].astype(np.float64).mean(0) / 5
    tot = b[15]
    print("%-34s total %8.0f kcyc/env-step | " % (names[k], tot / 1e3) + " ".join("%s %4.1f%%" % (n[:9], 100 * b[i] / tot) for i, n in enumerate(names_p)) + " | other %4.1f%%" % (100 * (tot - b[:15].sum()) / tot))
