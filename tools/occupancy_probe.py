#!/usr/bin/env python3
"""Diagnostic: how many workgroups of one morphology actually share a CU?  Times k_env_step for n = 256 * k environments
of a single morphology; the time steps up whenever k crosses a multiple of the resident workgroups per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgrl_amd.vec_env import BatchedModularVecEnv
name = sys.argv[1] if len(sys.argv) > 1 else "3d_walker_7_full"
for k in [int(a) for a in sys.argv[2:]] or [4, 5, 6, 7, 8, 12]:
    env = BatchedModularVecEnv([name], 256 * k, seed=1, device="cuda:0")
    env.reset_device()
    a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
    for _ in range(60): env.step_device(a)
    ms = env.time_steps(a, 10)
    print("%s: %2d workgroups per CU offered (lds %d B): %.3f ms per launch, %.3f ms per offered workgroup-per-CU" % (name, k, env.lds_bytes, ms, ms / k))
    del env
