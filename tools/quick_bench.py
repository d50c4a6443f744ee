#!/usr/bin/env python3
"""Quick engine-only timing on the GPU box: walker mix, random actions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgrl_amd.vec_env import BatchedModularVecEnv
names = sorted(["3d_walker_2_right_leg_left_knee", "3d_walker_3_left_leg_right_foot", "3d_walker_3_left_knee_right_knee",
         "3d_walker_4_right_knee_left_foot", "3d_walker_5_foot", "3d_walker_5_left_knee", "3d_walker_6_right_foot",
         "3d_walker_7_full"])
if os.environ.get("QB_FAMILY"):      # e.g. QB_FAMILY=hopper: every shipped morphology of that family
    from sgrl_amd import mjcf
    names = sorted(n for n in mjcf.list_assets() if n.split("_")[1] == os.environ["QB_FAMILY"])
if os.environ.get("QB_NAMES"):       # explicit comma-separated morphology list
    names = os.environ["QB_NAMES"].split(",")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kw = {}
if len(sys.argv) > 3: kw["max_rows"] = int(sys.argv[3])
if len(sys.argv) > 4: kw["pgs_iters"] = int(sys.argv[4])
env = BatchedModularVecEnv(names, per, seed=1, device="cuda:0", **kw)
print("n_env", env.num_envs, "lds_bytes", env.lds_bytes)
env.reset_device()
a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
WARM = int(os.environ.get('WARM', '30'))
for _ in range(WARM):
    a = (torch.rand((env.num_envs, env.action_max_len), device='cuda') * 2 - 1).contiguous()
    env.step_device(a)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
    env.step_device(a)
torch.cuda.synchronize()
dt = time.time() - t0
print("ms/step %.3f env-steps/s %.0f" % (dt / steps * 1e3, env.num_envs * steps / dt))
ms = env.time_steps(a, 10)
print("hip-event ms/launch %.3f -> %.0f env-steps/s" % (ms, env.num_envs / ms * 1e3))
rec, cnt = env.get_records()
import numpy as np
for k, sl in enumerate(env.morph_slices):
    c3 = cnt[sl, 3]
    print("  %-36s envs with matrix-free-PGS evals %5d (max evals %2d) | envs with block-pivot failures %4d" % (
        names[k], int(((c3 & 255) > 0).sum()), int((c3 & 255).max()), int(((c3 >> 8) > 0).sum())))
print("overflow envs", int((cnt[:, 2] > 0).sum()), "episodes mean", cnt[:, 1].mean())
