#!/usr/bin/env python3
"""The PCIe-inclusive rate of the drop-in boundary: BatchedModularVecEnv.step(list of host arrays) -> host obs / rewards / dones /
infos (the reference's SubprocVecEnv surface) against step_device (tensors stay in HBM), walker mix 8 x 1024, engine only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd.vec_env import BatchedModularVecEnv
from sgrl_amd import mjcf
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
env = BatchedModularVecEnv(names, 1024, seed=1, device="cuda:0")
env.reset()
rng = np.random.RandomState(0)
acts = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
for _ in range(30):
    env.step(list(acts))
torch.cuda.synchronize()
K = 30
t0 = time.time()
for _ in range(K):
    obs, rew, done, infos = env.step(list(acts))
dt_np = (time.time() - t0) / K
a = torch.from_numpy(acts).cuda()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(K):
    env.step_device(a)
torch.cuda.synchronize()
dt_dev = (time.time() - t0) / K
# the transfers alone: actions in, obs / reward / done / dist / truncated out
t0 = time.time()
for _ in range(K):
    env._act.copy_(torch.from_numpy(acts)); o = env.obs.cpu(); r = env.rew.cpu(); d = env.done.cpu(); x = env.dist.cpu(); y = env.trunc.cpu()
torch.cuda.synchronize()
dt_copy = (time.time() - t0) / K
print("NumPy VecEnv.step (host in, host out, 8192 info dicts): %.2f ms/step = %.0f env-steps/s | device surface %.2f ms/step = %.0f env-steps/s | "
      "transfers alone (0.7 MB in, 9.5 MB out) %.2f ms" % (dt_np * 1e3, env.num_envs / dt_np, dt_dev * 1e3, env.num_envs / dt_dev, dt_copy * 1e3))
