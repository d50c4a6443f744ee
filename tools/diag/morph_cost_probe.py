"""k_env_step cost per walker variant (8192 envs of ONE variant each) next to the 8-variant mix: how much of the mix's launch
time is tail / imbalance rather than work."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd import mjcf
from sgrl_amd.vec_env import BatchedModularVecEnv
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
def run(ns, per):
    env = BatchedModularVecEnv(ns, per, seed=1, device="cuda:0")
    env.reset_device()
    a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
    for _ in range(150):
        a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
        env.step_device(a)
    torch.cuda.synchronize()
    ms = env.time_steps(a, 10)
    lds = env.lds_bytes
    env.close()
    return ms, lds
tot = 0.0
for n in names:
    ms, lds = run([n], 8192)
    nv = mjcf.load_asset(n).nv
    tot += ms / 8
    print("%-36s nv %2d  lds %5d B  %.3f ms per 8192 envs  -> %.3f ms per 1024" % (n, nv, lds, ms, ms / 8), flush=True)
ms, lds = run(names, 1024)
print("sum of the per-variant shares %.3f ms; the mix in one launch %.3f ms (lds %d B)" % (tot, ms, lds))
