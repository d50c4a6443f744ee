import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd.vec_env import BatchedModularVecEnv
for n in sys.argv[1:]:
    env = BatchedModularVecEnv([n], 8192, seed=1, device="cuda:0")
    env.reset_device()
    for _ in range(150):
        a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
        env.step_device(a)
    torch.cuda.synchronize()
    print("%s light=%s: %.3f ms per 8192 envs (lds %d)" % (n, os.environ.get("SGRL_LIGHT", "1"), env.time_steps(a, 10), env.lds_bytes), flush=True)
    env.close()
