"""Free-running steps from contact-rich states (many-geom morphologies lying on the floor): exercises the > 64-row path
(pgs_big) for many steps.  usage: contact_stress.py [steps=300] [per=64]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from sgrl_amd.vec_env import BatchedModularVecEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
per = int(sys.argv[2]) if len(sys.argv) > 2 else 64
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["3d_cheetah_14_full", "3d_humanoid_9_full", "3d_cheetah_10_tail_leftbleg", "3d_humanoid_7_left_arm"]
env = BatchedModularVecEnv(names, per, seed=3, device="cuda:0")
env.reset_device()
rec, cnt = env.get_records()
quats = [[1, 0, 0, 0], [0.70710678, 0.70710678, 0, 0], [0.70710678, 0, 0.70710678, 0]]
for i in range(env.num_envs):
    m = env.models[env.env_morph[i]]
    fb = env._blobs[env.env_morph[i]][1]
    q = np.array(fb[16:16 + m.nq]); q[2] = 0.05 + 0.02 * (i % 5); q[3:7] = quats[i % 3]
    rec[i, :m.nq] = q; rec[i, m.nq:m.nq + m.nv] = 0
env.set_records(rec, cnt)
g = torch.Generator(device="cuda").manual_seed(1)
pgs = 0
for t in range(steps):
    a = (torch.rand((env.num_envs, env.action_max_len), device="cuda", generator=g) * 2 - 1).contiguous()
    env.step_device(a, auto_reset=(t % 50 != 0))
    if t % 25 == 24:
        torch.cuda.synchronize()
        c = env.get_counters()
        pgs += int((c[:, 3] & 0xFF).sum())
        print("step", t + 1, "pgs evals (last step)", int((c[:, 3] & 0xFF).sum()), "slab solves", int((c[:, 3] >> 16).sum()), "dropped", int((c[:, 2] > 0).sum()),
              "finite", bool(torch.isfinite(env.obs).all()), flush=True)
print("done; pgs evals seen", pgs, "groups", env.launch_groups, "fixed", env.fixed_dim_groups)
