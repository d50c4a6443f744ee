import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy
names = ["3d_hopper_3_shin", "3d_hopper_4_lower_shin", "3d_hopper_5_full"]
pol = make_policy(device="cuda:0").eval()
pol.load_state_dict(torch.load("build/hopper_actor.pt", map_location="cuda:0"))
ro = Rollout(names, 8, policy=pol, seed=5, device="cuda:0")
env = ro.env
ro.reset()
ended = {}
for t in range(140):
    a = (ro.policy_forward().clone() * ro.act_mask).contiguous()
    rec, cnt = env.get_records()
    ro.step(a)
    torch.cuda.synchronize()
    d = env.done.cpu().numpy()
    for i in np.nonzero(d)[0]:
        if i not in ended:
            m = env.models[env.env_morph[i]]
            q, v, _, _ = env.state_of(rec, i)
            ended[i] = (t + 1, float(q[2]), float(np.abs(v).max()), float(np.abs(q[7:]).max()))
    if t in (0, 50, 98):
        print("t", t, "action abs mean %.3f max %.3f" % (float(a.abs().mean()), float(a.abs().max())), "z", np.round(rec[::8, 2], 3))
for i in sorted(ended):
    print(names[env.env_morph[i]], "env", i, "ended at step %d: z before %.3f max|qvel| %.1f max|joint q| %.2f" % ended[i])
