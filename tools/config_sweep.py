#!/usr/bin/env python3
"""Run the engine (+ SET forward) on the other BASELINE.json configs' single-GPU shares and report health + speed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgrl_amd import mjcf
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy
A = mjcf.list_assets()
fam = lambda f: sorted(n for n in A if f in n)
CONFIGS = {
    "config2_hopper++_4096": (fam("hopper"), [1365, 1365, 1366]),
    "config4_humanoid++_share_4096": (fam("humanoid")[:6], [683, 683, 683, 683, 682, 682]),
    "cheetah_8x256": (fam("cheetah")[:8], [256] * 8),
    "config5_cwhh_share_8192": (None, None),
}
cw = sorted(fam("cheetah")[:8] + [n for n in fam("walker") if n not in ("3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot")] + fam("hopper") + fam("humanoid")[:6])
CONFIGS["config5_cwhh_share_8192"] = (cw, [8192 // len(cw)] * len(cw))
pol = make_policy(device="cuda:0").eval()
for name, (names, counts) in CONFIGS.items():
    ro = Rollout(names, counts, policy=pol, seed=3, device="cuda:0")
    env = ro.env
    ro.reset()
    for _ in range(120):
        ro.step(ro.random_actions())
    torch.cuda.synchronize()
    t0 = time.time()
    K = 10
    for _ in range(K):
        obs, rew, done, _ = ro.step(ro.random_actions())
        ro.policy_forward(obs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    ms_env = env.time_steps(ro.actions, 5)
    ms_set = ro.actor.time_forward(env.obs, ro.policy_actions, 3)
    rec, cnt = env.get_records()
    ok = bool(torch.isfinite(env.obs).all()) and bool(torch.isfinite(env.rew).all())
    print("%-34s morphs %2d envs %5d lds %6d B | step+SET %.2f ms (%.0f env-steps/s) | k_env_step %.2f ms, SET %.2f ms | finite %s overflow-envs %d episodes/env %.1f bpp-fail %d hbm-solve-envs %d" % (
        name, len(names), env.num_envs, env.lds_bytes, dt * 1e3, env.num_envs / dt, ms_env, ms_set, ok,
        int((cnt[:, 2] > 0).sum()), cnt[:, 1].mean(), int((((cnt[:, 3] >> 8) & 255) > 0).sum()), int(((cnt[:, 3] >> 16) > 0).sum())))
    env.close()
    del ro
