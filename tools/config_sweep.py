#!/usr/bin/env python3
"""Run the engine (+ SET forward) on the single-GPU shares of the other BASELINE.json configs and report health + speed.
Writes gpurun_out/config_sweep.json (copied to profiles/rN_config_sweep.json by hand)."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from sgrl_amd import mjcf
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy
A = mjcf.list_assets()
fam = lambda f: sorted(n for n in A if f in n)
HELD_OUT = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
            "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
cw = sorted(n for n in A if n not in HELD_OUT)
CONFIGS = {
    "config2_hopper++_4096": (fam("hopper"), [1365, 1365, 1366]),
    "config3_walker++_8192": (fam("walker"), [1024] * 8),
    "config4_humanoid++_share_4096": (fam("humanoid"), [512] * 8),
    "cheetah++_10x256": (fam("cheetah"), [256] * 10),
    "config5_cwhh_share_8188": (cw, [8192 // len(cw)] * len(cw)),
}
pol = make_policy(device="cuda:0").eval()
out = {}
out["_lib"] = os.environ.get("SGRL_HIP_LIB", "sgrl_amd/libsgrl_hip.so")
for name, (names, counts) in CONFIGS.items():
    if os.environ.get("SWEEP_ONLY") and not any(k in name for k in os.environ["SWEEP_ONLY"].split(",")):
        continue
    ro = Rollout(names, counts, policy=pol, seed=3, device="cuda:0")
    env = ro.env
    ro.reset()
    for _ in range(150):
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
    r = {"morphologies": len(names), "envs": env.num_envs, "lds_bytes": env.lds_bytes, "launch_groups": env.launch_groups,
         "ms_step_plus_set": round(dt * 1e3, 3), "env_steps_per_s": round(env.num_envs / dt, 1),
         "ms_k_env_step": round(ms_env, 3), "ms_set_forward": round(ms_set, 3), "set_nodes": ro.actor.num_nodes,
         "finite": ok, "row_overflow_envs": int((cnt[:, 2] > 0).sum()), "episodes_per_env": round(float(cnt[:, 1].mean()), 2),
         "block_pivot_failures": int((((cnt[:, 3] >> 8) & 255) > 0).sum()), "hbm_slab_solve_envs_last_step": int(((cnt[:, 3] >> 16) > 0).sum())}
    out[name] = r
    print(name, json.dumps(r), flush=True)
    env.close()
    del ro
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
with open(os.path.join(REPO, "gpurun_out", os.environ.get("SWEEP_OUT", "config_sweep.json")), "w") as f:
    json.dump(out, f, indent=1)
