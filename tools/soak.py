#!/usr/bin/env python3
"""Soak (VERDICT r2 item 4 ii): a big mixed batch driven by three kinds of actions -- uniform random (episodes of tens of
steps), a joint-space PD controller about the reset pose (sustained stance, episodes of hundreds of steps) and a SET policy
(the TD3-trained hopper actor of tools/learn_curve.py when build/hopper_actor.pt is present, random-init weights otherwise).
Every output must stay finite, no constraint row may be dropped, the block-pivot solver must not fail, episodes must keep
turning over.  Writes gpurun_out/soak.json (kept as profiles/r3_soak.json).

usage: soak.py [all|walker|...] [steps=20000]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from sgrl_amd import mjcf
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy

which = sys.argv[1] if len(sys.argv) > 1 else "all"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
names = sorted(n for n in mjcf.list_assets() if which == "all" or which in n)
per = max(1, 8192 // len(names))
policy = make_policy(device="cuda:0").eval()
wpath = os.path.join(REPO, "build", "hopper_actor.pt")
trained = os.path.exists(wpath)
if trained:
    sd = torch.load(wpath, map_location="cuda:0")
    policy.load_state_dict(sd)
ro = Rollout(names, per, policy=policy, seed=11, device="cuda:0")
env = ro.env
n, L = env.num_envs, env.obs_max_len // 41
BINS = [1, 10, 30, 100, 300, 1000, 10 ** 9]
out = {"batch": "%d morphologies x %d envs" % (len(names), per), "launch_groups": env.launch_groups,
       "fixed_dim_groups": env.fixed_dim_groups, "policy_weights": "TD3-trained hopper actor (tools/learn_curve.py)" if trained else "random init",
       "phases": []}


def joint_angles():
    return env.obs.view(n, L, 41)[:, :, 24:27]


def run_phase(driver, nsteps):
    ro.reset()
    q_ref = joint_angles().clone()
    q_prev = q_ref.clone()
    ep_len = torch.zeros(n, dtype=torch.long, device="cuda:0")
    hist = torch.zeros(len(BINS) - 1, dtype=torch.long, device="cuda:0")
    edges = torch.tensor(BINS, device="cuda:0")
    bad = torch.zeros((), dtype=torch.long, device="cuda:0")
    pivot = slab = pgs = 0
    longest = 0
    t0 = time.time()
    for t in range(nsteps):
        if driver == "random":
            a = ro.random_actions()
        elif driver == "pd":
            q = joint_angles()
            a = (4.0 * (q_ref - q) - 0.3 * (q - q_prev) / 0.008).reshape(n, 3 * L)
            q_prev = q.clone()
            a = ((a + 0.1 * torch.randn(a.shape, device=a.device, generator=ro.gen)).clamp_(-1, 1) * ro.act_mask).contiguous()
        else:
            a = (ro.policy_forward().clone() * ro.act_mask).contiguous()
        obs, rew, done, dist = ro.step(a)
        d = done.to(torch.bool)
        ep_len += 1
        idx = torch.bucketize(ep_len, edges, right=True) - 1
        hist += torch.bincount(idx[d].clamp_(0, len(BINS) - 2), minlength=len(BINS) - 1)
        ep_len = torch.where(d, torch.zeros_like(ep_len), ep_len)
        bad += (~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum() + (~torch.isfinite(dist)).sum()
        if driver == "pd":
            qn = joint_angles()
            q_ref = torch.where(d[:, None, None], qn, q_ref)
            q_prev = torch.where(d[:, None, None], qn, q_prev)
        if t % 100 == 99:          # solver diagnostics of the last step of every environment (sampled every 100 steps)
            c = env.get_counters()
            pgs += int((c[:, 3] & 0xFF).sum()); pivot += int(((c[:, 3] >> 8) & 0xFF).sum()); slab += int((c[:, 3] >> 16).sum())
            longest = max(longest, int(c[:, 0].max()))
        if t % 2000 == 1999:
            print("%s step %d (%.0f s)" % (driver, t + 1, time.time() - t0), flush=True)
    torch.cuda.synchronize()
    rec, cnt = env.get_records()
    ph = {"driver": driver, "steps": nsteps, "wall_s": round(time.time() - t0, 1), "non_finite_outputs": int(bad.item()),
          "non_finite_state_values": int((~np.isfinite(rec)).sum()), "envs_with_dropped_rows": int((cnt[:, 2] > 0).sum()),
          "block_pivot_failures_sampled": pivot, "matrix_free_pgs_evaluations_sampled": pgs, "hbm_slab_solves_sampled": slab,
          "episodes_finished": int(hist.sum().item()), "longest_running_episode_seen": longest,
          "episode_length_histogram": {"%d-%d" % (BINS[i], BINS[i + 1] - 1) if i < len(BINS) - 2 else ">=%d" % BINS[i]: int(hist[i].item())
                                       for i in range(len(BINS) - 1)},
          "envs_with_dropped_rows_by_family": {f: int(sum(int((cnt[sl, 2] > 0).sum()) for k, sl in enumerate(env.morph_slices) if f in names[k]))
                                               for f in ("hopper", "walker", "humanoid", "cheetah") if any(f in nm for nm in names)}}
    print(json.dumps(ph), flush=True)
    out["phases"].append(ph)


run_phase("random", steps // 4)
run_phase("pd", steps // 2)
run_phase("policy", steps // 4)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "soak.json"), "w"), indent=1)
