#!/usr/bin/env python3
"""Learning smoke + policy-driven state capture.

(1) `DeviceTrainer` on BASELINE.json config 2 (3D_Hopper++, three variants) for a wall-clock budget: the reference's
    schedule (trainer.py:143-286: warm-up with uniform actions, collection rounds, per_morph_iter TD3 updates per
    morphology after every round).  Writes gpurun_out/learning_curve.json: train return / episode length per round
    against environment steps, solver diagnostics (dropped rows, block-pivot failures, slab solves), range events.
(2) States along LONG, policy-driven episodes of every family, for the teacher-forced parity test
    tests/test_policy_states_gpu.py: hoppers under the policy trained in (1) (no exploration noise), walkers / humanoids /
    cheetahs under a joint-space PD controller that holds the reset pose (plus small exploration noise) -- snapshots of
    `sgrl_get_records` every 50 steps of environments whose episode has lasted at least 50 steps.  Writes
    gpurun_out/policy_states.npz (copied to tests/golden/ by hand).

usage: learn_curve.py [train_seconds=480] [envs_per_morph=64] [seed=3] [hopper|walker]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from sgrl_amd import mjcf
from sgrl_amd.rollout import Rollout
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer

OUT = os.path.join(REPO, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 480.0
per = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
family = sys.argv[4] if len(sys.argv) > 4 else "hopper"      # "walker" / "humanoid" / "cheetah" / "cwhh" (config 5): training only
HOPPERS = ["3d_hopper_3_shin", "3d_hopper_4_lower_shin", "3d_hopper_5_full"]


def diag(env):
    cnt = env.get_counters()
    return {"envs_with_dropped_rows": int((cnt[:, 2] > 0).sum())}


def train():
    args = default_train_args()
    held = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
            "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
    if family == "hopper":
        names = HOPPERS
    elif family == "cwhh":          # BASELINE.json config 5: every training morphology of the four families (23)
        names = sorted(n for n in mjcf.list_assets() if n not in held)
    else:
        names = sorted(n for n in mjcf.list_assets() if family in n)
    tr = DeviceTrainer(names, per, args=args, seed=seed, device="cuda:0", max_buffer_size=400000, graph_updates=True)
    env = tr.ro.env
    curve = []
    t0 = time.time()
    # random-policy level first: warm-up rounds with uniform actions (trainer.py:90-138); their returns are the baseline
    rand_returns, rand_lengths = [], []
    steps = 0
    while steps < 400:
        steps += 1
        if tr.collect_step(random_actions=True):
            rand_returns.append(tr.sink.collector.episode_reward.mean().item())
            rand_lengths.append(tr.sink.collector.episode_timesteps.float().mean().item())
            tr.begin_round()
    pivot_fail = slab = 0
    rnd = 0
    while time.time() - t0 < budget:
        s = tr.train_round()
        rnd += 1
        c = env.get_counters()
        pivot_fail += int(((c[:, 3] >> 8) & 0xFF).sum())
        slab += int((c[:, 3] >> 16).sum())
        s.update({"round": rnd, "wall_s": round(time.time() - t0, 1), "envs_with_dropped_rows": int((c[:, 2] > 0).sum())})
        curve.append(s)
        if rnd % 5 == 0:
            print("round %d: return %.2f length %.1f iters %d wall %.0f s" % (rnd, s["performance/train_return"],
                  s["performance/train_length"], s["per_morph_iter"], s["wall_s"]), flush=True)
    out = {"seed": seed, "config": "BASELINE.json config %s (%s) x %d envs" % ({"hopper": "2: 3D_Hopper++", "walker": "3: 3D_Walker++", "humanoid": "4: 3D_Humanoid++", "cwhh": "5: 3D_CWHH++"}.get(family, family), ", ".join(names), per),
           "schedule": "reference trainer.py:143-286 (per_morph_iter updates per morphology per round, batch %d (agent_batch_size, reference configs/default.py:61), lr 1e-4, expl_noise 0.126)" % tr.batch_size,
           "random_policy": {"train_return_mean": float(np.mean(rand_returns)) if rand_returns else None,
                             "train_length_mean": float(np.mean(rand_lengths)) if rand_lengths else None, "rounds": len(rand_returns)},
           "rounds": curve, "block_pivot_failures_last_step_sum": pivot_fail, "hbm_slab_solves_last_step_sum": slab,
           "summary": {"first5_return": float(np.mean([r["performance/train_return"] for r in curve[:5]])) if curve else None,
                       "last5_return": float(np.mean([r["performance/train_return"] for r in curve[-5:]])) if curve else None,
                       "first5_length": float(np.mean([r["performance/train_length"] for r in curve[:5]])) if curve else None,
                       "last5_length": float(np.mean([r["performance/train_length"] for r in curve[-5:]])) if curve else None}}
    json.dump(out, open(os.path.join(OUT, "learning_curve.json"), "w"), indent=1)
    print(json.dumps(out["random_policy"]), json.dumps(out["summary"]), flush=True)
    return tr


def pd_actions(ro, q_ref, q_prev, kp=4.0, kd=0.3, noise=0.1):
    """Joint-space PD about the reset pose: action slot 3 l + k drives joint k of limb l, whose angle is observation
    41 l + 24 + k (reference <env>.py:116-140; wrappers.py:30-46)."""
    env = ro.env
    L = env.obs_max_len // 41
    q = env.obs.view(env.num_envs, L, 41)[:, :, 24:27]
    a = kp * (q_ref - q) - kd * (q - q_prev) / 0.008
    a = a.reshape(env.num_envs, 3 * L)
    a = a + noise * torch.randn(a.shape, device=a.device, generator=ro.gen)
    return (a.clamp_(-1, 1) * ro.act_mask).contiguous(), q.clone()


def capture(names, driver, policy=None, steps=600, per_morph=8, tag=""):
    ro = Rollout(names, per_morph, policy=policy, seed=17, device="cuda:0")
    env = ro.env
    ro.reset()
    L = env.obs_max_len // 41
    q_ref = env.obs.view(env.num_envs, L, 41)[:, :, 24:27].clone()
    q_prev = q_ref.clone()
    recs, cnts, morphs, acts = [], [], [], []
    longest = np.zeros(env.num_envs, dtype=np.int64)
    for t in range(steps):
        if driver == "policy":
            a = ro.policy_forward().clone()
            a = (a * ro.act_mask).contiguous()
        else:
            a, q_prev = pd_actions(ro, q_ref, q_prev)
        if t % 50 == 49:
            torch.cuda.synchronize()
            rec, cnt = env.get_records()
            for i in range(env.num_envs):
                if cnt[i, 0] >= 50:
                    recs.append(rec[i].copy()); cnts.append(cnt[i].copy()); morphs.append(names[env.env_morph[i]])
                    acts.append(a[i].cpu().numpy().copy())
        obs, rew, done, dist = ro.step(a)
        if driver != "policy":
            # environments that were reset take the new pose as their reference
            d = done.to(torch.bool)
            if bool(d.any()):
                qn = env.obs.view(env.num_envs, L, 41)[:, :, 24:27]
                q_ref = torch.where(d[:, None, None], qn, q_ref)
                q_prev = torch.where(d[:, None, None], qn, q_prev)
        if t % 50 == 49:
            c = env.get_counters()
            longest = np.maximum(longest, c[:, 0])
    c = env.get_counters()
    print("%s %s: %d states captured, longest episode so far per env: median %d max %d, dropped-row envs %d" % (
        tag, driver, len(recs), int(np.median(longest)), int(longest.max()), int((c[:, 2] > 0).sum())), flush=True)
    return recs, cnts, morphs, acts, env.stride, env.action_max_len


def main():
    tr = train()
    if family != "hopper":
        return
    torch.save({k: v.detach().cpu() for k, v in tr.agent.actor.state_dict().items()}, os.path.join(OUT, "hopper_actor.pt"))
    A = mjcf.list_assets()
    fam = {"hopper": sorted(n for n in A if "hopper" in n), "walker": sorted(n for n in A if "walker" in n),
           "humanoid": sorted(n for n in A if "humanoid" in n), "cheetah": sorted(n for n in A if "cheetah" in n)}
    store = {}
    rng = np.random.RandomState(0)
    for f, names in fam.items():
        sets = [capture(names, "pd", tag=f)]
        if f == "hopper":
            sets.append(capture(names, "policy", policy=tr.agent.actor, tag=f))
        for si, (recs, cnts, morphs, acts, stride, amax) in enumerate(sets):
            if not recs:
                continue
            # keep the states with the longest history, at most 4 per morphology and driver
            order = np.argsort([-c[0] for c in cnts])
            kept = {}
            sel = []
            for j in order:
                if kept.get(morphs[j], 0) < 4:
                    kept[morphs[j]] = kept.get(morphs[j], 0) + 1
                    sel.append(j)
            key = "%s_%s" % (f, "policy" if si == 1 else "pd")
            store[key + "_rec"] = np.stack([np.pad(recs[j], (0, 128 - recs[j].size)) for j in sel])
            store[key + "_cnt"] = np.stack([cnts[j] for j in sel])
            store[key + "_act"] = np.stack([np.pad(acts[j], (0, 64 - acts[j].size)) for j in sel]).astype(np.float32)
            store[key + "_morph"] = np.array([morphs[j] for j in sel])
    np.savez_compressed(os.path.join(OUT, "policy_states.npz"), **store)
    print("saved", {k: v.shape for k, v in store.items() if k.endswith("_rec")}, flush=True)


if __name__ == "__main__":
    main()
