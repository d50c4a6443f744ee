#!/usr/bin/env python3
"""Config 5's take-off, seed by seed, on two arithmetic arms (VERDICT r5 "next round" item 1b).

BASELINE.json config 5 (cwhh: 23 morphologies x 24 environments, full TD3 loop at the reference's schedule and batch 256)
for a fixed number of ROUNDS per cell (the arms differ 3x in seconds per update, so a wall-clock budget would compare unequal
amounts of learning):

  shipped : this repository's defaults -- own exact-f32 training products, row-scaled two-piece rollout products, hipGraph replay
  control : SGRL_TRAIN_GEMM=0 (vendor GEMMs through PyTorch), SGRL_SET_GEMM=f32 (exact-f32 rollout products), eager updates

One child process per seed, all at once on the one GPU (the update is a chain of small launches: processes overlap well; the
parent never touches the GPU).  Each child appends one JSON line per round to gpurun_out/<tag>_<arm>_s<seed>.jsonl and a
summary to ..._summary.json; the parent prints a progress line per minute and, at the end, the table.

take-off := mean train return of the last five rounds > 1.5 x the random policy's (the warm-up rounds of the same process).

usage: takeoff_table.py <arm> <rounds> <wall_cap_s> <seed> [<seed> ...]          (parent)
       takeoff_table.py cells <rounds> <wall_cap_s> <arm>:<seed> [...]             (parent, mixed arms)
       takeoff_table.py --child <arm> <rounds> <wall_cap_s> <seed> <tag>          (one cell)
"""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
ARMS = {
    "shipped": {},
    "control": {"SGRL_TRAIN_GEMM": "0", "SGRL_SET_GEMM": "f32"},
    # the bisection of the gap the two arms above showed (the control takes off within 5-9 rounds, shipped within 25-30 or not at all)
    "eager": {},                                             # own kernels, eager updates: plain Agent.update, torch's Adam
    "vendor_graphed": {"SGRL_TRAIN_GEMM": "0"},              # vendor training ops under hipGraph replay + table optimizer
    "setf32": {"SGRL_SET_GEMM": "f32"},                      # exact-f32 rollout / target products, everything else shipped
    "vendor_eager": {"SGRL_TRAIN_GEMM": "0"},                # the control with the shipped rollout products
    # the perturbation test: the same arms with every initial weight multiplied by (1 + 1e-6 N(0, 1)) -- no arithmetic changes, only
    # a rounding-sized nudge of the starting point: does an arm's take-off survive it?
    "control_p": {"SGRL_TRAIN_GEMM": "0", "SGRL_SET_GEMM": "f32", "TAKEOFF_PERTURB": "1e-6"},
    "shipped_p": {"TAKEOFF_PERTURB": "1e-6"},
    "eager_f32": {"SGRL_SET_GEMM": "f32"},                   # own training kernels, eager, exact-f32 rollout / target products
    "eager_tableopt": {},                                    # own kernels, eager, but the trainer-path keywords + table optimizer
}
EAGER = {"control", "control_p", "eager", "vendor_eager", "eager_tableopt", "eager_f32"}
PER_MORPH = 24
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}


def child(arm, rounds, cap, seed, tag):
    sys.path.insert(0, REPO)
    import numpy as np
    import torch
    from sgrl_amd import mjcf
    from sgrl_amd.td3 import default_train_args
    from sgrl_amd.train_loop import DeviceTrainer
    names = sorted(n for n in mjcf.list_assets() if n not in HELD)
    per = PER_MORPH
    if os.environ.get("TAKEOFF_FAMILY"):          # a cheaper proxy: one family (BASELINE config 2 / 3 / 4), 64 environments per morphology
        names = sorted(n for n in mjcf.list_assets() if os.environ["TAKEOFF_FAMILY"] in n and n not in HELD)
        per = int(os.environ.get("TAKEOFF_PER_MORPH", "64"))
    t0 = time.time()
    tr = DeviceTrainer(names, per, args=default_train_args(), seed=seed, device="cuda:0", max_buffer_size=100000,
                       graph_updates=(arm not in EAGER), lag_flag=os.environ.get("TAKEOFF_LAG", "0") == "1")      # the immediate round flag in every cell of the table (same collection schedule); TAKEOFF_LAG=1: the trainer's default
    if os.environ.get("TAKEOFF_PERTURB"):
        eps = float(os.environ["TAKEOFF_PERTURB"])
        g = torch.Generator(device="cuda").manual_seed(4242)
        with torch.no_grad():
            for src, tgt in ((tr.agent.actor, tr.agent.actor_target), (tr.agent.critic, tr.agent.critic_target)):
                for p, q in zip(src.parameters(), tgt.parameters()):
                    p.mul_(1.0 + eps * torch.randn(p.shape, device=p.device, generator=g))
                    q.copy_(p)
        tr.ro.weights_changed()
    if arm == "eager_tableopt":
        for opt in (tr.agent.actor_optimizer, tr.agent.critic_optimizer):
            for g in opt.param_groups:
                g["capturable"] = True
        real_update = tr.agent.update
        tr.agent.update = lambda batch, it, **kw: real_update(batch, it, lazy_stats=True, skip_unused_critic_grads=True)
    rand = []
    for _ in range(400):            # the random policy's level: warm-up rounds with uniform actions (trainer.py:90-138)
        if tr.collect_step(random_actions=True):
            rand.append(tr.sink.collector.episode_reward.mean().item())
            tr.begin_round()
    rand_mean = float(np.mean(rand)) if rand else None
    path = os.path.join(OUT, "%s_%s_s%d" % (tag, arm, seed))
    curve = []
    with open(path + ".jsonl", "w") as f:
        for rnd in range(1, rounds + 1):
            if time.time() - t0 > cap:
                break
            s = tr.train_round()
            rec = {"round": rnd, "wall_s": round(time.time() - t0, 1), "return": s["performance/train_return"],
                   "length": s["performance/train_length"], "iters": s["per_morph_iter"]}
            curve.append(rec)
            f.write(json.dumps(rec) + "\n")
            f.flush()
    last5 = float(np.mean([r["return"] for r in curve[-5:]])) if curve else None
    out = {"arm": arm, "env": ARMS[arm], "graphed_updates": arm not in EAGER, "seed": seed, "rounds_done": len(curve),
           "random_policy_return": rand_mean, "first5_return": float(np.mean([r["return"] for r in curve[:5]])) if curve else None,
           "last5_return": last5, "last5_length": float(np.mean([r["length"] for r in curve[-5:]])) if curve else None,
           "took_off": bool(curve and rand_mean and last5 > 1.5 * rand_mean), "wall_s": round(time.time() - t0, 1),
           "updates": int(tr._updates), "device": torch.cuda.get_device_name(0)}
    json.dump(out, open(path + "_summary.json", "w"), indent=1)
    print(json.dumps(out), flush=True)


def parent(arm, rounds, cap, seeds, tag=os.environ.get("TAKEOFF_TAG", "r6_takeoff")):
    """arm: one arm for every seed, or "cells" with seeds given as arm:seed pairs (at most six processes may use the GPU)."""
    os.makedirs(OUT, exist_ok=True)
    cells = [(arm, int(s)) for s in seeds] if arm != "cells" else [(c.split(":")[0], int(c.split(":")[1])) for c in seeds]
    assert len(cells) <= 6
    procs = []
    for arm, s in cells:
        env = dict(os.environ)
        env.update(ARMS[arm])
        log = open(os.path.join(OUT, "%s_%s_s%d.log" % (tag, arm, s)), "w")
        procs.append(((arm, s), subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", arm, str(rounds), str(cap), str(s), tag],
                                                 env=env, stdout=log, stderr=subprocess.STDOUT), log))
    t0 = time.time()
    while any(p.poll() is None for _, p, _ in procs):
        time.sleep(60)
        state = []
        for (arm, s), p, _ in procs:
            path = os.path.join(OUT, "%s_%s_s%d.jsonl" % (tag, arm, s))
            last = None
            if os.path.exists(path):
                lines = open(path).read().strip().splitlines()
                last = json.loads(lines[-1]) if lines else None
            state.append("%s/s%d:%s" % (arm[:8], s, "r%d %.1f" % (last["round"], last["return"]) if last else "-"))
        print("[%4.0f s] %s" % (time.time() - t0, "  ".join(state)), flush=True)
    rc = 0
    for (arm, s), p, log in procs:
        log.close()
        rc |= p.returncode
        sp = os.path.join(OUT, "%s_%s_s%d_summary.json" % (tag, arm, s))
        if os.path.exists(sp):
            o = json.load(open(sp))
            print("%-8s seed %d: rounds %2d  random %.1f  first5 %.1f  last5 %.1f  took_off %s  (%d updates, %.0f s)" % (
                arm, s, o["rounds_done"], o["random_policy_return"], o["first5_return"], o["last5_return"], o["took_off"],
                o["updates"], o["wall_s"]), flush=True)
        else:
            print("%-8s seed %d: no summary (exit code %s)" % (arm, s, p.returncode), flush=True)
    return rc


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
    else:
        sys.exit(parent(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4:]))
