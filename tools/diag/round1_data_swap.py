#!/usr/bin/env python3
"""Does the rollout's product form act through the DATA it collects or through the forward that evaluates the updated policy?
  collect <file> : warm-up + round 1 of a config-5 run in this process's SGRL_SET_GEMM mode; the replay rings, the sampling stream and
                   the next round's first observations go to <file>
  update <file>  : a fresh trainer of the same seed takes those rings, runs the round's TD3 updates (torch-path targets: nothing in
                   the update depends on the mode) and evaluates the resulting actor on the stored observations twice: through
                   the rollout's HIP forward (this process's mode) and through PyTorch (mode independent)
usage: round1_data_swap.py collect|update <file> [seed=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer

what, path = sys.argv[1], sys.argv[2]
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = os.environ.get("SGRL_SET_GEMM", "f16x3")
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD)
tr = DeviceTrainer(names, 24, args=default_train_args(), seed=seed, device="cuda:0", max_buffer_size=100000, graph_updates=False, lag_flag=False)
if what == "collect":
    for _ in range(400):
        if tr.collect_step(random_actions=True):
            tr.begin_round()
    while not tr.collect_step():
        pass
    iters = tr.sink.total_episode_timesteps() // tr.num_envs_global
    tr.ro.reset()
    bufs = []
    for b in tr.buffers:
        n = b.max_sample_size
        bufs.append({"n": n, "curr": b.curr, "obs": b.obs_buffer[:n].cpu(), "act": b.action_buffer[:n].cpu(), "nxt": b.next_obs_buffer[:n].cpu(),
                     "rew": b.reward_buffer[:n].cpu(), "done": b.done_buffer[:n].cpu()})
    torch.save({"bufs": bufs, "gen": tr.gen.get_state(), "iters": iters, "obs": tr.ro.env.obs.cpu(), "mode": mode}, path)
    acts = torch.cat([b["act"].reshape(-1) for b in bufs])
    rews = torch.cat([b["rew"] for b in bufs])
    print("collected in mode %s: %d rows, iters %d; stored actions mean %.4f std %.4f |a|>0.99 share %.4f; reward mean %.4f std %.4f; done share %.4f" % (
        mode, sum(b["n"] for b in bufs), iters, float(acts.mean()), float(acts.std()), float((acts.abs() > 0.99).float().mean()), float(rews.mean()), float(rews.std()),
        float(torch.cat([b["done"] for b in bufs]).mean())), flush=True)
else:
    d = torch.load(path)
    for b, s in zip(tr.buffers, d["bufs"]):
        n = s["n"]
        b.obs_buffer[:n] = s["obs"].cuda(); b.action_buffer[:n] = s["act"].cuda(); b.next_obs_buffer[:n] = s["nxt"].cuda()
        b.reward_buffer[:n] = s["rew"].cuda(); b.done_buffer[:n] = s["done"].cuda()
        b.curr, b.max_sample_size = s["curr"], n
    tr.gen.set_state(d["gen"])
    ag = tr.agent
    ag.actor_target.use_hip = ag.critic_target.use_hip = False
    ag.models2train()
    for k, name in enumerate(names):
        ag.change_morphology(tr.graph_dicts[k])
        for it in range(d["iters"]):
            ag.update(tr.buffers[k].sample(tr.batch_size, generator=tr.gen), it)
    ag.models2eval()
    tr.ro.weights_changed()
    obs = d["obs"].cuda()
    a_hip = tr.ro.policy_forward(obs).clone()
    a_torch = torch.zeros_like(a_hip)
    env = tr.ro.env
    with torch.no_grad():
        for k, sl in enumerate(env.morph_slices):
            L = env.num_limbs[k]
            x = obs[sl, :41 * L].reshape(sl.stop - sl.start, L, 41)
            a_torch[sl, :3 * L] = (ag.actor.max_action * torch.tanh(ag.actor.actor(x, tr.graph_dicts[k], False))).reshape(sl.stop - sl.start, 3 * L)
    m = tr.ro.act_mask > 0
    print("data collected in mode %s, updates + HIP evaluation in mode %s: mean |a| HIP %.4f torch %.4f; saturated share HIP %.3f torch %.3f; max |HIP - torch| %.2e" % (
        d["mode"], mode, float(a_hip[m].abs().mean()), float(a_torch[m].abs().mean()), float((a_hip[m].abs() > 0.99).float().mean()),
        float((a_torch[m].abs() > 0.99).float().mean()), float((a_hip - a_torch).abs().max())), flush=True)
