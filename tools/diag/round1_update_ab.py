#!/usr/bin/env python3
"""Round 1 of a config-5 run collected ONCE; then the round's TD3 updates replayed from the same weights, optimizer state, replay rows
and sampling stream in several variants -- which ingredient of the update decides whether the actor comes out saturated?
  hip_targets      : the no-grad target networks on the HIP forwards (what the trainer runs)
  torch_targets    : the same update with the target networks on PyTorch's path (use_hip = False)
Prints, per variant, the share of saturated deterministic actions on the next round's first observations and the distance of the
resulting actor from the hip_targets one.  usage: round1_update_ab.py [seed=3]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import mjcf, set_policy
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD)
tr = DeviceTrainer(names, 24, args=default_train_args(), seed=seed, device="cuda:0", max_buffer_size=100000, graph_updates=False, lag_flag=False)
for _ in range(400):
    if tr.collect_step(random_actions=True):
        tr.begin_round()
while not tr.collect_step():
    pass
print("round 1 collected: return %.2f" % tr.sink.collector.episode_reward.mean().item(), flush=True)
sd0 = {k: v.detach().clone() for k, v in tr.agent.state_dict().items()}
gen0 = tr.gen.get_state()
cuda_rng0 = torch.cuda.get_rng_state()
iters = tr.sink.total_episode_timesteps() // tr.num_envs_global


def run(variant):
    ag = tr.agent
    ag.load_state_dict(sd0)
    ag.actor_optimizer = torch.optim.Adam(ag.actor.parameters(), lr=ag.args.lr)
    ag.critic_optimizer = torch.optim.Adam(ag.critic.parameters(), lr=ag.args.lr)
    tr.gen.set_state(gen0)
    torch.cuda.set_rng_state(cuda_rng0)
    hip = variant != "torch_targets"
    ag.actor_target.use_hip = hip
    ag.critic_target.use_hip = hip
    ag.models2train()
    stats = []
    for k, name in enumerate(names):
        ag.change_morphology(tr.graph_dicts[k])
        for it in range(iters):
            batch = tr.buffers[k].sample(tr.batch_size, generator=tr.gen)
            ag.update(batch, it)
    ag.models2eval()
    ag.actor_target.use_hip = ag.critic_target.use_hip = True
    tr.ro.weights_changed()
    tr.ro.reset()
    act = tr.ro.policy_forward(tr.ro.env.obs).clone()
    p = torch.cat([q.detach().reshape(-1) for q in ag.actor.parameters()]).double().cpu()
    return act.cpu(), p


out = {}
for v in ("hip_targets", "torch_targets", "hip_targets_again"):
    act, p = run(v)
    out[v] = (act, p)
    ref = out["hip_targets"]
    print("%-18s saturated share %.3f  mean |a| %.3f   vs hip_targets: actions max |d| %.3e mean %.3e, actor params max |d| %.3e rms %.3e" % (
        v, float((act.abs() > 0.99).float().mean()), float(act.abs().mean()), float((act - ref[0]).abs().max()), float((act - ref[0]).abs().mean()),
        float((p - ref[1]).abs().max()), float((p - ref[1]).pow(2).mean().sqrt())), flush=True)
print("SGRL_SET_GEMM=%s TWIN_TARGETS=%s iters per morphology %d" % (os.environ.get("SGRL_SET_GEMM", "(default f16x3)"), set_policy.TWIN_TARGETS, iters))
