"""Device-side TD3 training loop: BASELINE.json config 5 assembled end to end (e, f1, f2).

The batched counterpart of the reference's `Trainer.warmup` / `Trainer.train` (reference src/trainer.py:90-286): every
time step is ONE engine launch + ONE batched SET forward for all environments of this rank (rollout.py `Rollout`), the
per-environment bookkeeping of the reference loop runs as tensor ops (`RoundCollector`), transitions travel to the
learner rank in one gather and land in device-resident per-morphology ring buffers (`TransitionSink`,
`DeviceReplayBuffer`), and when every environment of every rank has finished its first episode of the round the learner
runs the reference's update schedule -- `per_morph_iter = sum(episode_timesteps) // num_envs` TD3 updates for each
morphology in turn (trainer.py:244-251) -- through `td3.Agent`, then all ranks reset and start the next round.

Multi-GPU: environments are sharded by rank (no collective in the step); the learner (rank `dst`) owns buffers and
optimizers; after its updates the actor's parameters are broadcast (one flat 18.9 MB `broadcast`) so that every
rank's rollout uses the new policy.  Single rank: no process group needed.
"""
import numpy as np
import torch

from . import graph as G
from .replay import DeviceReplayBuffer
from .rollout import TRAV, Rollout, TransitionSink
from .td3 import Agent, default_train_args


class DeviceTrainer(object):
    def __init__(self, env_names, envs_per_morph, args=None, seed=0, device="cuda:0", max_buffer_size=1000000,
                 batch_size=None, dst=0, graph_updates=False, tune_gemms=False, rollout=None, lag_flag=True, **env_kw):
        """graph_updates: replay the TD3 update from hipGraphs (td3.GraphedUpdates); tune_gemms: let PyTorch's TunableOp pick
        the rocBLAS / hipBLASLt algorithm per GEMM shape on first use (the defaults choose 256 x 256 tiles for the 700-row
        weight-gradient GEMMs of a 100-row update: 50 -> 38 ms per update, round 1).  batch_size: rows per TD3 update, default
        args.agent_batch_size = 256 (the reference's trainer.py:289-291).  lag_flag (GPU ranks): the round-finished flag is read one
        step late instead of synchronising every collection step (rollout.TransitionSink); False = the immediate flag."""
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.dst = dst
        self.args = args if args is not None else default_train_args()
        self.env_names = list(env_names)
        torch.manual_seed(seed)               # same initial weights on every rank
        self.agent = Agent(self.args, device=device)
        if tune_gemms:
            torch.cuda.tunable.enable(True)
        # `rollout`: a ready-made driver with the Rollout surface (env, graph_dicts, reset / step / random_actions / policy_forward /
        # add_exploration_noise, actions) or a callable (policy, rank) -> driver; the engine-backed Rollout by default.  The
        # multi-rank control flow of this class (round schedule, counters, actor broadcast) is covered on CPU ranks by
        # tests/test_trainer_multirank.py with a scripted driver injected here -- the product itself never builds one.
        if rollout is not None:
            self.ro = rollout(self.agent.actor, self.rank) if callable(rollout) else rollout
        else:
            # the actor's parameters change only in update_after_round (updates on the learner, the broadcast on the others): the
            # rollout holds its packed weights across a collection round
            self.ro = Rollout(self.env_names, envs_per_morph, policy=self.agent.actor, seed=seed, device=device, rank=self.rank,
                              max_episode_steps=self.args.max_episode_steps, hold_weights=True, **env_kw)
        env = self.ro.env
        self.device = env.device
        self.graph_dicts = self.ro.graph_dicts
        self.is_learner = self.rank == dst
        # rows per TD3 update: the reference's trainer samples `agent_batch_size` (configs/default.py:61 = 256, main.py:164-171,
        # trainer.py:289-291); args.batch_size (100) is only the policies' unused constructor argument
        self.batch_size = int(batch_size if batch_size is not None else getattr(self.args, "agent_batch_size", 256))
        self.buffers = None
        if self.is_learner:     # one ring buffer per morphology (reference main.py:141-155), rows cut to 41 L / 3 L
            self.buffers = [DeviceReplayBuffer(41 * L, 3 * L, max_buffer_size, device=self.device) for L in env.num_limbs]
        self.sink = TransitionSink(env.env_morph, env.num_limbs, env.obs_max_len, env.action_max_len,
                                   max_episode_steps=self.args.max_episode_steps, device=self.device, buffers=self.buffers,
                                   dst=dst, lag_flag=lag_flag)
        self.prev_obs = torch.zeros_like(env.obs)
        self.num_envs_global = env.num_envs * self.world
        self._updates = 0            # TD3 updates so far (the reference adds them to tot_env_steps: trainer.py:250)
        self._tot_synced = 0         # non-learner ranks: the learner's count as of the last round end
        self._tot_base = 0           # count restored from a snapshot (tot_env_steps setter)
        self.rounds = 0
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) * 7919 + 13)
        self.last_losses = {}
        self.graphed = None
        if graph_updates and self.is_learner:
            from .td3 import GraphedUpdates
            self.graphed = GraphedUpdates(self.agent, self.batch_size)
        self.begin_round()

    @property
    def tot_env_steps(self):
        """Stored transitions + updates (trainer.py:229, 250).  The learner's count: reading it there folds the device-side
        ingest counters in (one synchronisation -- nothing in the per-step path reads it); the other ranks hold the value the
        learner broadcast at the last round end."""
        if self.is_learner:
            return self._tot_base + self.sink.stored + self._updates
        return self._tot_synced

    @tot_env_steps.setter
    def tot_env_steps(self, v):
        """Resume from a snapshot (snapshot.load_snapshot returns the count, reference common/trainer.py:296-322): the restored
        value becomes the base the live counters are added to."""
        if self.is_learner:
            self._tot_base = int(v) - (self.sink.stored + self._updates)
        self._tot_synced = int(v)

    def sync_step_count(self):
        """Every rank learns the learner's count (called at the round ends and after the warm-up: rank-local logic that reads the
        count before the first update then agrees with the learner)."""
        if self.world > 1:
            t = torch.tensor([self.tot_env_steps if self.is_learner else 0], dtype=torch.long, device=self.device)
            self.dist.broadcast(t, src=self.dst)
            self._tot_synced = int(t.item())

    # ---- collection ----------------------------------------------------------------------------------
    def begin_round(self):
        """`obs_list = envs.reset()` + fresh done / timestep lists (trainer.py:155-160, 268-275)."""
        self.ro.reset()
        self.sink.begin_round()

    def collect_step(self, random_actions=False):
        """One time step of every environment + replay push.  Returns True when the collection round is complete."""
        env = self.ro.env
        self.prev_obs.copy_(env.obs)
        if random_actions:                       # Trainer.warmup: uniform actions (trainer.py:95-102)
            a = self.ro.random_actions()
        else:                                    # select_action + exploration noise (trainer.py:173-196)
            a = self.ro.policy_forward(self.prev_obs)
            if self.args.expl_noise != 0:
                a = self.ro.add_exploration_noise(a, self.args.expl_noise)
            self.ro.actions.copy_(a)
            a = self.ro.actions
        obs, rew, done, _ = self.ro.step(a)
        return self.sink.push(self.prev_obs, a, obs, rew, done)    # the round-finished flag (GPU: read one step late, no stall)

    def warmup(self, timesteps):
        """reference Trainer.warmup (trainer.py:90-138): `timesteps` batched steps of uniform random actions; finished
        rounds only reset the environments (no updates)."""
        for _ in range(int(timesteps)):
            if self.collect_step(random_actions=True):
                self.begin_round()
        self.sync_step_count()

    # ---- learning --------------------------------------------------------------------------------------
    def update_after_round(self, max_iters=None):
        """The update schedule of trainer.py:240-251 on the learner, then the weight broadcast.  Returns per_morph_iter."""
        per_morph_iter = self.sink.total_episode_timesteps() // self.num_envs_global
        if max_iters is not None:
            per_morph_iter = min(per_morph_iter, int(max_iters))
        if self.is_learner:
            self.agent.models2train()
            start = [0] * len(self.env_names)
            if self.graphed is not None:
                # every morphology runs eagerly before any graph bakes a pointer (capture protocol of td3.GraphedUpdates): the
                # FIRST iterations of its schedule serve as those eager runs -- same number of updates as the reference's
                # schedule, in the first such round the morphologies' first two iterations come before everybody's remaining ones
                for k in range(len(self.env_names)):
                    if k not in self.graphed.warmed and self.buffers[k].max_sample_size >= self.batch_size and per_morph_iter > 0:
                        n = min(2, per_morph_iter)
                        outs = self.graphed.warm(k, self.graph_dicts[k], self.ro.env.num_limbs[k],
                                                 lambda k=k: self.buffers[k].sample(self.batch_size, generator=self.gen), iters=n)
                        self.last_losses[self.env_names[k]] = outs[-1]
                        start[k] = n
                        self._updates += n
            for k, name in enumerate(self.env_names):
                self.agent.change_morphology(self.graph_dicts[k])
                for it in range(start[k], per_morph_iter):
                    batch = self.buffers[k].sample(self.batch_size, generator=self.gen)
                    if self.graphed is not None and k in self.graphed.warmed:
                        self.last_losses[name] = self.graphed.update(k, self.graph_dicts[k], self.ro.env.num_limbs[k], batch, it)
                    else:
                        self.last_losses[name] = self.agent.update(batch, it)
                    self._updates += 1                         # the reference counts updates too (trainer.py:250)
            self.agent.models2eval()
        self.sync_step_count()               # every rank reports the learner's step count (checkpoints, stopping rule)
        self.broadcast_actor()
        if hasattr(self.ro, "weights_changed"):
            self.ro.weights_changed()          # with the weights every rank rolls out next
        self.rounds += 1
        return per_morph_iter

    def broadcast_actor(self):
        if self.world == 1:
            return
        params = list(self.agent.actor.parameters())
        flat = torch.cat([p.data.reshape(-1) for p in params])
        self.dist.broadcast(flat, src=self.dst)
        if not self.is_learner:
            off = 0
            for p in params:
                n = p.numel()
                p.data.copy_(flat[off:off + n].view_as(p))     # in place: the HIP actor reads the live storage
                off += n

    def train_round(self, max_steps=None, max_iters=None):
        """Collect until every environment has finished one episode (or max_steps), update, reset.  Returns a summary."""
        steps = 0
        while True:
            steps += 1
            if self.collect_step() or (max_steps is not None and steps >= max_steps):
                break
        returns = self.sink.collector.episode_reward.mean().item()
        lengths = self.sink.collector.episode_timesteps.float().mean().item()
        iters = self.update_after_round(max_iters=max_iters)
        self.begin_round()
        return {"steps": steps, "per_morph_iter": iters, "performance/train_return": returns,
                "performance/train_length": lengths, "tot_env_steps": self.tot_env_steps}
