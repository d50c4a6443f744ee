"""Batched rollout driver: the build's counterpart of the reference's per-env Python loop
(`for i in range(num_envs_train): change_morphology; select_action; ...; envs.step`, reference trainer.py:173-236
and the random-action warm-up loop trainer.py:90-138), with the same per-environment semantics but ONE engine
launch and ONE batched SET forward per time step, plus the replay push as a single gather to the learner rank.
"""
import ctypes

import numpy as np
import torch

from . import graph as G
from .replay import DeviceReplayBuffer
from .set_hip import HipSetActor
from .vec_env import BatchedModularVecEnv

TRAV = ["pre", "inlcrs", "postlcrs"]
FUSED_RECORD = True      # False (tests, A/B): RoundCollector's bookkeeping as ~15 tensor operations on the GPU too.  Same rows, same flags; the config-5
                         # collection step is the same 9.68 ms either way (GPU-bound), the one launch saves 95 us of host time per step
FUSED_INGEST = True      # False (tests): the learner writes a gathered block morphology by morphology with indexed copies (the CPU path) on the GPU too


class Rollout(object):
    """Environments of one rank + the shared SET actor."""

    def __init__(self, env_names, envs_per_morph, policy=None, seed=0, device="cuda:0", rank=0, hold_weights=False, **env_kw):
        counts = [envs_per_morph] * len(env_names) if np.isscalar(envs_per_morph) else list(envs_per_morph)
        n_local = int(sum(counts))
        self.env = BatchedModularVecEnv(env_names, counts, seed=seed, device=device, env_id_base=rank * n_local, **env_kw)
        self.device = self.env.device
        self.policy = policy
        self.actor = None
        if policy is not None:
            self.graph_dicts = [G.getGraphDict(m.parents, TRAV, [], device=self.device) for m in self.env.models]
            self.actor = HipSetActor(policy, device=self.device)
            self.actor.configure(self.graph_dicts, counts)
            if hold_weights:      # the owner of the loop says when the policy's parameters change (weights_changed)
                self.actor.hold_weights(True)
        self.holds_weights = bool(hold_weights) and policy is not None
        self._weights_seen = self._weights_fingerprint()
        n, amax = self.env.num_envs, self.env.action_max_len
        self.actions = torch.zeros((n, amax), dtype=torch.float32, device=self.device)
        self.policy_actions = torch.zeros((n, amax), dtype=torch.float32, device=self.device)
        # padding mask: slots beyond 3*L of each morphology stay zero (reference trainer.py:191-195)
        self.act_mask = torch.zeros((n, amax), dtype=torch.float32, device=self.device)
        for k, sl in enumerate(self.env.morph_slices):
            self.act_mask[sl, :3 * self.env.num_limbs[k]] = 1.0
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) * 1000003 + rank)
        self.obs = None

    def _weights_fingerprint(self):
        """(storage address, in-place version counter) of every parameter: changes with optimizer steps, load_state_dict, copy_,
        soft updates, broadcasts into the parameters and re-assigned .data -- everything PyTorch itself can see.  Host-side only."""
        if self.policy is None:
            return None
        return tuple((p.data_ptr(), p._version) for p in self.policy.parameters())

    def weights_changed(self):
        """The policy's parameters were just updated (optimizer steps, soft updates, a broadcast): a rollout that holds its actor's
        packed weights (hold_weights=True) packs again on its next forward.  policy_forward() also notices by itself when a
        parameter's version counter or storage moved (load_state_dict, a snapshot restore, a manual update, and this library's
        own raw-pointer writers -- td3.clip_and_step, the table soft update, a GraphedUpdates replay -- which bump the counters
        themselves: td3._touched), so a forgotten call costs nothing but the check; only a foreign kernel writing into a
        parameter's memory behind PyTorch's back needs this call."""
        if self.holds_weights:
            self.actor.hold_weights(True)
            self._weights_seen = self._weights_fingerprint()

    def reset(self):
        self.obs = self.env.reset_device()
        return self.obs

    def random_actions(self):
        """i.i.d. U(-1, 1) per slot (reference trainer.py:95-102), zero in the padding slots."""
        self.actions.uniform_(-1.0, 1.0, generator=self.gen)
        self.actions.mul_(self.act_mask)
        return self.actions

    def policy_forward(self, obs=None):
        """One batched SET forward over every environment (replaces n x Agent.select_action, reference agent.py:189-198)."""
        if self.holds_weights and self._weights_fingerprint() != self._weights_seen:
            self.weights_changed()          # the held pack is stale: pack again, keep holding
        return self.actor.forward_batch(self.env.obs if obs is None else obs, out=self.policy_actions,
                                        act_ld=self.env.action_max_len)

    def add_exploration_noise(self, actions, expl_noise=0.126):
        """a + N(0, expl_noise) clipped to the action range (reference trainer.py:184-189)."""
        noise = torch.randn(actions.shape, device=self.device, generator=self.gen) * expl_noise
        return ((actions + noise).clamp_(-1.0, 1.0)) * self.act_mask

    def step(self, actions):
        return self.env.step_device(actions)


class ReplayGather(object):
    """The replay-buffer push as ONE collective: every rank contributes a fixed-size block of transition rows
        obs[obs_max_len] | action[action_max_len] | next_obs[obs_max_len] | reward | done | store | morph_id
    -- the arguments of ReplayBuffer.add_transition (reference common/buffer.py:75-84) for each of its environments, the
    flag saying whether the reference's loop would store the row at all (first episode of the round only, reference
    trainer.py:218) and the morphology the row belongs to (the reference keeps one buffer per morphology, main.py:141-155)
    -- and the learner rank receives them all with a single gather (RCCL over xGMI when the backend is 'nccl'; 'gloo'
    on CPU for tests).  Rows are padded to the widest morphology so that every rank sends the same number of bytes."""

    EXTRA = 4     # reward, done, store, morph_id

    def __init__(self, n_env_local, obs_max_len, action_max_len, device, dst=0, depth=1):
        """depth > 1: that many send blocks (and receive lists on the learner) used in turn, so that `push(wait=False)` can
        leave a gather in flight while the next steps run; a block is waited for only when its turn comes again."""
        import torch.distributed as dist
        self.dist = dist
        self.dst = dst
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.row = 2 * obs_max_len + action_max_len + self.EXTRA
        self.o, self.a = obs_max_len, action_max_len
        self.depth = int(depth)
        assert self.depth >= 1
        self.blocks = [torch.zeros((n_env_local, self.row), dtype=torch.float32, device=device) for _ in range(self.depth)]
        for b in self.blocks:
            b[:, 2 * obs_max_len + action_max_len + 2] = 1.0      # store flag defaults to "keep"
        self.block = self.blocks[0]        # the block packed last
        self.recvs = [None] * self.depth
        self.recv_flats = [None] * self.depth
        if self.rank == dst:
            # the learner receives into ONE contiguous [world * n, row] tensor per slot (rank order = global environment order);
            # the gather's per-rank list are views of it, so the ingest is one pair of launches over the whole thing whatever N
            self.recv_flats = [torch.zeros((self.world * n_env_local, self.row), dtype=torch.float32, device=device)
                               for _ in range(self.depth)]
            self.recvs = [list(f.split(n_env_local, dim=0)) for f in self.recv_flats]
        self.recv = self.recvs[0]
        self.recv_flat = self.recv_flats[0]
        self._works = [None] * self.depth
        self._k = 0

    def _wait(self, slot):
        if self._works[slot] is not None:
            self._works[slot].wait()       # nccl: the current stream waits for the collective; gloo: the host does
            self._works[slot] = None

    def drain(self):
        """Wait for every gather still in flight (call before reading the received blocks / before a timing barrier)."""
        for slot in range(self.depth):
            self._wait(slot)

    def _take_slot(self):
        slot = self._k % self.depth
        self._wait(slot)                   # the gather that last used this block (depth pushes ago) must be done
        self.block = self.blocks[slot]
        self.recv = self.recvs[slot]
        self.recv_flat = self.recv_flats[slot]
        return self.block

    def _write(self, b, obs, action, next_obs, reward, done, store, morph_id):
        """Columns of the block from their sources (None = leave as is).  Device blocks: ONE launch of the library's
        k_pack_transitions (include/sgrl.h sgrl_pack_transitions); host blocks (gloo ranks in the CPU tests): slice copies."""
        o, a = self.o, self.a
        if b.is_cuda:
            import ctypes
            from . import _lib
            L = _lib.lib()
            keep = []

            def rows(t, width):
                if t is None:
                    return None, 0
                if t.dtype != torch.float32 or t.stride(-1) != 1:
                    t = t.to(torch.float32).contiguous()
                assert t.shape[0] == b.shape[0] and t.shape[1] >= width and t.device == b.device
                keep.append(t)
                return ctypes.c_void_p(t.data_ptr()), int(t.stride(0))

            def col(t, dtype):
                if t is None:
                    return None
                if t.dtype == torch.bool and dtype == torch.uint8:
                    t = t.contiguous().view(torch.uint8)
                elif t.dtype != dtype or not t.is_contiguous():
                    t = t.to(dtype).contiguous()
                assert t.numel() == b.shape[0] and t.device == b.device
                keep.append(t)
                return ctypes.c_void_p(t.data_ptr())
            po, ldo = rows(obs, o)
            pa, lda = rows(action, a)
            pn, ldn = rows(next_obs, o)
            done_u8 = done is not None and done.dtype in (torch.bool, torch.uint8)
            _lib.check(L.sgrl_pack_transitions(po, ldo, pa, lda, pn, ldn, col(reward, torch.float32),
                                               None if done_u8 else col(done, torch.float32),
                                               col(done, torch.uint8) if done_u8 else None, col(store, torch.uint8),
                                               col(morph_id, torch.int64), ctypes.c_void_p(b.data_ptr()), int(b.shape[0]), o, a,
                                               ctypes.c_void_p(torch.cuda.current_stream(b.device).cuda_stream)),
                       "sgrl_pack_transitions")
            return
        if obs is not None:
            b[:, :o] = obs[:, :o]
        if action is not None:
            b[:, o:o + a] = action[:, :a]
        if next_obs is not None:
            b[:, o + a:2 * o + a] = next_obs[:, :o]
        if reward is not None:
            b[:, 2 * o + a] = reward
        if done is not None:
            b[:, 2 * o + a + 1] = done.to(torch.float32)
        if store is not None:
            b[:, 2 * o + a + 2] = store.to(torch.float32)
        if morph_id is not None:
            b[:, 2 * o + a + 3] = morph_id.to(torch.float32)     # small integers: exact in float32

    def stage_obs(self, obs):
        """First half of a row, written BEFORE the step overwrites the observation buffer in place (saves the copy of the
        previous observations); the following `pack(None, ...)` completes the same block."""
        b = self._take_slot()
        self._write(b, obs, None, None, None, None, None, None)
        self._staged = True
        return b

    def pack(self, obs, action, next_obs, reward, done, store=None, morph_id=None):
        if getattr(self, "_staged", False):
            assert obs is None, "stage_obs() already wrote this block's observation columns"
            self._staged = False
            b = self.block
        else:
            b = self._take_slot()
        self._write(b, obs, action, next_obs, reward, done, store, morph_id)
        return b

    def push(self, wait=True):
        """Gather the block packed last.  Returns the list of per-rank blocks on the learner rank, None elsewhere.  wait=False
        leaves the gather in flight (the returned blocks are complete after `drain()`, or once `depth` further pushes have been
        packed): the next step's kernels overlap the transfer."""
        slot = self._k % self.depth
        self._k += 1
        if self.world == 1 and not self.dist.is_initialized():
            return [self.block]            # no process group: the learner's own block is the whole gather
        work = self.dist.gather(self.block, self.recv if self.rank == self.dst else None, dst=self.dst, async_op=True)
        if wait:
            work.wait()
        else:
            self._works[slot] = work
        return self.recv

    def unpack(self, block):
        """(obs, action, next_obs, reward, done, store bool, morph_id long) views / columns of a block."""
        o, a = self.o, self.a
        return (block[:, :o], block[:, o:o + a], block[:, o + a:2 * o + a], block[:, 2 * o + a], block[:, 2 * o + a + 1],
                block[:, 2 * o + a + 2] > 0.5, block[:, 2 * o + a + 3].to(torch.long))

    def bytes_per_step(self):
        return self.block.numel() * 4


class RoundCollector(object):
    """Collection-round bookkeeping of the reference trainer for ALL environments at once (tensors on any device).

    The reference keeps stepping every env (auto-reset) but stores and counts only the FIRST episode of each env per
    round; a round ends when every env has finished once, then the TD3 updates run and `envs.reset()` starts the next
    round (reference src/trainer.py:155-160, 205-275; same rule in warmup, :108-138).  Per step and env i:
        done_bool = curr_done[i];  if episode_timesteps[i] + 1 == max_episode_steps: done_bool = 0, curr_done[i] = True
        if not done_list[i]:  episode_timesteps[i] += 1;  store (obs, action, next_obs, reward, done_bool);
                              done_list[i] |= curr_done[i]
    """

    def __init__(self, n_env, max_episode_steps=1000, device="cpu"):
        self.n = n_env
        self.max_episode_steps = int(max_episode_steps)
        self.device = torch.device(device)
        self.begin_round()

    def begin_round(self):
        self.done_list = torch.zeros(self.n, dtype=torch.bool, device=self.device)
        self.episode_timesteps = torch.zeros(self.n, dtype=torch.long, device=self.device)
        self.episode_reward = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self._reward_buf = torch.zeros(self.n, dtype=torch.float32, device=self.device)

    def record(self, reward, curr_done, sync=True):
        """reward float[n], curr_done bool/uint8[n] as returned by VecEnv.step.  Returns (store_mask bool[n],
        done_to_store float[n], round_finished).  Rows where store_mask is False are dropped (trainer.py:218).
        round_finished: a Python bool (ONE host synchronisation), or with sync=False the 0-dim bool tensor it would be read from
        (TransitionSink reads it a step late through pinned memory instead of stalling the step on it)."""
        if self.device.type == "cuda" and FUSED_RECORD:
            return self._record_hip(reward, curr_done, sync)
        curr_done = curr_done.to(torch.bool).clone()
        done_bool = curr_done.to(torch.float32)
        timeout = (self.episode_timesteps + 1) == self.max_episode_steps
        done_bool = torch.where(timeout, torch.zeros_like(done_bool), done_bool)
        curr_done |= timeout
        self._reward_buf += reward.to(torch.float32)
        first = curr_done & (self.episode_reward == 0)
        self.episode_reward = torch.where(first, self._reward_buf, self.episode_reward)
        self._reward_buf = torch.where(first, torch.zeros_like(self._reward_buf), self._reward_buf)
        store = ~self.done_list
        self.episode_timesteps += store.to(torch.long)
        self.done_list |= store & curr_done
        fin = self.done_list.all()
        return store, done_bool, (bool(fin) if sync else fin)

    def _record_hip(self, reward, curr_done, sync):
        """The same rule as one launch (include/sgrl.h sgrl_round_record; the tensor form above is ~15 small launches per step)."""
        from . import _lib
        L = _lib.lib()
        r = reward if (reward.dtype == torch.float32 and reward.is_contiguous()) else reward.to(torch.float32).contiguous()
        d = curr_done
        if d.dtype == torch.bool:
            d = d.contiguous().view(torch.uint8)
        elif d.dtype != torch.uint8 or not d.is_contiguous():
            d = (d != 0).contiguous().view(torch.uint8)
        if getattr(self, "_store_u8", None) is None:
            self._store_u8 = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            self._done_bool = torch.zeros(self.n, dtype=torch.float32, device=self.device)
            self._all_done = torch.zeros(1, dtype=torch.int32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(L.sgrl_round_record(p(r), p(d), p(self.done_list), p(self.episode_timesteps), p(self.episode_reward), p(self._reward_buf),
                                       p(self._store_u8), p(self._done_bool), p(self._all_done), self.n, self.max_episode_steps,
                                       ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "sgrl_round_record")
        return self._store_u8.view(torch.bool), self._done_bool, (bool(self._all_done.item()) if sync else self._all_done)

    def per_morph_iter(self):
        """Number of TD3 updates per morphology after the round (reference trainer.py:244)."""
        return int(self.episode_timesteps.sum().item()) // self.n


class TransitionSink(object):
    """Everything that happens to a batch of transitions after `VecEnv.step` (SURVEY 8 a15 + a16), for one rank:

        RoundCollector.record          which rows the reference's loop would store, with which `done` (trainer.py:205-232)
        ReplayGather.pack / push       ONE gather of the rows (+ store flag + morphology id) to the learner rank
        learner: ingest                rank by rank, env by env -- the order of the reference's `for i in range(num_envs)`
                                       loop -- every kept row goes to the replay buffer of ITS morphology, cut to that
                                       morphology's 41 L / 3 L columns (`add_transition`, common/buffer.py:75-84)
        all ranks: round_finished      `all(done_list)` over the environments of every rank (one 4-byte all-reduce)

    `buffers`: list indexed by morphology id of objects with `add_transitions(obs, action, next_obs, reward, done, mask)`
    (sgrl_amd.replay.DeviceReplayBuffer) on the learner rank, None elsewhere."""

    def __init__(self, env_morph, num_limbs, obs_max_len, action_max_len, max_episode_steps=1000, device="cpu", buffers=None,
                 dst=0, lag_flag=False):
        """lag_flag: push() answers with the round-finished flag of the PREVIOUS step, fetched through pinned memory, instead of
        synchronising the host with the device on every step (and, at N > 1, waiting for a 4-byte all-reduce).  The round then ends
        one step late; that step's rows carry store = False on every rank (every environment had finished), so buffers, episode
        statistics and update counts are exactly those of the immediate flag -- only the environments take one more, discarded,
        step before the round's reset (reference trainer.py:205-275 has no observable for it)."""
        import torch.distributed as dist
        self.dist = dist
        self.device = torch.device(device)
        self.env_morph = torch.as_tensor(env_morph, dtype=torch.long, device=self.device)
        self.num_limbs = [int(l) for l in num_limbs]
        n = int(self.env_morph.numel())
        self.collector = RoundCollector(n, max_episode_steps, device=self.device)
        self.gather = ReplayGather(n, obs_max_len, action_max_len, self.device, dst=dst)
        self.is_learner = self.gather.rank == dst
        self.buffers = buffers
        if self.is_learner and buffers is None:
            raise ValueError("the learner rank needs the per-morphology replay buffers")
        self.lag_flag = bool(lag_flag) and self.device.type == "cuda"
        self._lag = None            # (pinned int32 flag of the previous step, the event after its copy)
        self._lag_spare = []
        self._stored = 0            # host-side count, current after fold_counters()
        # rows per morphology written by the one-launch ingest since the rings' host-side pointers were last brought up to date
        # (device tensor: the ingest itself never synchronises; readers of `curr` / `max_sample_size` / `stored` fold it in)
        self._pend = None
        if self.is_learner and buffers is not None:
            for b in buffers:
                if isinstance(b, DeviceReplayBuffer):
                    b._sync_hook = self.fold_counters

    def fold_counters(self):
        """Bring the rings' host-side write pointers / fill levels and `stored` up to date with what the device-side ingest has
        written (ONE host synchronisation, when somebody looks: round end, sampling, snapshots, tests -- not per step)."""
        if self._pend is None:
            return
        pend, self._pend = self._pend, None
        counts = pend.tolist()
        for c, b in zip(counts, self.buffers):
            if c:
                b._curr = (b._curr + c) % b.max_buffer_size
                b._fill = min(b._fill + c, b.max_buffer_size)
        self._stored += int(sum(counts))
        self._pos_host = tuple(b._curr for b in self.buffers)

    @property
    def stored(self):
        """Transitions written on the learner so far (tot_env_steps bookkeeping, trainer.py:229); reading it synchronises."""
        self.fold_counters()
        return int(self._stored)

    def begin_round(self):
        self.collector.begin_round()
        if self._lag is not None:           # a flag of the round that just ended says nothing about the new one
            self._lag_spare.append(self._lag)
            self._lag = None

    def push(self, prev_obs, action, next_obs, reward, done):
        """Returns True when every environment of every rank has finished its first episode of this round (lag_flag: as of the
        previous step)."""
        store, done_bool, finished = self.collector.record(reward, done, sync=not self.lag_flag)
        self.gather.pack(prev_obs, action, next_obs, reward, done_bool, store, self.env_morph)
        blocks = self.gather.push()
        if self.is_learner:
            self.ingest(blocks)
        if self.lag_flag:
            flag = finished if finished.dtype == torch.int32 else finished.to(torch.int32).reshape(1)      # (the fused record hands out its int32 flag)
            if self.gather.world > 1:
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)      # on the stream; nobody waits for it on the host
            host, ev = self._lag_spare.pop() if self._lag_spare else (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
            host.copy_(flag, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.device))
            prev, self._lag = self._lag, (host, ev)
            if prev is None:
                return False
            prev[1].synchronize()           # recorded a whole step ago
            out = bool(int(prev[0][0]))
            self._lag_spare.append(prev)
            return out
        if self.gather.world > 1:
            flag = torch.tensor([1 if finished else 0], dtype=torch.int32, device=self.device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
            finished = bool(flag.item())
        return finished

    # ---- learner: one pair of launches per step (CUDA) -------------------------------------------------------------------
    def _ring_table(self):
        """Device array of include/sgrl.h sgrl_ring descriptors (one per morphology) + the capacities; rebuilt if a buffer's
        storage has moved."""
        key = tuple((b.obs_buffer.data_ptr(), b.action_buffer.data_ptr(), b.next_obs_buffer.data_ptr(), b.reward_buffer.data_ptr(),
                     b.done_buffer.data_ptr()) for b in self.buffers)
        if getattr(self, "_ring_key", None) != key:
            desc = np.zeros(len(self.buffers), dtype=np.dtype([("obs", "<u8"), ("action", "<u8"), ("next_obs", "<u8"), ("reward", "<u8"),
                                                               ("done", "<u8"), ("obs_dim", "<i4"), ("act_dim", "<i4")]))
            assert desc.dtype.itemsize == 48
            for k, b in enumerate(self.buffers):
                desc[k] = key[k] + (b.obs_dim, b.action_dim)
            self._ring_dev = torch.from_numpy(desc.view(np.uint8).copy()).to(self.device)
            self._ring_cap = torch.tensor([b.max_buffer_size for b in self.buffers], dtype=torch.long, device=self.device)
            self._ring_ids = torch.arange(len(self.buffers), device=self.device)
            self._ring_key = key
        return self._ring_dev

    def _ingest_block_hip(self, blk):
        """The rows of one block into their morphologies' rings with ONE launch (include/sgrl.h sgrl_ingest_rows) and NO host
        synchronisation: the rings' write pointers are mirrored on the device, the per-morphology row counts stay there too
        (`_pend`) until somebody reads the host-side pointers (fold_counters: once per round).  Slot of a stored row = write
        pointer of its ring + its rank among the stored rows of its morphology in this block (row order), modulo the capacity
        -- what add_transition row by row produces.  Returns False (nothing written) when a block has more rows than the
        smallest ring holds (it could wrap a ring onto itself): the row-by-row path handles that."""
        from . import _lib
        if int(blk.shape[0]) > min(b.max_buffer_size for b in self.buffers):
            return False
        rings = self._ring_table()
        if self._pend is None:
            host_pos = tuple(b._curr for b in self.buffers)
            if getattr(self, "_pos_host", None) != host_pos:      # first block, or the buffers were written / loaded elsewhere
                self._pos_dev = torch.tensor(host_pos, dtype=torch.long, device=self.device)
        blk = blk if blk.is_contiguous() else blk.contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L = _lib.lib()
        if len(self.buffers) <= 32:
            # slots, write pointers and the pending counts on the device (include/sgrl.h sgrl_ingest_block): three launches
            if self._pend is None:
                self._pend_buf = getattr(self, "_pend_buf", None)
                if self._pend_buf is None:
                    self._pend_buf = torch.zeros(len(self.buffers), dtype=torch.long, device=self.device)
                else:
                    self._pend_buf.zero_()
                self._pend = self._pend_buf
            need = int(L.sgrl_ingest_ws_words(int(blk.shape[0])))      # include/sgrl.h sgrl_ingest_block: workspace
            if getattr(self, "_slot_ws", None) is None or self._slot_ws.numel() < need:
                self._slot_ws = torch.empty(need, dtype=torch.long, device=self.device)      # contents do not matter
            _lib.check(L.sgrl_ingest_block(ctypes.c_void_p(blk.data_ptr()), int(blk.shape[0]), int(self.gather.o), int(self.gather.a),
                                           ctypes.c_void_p(rings.data_ptr()), len(self.buffers), ctypes.c_void_p(self._pos_dev.data_ptr()),
                                           ctypes.c_void_p(self._ring_cap.data_ptr()), ctypes.c_void_p(self._pend.data_ptr()),
                                           ctypes.c_void_p(self._slot_ws.data_ptr()), ctypes.c_void_p(stream)), "sgrl_ingest_block")
            return True
        _, _, _, _, _, store, morph = self.gather.unpack(blk)
        m = morph.clamp(0, len(self.buffers) - 1)
        hit = (self._ring_ids.unsqueeze(1) == m.unsqueeze(0)) & store.unsqueeze(0)              # [morphologies, rows]
        cs = torch.cumsum(hit, dim=1, dtype=torch.int64)                                          # scans along the contiguous axis
        counts_dev = cs[:, -1]
        rank = cs.gather(0, m.unsqueeze(0)).squeeze(0) - 1
        slot = torch.where(store, (self._pos_dev[m] + rank) % self._ring_cap[m], rank.new_full((), -1))
        _lib.check(L.sgrl_ingest_rows(ctypes.c_void_p(blk.data_ptr()), int(blk.shape[0]), int(self.gather.o), int(self.gather.a),
                                      ctypes.c_void_p(slot.data_ptr()), ctypes.c_void_p(rings.data_ptr()), len(self.buffers),
                                      ctypes.c_void_p(stream)), "sgrl_ingest_rows")
        self._pos_dev = (self._pos_dev + counts_dev) % self._ring_cap      # the device copy follows without an upload
        self._pend = counts_dev if self._pend is None else self._pend + counts_dev
        return True

    def ingest(self, blocks):
        fast = FUSED_INGEST and self.device.type == "cuda" and all(isinstance(b, DeviceReplayBuffer) for b in self.buffers)
        # the learner's whole gather at once: the per-rank blocks are views of one contiguous tensor in rank order (ReplayGather),
        # so N ranks cost three launches per step, not 2 N (k_ingest_keys + k_ingest_slots + k_ingest_rows)
        flat = getattr(blocks[0], "_base", None) if len(blocks) > 1 else None      # the tensor the per-rank views were split from
        if (fast and flat is not None and flat.dim() == 2 and flat.is_contiguous() and blocks[0].data_ptr() == flat.data_ptr()
                and all(getattr(b, "_base", None) is flat for b in blocks)
                and all(b.data_ptr() == flat.data_ptr() + 4 * flat.shape[1] * sum(int(x.shape[0]) for x in blocks[:i]) for i, b in enumerate(blocks))
                and sum(int(b.shape[0]) for b in blocks) == int(flat.shape[0]) and self._ingest_block_hip(flat)):
            return
        for blk in blocks:                                   # rank order = global environment order
            if fast and self._ingest_block_hip(blk):
                continue
            self.fold_counters()                             # the row-by-row path works on the host-side pointers
            obs, act, nxt, rew, done, store, morph = self.gather.unpack(blk)
            rows = torch.nonzero(store, as_tuple=False).flatten()          # host sync 1 of 2 per block (row count)
            if rows.numel() == 0:
                continue
            m = morph[rows]
            rows = rows[torch.argsort(m, stable=True)]                     # grouped by morphology, env order kept inside
            counts = torch.bincount(m, minlength=len(self.buffers)).tolist()   # host sync 2 (the ring pointers live on the host)
            self._stored += int(rows.numel())
            off = 0
            for k, cnt in enumerate(counts):
                if cnt == 0:
                    continue
                r = rows[off:off + cnt]
                off += cnt
                L = self.num_limbs[k]
                self.buffers[k].add_transitions(obs[r, :41 * L], act[r, :3 * L], nxt[r, :41 * L], rew[r], done[r])

    def total_episode_timesteps(self):
        """sum(episode_timesteps_list) over all ranks (the numerator of per_morph_iter, trainer.py:244)."""
        t = self.collector.episode_timesteps.sum().reshape(1)
        if self.gather.world > 1:
            t = t.clone()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())
