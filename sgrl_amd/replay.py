"""Device-resident replay ring buffer (reference src/common/buffer.py:35-126, one buffer per morphology,
constructed with modular=True in src/main.py:141-155).

Same state and semantics as the reference's NumPy buffer -- five arrays (obs, action, next_obs, reward, done), write
pointer `curr`, fill level `max_sample_size`, wrap-around overwrite -- but the storage lives on the GPU so that the
rollout's transition block never leaves HBM: `add_transitions` scatters a whole batch of rows in one indexed copy
(the reference's `add_transition` row by row is also kept).  `state_arrays()` / `load_state_arrays()` exchange the
exact `.npy` payloads of the reference's snapshot (reference common/trainer.py:261-322).
"""
import numpy as np
import torch


class DeviceReplayBuffer(object):
    def __init__(self, obs_dim, action_dim, max_buffer_size=1000000, device="cpu"):
        """obs_dim = 41 * L, action_dim = 3 * L (the reference adds the 3 torso slots: buffer.py:49-50)."""
        self.max_buffer_size = int(max_buffer_size)
        self.obs_dim, self.action_dim = int(obs_dim), int(action_dim)
        self.device = torch.device(device)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.device)
        self.obs_buffer = z(self.max_buffer_size, self.obs_dim)
        self.action_buffer = z(self.max_buffer_size, self.action_dim)
        self.next_obs_buffer = z(self.max_buffer_size, self.obs_dim)
        self.reward_buffer = z(self.max_buffer_size)
        self.done_buffer = z(self.max_buffer_size)
        self._curr = 0
        self._fill = 0
        self._sync_hook = None      # set by a writer that advances the ring on the device (rollout.TransitionSink): called
                                    # before the host-side pointers are read, so they are always current when looked at

    # write pointer / fill level (reference buffer.py:60-61); a device-side writer may hold increments not yet folded in
    @property
    def curr(self):
        if self._sync_hook is not None:
            self._sync_hook()
        return self._curr

    @curr.setter
    def curr(self, v):
        if self._sync_hook is not None:      # increments a device-side writer still holds belong to the pointer being replaced
            self._sync_hook()
        self._curr = int(v)

    @property
    def max_sample_size(self):
        if self._sync_hook is not None:
            self._sync_hook()
        return self._fill

    @max_sample_size.setter
    def max_sample_size(self, v):
        if self._sync_hook is not None:
            self._sync_hook()
        self._fill = int(v)

    def clear(self):
        self.curr = 0
        self.max_sample_size = 0

    def add_transition(self, obs, action, next_obs, reward, done):
        self.add_transitions(torch.as_tensor(obs).reshape(1, -1), torch.as_tensor(action).reshape(1, -1),
                             torch.as_tensor(next_obs).reshape(1, -1), torch.as_tensor([reward]), torch.as_tensor([done]))

    def add_transitions(self, obs, action, next_obs, reward, done, mask=None):
        """Append the rows where `mask` is True (all rows if None), in row order -- identical to calling the reference's
        add_transition for each kept row."""
        if mask is not None:
            keep = torch.nonzero(mask.to(self.device), as_tuple=False).flatten()
            obs, action, next_obs = obs[keep], action[keep], next_obs[keep]
            reward, done = reward[keep], done[keep]
        k = int(obs.shape[0])
        if k == 0:
            return
        cap = self.max_buffer_size
        if k > cap:   # only the last `cap` rows survive a wrap of the whole ring
            drop = k - cap
            self.curr = (self.curr + drop) % cap
            obs, action, next_obs, reward, done = obs[drop:], action[drop:], next_obs[drop:], reward[drop:], done[drop:]
            self.max_sample_size = cap
            k = cap
        idx = (self.curr + torch.arange(k, device=self.device)) % cap
        f = lambda t: t.to(self.device, torch.float32)
        self.obs_buffer[idx] = f(obs)[:, :self.obs_dim]
        self.action_buffer[idx] = f(action)[:, :self.action_dim]
        self.next_obs_buffer[idx] = f(next_obs)[:, :self.obs_dim]
        self.reward_buffer[idx] = f(reward).reshape(-1)
        self.done_buffer[idx] = f(done).reshape(-1)
        self.curr = (self.curr + k) % cap
        self.max_sample_size = min(self.max_sample_size + k, cap)

    # fill levels from which the draw no longer permutes the whole buffer (O(fill) work for a 100-row batch: at the
    # reference's 1 M-row buffers 8 MB of traffic per update)
    SPARSE_DRAW_FACTOR = 32

    def draw_indices(self, batch_size, generator=None):
        """`batch_size` distinct row indices, uniform over the filled part, in random order: np.random.choice(fill, batch,
        replace=False) of reference buffer.py:87-126.  Small fills: a permutation.  Large fills (fill >= 32 x batch): the first
        `batch` DISTINCT values of 2 x batch i.i.d. uniform draws -- sequential rejection of repeats, i.e. exactly a uniform
        ordered sample without replacement -- in O(batch log batch) work and without a host synchronisation (fewer than
        `batch` distinct values among 2 x batch draws needs >= batch collisions: probability < 1e-50 at this ratio; the
        unfilled tail would then repeat row 0)."""
        fill = self.max_sample_size
        k = min(fill, int(batch_size))
        if fill < self.SPARSE_DRAW_FACTOR * max(k, 1):
            return torch.randperm(fill, device=self.device, generator=generator)[:k]
        m = 2 * k
        c = torch.randint(0, fill, (m,), device=self.device, generator=generator)
        srt, perm = torch.sort(c, stable=True)                       # equal values keep their draw order
        later = torch.zeros(m, dtype=torch.bool, device=self.device)
        later[1:] = srt[1:] == srt[:-1]                              # not the first occurrence of its value
        first = torch.empty(m, dtype=torch.bool, device=self.device)
        first[perm] = ~later
        pos = torch.cumsum(first, 0) - 1                             # rank among the distinct values, in draw order
        dst = torch.where(first & (pos < k), pos, pos.new_full((), k))
        out = torch.zeros(k + 1, dtype=torch.long, device=self.device)
        out.scatter_(0, dst, c)                                      # slot k collects everything that is not kept
        return out[:k]

    def sample(self, batch_size, generator=None):
        """Uniform sample without replacement from the filled part (reference buffer.py:87-126 default path)."""
        idx = self.draw_indices(batch_size, generator=generator)
        return dict(obs=self.obs_buffer[idx], action=self.action_buffer[idx], next_obs=self.next_obs_buffer[idx],
                    reward=self.reward_buffer[idx].reshape(-1, 1), done=self.done_buffer[idx].reshape(-1, 1))

    # ---- snapshot interchange with the reference's .npy files --------------------------------------------
    def state_arrays(self):
        n = lambda t: t.detach().cpu().numpy()
        return {"obs_buffer": n(self.obs_buffer), "action_buffer": n(self.action_buffer),
                "next_obs_buffer": n(self.next_obs_buffer), "reward_buffer": n(self.reward_buffer),
                "done_buffer": n(self.done_buffer), "curr": self.curr, "max_sample_size": self.max_sample_size}

    def load_state_arrays(self, d):
        for k in ("obs_buffer", "action_buffer", "next_obs_buffer", "reward_buffer", "done_buffer"):
            getattr(self, k).copy_(torch.from_numpy(np.asarray(d[k], dtype=np.float32)))
        if self._sync_hook is not None:
            self._sync_hook()               # pending device-side increments belong to the state being replaced
        self.curr, self.max_sample_size = int(d["curr"]), int(d["max_sample_size"])
