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
        self.curr = 0
        self.max_sample_size = 0

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

    def sample(self, batch_size, generator=None):
        """Uniform sample without replacement from the filled part (reference buffer.py:87-126 default path)."""
        batch_size = min(self.max_sample_size, int(batch_size))
        idx = torch.randperm(self.max_sample_size, device=self.device, generator=generator)[:batch_size]
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
        self.curr, self.max_sample_size = int(d["curr"]), int(d["max_sample_size"])
