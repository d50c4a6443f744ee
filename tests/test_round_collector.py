"""RoundCollector against a scalar re-enactment of the reference's bookkeeping rules
(reference src/trainer.py:205-232): first episode per env per round, time-limit transitions stored with done = 0."""
import numpy as np
import torch

from sgrl_amd.rollout import RoundCollector


def _scalar_round(rewards, dones, max_steps):
    n = rewards.shape[1]
    done_list = [False] * n
    steps = [0] * n
    ep_rew = [0.0] * n
    buf = [0.0] * n
    stored = []
    for t in range(rewards.shape[0]):
        curr = [bool(d) for d in dones[t]]
        row = []
        for i in range(n):
            buf[i] += float(rewards[t, i])
            done_bool = float(curr[i])
            if steps[i] + 1 == max_steps:
                done_bool = 0.0
                curr[i] = True
            if curr[i] and ep_rew[i] == 0:
                ep_rew[i] = buf[i]
                buf[i] = 0.0
            if not done_list[i]:
                steps[i] += 1
                row.append((i, done_bool))
                done_list[i] = done_list[i] or curr[i]
        stored.append(row)
        if all(done_list):
            return stored, steps, ep_rew, t + 1
    return stored, steps, ep_rew, rewards.shape[0]


def test_matches_scalar_reenactment():
    rng = np.random.RandomState(0)
    for trial in range(20):
        n, T, max_steps = 7, 60, int(rng.choice([5, 12, 1000]))
        rewards = rng.normal(size=(T, n)).astype(np.float32)
        dones = rng.rand(T, n) < 0.08
        ref_stored, ref_steps, ref_rew, ref_T = _scalar_round(rewards, dones, max_steps)
        rc = RoundCollector(n, max_steps)
        for t in range(T):
            store, done_bool, finished = rc.record(torch.from_numpy(rewards[t]), torch.from_numpy(dones[t]))
            got = [(i, float(done_bool[i])) for i in range(n) if bool(store[i])]
            assert got == ref_stored[t], (trial, t)
            if finished:
                assert t + 1 == ref_T
                break
        assert rc.episode_timesteps.tolist() == ref_steps
        np.testing.assert_allclose(rc.episode_reward.numpy(), np.array(ref_rew, dtype=np.float32), rtol=1e-5, atol=1e-6)
        assert rc.per_morph_iter() == sum(ref_steps) // n


def test_begin_round_resets():
    rc = RoundCollector(3, 4)
    rc.record(torch.ones(3), torch.tensor([True, False, False]))
    rc.begin_round()
    assert not rc.done_list.any() and int(rc.episode_timesteps.sum()) == 0
