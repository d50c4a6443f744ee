"""The N > 1 tail of a rollout step: RoundCollector -> ONE gather -> learner-side routing of every kept
row into the replay buffer of its morphology.  world_size 2 and 4 on gloo (CPU); the checker is a scalar re-enactment of
the reference's own loop (reference src/trainer.py:205-236: per env, in env order, first episode of the round only,
time-limit rows stored with done = 0, rows cut to the morphology's 41 L / 3 L columns) fed with the same streams."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sgrl_amd.replay import DeviceReplayBuffer
from sgrl_amd.rollout import TransitionSink

LIMBS = [3, 7, 5]            # three morphologies, ragged
PER = [2, 1, 3]              # environments of each per rank
OBS, ACT = 41 * 7, 3 * 7
MAX_STEPS = 9
T = 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _streams(rank, n):
    rng = np.random.RandomState(1000 + rank)
    return dict(obs=rng.rand(T + 1, n, OBS).astype(np.float32), act=rng.rand(T, n, ACT).astype(np.float32),
                rew=rng.normal(size=(T, n)).astype(np.float32), done=rng.rand(T, n) < 0.12)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env_morph = np.repeat(np.arange(len(LIMBS)), PER)
        n = env_morph.size
        buffers = [DeviceReplayBuffer(41 * L, 3 * L, max_buffer_size=64) for L in LIMBS] if rank == 0 else None
        sink = TransitionSink(env_morph, LIMBS, OBS, ACT, max_episode_steps=MAX_STEPS, buffers=buffers, dst=0)
        s = _streams(rank, n)
        rounds = []
        for t in range(T):
            fin = sink.push(torch.from_numpy(s["obs"][t]), torch.from_numpy(s["act"][t]), torch.from_numpy(s["obs"][t + 1]),
                            torch.from_numpy(s["rew"][t]), torch.from_numpy(s["done"][t]))
            if fin:
                rounds.append((t, sink.total_episode_timesteps()))
                sink.begin_round()
        np.save(os.path.join(out_dir, "rounds_%d.npy" % rank), np.array(rounds, dtype=np.int64).reshape(-1, 2))
        if rank == 0:
            for k, b in enumerate(buffers):
                st = b.state_arrays()
                np.savez(os.path.join(out_dir, "buf_%d.npz" % k), **st)
            np.save(os.path.join(out_dir, "stored.npy"), np.array([sink.stored]))
    finally:
        dist.destroy_process_group()


def _reenact(world):
    """The reference's loop over the GLOBAL environment list (rank-major), scalar code."""
    env_morph = list(np.repeat(np.arange(len(LIMBS)), PER))
    n = len(env_morph)
    S = [_streams(r, n) for r in range(world)]
    N = world * n
    rows = [[] for _ in LIMBS]
    done_list, steps = [False] * N, [0] * N
    rounds = []
    for t in range(T):
        for g in range(N):
            r, i = divmod(g, n)
            curr = bool(S[r]["done"][t, i])
            done_bool = float(curr)
            if steps[g] + 1 == MAX_STEPS:
                done_bool = 0.0
                curr = True
            if not done_list[g]:
                steps[g] += 1
                k = env_morph[i]
                L = LIMBS[k]
                rows[k].append((S[r]["obs"][t, i, :41 * L], S[r]["act"][t, i, :3 * L], S[r]["obs"][t + 1, i, :41 * L],
                                S[r]["rew"][t, i], done_bool))
                done_list[g] = done_list[g] or curr
        if all(done_list):
            rounds.append((t, sum(steps)))
            done_list, steps = [False] * N, [0] * N
    return rows, rounds


@pytest.mark.parametrize("world", [2, 4])
def test_learner_buffers_equal_the_reference_loop(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rows, rounds = _reenact(world)
    assert len(rounds) >= 2                                    # several collection rounds were exercised
    for r in range(world):                                     # every rank saw the same global round boundaries
        got = np.load(tmp_path / ("rounds_%d.npy" % r))
        assert got.tolist() == [list(x) for x in rounds], r
    assert int(np.load(tmp_path / "stored.npy")[0]) == sum(len(x) for x in rows)
    for k, L in enumerate(LIMBS):
        b = np.load(tmp_path / ("buf_%d.npz" % k))
        cap = 64
        total = len(rows[k])
        assert total > cap or k != 2                           # morphology 2 (most envs) wraps the 64-row ring
        assert int(b["curr"]) == total % cap and int(b["max_sample_size"]) == min(total, cap)
        for j in range(max(0, total - cap), total):           # every row still in the ring, at its ring position
            o, a, nx, rw, d = rows[k][j]
            p = j % cap
            assert np.array_equal(b["obs_buffer"][p], o) and np.array_equal(b["action_buffer"][p], a)
            assert np.array_equal(b["next_obs_buffer"][p], nx)
            assert b["reward_buffer"][p] == rw and b["done_buffer"][p] == d


def test_single_rank_needs_no_process_group():
    env_morph = np.repeat(np.arange(len(LIMBS)), PER)
    buffers = [DeviceReplayBuffer(41 * L, 3 * L, max_buffer_size=64) for L in LIMBS]
    sink = TransitionSink(env_morph, LIMBS, OBS, ACT, max_episode_steps=MAX_STEPS, buffers=buffers)
    s = _streams(0, env_morph.size)
    for t in range(5):
        sink.push(torch.from_numpy(s["obs"][t]), torch.from_numpy(s["act"][t]), torch.from_numpy(s["obs"][t + 1]),
                  torch.from_numpy(s["rew"][t]), torch.from_numpy(s["done"][t]))
    assert sink.stored == sum(b.max_sample_size for b in buffers) > 0
    with pytest.raises(ValueError):
        TransitionSink(env_morph, LIMBS, OBS, ACT, buffers=None)
