"""The N > 1 path: environments shard across ranks with no data-path collective except ONE gather of the replay
block to the learner rank.  Covered here with world_size = 2 on the gloo backend (CPU); on the GPU box the same code
runs over RCCL (backend 'nccl')."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sgrl_amd.rollout import ReplayGather


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_env, obs_len, act_len, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = ReplayGather(n_env, obs_len, act_len, "cpu", dst=0)
        rng = np.random.RandomState(100 + rank)
        for step in range(3):
            obs = torch.from_numpy(rng.rand(n_env, obs_len).astype(np.float32))
            act = torch.from_numpy(rng.rand(n_env, act_len).astype(np.float32))
            nxt = torch.from_numpy(rng.rand(n_env, obs_len).astype(np.float32))
            rew = torch.from_numpy(rng.rand(n_env).astype(np.float32))
            done = torch.from_numpy((rng.rand(n_env) > 0.5).astype(np.uint8))
            g.pack(obs, act, nxt, rew, done)
            blocks = g.push()
            if rank == 0:
                assert len(blocks) == world
                np.save(os.path.join(out_dir, "recv_%d.npy" % step), torch.stack(blocks).numpy())
            else:
                assert blocks is None
            np.save(os.path.join(out_dir, "sent_%d_%d.npy" % (rank, step)), g.block.numpy().copy())
        assert g.bytes_per_step() == n_env * (2 * obs_len + act_len + 4) * 4
    finally:
        dist.destroy_process_group()


def _worker_pipelined(rank, world, port, n_env, obs_len, act_len, out_dir):
    """depth 2, push(wait=False): the gather of step t is in flight while step t + 1 is packed into the other block."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = ReplayGather(n_env, obs_len, act_len, "cpu", dst=0, depth=2)
        rng = np.random.RandomState(200 + rank)
        pending = []
        for step in range(5):
            obs = torch.from_numpy(rng.rand(n_env, obs_len).astype(np.float32))
            act = torch.from_numpy(rng.rand(n_env, act_len).astype(np.float32))
            nxt = torch.from_numpy(rng.rand(n_env, obs_len).astype(np.float32))
            rew = torch.from_numpy(rng.rand(n_env).astype(np.float32))
            done = torch.from_numpy((rng.rand(n_env) > 0.5).astype(np.uint8))
            blk = g.pack(obs, act, nxt, rew, done)          # waits for the gather that used this block two pushes ago
            assert blk is g.blocks[step % 2]
            np.save(os.path.join(out_dir, "sent_%d_%d.npy" % (rank, step)), blk.numpy().copy())
            recv = g.push(wait=False)
            pending.append((step, recv))
            if len(pending) == 2:                            # the older one is complete once its slot comes up again: drain to read it
                g.drain()
                for st, rv in pending:
                    if rank == 0:
                        np.save(os.path.join(out_dir, "recv_%d.npy" % st), torch.stack(rv).numpy())
                    else:
                        assert rv is None
                pending = []
        g.drain()
        for st, rv in pending:
            if rank == 0:
                np.save(os.path.join(out_dir, "recv_%d.npy" % st), torch.stack(rv).numpy())
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_world_size_2(tmp_path):
    world, n_env, obs_len, act_len = 2, 6, 41 * 2, 6
    mp.spawn(_worker_pipelined, args=(world, _free_port(), n_env, obs_len, act_len, str(tmp_path)), nprocs=world, join=True)
    for step in range(5):
        recv = np.load(tmp_path / ("recv_%d.npy" % step))
        for r in range(world):
            assert np.array_equal(recv[r], np.load(tmp_path / ("sent_%d_%d.npy" % (r, step))))


def test_gather_world_size_2(tmp_path):
    world, n_env, obs_len, act_len = 2, 5, 41 * 3, 9
    mp.spawn(_worker, args=(world, _free_port(), n_env, obs_len, act_len, str(tmp_path)), nprocs=world, join=True)
    for step in range(3):
        recv = np.load(tmp_path / ("recv_%d.npy" % step))
        for r in range(world):
            sent = np.load(tmp_path / ("sent_%d_%d.npy" % (r, step)))
            assert np.array_equal(recv[r], sent)      # byte-exact transport, rank order preserved


def test_pack_unpack_round_trip_single_rank():
    g = ReplayGather(4, 82, 6, "cpu")
    obs, act, nxt = torch.rand(4, 82), torch.rand(4, 6), torch.rand(4, 82)
    rew, done = torch.rand(4), torch.tensor([0, 1, 0, 1], dtype=torch.uint8)
    blk = g.pack(obs, act, nxt, rew, done)
    assert g.push()[0] is blk
    o, a, n, r, d, st, mid = g.unpack(blk)
    assert torch.equal(o, obs) and torch.equal(a, act) and torch.equal(n, nxt) and torch.equal(r, rew)
    assert torch.equal(d, done.float())
    assert st.all() and (mid == 0).all()                 # defaults: keep every row, morphology 0
    g.pack(obs, act, nxt, rew, done, store=torch.tensor([True, False, True, False]), morph_id=torch.tensor([3, 1, 0, 22]))
    o, a, n, r, d, st, mid = g.unpack(g.block)
    assert st.tolist() == [True, False, True, False] and mid.tolist() == [3, 1, 0, 22]


def test_env_id_sharding_is_disjoint():
    """rank r owns global env ids [r*n, (r+1)*n): the counter-RNG streams of different ranks never coincide."""
    from oracle import physics_ref
    a = physics_ref.lib().sgrl_oracle_rng_uniform01(7, 8191, 0, 0, 0)
    b = physics_ref.lib().sgrl_oracle_rng_uniform01(7, 8192, 0, 0, 0)
    assert a != b and 0 < a < 1 and 0 < b < 1


@pytest.mark.gpu
def test_rccl_gather_in_flight_single_rank():
    """The pipelined push on the 'nccl' (= RCCL) backend with device tensors: a one-rank process group still runs the real
    collective (asynchronous gather, stream-side wait), which is all a 1-GPU box can exercise of the N > 1 path."""
    assert torch.cuda.is_available()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        g = ReplayGather(64, 287, 21, "cuda:0", dst=0, depth=2)
        gen = torch.Generator(device="cuda:0").manual_seed(3)
        sent, got = [], []
        for step in range(5):
            obs, nxt = torch.rand(64, 287, device="cuda:0", generator=gen), torch.rand(64, 287, device="cuda:0", generator=gen)
            act, rew = torch.rand(64, 21, device="cuda:0", generator=gen), torch.rand(64, device="cuda:0", generator=gen)
            done = torch.rand(64, device="cuda:0", generator=gen) > 0.5
            sent.append(g.pack(obs, act, nxt, rew, done).clone())
            recv = g.push(wait=False)
            assert recv is g.recvs[step % 2]
            if step % 2 == 1:
                g.drain()
                got += [g.recvs[0][0].clone(), g.recvs[1][0].clone()]
        g.drain()
        got.append(g.recvs[0][0].clone())
        torch.cuda.synchronize()
        for a, b in zip(sent, got):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_device_pack_kernel_equals_the_slice_copies():
    """sgrl_pack_transitions (one launch per block on the device) against the host form of the same `pack` on the same data:
    byte-equal rows, for done flags given as bool / uint8 / float32, rows cut out of wider tensors (row stride > width), the
    staged form (observation half first, the rest after the step), and columns left untouched by None sources."""
    n, o, a = 37, 41 * 7, 21
    gen = torch.Generator().manual_seed(11)
    wide_obs = torch.rand(n, o + 13, generator=gen)
    act, nxt, rew = torch.rand(n, a, generator=gen) * 2 - 1, torch.rand(n, o, generator=gen), torch.randn(n, generator=gen)
    done_b = torch.rand(n, generator=gen) > 0.5
    store = torch.rand(n, generator=gen) > 0.3
    morph = torch.randint(0, 23, (n,), generator=gen)
    host = ReplayGather(n, o, a, "cpu")
    dev = ReplayGather(n, o, a, "cuda:0", depth=2)
    c = lambda t: None if t is None else t.cuda()
    for done in (done_b, done_b.to(torch.uint8), done_b.to(torch.float32)):
        ref = host.pack(wide_obs[:, :o], act, nxt, rew, done, store, morph).clone()
        got = dev.pack(c(wide_obs)[:, :o], c(act), c(nxt), c(rew), c(done), c(store), c(morph))
        dev.push()
        assert torch.equal(got.cpu(), ref)
    # staged: observation half first; the rest later; store / morph_id left as the previous use of that block wrote them
    blk = dev.stage_obs(c(nxt))
    before = blk.clone()
    assert torch.equal(before[:, :o].cpu(), nxt) and torch.equal(before[:, o:].cpu(), dev.blocks[dev._k % 2][:, o:].cpu())
    got = dev.pack(None, c(act), c(wide_obs)[:, :o], c(rew), c(done_b))
    assert got is blk
    exp = before.cpu()
    exp[:, o:o + a], exp[:, o + a:2 * o + a], exp[:, 2 * o + a], exp[:, 2 * o + a + 1] = act, wide_obs[:, :o], rew, done_b.float()
    assert torch.equal(got.cpu(), exp)
    with pytest.raises(AssertionError):
        dev.stage_obs(c(nxt))
        dev.pack(c(nxt), c(act), c(nxt), c(rew), c(done_b))
