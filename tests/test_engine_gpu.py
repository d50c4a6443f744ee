"""GPU parity tests of the rollout engine (through the C ABI of libsgrl_hip.so) against the CPU oracle.

Tolerances: the engine computes in float64 like the oracle; the two differ only in summation order and FMA
contraction, so per-step (teacher-forced) agreement is asserted at 1e-9 relative and free-running episodes at the
north_star bound of 1e-4 relative on qpos/qvel.  Integer/byte outputs (done flags, counters, padding) are bit exact."""
import numpy as np
import pytest

from helpers import packed, WALKERS, HOPPERS

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _oracle_envs(env, names, seed):
    from oracle import physics_ref
    out = []
    for i in range(env.num_envs):
        ib, fb = env._blobs[env.env_morph[i]]      # the very blobs the engine was created with (row caps included)
        out.append(physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=seed, env_id=i))
    return out


def _make(names, per, seed=5, **kw):
    from sgrl_amd.vec_env import BatchedModularVecEnv
    env = BatchedModularVecEnv(names, per, seed=seed, device="cuda:0", **kw)
    env.enable_f64_outputs()
    return env


def test_native_library_is_the_one_loaded():
    from sgrl_amd import _lib
    L = _lib.lib()
    assert b"gfx950" in L.sgrl_version()
    maps = open("/proc/self/maps").read()
    assert "libsgrl_hip.so" in maps


@pytest.mark.parametrize("names", [HOPPERS, WALKERS, ["3d_humanoid_9_full", "3d_humanoid_7_left_arm"],
                                   ["3d_cheetah_14_full", "3d_cheetah_10_tail_leftbleg"], ["3d_walker_v2_7_full"]])
def test_reset_matches_oracle(names):
    torch = _torch()
    env = _make(names, 3)
    env.reset_device()
    torch.cuda.synchronize()
    obs = env.obs64.cpu().numpy()
    rec, cnt = env.get_records()
    for i, oe in enumerate(_oracle_envs(env, names, 5)):
        o = oe.reset()
        L = o.size
        np.testing.assert_allclose(obs[i, :L], o, rtol=0, atol=1e-12)
        assert (obs[i, L:] == 0).all()
        q, v, xy, tg = env.state_of(rec, i)
        np.testing.assert_allclose(q, oe.qpos, rtol=0, atol=1e-15)
        np.testing.assert_allclose(v, oe.qvel, rtol=0, atol=1e-15)
        np.testing.assert_allclose(tg, oe.target, rtol=1e-12)  # device vs libm cos/sin
        np.testing.assert_allclose(xy, oe.torso_xy_stale, rtol=0, atol=1e-15)
        assert cnt[i, 0] == 0 and cnt[i, 1] == 0
    assert np.array_equal(env.obs.cpu().numpy(), obs.astype(np.float32))


@pytest.mark.parametrize("names", [HOPPERS, WALKERS, ["3d_humanoid_9_full"], ["3d_cheetah_14_full"]])
def test_teacher_forced_step_parity(names):
    """Every step starts from the oracle's state (sgrl_set_records), so errors cannot accumulate."""
    torch = _torch()
    env = _make(names, 2)
    env.reset_device()
    oes = _oracle_envs(env, names, 5)
    for oe in oes:
        oe.reset()
    rng = np.random.RandomState(0)
    n_done = 0
    for t in range(90):
        rec, cnt = env.get_records()
        for i, oe in enumerate(oes):
            m = env.models[env.env_morph[i]]
            rec[i, :m.nq] = oe.qpos
            rec[i, m.nq:m.nq + m.nv] = oe.qvel
            rec[i, m.nq + m.nv:m.nq + m.nv + 2] = oe.torso_xy_stale
            rec[i, m.nq + m.nv + 2:m.nq + m.nv + 4] = oe.target
            cnt[i, 0], cnt[i, 1] = oe.counters[0], oe.counters[1]
        env.set_records(rec, cnt)
        a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
        env.step_device(torch.from_numpy(a).cuda(), auto_reset=False)
        torch.cuda.synchronize()
        obs = env.obs64.cpu().numpy()
        rew = env.rew64.cpu().numpy()
        done = env.done.cpu().numpy()
        dist = env.dist.cpu().numpy()
        rec2, cnt2 = env.get_records()
        for i, oe in enumerate(oes):
            o, r, d, info = oe.step(a[i].astype(np.float64), auto_reset=False)
            q, v, xy, tg = env.state_of(rec2, i)
            scale_q, scale_v = 1 + np.abs(oe.qpos).max(), 1 + np.abs(oe.qvel).max()
            tol = 1e-7 if "cheetah" in names[0] else 1e-9
            assert np.abs(q - oe.qpos).max() < tol * scale_q, (t, i)
            assert np.abs(v - oe.qvel).max() < tol * scale_v, (t, i)
            assert np.abs(obs[i, :o.size] - o).max() < tol * (1 + np.abs(o).max()), (t, i)
            assert abs(rew[i] - r) < tol * 100 * (1 + abs(r))
            assert bool(done[i]) == d, (t, i)
            assert abs(dist[i] - info["dist"]) < 1e-3 * (1 + info["dist"])
            assert cnt2[i, 2] == 0  # no constraint-row overflow
            n_done += d
            if d:
                oe.counters[1] += 1
                oe.reset()
    assert n_done > 0


@pytest.mark.parametrize("names,per", [(HOPPERS, 4), (WALKERS, 2), (["3d_humanoid_9_full", "3d_humanoid_7_left_arm", "3d_humanoid_8_left_knee"], 2),
                                       (["3d_cheetah_14_full", "3d_cheetah_11_leftfleg"], 2)])
def test_free_running_1000_steps_within_1e4(names, per):
    """north_star: qpos/qvel within 1e-4 relative over 1000 free-running steps (auto-reset on, same counter RNG); ASSERTED at 1e-6
    (measured: <= 4e-9) so that a regression of three orders of magnitude cannot pass unnoticed."""
    torch = _torch()
    env = _make(names, per)
    env.reset_device()
    oes = _oracle_envs(env, names, 5)
    for oe in oes:
        oe.reset()
    rng = np.random.RandomState(1)
    worst = 0.0
    episodes = 0
    for t in range(1000):
        a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
        env.step_device(torch.from_numpy(a).cuda())
        if t % 50 == 49 or t == 999:
            torch.cuda.synchronize()
            done = env.done.cpu().numpy()
            rec, cnt = env.get_records()
        ods = [oe.step(a[i].astype(np.float64)) for i, oe in enumerate(oes)]
        episodes += sum(od[2] for od in ods)
        if t % 50 == 49 or t == 999:
            for i, oe in enumerate(oes):
                q, v, xy, tg = env.state_of(rec, i)
                assert cnt[i, 1] == oe.counters[1], "episode count diverged at step %d env %d" % (t, i)
                assert cnt[i, 0] == oe.counters[0]
                assert bool(done[i]) == ods[i][2]
                eq = np.abs(q - oe.qpos).max() / (1 + np.abs(oe.qpos).max())
                ev = np.abs(v - oe.qvel).max() / (1 + np.abs(oe.qvel).max())
                worst = max(worst, eq, ev)
    print("free-running worst relative deviation:", names[0], worst, "episodes", episodes)
    assert episodes > 5
    assert worst < 1e-6, worst


def test_numpy_vecenv_surface_matches_reference_conventions():
    torch = _torch()
    from sgrl_amd.vec_env import BatchedModularVecEnv
    names = sorted(WALKERS)
    env = BatchedModularVecEnv(names, 1, seed=0, device="cuda:0")
    assert env.num_envs == 8 and env.obs_max_len == 287 and env.action_max_len == 21
    assert env.action_space.low[0] == -1.0 and env.action_space.high[0] == 1.0
    obs = env.reset()
    assert obs.shape == (8, 287) and obs.dtype == np.float32
    for i, L in enumerate(env.num_limbs):
        assert (obs[i, 41 * L:] == 0).all() and np.abs(obs[i, :41 * L]).max() > 0
    acts = [np.random.uniform(env.action_space.low[0], env.action_space.high[0], size=env.action_max_len) for _ in range(8)]
    assert not env.waiting
    env.step_async(acts)
    assert env.waiting
    o, r, d, infos = env.step_wait()
    assert not env.waiting
    assert o.shape == (8, 287) and r.shape == (8,) and d.shape == (8,) and d.dtype == bool and len(infos) == 8
    assert "dist" in infos[0]
    r.astype(np.float32)
    d2 = d.astype(np.float32)
    d2[0] = True  # item assignment like trainer.py:212
    o_keep, r_keep, dist_keep = o.copy(), r.copy(), [infos[i]["dist"] for i in range(8)]
    o2, r2, d3, infos2 = env.step(acts)
    assert np.isfinite(o2).all()
    # every step hands out its OWN host arrays (views of one pinned block per step): the previous step's are untouched
    assert np.array_equal(o, o_keep) and np.array_equal(r, r_keep) and [infos[i]["dist"] for i in range(8)] == dist_keep
    assert not np.array_equal(o2, o) and isinstance(infos2[3], dict) and len(tuple(infos2)) == 8
    # the three forms a trainer may hand actions in: a list of float32 arrays (buffer-protocol flatten), of float64 arrays, an ndarray
    env_b = BatchedModularVecEnv(names, 1, seed=0, device="cuda:0")
    env_c = BatchedModularVecEnv(names, 1, seed=0, device="cuda:0")
    env_d = BatchedModularVecEnv(names, 1, seed=0, device="cuda:0")
    for e in (env_b, env_c, env_d):
        e.reset()
    a32 = [np.random.RandomState(i).uniform(-1, 1, size=env.action_max_len).astype(np.float32) for i in range(8)]
    ob, rb, _, _ = env_b.step(a32)
    oc, rc, _, _ = env_c.step([x.astype(np.float64) for x in a32])
    od, rd, _, _ = env_d.step(np.stack(a32))
    assert np.array_equal(ob, oc) and np.array_equal(ob, od) and np.array_equal(rb, rc) and np.array_equal(rb, rd)
    for e in (env_b, env_c, env_d):
        e.close()
    with pytest.raises(ValueError):
        env.step_async([np.zeros(5)] * 8)
    with pytest.raises(ValueError):
        env.step_async([np.zeros(env.action_max_len, dtype=np.float32)] * 7)
    env.close()
    assert env.closed


def test_full_size_batch_properties():
    """Config 3 size (8 walkers x 1024 = 8192 envs): determinism, env independence, padding, auto-reset semantics."""
    torch = _torch()
    names = sorted(WALKERS)
    envA = _make(names, 1024, seed=9)
    envB = _make(names, 1024, seed=9)
    small_ids = [0, 1023, 1024, 4095, 8191]
    envA.reset_device()
    envB.reset_device()
    g = torch.Generator(device="cuda").manual_seed(3)
    ndone_total = 0
    for t in range(25):
        a = (torch.rand((8192, 21), device="cuda", generator=g) * 2 - 1).contiguous()
        oA, rA, dA, _ = envA.step_device(a)
        oB, rB, dB, _ = envB.step_device(a)
        torch.cuda.synchronize()
        assert torch.equal(oA, oB) and torch.equal(rA, rB) and torch.equal(dA, dB)   # run-to-run bit identical
        assert torch.isfinite(oA).all() and torch.isfinite(rA).all()
        ndone_total += int(dA.sum())
        # padding: rows of morphology k are zero beyond 41*L_k
        for k, L in enumerate(envA.num_limbs):
            sl = envA.morph_slices[k]
            assert float(oA[sl, 41 * L:].abs().max()) == 0.0 if 41 * L < 287 else True
        # auto-reset: a done env's row is a fresh reset observation: step counter back to 0
        if int(dA.sum()) > 0:
            rec, cnt = envA.get_records()
            idx = np.nonzero(dA.cpu().numpy())[0]
            assert (cnt[idx, 0] == 0).all() and (cnt[idx, 1] >= 1).all()
    assert ndone_total > 0
    recA, cntA = envA.get_records()
    assert (cntA[:, 2] == 0).all(), "constraint-row overflow in %d envs" % int((cntA[:, 2] > 0).sum())
    # env independence: env i of the big batch == the same (morphology, env_id) stepped alone by the oracle
    from oracle import physics_ref
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = [(torch.rand((8192, 21), device="cuda", generator=g) * 2 - 1).cpu().numpy() for _ in range(25)]
    for i in small_ids:
        ib, fb = envA._blobs[envA.env_morph[i]]
        oe = physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=9, env_id=i)
        oe.reset()
        for t in range(25):
            oe.step(acts[t][i].astype(np.float64))
        q, v, xy, tg = envA.state_of(recA, i)
        assert cntA[i, 1] == oe.counters[1] and cntA[i, 0] == oe.counters[0]
        assert np.abs(q - oe.qpos).max() < 1e-6 * (1 + np.abs(oe.qpos).max())
        assert np.abs(v - oe.qvel).max() < 1e-6 * (1 + np.abs(oe.qvel).max())


def test_mixed_families_use_separate_launch_groups_and_still_match_the_oracle():
    """hopper (10 KB LDS slab), walker (20 KB) and cheetah (41 KB) fall into different occupancy classes: the engine
    issues one launch per class on forked streams.  Same parity as a single launch."""
    torch = _torch()
    names = ["3d_cheetah_14_full", "3d_hopper_3_shin", "3d_walker_7_full"]
    env = _make(names, 3)
    assert env.lds_bytes > 32 * 1024 and env.launch_groups >= 2   # largest slab (cheetah_14); a dispatch per occupancy class
    env.reset_device()
    oes = _oracle_envs(env, names, 5)
    for oe in oes:
        oe.reset()
    rng = np.random.RandomState(2)
    for t in range(30):
        a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
        env.step_device(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        obs = env.obs64.cpu().numpy()
        done = env.done.cpu().numpy()
        for i, oe in enumerate(oes):
            o, r, d, info = oe.step(a[i].astype(np.float64))
            assert bool(done[i]) == d, (t, i)
            assert np.abs(obs[i, :o.size] - o).max() < 1e-5 * (1 + np.abs(o).max()), (t, i)
            assert (obs[i, o.size:] == 0).all()


def test_limits_are_rejected_loudly():
    """An observation row too narrow for a morphology is an error at create time (never a silent truncation)."""
    from sgrl_amd._lib import SgrlError
    from sgrl_amd.vec_env import BatchedModularVecEnv
    with pytest.raises(SgrlError):
        BatchedModularVecEnv(["3d_cheetah_14_full"], 2, seed=0, device="cuda:0", obs_max_len=41 * 3)


def test_contact_rich_states_keep_every_constraint_row():
    """Many-geom morphologies lying on the floor want far more than 64 constraint rows (profiles/r3_soak.json: 64 % of the
    cheetah environments dropped rows under a flailing policy at the round-2 cap of 64).  The row cap of those morphologies is
    now their geometric worst case (cheetah_14: 211, humanoid_9: 140; MuJoCo's default njmax = -1 never drops either): no
    row is dropped, evaluations beyond the 64 rows of the exact block-pivot solve run Gauss-Seidel over the rows in the HBM
    slab (wave_hip.h pgs_big), and the result follows the CPU oracle's (same dispatch rule, same sweep order; the iteration is
    stopped by its tolerance, so agreement is to 1e-6, not to rounding)."""
    import torch
    from oracle import physics_ref
    from sgrl_amd.vec_env import BatchedModularVecEnv
    names = ["3d_cheetah_14_full", "3d_cheetah_10_tail_leftbleg", "3d_humanoid_9_full"]
    quats = [[1, 0, 0, 0], [0.70710678, 0.70710678, 0, 0], [0.70710678, 0, 0.70710678, 0]]
    env = BatchedModularVecEnv(names, len(quats), seed=2, device="cuda:0")
    env.enable_f64_outputs()
    env.reset_device()
    assert [int(b[0][16]) for b in env._blobs] == [211, 159, 140]
    oes = []
    rec, cnt = env.get_records()
    for i in range(env.num_envs):
        ib, fb = env._blobs[env.env_morph[i]]
        m = env.models[env.env_morph[i]]
        oe = physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=2, env_id=i)
        oe.reset()
        q = np.array(fb[16:16 + m.nq])
        q[2] = 0.05
        q[3:7] = quats[i % len(quats)]
        oe.qpos[:] = q
        oe.qvel[:] = 0
        oes.append(oe)
    rng = np.random.RandomState(4)
    pgs_evals = 0
    for t in range(6):
        rec, cnt = env.get_records()
        for i, oe in enumerate(oes):
            m = env.models[env.env_morph[i]]
            rec[i, :m.nq] = oe.qpos
            rec[i, m.nq:m.nq + m.nv] = oe.qvel
            rec[i, m.nq + m.nv:m.nq + m.nv + 2] = oe.torso_xy_stale
            rec[i, m.nq + m.nv + 2:m.nq + m.nv + 4] = oe.target
            cnt[i, 0], cnt[i, 1] = oe.counters[0], oe.counters[1]
        env.set_records(rec, cnt)
        a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
        env.step_device(torch.from_numpy(a).cuda(), auto_reset=False)
        torch.cuda.synchronize()
        rec2, cnt2 = env.get_records()
        obs = env.obs64.cpu().numpy()
        for i, oe in enumerate(oes):
            o, r, d, info = oe.step(a[i].astype(np.float64), auto_reset=False)
            assert info["overflow"] == 0 and cnt2[i, 2] == 0, (names[env.env_morph[i]], t)
            q, v, _, _ = env.state_of(rec2, i)
            assert np.abs(q - oe.qpos).max() < 1e-6 * (1 + np.abs(oe.qpos).max()), (names[env.env_morph[i]], t)
            assert np.abs(v - oe.qvel).max() < 1e-6 * (1 + np.abs(oe.qvel).max()), (names[env.env_morph[i]], t)
            assert np.abs(obs[i, :o.size] - o).max() < 1e-6 * (1 + np.abs(o).max())
        pgs_evals += int((cnt2[:, 3] & 0xFF).sum())
    assert pgs_evals > 0, "no evaluation went beyond the 64 rows of the block-pivot solve: the test state is too tame"


def test_contact_rich_free_running_on_both_kernel_kinds(monkeypatch):
    """Regression (round 3): a humanoid_7 lying on the floor took the slab path with 33..64 rows, where the factor scratch was a
    run-time select between an LDS and a global pointer -- flat accesses that FAULTED in the fixed-dimension kernels (memory
    aperture violation).  The address space is now static.  64 environments per morphology from lying poses, free-running
    with auto-reset, on the family kernels and on the generic kernel: finite, no dropped rows, both kinds agree."""
    import torch
    from sgrl_amd.vec_env import BatchedModularVecEnv
    names = ["3d_cheetah_14_full", "3d_humanoid_9_full", "3d_cheetah_10_tail_leftbleg", "3d_humanoid_7_left_arm", "3d_walker_7_full"]
    quats = [[1, 0, 0, 0], [0.70710678, 0.70710678, 0, 0], [0.70710678, 0, 0.70710678, 0]]
    for specs in ("1", "0"):
        monkeypatch.setenv("SGRL_SPECS", specs)
        env = BatchedModularVecEnv(names, 64, seed=3, device="cuda:0")
        assert (env.fixed_dim_groups > 0) == (specs == "1")
        env.enable_f64_outputs()
        env.reset_device()
        rec, cnt = env.get_records()
        for i in range(env.num_envs):
            m = env.models[env.env_morph[i]]
            q = np.array(env._blobs[env.env_morph[i]][1][16:16 + m.nq])
            q[2] = 0.05 + 0.02 * (i % 5)
            q[3:7] = quats[i % 3]
            rec[i, :m.nq] = q
            rec[i, m.nq:m.nq + m.nv] = 0
        env.set_records(rec, cnt)
        g = torch.Generator(device="cuda").manual_seed(1)
        slab = 0
        for t in range(40):
            a = (torch.rand((env.num_envs, env.action_max_len), device="cuda", generator=g) * 2 - 1).contiguous()
            env.step_device(a, auto_reset=True)
            if t % 10 == 9:
                torch.cuda.synchronize()
                c = env.get_counters()
                slab += int((c[:, 3] >> 16).sum())
                assert (c[:, 2] == 0).all() and bool(torch.isfinite(env.obs).all())
        assert slab > 0, "no evaluation took the slab path: the poses are too tame"
        rec = env.get_records()[0]
        for i in range(env.num_envs):             # the used part of every record (rows are padded to the widest morphology)
            assert all(np.isfinite(x).all() for x in env.state_of(rec, i)), (specs, names[env.env_morph[i]])
        env.close()


def test_fixed_dimension_kernels_agree_with_the_generic_kernel(monkeypatch):
    """The family kernels of csrc/step_spec.hip (dimensions as compile-time constants, light families at four waves per SIMD)
    and the generic kernel are the same source: same states after 30 free-running steps to rounding (contraction of
    multiply-adds may differ between instances), identical flags -- and the batch really ran on the walker family's fixed-dimension kernel.  A custom row cap has no instance:
    generic kernel."""
    import torch
    from sgrl_amd.vec_env import BatchedModularVecEnv
    names = sorted(n for n in __import__("sgrl_amd.mjcf", fromlist=["x"]).list_assets() if "walker" in n)
    outs = []
    for specs in ("0", "1"):
        monkeypatch.setenv("SGRL_SPECS", specs)
        env = BatchedModularVecEnv(names, 3, seed=21, device="cuda:0")
        assert env.launch_groups == 1 and env.fixed_dim_groups == int(specs)
        env.enable_f64_outputs()
        env.reset_device()
        g = torch.Generator(device="cuda").manual_seed(5)
        dones = []
        for t in range(30):
            a = (torch.rand((env.num_envs, env.action_max_len), device="cuda", generator=g) * 2 - 1).contiguous()
            env.step_device(a)
            dones.append(env.done.clone())
        torch.cuda.synchronize()
        rec, cnt = env.get_records()
        outs.append((rec, cnt, env.obs64.cpu().numpy(), torch.stack(dones).cpu().numpy()))
        env.close()
    (r0, c0, o0, d0), (r1, c1, o1, d1) = outs
    envc = BatchedModularVecEnv(names[:2], 2, seed=21, device="cuda:0", max_rows=16)
    assert envc.fixed_dim_groups == 0
    envc.close()
    assert (d0 == d1).all() and (c0[:, :3] == c1[:, :3]).all()
    assert abs(r0 - r1).max() < 1e-9 * (1 + abs(r0).max()) and abs(o0 - o1).max() < 1e-9 * (1 + abs(o0).max())
