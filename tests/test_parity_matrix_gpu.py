"""GPU parity matrix of the rollout engine: EVERY shipped morphology (29) plus one `_v2_` task per family, teacher-forced
and free-running against the CPU oracle, and full-size property tests of the single-GPU shares of BASELINE.json's
configs 2, 4 and 5 (config 3 lives in test_engine_gpu.py).  Through the C ABI of libsgrl_hip.so.

Tolerances: float64 on both sides, differing in summation order / FMA contraction only -- 1e-9 relative per
teacher-forced step (1e-7 cheetah: Euler with stiff tendons amplifies rounding within the step), 1e-6 relative over 1000
free-running steps (north_star's bar is 1e-4; measured <= 4e-9).  Flags, counters and padding are bit exact."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from sgrl_amd import mjcf

pytestmark = pytest.mark.gpu

_A = mjcf.list_assets()
FAMILIES = {
    "hopper": sorted(n for n in _A if "hopper" in n) + ["3d_hopper_v2_5_full"],
    "walker": sorted(n for n in _A if "walker" in n) + ["3d_walker_v2_7_full"],
    "humanoid": sorted(n for n in _A if "humanoid" in n) + ["3d_humanoid_v2_9_full"],
    "cheetah": sorted(n for n in _A if "cheetah" in n) + ["3d_cheetah_v2_14_full"],
}
assert sum(len(v) for v in FAMILIES.values()) == 29 + 4
# BASELINE.json config 5: 3d_cwhh = 8 cheetahs + 6 walkers + 3 hoppers + 6 humanoids (the training sets; reference
# src/environments/3d_cwhh/); the two held-out XMLs of each family live under zero_shot/
_HELD_OUT = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
             "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
_POOL = ThreadPoolExecutor(max_workers=16)     # the oracle is a C library called through ctypes: the GIL is released


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _make(names, per, seed=5, **kw):
    from sgrl_amd.vec_env import BatchedModularVecEnv
    env = BatchedModularVecEnv(names, per, seed=seed, device="cuda:0", **kw)
    env.enable_f64_outputs()
    return env


def _oracle_envs(env, seed, ids=None):
    from oracle import physics_ref
    out = []
    for i in (range(env.num_envs) if ids is None else ids):
        ib, fb = env._blobs[env.env_morph[i]]      # the very blobs the engine was created with (row caps included)
        out.append(physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=seed, env_id=i))
    return out


def test_config5_training_set_is_what_the_reference_ships():
    names = sorted(n for n in _A if n not in _HELD_OUT)
    assert len(names) == 23
    assert sum("cheetah" in n for n in names) == 8 and sum("walker" in n for n in names) == 6
    assert sum("hopper" in n for n in names) == 3 and sum("humanoid" in n for n in names) == 6


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_teacher_forced_step_parity_every_morphology(family):
    """Every step starts from the oracle's state (sgrl_set_records), so errors cannot accumulate."""
    torch = _torch()
    names = FAMILIES[family]
    env = _make(names, 1)
    env.reset_device()
    oes = _oracle_envs(env, 5)
    for oe in oes:
        oe.reset()
    rng = np.random.RandomState(0)
    tol = 1e-7 if family == "cheetah" else 1e-9
    n_done = np.zeros(len(names), dtype=int)
    for t in range(100):
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
        obs, rew = env.obs64.cpu().numpy(), env.rew64.cpu().numpy()
        done, dist = env.done.cpu().numpy(), env.dist.cpu().numpy()
        rec2, cnt2 = env.get_records()
        ods = list(_POOL.map(lambda ia: ia[1].step(a[ia[0]].astype(np.float64), auto_reset=False), enumerate(oes)))
        for i, oe in enumerate(oes):
            o, r, d, info = ods[i]
            q, v, xy, tg = env.state_of(rec2, i)
            assert np.abs(q - oe.qpos).max() < tol * (1 + np.abs(oe.qpos).max()), (names[i], t)
            assert np.abs(v - oe.qvel).max() < tol * (1 + np.abs(oe.qvel).max()), (names[i], t)
            assert np.abs(obs[i, :o.size] - o).max() < tol * (1 + np.abs(o).max()), (names[i], t)
            assert (obs[i, o.size:] == 0).all()
            assert abs(rew[i] - r) < tol * 100 * (1 + abs(r)), (names[i], t)
            assert bool(done[i]) == d, (names[i], t)
            assert abs(dist[i] - info["dist"]) < 1e-3 * (1 + info["dist"])
            assert cnt2[i, 2] == 0, "constraint rows dropped: " + names[i]
            n_done[i] += d
            if d:
                oe.counters[1] += 1
                oe.reset()
    assert n_done.sum() >= 2 and (n_done > 0).sum() * 2 >= len(names), dict(zip(names, n_done))   # terminations were exercised


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_free_running_1000_steps_every_morphology(family):
    """north_star: qpos/qvel within 1e-4 relative over 1000 free-running steps (auto-reset on, same counter RNG);
    asserted at 1e-6."""
    torch = _torch()
    names = FAMILIES[family]
    env = _make(names, 1)
    env.reset_device()
    oes = _oracle_envs(env, 5)
    for oe in oes:
        oe.reset()
    rng = np.random.RandomState(1)
    worst = np.zeros(len(names))
    episodes = np.zeros(len(names), dtype=int)
    for t in range(1000):
        a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
        env.step_device(torch.from_numpy(a).cuda())
        check = t % 50 == 49 or t == 999
        if check:
            torch.cuda.synchronize()
            done = env.done.cpu().numpy()
            rec, cnt = env.get_records()
        ods = list(_POOL.map(lambda ia: ia[1].step(a[ia[0]].astype(np.float64)), enumerate(oes)))
        episodes += np.array([od[2] for od in ods], dtype=int)
        if check:
            for i, oe in enumerate(oes):
                q, v, xy, tg = env.state_of(rec, i)
                assert cnt[i, 1] == oe.counters[1], "episode count diverged at step %d: %s" % (t, names[i])
                assert cnt[i, 0] == oe.counters[0]
                assert bool(done[i]) == ods[i][2]
                assert cnt[i, 2] == 0, "constraint rows dropped: " + names[i]
                eq = np.abs(q - oe.qpos).max() / (1 + np.abs(oe.qpos).max())
                ev = np.abs(v - oe.qvel).max() / (1 + np.abs(oe.qvel).max())
                worst[i] = max(worst[i], eq, ev)
    print("free-running worst relative deviation per morphology (%s):" % family,
          {n: float("%.2e" % w) for n, w in zip(names, worst)}, "episodes", int(episodes.sum()))
    assert (episodes > 0).all()
    assert worst.max() < 1e-6, dict(zip(names, worst))


def _full_size_properties(names, counts, steps, sample_ids, seed=9):
    """Determinism (two engines, same seed, same actions: bit identical), finiteness, zero padding, auto-reset semantics,
    no dropped constraint rows, and env independence: sampled envs of the big batch equal the same (morphology, env id)
    stepped alone by the oracle."""
    torch = _torch()
    envA = _make(names, counts, seed=seed)
    envB = _make(names, counts, seed=seed)
    n, amax, omax = envA.num_envs, envA.action_max_len, envA.obs_max_len
    envA.reset_device()
    envB.reset_device()
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = []
    ndone_total = 0
    for t in range(steps):
        a = (torch.rand((n, amax), device="cuda", generator=g) * 2 - 1).contiguous()
        acts.append(a[sample_ids].cpu().numpy())
        oA, rA, dA, _ = envA.step_device(a)
        oB, rB, dB, _ = envB.step_device(a)
        torch.cuda.synchronize()
        assert torch.equal(oA, oB) and torch.equal(rA, rB) and torch.equal(dA, dB)   # run-to-run bit identical
        assert torch.isfinite(oA).all() and torch.isfinite(rA).all()
        ndone_total += int(dA.sum())
        for k, L in enumerate(envA.num_limbs):
            if 41 * L < omax:
                assert float(oA[envA.morph_slices[k], 41 * L:].abs().max()) == 0.0
        if int(dA.sum()) > 0:       # a done env's row is a fresh reset observation: step counter back to 0
            cnt = envA.get_counters()
            idx = np.nonzero(dA.cpu().numpy())[0]
            assert (cnt[idx, 0] == 0).all() and (cnt[idx, 1] >= 1).all()
    assert ndone_total > 0
    recA, cntA = envA.get_records()
    n_over = int((cntA[:, 2] > 0).sum())
    assert n_over == 0, "constraint rows dropped in %d of %d envs: %s" % (
        n_over, n, {names[k]: int((cntA[envA.morph_slices[k], 2] > 0).sum()) for k in range(len(names))})
    assert envA.row_overflow_envs() == 0
    assert int((((cntA[:, 3] >> 8) & 255) > 0).sum()) == 0, "block pivoting gave up somewhere"
    oes = _oracle_envs(envA, seed, ids=sample_ids)
    for oe in oes:
        oe.reset()

    def run(j):
        for t in range(steps):
            oes[j].step(acts[t][j].astype(np.float64))
    list(_POOL.map(run, range(len(oes))))
    for j, i in enumerate(sample_ids):
        oe = oes[j]
        q, v, xy, tg = envA.state_of(recA, i)
        assert cntA[i, 1] == oe.counters[1] and cntA[i, 0] == oe.counters[0], i
        assert np.abs(q - oe.qpos).max() < 1e-6 * (1 + np.abs(oe.qpos).max()), i
        assert np.abs(v - oe.qvel).max() < 1e-6 * (1 + np.abs(oe.qvel).max()), i
    envA.close()
    envB.close()


def test_config2_hopper_4096_envs_properties():
    """BASELINE.json config 2: 3D_Hopper++ (3 variants), 4096 envs on one MI355X."""
    names = sorted(n for n in _A if "hopper" in n)
    _full_size_properties(names, [1365, 1365, 1366], 30, [0, 1364, 1365, 2729, 2730, 4095])


def test_config4_humanoid_4096_envs_properties():
    """BASELINE.json config 4, one GPU's share: 3D_Humanoid++ (6 training + 2 held-out morphologies), 4096 envs."""
    names = sorted(n for n in _A if "humanoid" in n)
    assert len(names) == 8
    _full_size_properties(names, [512] * 8, 30, [0, 511, 512, 1700, 2047, 2048, 3000, 4095])


def test_config5_cwhh_8188_envs_properties():
    """BASELINE.json config 5, one GPU's share of the environments: the 23 cwhh training morphologies x 356 envs
    (four occupancy classes -> several concurrent launch groups)."""
    names = sorted(n for n in _A if n not in _HELD_OUT)
    per = 8192 // len(names)
    ids = [0, per - 1, per, 5 * per + 7, 9 * per + 100, 12 * per + 1, 15 * per + 300, 18 * per + 5, 22 * per, 23 * per - 1]
    _full_size_properties(names, [per] * len(names), 25, ids)
