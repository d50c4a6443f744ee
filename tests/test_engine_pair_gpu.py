"""Two environments per wavefront on the MI355X (sgrl_amd/csrc/wave_half.h, engine_kernel.h env_step_pair): the light morphologies
(walker_2, walker_3, walker_4, hopper_3, hopper_4: up to 15 dofs) of a batch on a fixed-dimension kernel step in pairs -- lanes 0..31 one environment, lanes
32..63 its neighbour of the same morphology.  Checked against the oracle, against the same engine with SGRL_PAIR=0 (one
environment per wavefront), with odd counts (the last environment steps alone) and with the two halves of a wavefront in very
different contact situations (their data-dependent branches diverge)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["3d_hopper_3_shin", "3d_walker_2_right_leg_left_knee", "3d_walker_3_left_knee_right_knee", "3d_walker_3_left_leg_right_foot",
         "3d_walker_7_full", "3d_walker_v2_3_left_leg_right_foot",      # a `_v2_` task (near targets, resampled on arrival) of a paired set
         "3d_walker_4_right_knee_left_foot",                             # nv = 15: pairs on the dieted slab (20 LDS rows)
         "3d_hopper_4_lower_shin"]                                       # nv = 15, 14 contact candidates: 17 LDS rows
COUNTS = [4, 3, 5, 2, 2, 2, 3, 4]


def _make(monkeypatch, pair, names=NAMES, counts=COUNTS, seed=9, **kw):
    from sgrl_amd.vec_env import BatchedModularVecEnv
    monkeypatch.setenv("SGRL_PAIR", "1" if pair else "0")
    monkeypatch.setenv("SGRL_SPECS", "1")         # hoppers and walkers in one batch: one fixed-dimension launch per family
    env = BatchedModularVecEnv(names, counts, seed=seed, device="cuda:0", **kw)
    env.enable_f64_outputs()
    return env


def _oracles(env, seed):
    from oracle import physics_ref
    return [physics_ref.OracleEnv(physics_ref.OracleModel(*env._blobs[env.env_morph[i]]), seed=seed, env_id=i) for i in range(env.num_envs)]


def test_pairing_is_what_the_engine_reports(monkeypatch):
    env = _make(monkeypatch, True)
    assert env.fixed_dim_groups == 2
    assert env.paired_envs == 4 + 2 + 4 + 2 + 2 + 2 + 4  # hopper_3 4 of 4, walker_2 2 of 3, walker_3 4 of 5, 2 of 2 and (v2) 2 of 2, walker_4 2 of 3, hopper_4 4 of 4; walker_7 never
    assert env.lds_bytes <= 20480                    # the slab pairs keep eight workgroups per CU
    env.close()
    env = _make(monkeypatch, False)
    assert env.paired_envs == 0
    env.close()


def test_paired_free_running_matches_oracle_and_the_unpaired_engine(monkeypatch):
    import torch
    envs = [_make(monkeypatch, True), _make(monkeypatch, False)]
    for e in envs:
        e.reset_device()
    oes = _oracles(envs[0], 9)
    for oe in oes:
        oe.reset()
    rng = np.random.RandomState(3)
    episodes, worst, worst_pair = 0, 0.0, 0.0
    for t in range(300):
        a = rng.uniform(-1, 1, size=(envs[0].num_envs, envs[0].action_max_len)).astype(np.float32)
        for e in envs:
            e.step_device(torch.from_numpy(a).cuda())
        ods = [oe.step(a[i].astype(np.float64)) for i, oe in enumerate(oes)]
        episodes += sum(od[2] for od in ods)
        if t % 20 == 19:
            torch.cuda.synchronize()
            recs = [e.get_records() for e in envs]
            dones = [e.done.cpu().numpy() for e in envs]
            obs = [e.obs64.cpu().numpy() for e in envs]
            assert np.array_equal(dones[0], dones[1]) and np.array_equal(recs[0][1][:, :2], recs[1][1][:, :2])
            worst_pair = max(worst_pair, float(np.abs(obs[0] - obs[1]).max()))
            for i, oe in enumerate(oes):
                q, v, xy, tg = envs[0].state_of(recs[0][0], i)
                assert recs[0][1][i, 1] == oe.counters[1] and recs[0][1][i, 0] == oe.counters[0], (t, i)
                assert bool(dones[0][i]) == ods[i][2]
                assert recs[0][1][i, 2] == 0
                worst = max(worst, np.abs(q - oe.qpos).max() / (1 + np.abs(oe.qpos).max()), np.abs(v - oe.qvel).max() / (1 + np.abs(oe.qvel).max()))
                L = ods[i][0].size
                assert (obs[0][i, L:] == 0).all()
    print("paired engine vs oracle %.2e, vs one environment per wavefront %.2e, episodes %d" % (worst, worst_pair, episodes))
    assert episodes > 5 and worst < 1e-6 and worst_pair < 1e-6
    for e in envs:
        e.close()


def test_teacher_forced_parity_with_divergent_halves(monkeypatch):
    """Even environments of every morphology start lying on their side on the floor (many contact rows, evaluations that leave the
    LDS row arrays for the HBM slab), odd ones are dropped from the air: the two halves of every paired wavefront solve very
    different constraint problems.  Every step starts from the oracle's state; agreement per step to 1e-9."""
    import torch
    env = _make(monkeypatch, True, seed=4)
    assert env.paired_envs > 0
    env.reset_device()
    oes = _oracles(env, 4)
    for i, oe in enumerate(oes):
        oe.reset()
        if i % 2 == 0:        # pressed flat into the floor, every hinge beyond its limit: 18 .. 30 rows (walker_3 / hopper_3: the slab path)
            oe.qpos[2] = 0.03
            oe.qpos[3:7] = [0.70710678, 0, 0.70710678, 0] if "hopper" in NAMES[env.env_morph[i]] else [1, 0, 0, 0]
            oe.qpos[7:env.models[env.env_morph[i]].nq] = 0.8
        else:
            oe.qpos[2] += 0.8
        oe.qvel[:] = 0
        oe.refresh()
    rng = np.random.RandomState(0)
    slab_evals = 0
    for t in range(40):
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
        obs, rew, done = env.obs64.cpu().numpy(), env.rew64.cpu().numpy(), env.done.cpu().numpy()
        rec2, cnt2 = env.get_records()
        slab_evals += int((cnt2[:, 3] >> 16).sum())
        for i, oe in enumerate(oes):
            o, r, d, info = oe.step(a[i].astype(np.float64), auto_reset=False)
            q, v, xy, tg = env.state_of(rec2, i)
            assert np.abs(q - oe.qpos).max() < 1e-9 * (1 + np.abs(oe.qpos).max()), (t, i)
            assert np.abs(v - oe.qvel).max() < 1e-9 * (1 + np.abs(oe.qvel).max()), (t, i)
            assert np.abs(obs[i, :o.size] - o).max() < 1e-9 * (1 + np.abs(o).max()), (t, i)
            assert abs(rew[i] - r) < 1e-7 * (1 + abs(r)) and bool(done[i]) == d and cnt2[i, 2] == 0
            if d:
                oe.counters[1] += 1
                oe.reset()
    print("evaluations on the HBM-slab row path:", slab_evals)
    assert slab_evals > 0
    env.close()


def test_all_light_batches_take_the_paired_family_kernels(monkeypatch):
    """A batch of light morphologies only used to run on the four-waves-per-SIMD light kernel; with every morphology pairable it
    now takes the family kernels two environments per wavefront (profiles/r5_light_pair_probe.txt: 21-28 % faster).
    SGRL_PAIR=0 keeps the light kernel.  Both agree with the oracle."""
    import torch
    from sgrl_amd.vec_env import BatchedModularVecEnv
    names = ["3d_hopper_3_shin", "3d_walker_3_left_knee_right_knee"]
    monkeypatch.delenv("SGRL_SPECS", raising=False)
    outs = []
    for lk in (None, "1"):
        monkeypatch.setenv("SGRL_PAIR", "1" if lk is None else "0")
        env = BatchedModularVecEnv(names, 4, seed=6, device="cuda:0")
        assert env.fixed_dim_groups == 2 and env.paired_envs == (8 if lk is None else 0)
        env.enable_f64_outputs()
        env.reset_device()
        oes = _oracles(env, 6)
        for oe in oes:
            oe.reset()
        rng = np.random.RandomState(1)
        for t in range(40):
            a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
            env.step_device(torch.from_numpy(a).cuda())
            for i, oe in enumerate(oes):
                oe.step(a[i].astype(np.float64))
        torch.cuda.synchronize()
        rec, cnt = env.get_records()
        for i, oe in enumerate(oes):
            q, v, xy, tg = env.state_of(rec, i)
            assert cnt[i, 1] == oe.counters[1] and np.abs(q - oe.qpos).max() < 1e-7 * (1 + np.abs(oe.qpos).max())
        outs.append(rec)
        env.close()
    assert np.abs(outs[0] - outs[1]).max() < 1e-7
