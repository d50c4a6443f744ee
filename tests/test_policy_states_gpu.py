"""Teacher-forced parity of the rollout engine FROM STATES ALONG LONG, POLICY-DRIVEN EPISODES (VERDICT r2 item 4 i).

tests/golden/policy_states.npz holds `sgrl_get_records` snapshots taken every 50 steps of episodes that had already lasted
80 ... 600 steps (tools/learn_curve.py, run on the MI355X; profiles/r3_learning_curve.log): hoppers under the TD3-trained SET
policy and under a joint-space PD controller, walkers / humanoids / cheetahs under the PD controller -- sustained stance,
long contact sequences, joint limits engaged -- together with the action the driver applied next.  Every other GPU parity
test starts at `reset_model` and applies U(-1, 1) actions; here the engine and the CPU oracle are started from the SAME
captured state and must agree to the same bounds (1e-9 relative per step, 1e-7 cheetah), through the C ABI."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_states.npz")
_POOL = ThreadPoolExecutor(max_workers=16)
SETS = ["hopper_pd", "hopper_policy", "walker_pd", "humanoid_pd", "cheetah_pd"]


def test_fixture_covers_long_episodes_of_every_family():
    g = np.load(GOLD)
    for s in SETS:
        steps = g[s + "_cnt"][:, 0]
        assert len(steps) >= 9 and steps.min() >= 50 and steps.max() >= 110, (s, steps)


@pytest.mark.parametrize("which", SETS)
def test_teacher_forced_parity_from_policy_driven_states(which):
    import torch
    from oracle import physics_ref
    from sgrl_amd.vec_env import BatchedModularVecEnv
    g = np.load(GOLD)
    names = [str(n) for n in g[which + "_morph"]]
    rec0, cnt0, act0 = g[which + "_rec"], g[which + "_cnt"], g[which + "_act"]
    env = BatchedModularVecEnv(names, 1, seed=17, device="cuda:0")
    env.enable_f64_outputs()
    env.reset_device()
    oes = []
    for i in range(env.num_envs):
        ib, fb = env._blobs[env.env_morph[i]]
        oe = physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=17, env_id=i)
        oe.reset()
        m = env.models[env.env_morph[i]]
        oe.qpos[:] = rec0[i, :m.nq]
        oe.qvel[:] = rec0[i, m.nq:m.nq + m.nv]
        oe.torso_xy_stale[:] = rec0[i, m.nq + m.nv:m.nq + m.nv + 2]
        oe.target[:] = rec0[i, m.nq + m.nv + 2:m.nq + m.nv + 4]
        oe.counters[0], oe.counters[1] = cnt0[i, 0], cnt0[i, 1]
        oes.append(oe)
    tol = 1e-7 if "cheetah" in which else 1e-9
    rng = np.random.RandomState(3)
    rows_seen = 0
    for t in range(20):
        rec, cnt = env.get_records()
        for i, oe in enumerate(oes):
            m = env.models[env.env_morph[i]]
            rec[i, :m.nq] = oe.qpos
            rec[i, m.nq:m.nq + m.nv] = oe.qvel
            rec[i, m.nq + m.nv:m.nq + m.nv + 2] = oe.torso_xy_stale
            rec[i, m.nq + m.nv + 2:m.nq + m.nv + 4] = oe.target
            cnt[i, 0], cnt[i, 1] = oe.counters[0], oe.counters[1]
        env.set_records(rec, cnt)
        a = np.zeros((env.num_envs, env.action_max_len), dtype=np.float32)
        if t == 0:
            a[:] = act0[:, :env.action_max_len]          # what the driver applied in this state
        else:
            a[:] = np.clip(act0[:, :env.action_max_len] + 0.3 * rng.randn(*a.shape), -1, 1)      # stays near the driver's regime
        for i in range(env.num_envs):
            a[i, 3 * env.num_limbs[env.env_morph[i]]:] = 0
        env.step_device(torch.from_numpy(a).cuda(), auto_reset=False)
        torch.cuda.synchronize()
        obs, rew, done = env.obs64.cpu().numpy(), env.rew64.cpu().numpy(), env.done.cpu().numpy()
        rec2, cnt2 = env.get_records()
        ods = list(_POOL.map(lambda ia: ia[1].step(a[ia[0]].astype(np.float64), auto_reset=False), enumerate(oes)))
        for i, oe in enumerate(oes):
            o, r, d, info = ods[i]
            q, v, xy, tg = env.state_of(rec2, i)
            assert np.abs(q - oe.qpos).max() < tol * (1 + np.abs(oe.qpos).max()), (names[i], t)
            assert np.abs(v - oe.qvel).max() < tol * (1 + np.abs(oe.qvel).max()), (names[i], t)
            assert np.abs(obs[i, :o.size] - o).max() < tol * (1 + np.abs(o).max()), (names[i], t)
            assert abs(rew[i] - r) < tol * 100 * (1 + abs(r)), (names[i], t)
            assert bool(done[i]) == d, (names[i], t)
            assert cnt2[i, 2] == 0, "constraint rows dropped: " + names[i]
            rows_seen += int(cnt2[i, 3] >> 16)
            if d:                           # keep going from a fresh episode of the oracle
                oe.counters[1] += 1
                oe.reset()
