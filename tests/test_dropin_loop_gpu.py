"""Drop-in check: a miniature of the reference's own driver loops run unchanged in structure against the new VecEnv and
SEPolicy -- `Trainer.warmup` (reference src/trainer.py:90-138: random actions, per-env bookkeeping, reset when all envs
finished) and the per-env `select_action` loop of `Trainer.train` (src/trainer.py:173-236) with `Agent.select_action`
(src/agent.py:189-198) and `change_morphology` (src/agent.py:201-205).  One env per morphology, NumPy in/out."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Buffer(object):
    """ReplayBuffer.add_transition layout (reference common/buffer.py:36-84, modular=True: action_dim = nu + 3)."""

    def __init__(self, obs_dim, action_dim, size=4096):
        self.obs = np.zeros((size, obs_dim), np.float32)
        self.act = np.zeros((size, action_dim), np.float32)
        self.nxt = np.zeros((size, obs_dim), np.float32)
        self.rew = np.zeros(size, np.float32)
        self.done = np.zeros(size, np.float32)
        self.curr = 0

    def add_transition(self, obs, action, next_obs, reward, done):
        i = self.curr
        self.obs[i], self.act[i], self.nxt[i], self.rew[i], self.done[i] = obs, action, next_obs, reward, done
        self.curr += 1


def test_reference_style_warmup_and_policy_loops():
    import torch
    from sgrl_amd import graph as G
    from sgrl_amd.set_policy import make_policy
    from sgrl_amd.vec_env import BatchedModularVecEnv
    names = sorted(["3d_hopper_3_shin", "3d_hopper_4_lower_shin", "3d_hopper_5_full"])   # config 2's morphologies
    graphs = {n: None for n in names}
    envs_train = BatchedModularVecEnv(names, 1, seed=0, device="cuda:0", max_episode_steps=1000)
    num_envs = envs_train.num_envs
    limb_obs_size, limb_action_size = envs_train.limb_obs_size, envs_train.limb_action_size
    action_max_len = envs_train.action_max_len
    for n, m in zip(names, envs_train.models):
        graphs[n] = m.parents
    graph_dicts = {n: G.getGraphDict(graphs[n], ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cuda:0")) for n in names}
    buffers = {n: _Buffer(41 * len(graphs[n]), 3 * len(graphs[n])) for n in names}
    rng = np.random.RandomState(0)

    # ---- Trainer.warmup (trainer.py:90-138) ----
    obs_list = envs_train.reset()
    done_list = [False] * num_envs
    episode_timesteps_list = [0] * num_envs
    resets = 0
    for step in range(150):
        action_list = [rng.uniform(low=envs_train.action_space.low[0], high=envs_train.action_space.high[0],
                                   size=action_max_len) for _ in range(num_envs)]
        new_obs_list, reward_list, curr_done_list, _ = envs_train.step(action_list)
        reward_list = reward_list.astype(np.float32)
        curr_done_list = curr_done_list.astype(np.float32)
        for i in range(num_envs):
            done_bool = curr_done_list[i]
            if episode_timesteps_list[i] + 1 == 1000:
                done_bool = 0
                curr_done_list[i] = True
            if not done_list[i]:
                episode_timesteps_list[i] += 1
                num_limbs = len(graphs[names[i]])
                obs = np.array(obs_list[i][:limb_obs_size * num_limbs]).astype(np.float32)
                new_obs = np.array(new_obs_list[i][:limb_obs_size * num_limbs]).astype(np.float32)
                action = np.array(action_list[i][:limb_action_size * num_limbs]).astype(np.float32)
                buffers[names[i]].add_transition(obs, action, new_obs, reward_list[i], done_bool)
                done_list[i] = done_list[i] or curr_done_list[i]
        obs_list = new_obs_list
        if all(done_list):
            obs_list = envs_train.reset()
            done_list = [False] * num_envs
            episode_timesteps_list = [0] * num_envs
            resets += 1
    assert resets >= 1                                  # random actions: every hopper falls well within 150 steps
    assert all(b.curr > 10 for b in buffers.values())
    for n, b in buffers.items():
        assert np.isfinite(b.obs[:b.curr]).all() and np.isfinite(b.rew[:b.curr]).all()
        assert b.done[:b.curr].sum() >= 1               # terminal transitions were stored with done = 1
        # stored next_obs of a terminal transition is the RESET observation (reference subproc_vec_env.py:12-15)
        assert (b.act[:b.curr, :3] != 0).any()          # torso dummy slots carry the random values, as in the reference

    # ---- per-env select_action loop (trainer.py:173-200, agent.py:189-205) ----
    actor = make_policy(device="cuda:0").eval()

    @torch.no_grad()
    def select_action(obs):
        if len(obs.shape) == 1:
            obs = obs[None, ]
        obs = torch.FloatTensor(obs).to("cuda:0")
        return actor(obs).cpu().numpy()

    obs_list = envs_train.reset()
    for env_step in range(5):
        action_list = []
        for i in range(num_envs):
            actor.change_morphology(graph_dicts[names[i]])
            obs = np.array(obs_list[i][:limb_obs_size * len(graphs[names[i]])])
            action = select_action(obs)
            assert action.shape == (1, 3 * len(graphs[names[i]]))
            action = (action + rng.normal(0, 0.126, size=action.size)).clip(envs_train.action_space.low[0],
                                                                          envs_train.action_space.high[0])
            action = np.append(action, np.array([0 for _ in range(action_max_len - action.size)]))
            action_list.append(action)
        new_obs_list, reward_list, curr_done_list, infos = envs_train.step(action_list)
        assert new_obs_list.shape == (num_envs, envs_train.obs_max_len)
        assert "dist" in infos[0]
        obs_list = new_obs_list
    assert actor._hip is not None   # select_action ran on the HIP path
    envs_train.close()


def test_batched_evaluator_on_the_engine_matches_a_per_env_replay_of_the_reference_loop():
    """sgrl_amd.evaluate.BatchedEvaluator over Rollout (device tensors, thousands of envs at once) vs the reference's
    per-env bookkeeping (trainer.py:80-146) replayed in plain Python on the recorded reward / done streams."""
    import torch
    from sgrl_amd.evaluate import BatchedEvaluator
    from sgrl_amd.rollout import Rollout
    from sgrl_amd.set_policy import make_policy
    names = sorted(["3d_hopper_3_shin", "3d_hopper_4_lower_shin", "3d_hopper_5_full"])
    torch.manual_seed(0)
    ro = Rollout(names, 8, policy=make_policy(device="cuda:0").eval(), seed=3, device="cuda:0", max_episode_steps=60)

    class Rec(object):
        def __init__(self): self.trajs = []
        def reset(self):
            self.trajs.append([])
            return ro.reset()
        def step(self, a):
            obs, rew, done, dist = ro.step(a)
            self.trajs[-1].append((rew.double().cpu().numpy().copy(), done.cpu().numpy().astype(bool).copy()))
            return obs, rew, done, dist

    rec = Rec()
    out = BatchedEvaluator(rec, ro.policy_forward, num_eval_trajectories=2, max_trajectory_length=80, max_episode_steps=60).evaluate()
    n = ro.env.num_envs
    rets, lens = [], []
    for traj in rec.trajs:
        done_list, ep_rew, ep_t, buf = [False] * n, [0.0] * n, [0] * n, [0.0] * n
        for rew, cur in traj:
            cur = list(cur)
            for i in range(n):
                buf[i] += rew[i]
                if ep_t[i] + 1 == 60:
                    cur[i] = True
                if cur[i] and ep_rew[i] == 0:
                    ep_rew[i] = buf[i]
                    buf[i] = 0
                if not done_list[i]:
                    ep_t[i] += 1
                    done_list[i] = done_list[i] or cur[i]
            if all(done_list):
                lens.extend(ep_t)
                rets.extend(ep_rew)
                break
    assert len(lens) == 2 * n                      # the 60-step time limit guarantees completion within 80 steps
    assert out["performance/eval_length"] == pytest.approx(np.mean(lens), rel=1e-12)
    assert out["performance/eval_return"] == pytest.approx(np.mean(rets), rel=1e-9)
    assert np.isfinite(out["performance/eval_return"])
