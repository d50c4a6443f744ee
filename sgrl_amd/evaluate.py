"""Batched evaluator: the bookkeeping of the reference's `BaseTrainer.evaluate`
(reference src/common/trainer.py:80-146) over a whole batch of environments at once.

The reference steps `num_envs_train` sub-environments (one per morphology) through an auto-resetting VecEnv for up to
`max_trajectory_length` steps per trajectory and keeps, per sub-environment,

  * `episode_timesteps`: steps until the FIRST done (the time limit `max_episode_steps` counts as done, :124-125),
  * `episode_reward`: the reward accumulated up to the first done -- latched only while it is still exactly 0 (:126-128;
    later dones of the auto-reset episodes re-latch only if the latched value is 0, with the accumulator restarted),
  * a trajectory contributes its per-env (length, return) pairs only if EVERY sub-environment was done at least once
    before `max_trajectory_length` (:139-145); otherwise it contributes nothing.

`evaluate()` returns the same dictionary (`performance/eval_return`, `performance/eval_length`: means over all
contributed pairs).  Arrays stay on the device of the environment; the only host sync per step is the `all(done)` test,
which the reference also performs.
"""
import numpy as np
import torch


class BatchedEvaluator(object):
    def __init__(self, env, act_fn, num_eval_trajectories=10, max_trajectory_length=1000, max_episode_steps=1000):
        """env: object with reset() -> obs [n, obs_len] and step(actions) -> (obs, reward [n], done [n], info);
        act_fn(obs) -> actions [n, action_len] (deterministic policy: Agent.select_action, reference agent.py:189-198).
        Tensors or NumPy arrays are accepted (NumPy is converted)."""
        self.env, self.act_fn = env, act_fn
        self.num_eval_trajectories = int(num_eval_trajectories)
        self.max_trajectory_length = int(max_trajectory_length)
        self.max_episode_steps = int(max_episode_steps)

    @torch.no_grad()
    def evaluate(self):
        returns, lengths = [], []
        for _ in range(self.num_eval_trajectories):
            obs = self.env.reset()
            n = int(obs.shape[0])
            dev = obs.device if torch.is_tensor(obs) else torch.device("cpu")
            done_ever = torch.zeros(n, dtype=torch.bool, device=dev)
            ep_reward = torch.zeros(n, dtype=torch.float64, device=dev)      # episode_reward_list
            ep_steps = torch.zeros(n, dtype=torch.int64, device=dev)         # episode_timesteps_list
            acc = torch.zeros(n, dtype=torch.float64, device=dev)            # episode_reward_list_buffer
            for _step in range(self.max_trajectory_length):
                obs, rew, done, _info = self.env.step(self.act_fn(obs))
                rew = torch.as_tensor(rew, device=dev).to(torch.float64).reshape(n)
                cur = torch.as_tensor(done, device=dev).to(torch.bool).reshape(n).clone()
                acc += rew
                cur |= (ep_steps + 1) == self.max_episode_steps
                latch = cur & (ep_reward == 0)
                ep_reward = torch.where(latch, acc, ep_reward)
                acc = torch.where(latch, torch.zeros_like(acc), acc)
                ep_steps += (~done_ever).to(torch.int64)
                done_ever |= cur
                if bool(done_ever.all()):
                    lengths.extend(ep_steps.tolist())
                    returns.extend(ep_reward.tolist())
                    break
        # np.mean of an empty list is nan (with a warning) in the reference as well
        return {"performance/eval_return": float(np.mean(returns)) if returns else float("nan"),
                "performance/eval_length": float(np.mean(lengths)) if lengths else float("nan")}
