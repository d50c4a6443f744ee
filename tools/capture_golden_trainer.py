#!/usr/bin/env python3
"""Golden vectors for the evaluator bookkeeping and the snapshot format: executes the reference's
BaseTrainer.evaluate / snapshot / load_snapshot (reference src/common/trainer.py:80-146, 249-322) on scripted
stand-ins for the environment, the agent and the logger, and stores inputs + outputs.  Build container only."""
import os, sys, types, tempfile, json
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import numpy as np
import torch
import refstub
refstub.install()
for name in ("cv2", "imageio", "wandb"):          # imported by common/trainer.py, unused by the methods captured here
    sys.modules.setdefault(name, types.ModuleType(name))
from common import trainer as ref_trainer
from common import util as ref_util
from common.buffer import ReplayBuffer
from gym.spaces import Box

# ---------------------------------------------------------------- evaluate()
names = ["m0", "m1", "m2", "m3"]
limbs = {"m0": 2, "m1": 3, "m2": 3, "m3": 5}
obs_max, act_max = 41 * 5, 3 * 5
cases = {}
rng = np.random.RandomState(11)
for case, (n_traj, max_len, max_ep, p_done) in {"mixed": (3, 40, 25, 0.08), "time_limit": (2, 30, 12, 0.0),
                                                "never_all_done": (2, 10, 1000, 0.02), "zero_reward_relatch": (2, 30, 20, 0.15)}.items():
    rew = rng.randn(n_traj, max_len, 4)
    if case == "zero_reward_relatch":
        rew[:, :8, 1] = 0.0              # env 1: first episode returns exactly 0 -> the reference re-latches later
    done = rng.rand(n_traj, max_len, 4) < p_done
    if case == "zero_reward_relatch":
        done[:, 3, 1] = True

    class Env(object):
        def __init__(self): self.t = -1; self.k = 0
        def reset(self):
            self.t += 1; self.k = 0
            return [np.zeros(obs_max) for _ in names]
        def step(self, actions):
            assert all(a.size == act_max for a in actions)
            r, d = rew[self.t, self.k], done[self.t, self.k]
            self.k += 1
            return [np.zeros(obs_max) for _ in names], list(r), list(d), [{} for _ in names]

    class Agent(object):
        def change_morphology(self, g): self.L = g
        def select_action(self, obs): return np.zeros(3 * self.L)

    args = types.SimpleNamespace(num_envs_train=4, envs_train_names=names, graph_dicts=limbs, limb_obs_size=41,
                                 graphs={k: [0] * v for k, v in limbs.items()}, action_max_len=act_max, max_episode_steps=max_ep)
    fake = types.SimpleNamespace(num_eval_trajectories=n_traj, eval_env=Env(), args=args, agent=Agent(), max_trajectory_length=max_len)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = ref_trainer.BaseTrainer.evaluate(fake)
    cases[case] = dict(rew=rew, done=done, n_traj=n_traj, max_len=max_len, max_ep=max_ep,
                       eval_return=out["performance/eval_return"], eval_length=out["performance/eval_length"])
    print(case, out)
flat = {}
for c, d in cases.items():
    for k, v in d.items():
        flat[c + "__" + k] = np.asarray(v)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "evaluator.npz"), **flat)

# ---------------------------------------------------------------- snapshot() / load_snapshot()
tmp = tempfile.mkdtemp()
ref_util.logger = types.SimpleNamespace(log_path=tmp)
ref_util.device = torch.device("cpu")
env_names = ["3d_walker_3_x", "3d_hopper_3_shin"]
bufs, script = {}, {}
rng = np.random.RandomState(5)
for nm, L in zip(env_names, (3, 3)):
    rb = ReplayBuffer(Box(-np.ones(41 * L), np.ones(41 * L)), Box(-np.ones(3 * (L - 1)), np.ones(3 * (L - 1))), max_buffer_size=6, modular=True)
    n = 8
    tr = dict(obs=rng.rand(n, 41 * L).astype(np.float32), act=rng.rand(n, 3 * L).astype(np.float32), nxt=rng.rand(n, 41 * L).astype(np.float32),
              rew=rng.rand(n).astype(np.float32), done=(rng.rand(n) > 0.5).astype(np.float32))
    for i in range(n):
        rb.add_transition(tr["obs"][i], tr["act"][i], tr["nxt"][i], tr["rew"][i], tr["done"][i])
    bufs[nm] = rb; script[nm] = tr
state = {"actor": {"w": torch.arange(6.).reshape(2, 3)}, "critic": {"b": torch.ones(2)}}
agent = types.SimpleNamespace(state_dict=lambda: state, load_state_dict=lambda s: None)
fake = types.SimpleNamespace(agent=agent, tot_env_steps=12345, env_buffer=bufs,
                             args=types.SimpleNamespace(num_envs_train=2, envs_train_names=env_names, load_buffer=True))
ref_trainer.BaseTrainer.snapshot(fake, 0)
files = sorted(os.listdir(os.path.join(tmp, "models")))
ck = torch.load(os.path.join(tmp, "models", "save.pth"), weights_only=False)
meta = {"files": files, "checkpoint_keys": sorted(ck.keys()), "env_names": env_names, "tot_env_steps": 12345,
        "scalars": {k: int(ck[k]) for k in ck if k.endswith("curr") or k.endswith("max_sample_size")},
        "npy": {}}
arrs = {}
for f in files:
    if f.endswith(".npy"):
        a = np.load(os.path.join(tmp, "models", f))
        meta["npy"][f] = {"dtype": str(a.dtype), "shape": list(a.shape)}
        arrs["file__" + f] = a
for nm in env_names:
    for k, v in script[nm].items():
        arrs["script__%s__%s" % (nm, k)] = v
np.savez_compressed(os.path.join(REPO, "tests", "golden", "snapshot.npz"), **arrs)
json.dump(meta, open(os.path.join(REPO, "tests", "golden", "snapshot_meta.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(meta)[:600])
