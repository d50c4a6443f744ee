"""Batched evaluator and snapshot format against vectors produced by executing the reference's
BaseTrainer.evaluate / snapshot (tools/capture_golden_trainer.py; reference src/common/trainer.py:80-146, 249-322)."""
import json
import os

import numpy as np
import pytest
import torch

from sgrl_amd.evaluate import BatchedEvaluator
from sgrl_amd.replay import DeviceReplayBuffer
from sgrl_amd.snapshot import load_snapshot, save_snapshot

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _ScriptedEnv(object):
    """Batched stand-in with the VecEnv surface: rewards / dones replayed from the fixture."""

    def __init__(self, rew, done, as_tensor):
        self.rew, self.done, self.as_tensor = rew, done, as_tensor
        self.t, self.k = -1, 0

    def _wrap(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)) if self.as_tensor else a

    def reset(self):
        self.t += 1
        self.k = 0
        return self._wrap(np.zeros((self.rew.shape[2], 7), dtype=np.float32))

    def step(self, actions):
        r, d = self.rew[self.t, self.k], self.done[self.t, self.k]
        self.k += 1
        return self._wrap(np.zeros((self.rew.shape[2], 7), dtype=np.float32)), self._wrap(r), self._wrap(d), {}


@pytest.mark.parametrize("case", ["mixed", "time_limit", "never_all_done", "zero_reward_relatch"])
@pytest.mark.parametrize("as_tensor", [False, True])
def test_batched_evaluator_reproduces_the_reference_bookkeeping(case, as_tensor):
    z = np.load(os.path.join(GOLD, "evaluator.npz"))
    g = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(case + "__")}
    env = _ScriptedEnv(g["rew"], g["done"], as_tensor)
    ev = BatchedEvaluator(env, lambda obs: obs, num_eval_trajectories=int(g["n_traj"]), max_trajectory_length=int(g["max_len"]),
                          max_episode_steps=int(g["max_ep"]))
    out = ev.evaluate()
    for key, ref in (("performance/eval_return", float(g["eval_return"])), ("performance/eval_length", float(g["eval_length"]))):
        if np.isnan(ref):
            assert np.isnan(out[key]), (case, key)
        else:
            assert out[key] == pytest.approx(ref, rel=1e-12, abs=1e-12), (case, key)


def _filled_buffers(z, names):
    bufs = {}
    for nm in names:
        tr = {k: z["script__%s__%s" % (nm, k)] for k in ("obs", "act", "nxt", "rew", "done")}
        b = DeviceReplayBuffer(tr["obs"].shape[1], tr["act"].shape[1], max_buffer_size=6)
        for i in range(tr["obs"].shape[0]):
            b.add_transition(tr["obs"][i], tr["act"][i], tr["nxt"][i], tr["rew"][i], tr["done"][i])
        bufs[nm] = b
    return bufs


def test_snapshot_files_keys_and_arrays_match_the_reference(tmp_path):
    z = np.load(os.path.join(GOLD, "snapshot.npz"))
    meta = json.load(open(os.path.join(GOLD, "snapshot_meta.json")))
    names = meta["env_names"]
    bufs = _filled_buffers(z, names)
    state = {"actor": {"w": torch.arange(6.).reshape(2, 3)}, "critic": {"b": torch.ones(2)}}
    d = str(tmp_path / "models")
    path = save_snapshot(d, state, meta["tot_env_steps"], names, bufs)
    assert sorted(os.listdir(d)) == meta["files"]
    ck = torch.load(path, weights_only=False)
    assert sorted(ck.keys()) == meta["checkpoint_keys"]
    assert ck["tot_env_steps"] == meta["tot_env_steps"]
    for k, v in meta["scalars"].items():
        assert int(ck[k]) == v, k
    for f, info in meta["npy"].items():
        a = np.load(os.path.join(d, f))
        assert str(a.dtype) == info["dtype"] and list(a.shape) == info["shape"], f
        assert np.array_equal(a, z["file__" + f]), f


def test_snapshot_round_trip_restores_buffers_and_pointers(tmp_path):
    z = np.load(os.path.join(GOLD, "snapshot.npz"))
    meta = json.load(open(os.path.join(GOLD, "snapshot_meta.json")))
    names = meta["env_names"]
    bufs = _filled_buffers(z, names)
    d = str(tmp_path / "models")
    path = save_snapshot(d, {"actor": {"w": torch.ones(1)}}, 77, names, bufs)
    fresh = {nm: DeviceReplayBuffer(bufs[nm].obs_dim, bufs[nm].action_dim, max_buffer_size=6) for nm in names}
    state, steps = load_snapshot(path, names, fresh)
    assert steps == 77 and torch.equal(state["actor"]["w"], torch.ones(1))
    for nm in names:
        assert fresh[nm].curr == bufs[nm].curr and fresh[nm].max_sample_size == bufs[nm].max_sample_size
        for f in ("obs_buffer", "action_buffer", "next_obs_buffer", "reward_buffer", "done_buffer"):
            assert torch.equal(getattr(fresh[nm], f), getattr(bufs[nm], f)), (nm, f)
    with pytest.raises(FileNotFoundError):
        load_snapshot(os.path.join(d, "missing.pth"))
