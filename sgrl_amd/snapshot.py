"""Snapshot / resume in the reference's on-disk format (reference src/common/trainer.py:249-322).

    <dir>/save.pth                              torch.save({"agent": state_dict, "tot_env_steps": int,
                                                            "<env>max_sample_size": int, "<env>curr": int, ...})
    <dir>/save_<env>_{obs,action,next_obs,reward,done}_buffer.npy      np.save(..., allow_pickle=False)

so that a run started with the reference can be resumed here and vice versa.  Buffers are anything exposing the
reference ReplayBuffer's fields (`obs_buffer`, `action_buffer`, `next_obs_buffer`, `reward_buffer`, `done_buffer`,
`curr`, `max_sample_size`) -- `sgrl_amd.replay.DeviceReplayBuffer` does, with device tensors.
"""
import os

import numpy as np
import torch

_FIELDS = ("obs_buffer", "action_buffer", "next_obs_buffer", "reward_buffer", "done_buffer")


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def save_snapshot(save_dir, agent_state_dict, tot_env_steps, env_names, buffers):
    """buffers: dict env name -> replay buffer.  Returns the path of save.pth."""
    os.makedirs(save_dir, exist_ok=True)
    model_path = os.path.join(save_dir, "save.pth")
    rb_path = os.path.join(save_dir, "save_")
    checkpoint = {"agent": agent_state_dict, "tot_env_steps": tot_env_steps}
    for name in env_names:
        b = buffers[name]
        checkpoint[name + "max_sample_size"] = b.max_sample_size
        checkpoint[name + "curr"] = b.curr
        for f in _FIELDS:
            np.save(rb_path + name + "_" + f + ".npy", _np(getattr(b, f)), allow_pickle=False)
    torch.save(checkpoint, model_path)
    return model_path


def load_snapshot(load_path, env_names=(), buffers=None, map_location="cpu"):
    """Returns (agent_state_dict, tot_env_steps).  With `buffers` (dict env name -> buffer) the replay contents and
    ring pointers are restored as the reference does under --load_buffer (obs cast to float32, trainer.py:305-307)."""
    if not os.path.exists(load_path):
        raise FileNotFoundError("snapshot not found: %s" % load_path)
    checkpoint = torch.load(load_path, map_location=map_location, weights_only=False)
    if buffers is not None:
        rb_path = load_path.replace(".pth", "_")
        for name in env_names:
            b = buffers[name]
            b.max_sample_size = int(checkpoint[name + "max_sample_size"])
            b.curr = int(checkpoint[name + "curr"])
            for f in _FIELDS:
                arr = np.load(rb_path + name + "_" + f + ".npy")
                if f == "obs_buffer":
                    arr = arr.astype(np.float32)
                cur = getattr(b, f)
                if torch.is_tensor(cur):
                    cur.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(cur.dtype))
                else:
                    setattr(b, f, arr)
    return checkpoint["agent"], checkpoint["tot_env_steps"]
