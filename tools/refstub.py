"""Import harness for the read-only reference at /root/reference (THIS CONTAINER ONLY).

The reference (alpc91/SGRL) needs gym 0.17.2, mujoco-py, baselines, xmltodict and wandb,
none of which exist in this image.  This module injects the smallest stand-ins that let the
reference's *own* Python run for the parts of the hot path that are pure arithmetic:

  * utils.getGraphStructure / getGraphDict / getGraphJoints / getMotorJoints   (src/utils.py:236-484)
  * wrappers.ModularEnvWrapper (action_order, padding)                         (src/wrappers.py:7-65)
  * SEActor.SEPolicy forward                                                   (src/SEActor.py:290-356)
  * environments/<name>.py ModularEnv.step/_get_obs/reset_model on a FAKE simulator
    (kinematic quantities are injected; MuJoCo itself is not available -> physics parity unpinned)

Nothing from the reference is copied: it is imported from where it lies, executed, and only
its numeric inputs/outputs are written to tests/golden/ by tools/capture_golden.py.
This file never travels to the GPU box in any useful sense (it needs /root/reference).
"""
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

REF_ROOT = "/root/reference"
REF_SRC = os.path.join(REF_ROOT, "src")


def _etree_to_xmltodict(elem):
    """xmltodict.parse() shape: '@attr' keys, repeated children -> list, single child -> dict."""
    d = {}
    for k, v in elem.attrib.items():
        d["@" + k] = v
    for child in elem:
        if not isinstance(child.tag, str):
            continue  # comments / PIs
        c = _etree_to_xmltodict(child)
        if child.tag in d:
            if not isinstance(d[child.tag], list):
                d[child.tag] = [d[child.tag]]
            d[child.tag].append(c)
        else:
            d[child.tag] = c
    if not d:
        return None
    return d


def _xmltodict_parse(text):
    root = ET.fromstring(text)
    return {root.tag: _etree_to_xmltodict(root)}


class _Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = dtype


class _Wrapper(object):
    """gym.Wrapper stand-in: stores env and forwards attribute access (gym 0.17.2 behaviour)."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)


class _EzPickle(object):
    def __init__(self, *a, **k):
        pass


class _FakeMujocoEnv(object):
    """gym.envs.mujoco.mujoco_env.MujocoEnv stand-in: NO physics.

    A test harness installs `sim`, `data`, `model`, `np_random`, `init_qpos`, `init_qvel` and a
    `_after` snapshot that do_simulation() swaps in (the reference env only *reads* kinematic
    quantities from the simulator around the do_simulation call).
    """

    def __init__(self, xml, frame_skip):
        self.frame_skip = frame_skip
        self.fullpath = xml

    @property
    def dt(self):
        return self.model.opt.timestep * self.frame_skip

    def do_simulation(self, ctrl, n_frames):
        self._ctrl = np.array(ctrl, dtype=np.float64)
        self._n_frames = n_frames
        self.sim.data = self._after
        self.data = self._after

    def state_vector(self):
        return np.concatenate([self.sim.data.qpos.flat, self.sim.data.qvel.flat])

    def set_state(self, qpos, qvel):
        self._set_state_args = (np.array(qpos, dtype=np.float64), np.array(qvel, dtype=np.float64))
        if hasattr(self, "_on_set_state"):
            self._on_set_state(*self._set_state_args)


_INSTALLED = False


def install():
    """Inject the stand-in modules and put the reference on sys.path.  Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference not present at %s (this tool only runs in the build container)" % REF_SRC)
    sys.dont_write_bytecode = True  # never leave __pycache__ in the read-only tree
    import torch  # noqa: F401  (reference imports it)

    xd = types.ModuleType("xmltodict")
    xd.parse = _xmltodict_parse

    gym = types.ModuleType("gym")
    gym.Wrapper = _Wrapper
    gym.Space = object
    gym.Env = object
    gym.make = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("gym.make is not available in the stub"))
    gym_spaces = types.ModuleType("gym.spaces")
    gym_spaces.Box = _Box
    gym_spaces.Discrete = type("Discrete", (), {})
    gym_spaces.MultiBinary = type("MultiBinary", (), {})     # imported by common/networks.py:10, unused by the SET path
    gym_space = types.ModuleType("gym.spaces.space")
    gym_space.Space = object
    gym_spaces.space = gym_space
    gym_box = types.ModuleType("gym.spaces.box")
    gym_box.Box = _Box
    gym_disc = types.ModuleType("gym.spaces.discrete")
    gym_disc.Discrete = gym_spaces.Discrete
    gym_spaces.box = gym_box
    gym_spaces.discrete = gym_disc
    gym.spaces = gym_spaces
    gym_utils = types.ModuleType("gym.utils")
    gym_utils.EzPickle = _EzPickle
    gym.utils = gym_utils
    gym_envs = types.ModuleType("gym.envs")
    gym_reg = types.ModuleType("gym.envs.registration")
    gym_reg.register = lambda **k: None
    gym_muj = types.ModuleType("gym.envs.mujoco")
    gym_mujenv = types.ModuleType("gym.envs.mujoco.mujoco_env")
    gym_mujenv.MujocoEnv = _FakeMujocoEnv
    gym_muj.mujoco_env = gym_mujenv
    gym_envs.registration = gym_reg
    gym_envs.mujoco = gym_muj
    gym.envs = gym_envs

    sys.modules.update({
        "xmltodict": xd,
        "gym": gym,
        "gym.spaces": gym_spaces,
        "gym.spaces.box": gym_box,
        "gym.spaces.discrete": gym_disc,
        "gym.spaces.space": gym_space,
        "gym.utils": gym_utils,
        "gym.envs": gym_envs,
        "gym.envs.registration": gym_reg,
        "gym.envs.mujoco": gym_muj,
        "gym.envs.mujoco.mujoco_env": gym_mujenv,
    })
    sys.path.insert(0, REF_SRC)
    os.chdir(REF_SRC)
    from common import util
    import torch
    util.device = torch.device("cpu")
    _INSTALLED = True


def all_xmls():
    """name -> path for the distinct morphologies shipped by the reference (29)."""
    base = os.path.join(REF_SRC, "environments")
    out = {}
    for sub in ["3d_hoppers", "3d_walkers", "3d_humanoids", "3d_cheetahs", "zero_shot"]:
        d = os.path.join(base, sub)
        for f in sorted(os.listdir(d)):
            if f.endswith(".xml"):
                out[f[:-4]] = os.path.join(d, f)
    return out
