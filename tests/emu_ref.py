"""ctypes front-end of tests/emu/libsgrl_emu.so: the engine source (sgrl_amd/csrc/step_body.h) compiled with a
serial lane emulator.  TEST HARNESS ONLY -- lets the CPU-only container check the kernel logic against the oracle."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_f64p = ctypes.POINTER(ctypes.c_double)
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def lib():
    global _LIB
    if _LIB is None:
        d = os.path.join(_HERE, "emu")
        so = os.path.join(d, "libsgrl_emu.so")
        srcs = [os.path.join(d, "emu_step.cpp"), os.path.join(_HERE, "..", "sgrl_amd", "csrc", "step_body.h"),
                os.path.join(_HERE, "..", "include", "sgrl_model.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                                   srcs[0], "-lm"])
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class EmuEnv(object):
    def __init__(self, ib, fb, seed=0, env_id=0, max_episode_steps=1000, obs_max_len=None):
        self.ib = np.ascontiguousarray(ib, dtype=np.int32)
        self.fb = np.ascontiguousarray(fb, dtype=np.float64)
        self.nbody, self.njnt, self.nq, self.nv, self.nu = [int(v) for v in self.ib[1:6]]
        self.L = self.nbody - 1
        self.obs_max_len = obs_max_len or 41 * self.L
        self.rec = np.zeros(self.nq + self.nv + 4)
        self.cnt = np.zeros(4, dtype=np.int32)
        self.seed, self.env_id, self.max_episode_steps = seed, env_id, max_episode_steps

    qpos = property(lambda s: s.rec[:s.nq])
    qvel = property(lambda s: s.rec[s.nq:s.nq + s.nv])
    torso_xy_stale = property(lambda s: s.rec[s.nq + s.nv:s.nq + s.nv + 2])
    target = property(lambda s: s.rec[s.nq + s.nv + 2:s.nq + s.nv + 4])

    def _call(self, op, action=None, auto_reset=True):
        obs32 = np.zeros(self.obs_max_len, dtype=np.float32)
        obs64 = np.zeros(self.obs_max_len)
        rew = np.zeros(1)
        done = np.zeros(1, dtype=np.uint8)
        trunc = np.zeros(1, dtype=np.uint8)
        dist = np.zeros(1, dtype=np.float32)
        act = None if action is None else np.ascontiguousarray(action, dtype=np.float32)
        rc = lib().sgrl_emu_env(op, _p(self.ib, _i32p), _p(self.fb, _f64p), _p(self.rec, _f64p), _p(self.cnt, _i32p),
                                _p(act, _f32p), _p(obs32, _f32p), _p(obs64, _f64p), self.obs_max_len,
                                ctypes.c_uint64(self.seed), ctypes.c_uint32(self.env_id), self.max_episode_steps,
                                int(auto_reset), _p(rew, _f64p), _p(done, _u8p), _p(dist, _f32p), _p(trunc, _u8p))
        assert rc == 0
        return obs32, obs64, float(rew[0]), bool(done[0]), float(dist[0]), bool(trunc[0])

    def reset(self):
        return self._call(0)[1]

    def refresh(self):
        return self._call(2)[1]

    def step(self, action, auto_reset=True):
        o32, o64, r, d, dist, tr = self._call(1, action, auto_reset)
        return o64, r, d, {"dist": dist, "TimeLimit.truncated": tr, "obs32": o32, "overflow": int(self.cnt[2])}


def forward(ib, fb, qpos, qvel, ctrl):
    ib = np.ascontiguousarray(ib, dtype=np.int32)
    fb = np.ascontiguousarray(fb, dtype=np.float64)
    nv = int(ib[4])
    qpos = np.array(qpos, dtype=np.float64)
    qvel = np.ascontiguousarray(qvel, dtype=np.float64)
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
    qacc = np.zeros(nv)
    diag = np.zeros(4)
    rc = lib().sgrl_emu_forward(_p(ib, _i32p), _p(fb, _f64p), _p(qpos, _f64p), _p(qvel, _f64p), _p(ctrl, _f64p),
                                _p(qacc, _f64p), _p(diag, _f64p))
    assert rc == 0
    return qacc, {"ncon": int(diag[0]), "nrow": int(diag[1]), "nrow_wanted": int(diag[2]), "qpos": qpos}


def set_reverse(flag):
    lib().sgrl_emu_set_reverse(int(bool(flag)))


def set_linv(flag):
    """True (default): small systems take the explicit-inverse path, as HipWave does; False: always the L path."""
    lib().sgrl_emu_set_linv(int(bool(flag)))


# ---- two environments per wavefront (sgrl_amd/csrc/wave_half.h) on the SIMT fiber emulator (tests/emu/emu_pair.cpp) --------------
_PAIR_LIB = None


def pair_lib():
    global _PAIR_LIB
    if _PAIR_LIB is None:
        d = os.path.join(_HERE, "emu")
        so = os.path.join(d, "libsgrl_emu_pair.so")
        csrc = os.path.join(_HERE, "..", "sgrl_amd", "csrc")
        srcs = [os.path.join(d, "emu_pair.cpp"), os.path.join(csrc, "step_body.h"), os.path.join(csrc, "wave_half.h"),
                os.path.join(_HERE, "..", "include", "sgrl_model.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, srcs[0], "-lm"])
        _PAIR_LIB = ctypes.CDLL(so)
    return _PAIR_LIB


class PairEmu(object):
    """Two environments of ONE morphology stepped by one emulated wavefront (lanes 0..31 / 32..63)."""

    def __init__(self, ib, fb, seed=0, env_ids=(0, 1), max_episode_steps=1000, obs_max_len=None):
        self.envs = [EmuEnv(ib, fb, seed=seed, env_id=e, max_episode_steps=max_episode_steps, obs_max_len=obs_max_len) for e in env_ids]
        self.ib, self.fb = self.envs[0].ib, self.envs[0].fb
        self.seed, self.max_episode_steps = seed, max_episode_steps
        self.obs_max_len = self.envs[0].obs_max_len

    def layout_bytes(self):
        return pair_lib().sgrl_emu_pair_layout_bytes(_p(self.ib, _i32p), _p(self.fb, _f64p))

    def _call(self, op, actions=None, auto_reset=True):
        n = self.obs_max_len
        obs32 = [np.zeros(n, dtype=np.float32) for _ in range(2)]
        obs64 = [np.zeros(n) for _ in range(2)]
        rew = [np.zeros(1) for _ in range(2)]
        done = [np.zeros(1, dtype=np.uint8) for _ in range(2)]
        trunc = [np.zeros(1, dtype=np.uint8) for _ in range(2)]
        dist = [np.zeros(1, dtype=np.float32) for _ in range(2)]
        acts = None if actions is None else [np.ascontiguousarray(a, dtype=np.float32) for a in actions]

        def arr(xs, t):
            return (t * 2)(*[x.ctypes.data_as(t) for x in xs])
        ids = (ctypes.c_uint32 * 2)(*[e.env_id for e in self.envs])
        rc = pair_lib().sgrl_emu_pair_env(
            op, _p(self.ib, _i32p), _p(self.fb, _f64p), arr([e.rec for e in self.envs], _f64p), arr([e.cnt for e in self.envs], _i32p),
            None if acts is None else arr(acts, _f32p), arr(obs32, _f32p), arr(obs64, _f64p), n, ctypes.c_uint64(self.seed), ids,
            self.max_episode_steps, int(auto_reset), arr(rew, _f64p), arr(done, _u8p), arr(dist, _f32p), arr(trunc, _u8p))
        assert rc == 0, {-1: "bad model", -2: "the lanes of a half diverged (deadlock)", -3: "an access left its slab"}.get(rc, rc)
        return [(obs64[h], float(rew[h][0]), bool(done[h][0]), {"dist": float(dist[h][0]), "TimeLimit.truncated": bool(trunc[h][0]),
                                                                "obs32": obs32[h]}) for h in range(2)]

    def reset(self):
        return [r[0] for r in self._call(0)]

    def refresh(self):
        return [r[0] for r in self._call(2)]

    def step(self, actions, auto_reset=True):
        return self._call(1, actions, auto_reset)
