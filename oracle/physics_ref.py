"""ctypes front-end of oracle/libsgrl_oracle.so (TEST INFRASTRUCTURE; see oracle/physics.c header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libsgrl_oracle.so")
    src = os.path.join(_HERE, "physics.c")
    hdr = os.path.join(_HERE, "..", "include", "sgrl_model.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsgrl_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.sgrl_oracle_rng_uniform01.restype = ctypes.c_double
        _LIB.sgrl_oracle_rng_uniform01.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                                   ctypes.c_uint32]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _f(a):
    return None if a is None else a.ctypes.data_as(_f64p)


class OracleModel(object):
    def __init__(self, ib, fb):
        self.ib = np.ascontiguousarray(ib, dtype=np.int32)
        self.fb = np.ascontiguousarray(fb, dtype=np.float64)
        self.nbody, self.njnt, self.nq, self.nv, self.nu = [int(v) for v in self.ib[1:6]]
        self.L = self.nbody - 1

    def _m(self):
        return _p(self.ib, _i32p), _p(self.fb, _f64p)

    def forward(self, qpos, qvel, ctrl):
        qpos = np.array(qpos, dtype=np.float64)
        qvel = np.ascontiguousarray(qvel, dtype=np.float64)
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
        qacc = np.zeros(self.nv)
        M = np.zeros((self.nv, self.nv))
        diag = np.zeros(8)
        rc = lib().sgrl_oracle_forward(*self._m(), _f(qpos), _f(qvel), _f(ctrl), _f(qacc), _f(M), _f(diag))
        assert rc == 0
        return qacc, M, {"ncon": int(diag[0]), "nrow": int(diag[1]), "nrow_wanted": int(diag[2]),
                         "pgs_last_change": diag[3], "com": diag[4:7].copy(), "qpos": qpos,
                         "pgs_iters": int(diag[7])}

    def mj_step(self, qpos, qvel, ctrl, nsteps=1):
        qpos = np.array(qpos, dtype=np.float64)
        qvel = np.array(qvel, dtype=np.float64)
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
        kin = np.zeros(3 * self.nbody * 3 + 3 * self.njnt)
        ov = lib().sgrl_oracle_mj_step(*self._m(), _f(qpos), _f(qvel), _f(ctrl), int(nsteps), _f(kin))
        nb, nj = self.nbody, self.njnt
        out = {"xpos": kin[:3 * nb].reshape(nb, 3), "xaxis": kin[3 * nb:3 * nb + 3 * nj].reshape(nj, 3),
               "xvelp": kin[3 * nb + 3 * nj:6 * nb + 3 * nj].reshape(nb, 3),
               "xvelr": kin[6 * nb + 3 * nj:].reshape(nb, 3), "overflow": ov}
        return qpos, qvel, out

    def energy(self, qpos, qvel):
        qpos = np.array(qpos, dtype=np.float64)
        out = np.zeros(2)
        lib().sgrl_oracle_energy(*self._m(), _f(qpos), _f(np.ascontiguousarray(qvel, dtype=np.float64)), _f(out))
        return out[0], out[1]

    def env_epilogue(self, quat_before, pos_before, env_action, xpos, xvelp, xvelr, xaxis, qpos, qvel, target):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        obs = np.zeros(41 * self.L)
        rew = np.zeros(1)
        dist = np.zeros(1)
        done = lib().sgrl_oracle_env_epilogue(*self._m(), _f(c(quat_before)), _f(c(pos_before)), _f(c(env_action)),
                                              _f(c(xpos)), _f(c(xvelp)), _f(c(xvelr)), _f(c(xaxis)), _f(c(qpos)),
                                              _f(c(qvel)), _f(c(target)), _f(obs), _f(rew), _f(dist))
        return obs, float(rew[0]), bool(done), float(dist[0])


class OracleEnv(object):
    """One environment stepped by the oracle (mirrors one SubprocVecEnv worker, reference subproc_vec_env.py:6-30)."""

    def __init__(self, model, seed=0, env_id=0, max_episode_steps=1000):
        self.m = model
        n = lib().sgrl_oracle_sizeof_env()
        self.buf = np.zeros(n // 8 + 2, dtype=np.float64)
        self.seed, self.env_id, self.max_episode_steps = int(seed), int(env_id), int(max_episode_steps)

    # raw views into the C struct: qpos[49], qvel[48], torso_xy_stale[2], target[2], (int32 step_count, episode)
    @property
    def qpos(self):
        return self.buf[0:self.m.nq]

    @property
    def qvel(self):
        return self.buf[49:49 + self.m.nv]

    @property
    def torso_xy_stale(self):
        return self.buf[97:99]

    @property
    def target(self):
        return self.buf[99:101]

    @property
    def counters(self):
        return self.buf[101:102].view(np.int32)

    def reset(self):
        obs = np.zeros(41 * self.m.L)
        lib().sgrl_oracle_env_reset(*self.m._m(), ctypes.c_void_p(self.buf.ctypes.data), ctypes.c_uint64(self.seed),
                                    ctypes.c_uint32(self.env_id), _f(obs))
        return obs

    def refresh(self):
        obs = np.zeros(41 * self.m.L)
        lib().sgrl_oracle_env_refresh(*self.m._m(), ctypes.c_void_p(self.buf.ctypes.data), _f(obs))
        return obs

    def step(self, action, auto_reset=True):
        action = np.ascontiguousarray(action, dtype=np.float64)
        assert action.size >= 3 * self.m.L
        obs = np.zeros(41 * self.m.L)
        rew = np.zeros(1)
        info = np.zeros(4)
        done = lib().sgrl_oracle_env_step(*self.m._m(), ctypes.c_void_p(self.buf.ctypes.data), _f(action),
                                          ctypes.c_uint64(self.seed), ctypes.c_uint32(self.env_id),
                                          int(self.max_episode_steps), int(bool(auto_reset)), _f(obs), _f(rew), _f(info))
        assert done >= 0
        return obs, float(rew[0]), bool(done), {"dist": float(info[0]), "overflow": int(info[1]),
                                                "TimeLimit.truncated": bool(info[2])}
