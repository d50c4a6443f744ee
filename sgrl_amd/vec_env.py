"""Batched VecEnv over the HIP rollout engine.

Keeps the Gym/baselines VecEnv surface the reference's trainer uses (reference src/subproc_vec_env.py:33-90,
base class baselines.common.vec_env.VecEnv): `num_envs`, `observation_space`, `action_space`, `reset()`,
`step_async()`, `step_wait()`, `step()`, `get_images()`, `reset_task()`, `close()`, flags `waiting` / `closed`.
Where SubprocVecEnv forks one MuJoCo process per environment, this class owns ONE engine handle that advances
all environments in a single kernel launch per step; observation padding and the action un-pad / reorder of
ModularEnvWrapper (reference src/wrappers.py:39-65) happen inside the kernel.

Two ways to drive it:
  * NumPy surface (drop-in for trainer.py): step(list of float arrays) -> (obs f32[n, obs_max_len], rews f32[n],
    dones bool[n], infos tuple of {"dist": ...}).  The reference returns float64; its callers cast to float32
    immediately (reference trainer.py:105-106,121-123).
  * Device surface (hot path): step_device(actions: torch.cuda.FloatTensor[n, action_max_len]) -> tensors on the GPU.
"""
import collections.abc
import ctypes

import numpy as np

from . import _lib, mjcf, model_pack
from .env_spec import env_spec_for


class Box(object):
    """Minimal stand-in for gym.spaces.Box (only .low/.high/.shape/.dtype are read by the reference trainer)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


class StepInfos(collections.abc.Sequence):
    """The `infos` of one step: a sequence of n dicts {"dist": float[, "TimeLimit.truncated": True][, "constraint_rows_dropped": k]}
    -- what the reference's workers send back (reference src/subproc_vec_env.py:12-16, the TimeLimit wrapper of utils.py:66-71) --
    whose dicts are MADE WHEN READ.  The reference's trainer reads `infos[i]['dist']` in the video path only
    (reference common/trainer.py:216); building 8192 dicts per step was 1.9 ms of a 4.6 ms step.  Indexing, slicing, iteration,
    len() and equality with a list / tuple of dicts behave like the tuple the reference returns."""
    __slots__ = ("_dist", "_trunc", "_extra")

    def __init__(self, dist, trunc, extra=None):
        self._dist, self._trunc, self._extra = dist, trunc, extra

    def __len__(self):
        return int(self._dist.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(len(self))))
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        d = {"dist": float(self._dist[i])}
        if self._trunc[i]:
            d["TimeLimit.truncated"] = True
        if self._extra and i in self._extra:
            d.update(self._extra[i])
        return d

    def __eq__(self, other):
        if isinstance(other, (list, tuple, StepInfos)):
            return len(other) == len(self) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __repr__(self):
        return "StepInfos(n=%d)" % len(self)


def _round_up(x, a=256):
    return (x + a - 1) // a * a


class VecEnv(object):
    """The baselines VecEnv contract (un-vendored third party in the reference): step = step_async + step_wait."""

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()


def resolve_models(env_names, xml_paths=None):
    """name -> compiled Model (from an XML path when given, else from the packaged assets)."""
    models = []
    for i, name in enumerate(env_names):
        if xml_paths is not None and xml_paths[i] is not None:
            models.append(mjcf.compile_mjcf(xml_paths[i], name=name.replace("_v2_", "_")))
        else:
            models.append(mjcf.load_asset(name.replace("_v2_", "_")))
    return models


class BatchedModularVecEnv(VecEnv):
    def __init__(self, env_names, envs_per_morph, obs_max_len=None, seed=0, device=None, max_episode_steps=1000,
                 xml_paths=None, env_id_base=0, max_rows=None,
                 pgs_iters=model_pack.DEFAULT_PGS_ITERS, pgs_tol=model_pack.DEFAULT_PGS_TOL):
        """env_names: morphology / environment names (e.g. '3d_walker_7_full'), in the order the reference sorts them
        (main.py:99); envs_per_morph: int or list; env i of morphology k has global index sum(counts[:k]) + i."""
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.SgrlError("BatchedModularVecEnv needs an MI355X: torch.cuda.is_available() is False "
                                 "(the engine has no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda:0")
        torch.cuda.set_device(self.device)
        self.env_names = list(env_names)
        counts = [envs_per_morph] * len(env_names) if np.isscalar(envs_per_morph) else list(envs_per_morph)
        assert len(counts) == len(env_names)
        self.counts = [int(c) for c in counts]
        self.models = resolve_models(self.env_names, xml_paths)
        self.num_limbs = [m.num_limbs for m in self.models]
        max_limbs = max(self.num_limbs)
        self.obs_max_len = int(obs_max_len) if obs_max_len else 41 * max_limbs
        self.action_max_len = 3 * max_limbs
        self.limb_obs_size, self.limb_action_size, self.max_action = 41, 3, 1.0
        n = sum(self.counts)
        self.env_morph = np.repeat(np.arange(len(env_names)), self.counts)
        self.morph_slices = []
        off = 0
        for c in self.counts:
            self.morph_slices.append(slice(off, off + c))
            off += c
        # spaces of worker 0 = first morphology (reference subproc_vec_env.py:50-52)
        L0 = self.num_limbs[0]
        VecEnv.__init__(self, n, Box(-np.inf, np.inf, (41 * L0,), np.float64), Box(-1.0, 1.0, (3 * (L0 - 1),), np.float32))
        self.waiting = False
        self.closed = False
        # constraint-row cap per morphology: multi-geom bodies (humanoid, cheetah) can touch the floor in many places
        rows_of = lambda m: max_rows if max_rows is not None else _lib.default_max_rows(m)
        self._blobs = [model_pack.pack_model(m, spec=env_spec_for(nm), max_rows=rows_of(m), pgs_iters=pgs_iters,
                                             pgs_tol=pgs_tol) for m, nm in zip(self.models, self.env_names)]
        L = _lib.lib()
        k = len(self._blobs)
        ibp = (ctypes.POINTER(ctypes.c_int32) * k)(*[b[0].ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) for b in self._blobs])
        fbp = (ctypes.POINTER(ctypes.c_double) * k)(*[b[1].ctypes.data_as(ctypes.POINTER(ctypes.c_double)) for b in self._blobs])
        ibl = (ctypes.c_int32 * k)(*[len(b[0]) for b in self._blobs])
        fbl = (ctypes.c_int32 * k)(*[len(b[1]) for b in self._blobs])
        cnt = (ctypes.c_int32 * k)(*self.counts)
        h = ctypes.c_void_p()
        _lib.check(L.sgrl_engine_create(k, ibp, ibl, fbp, fbl, cnt, self.obs_max_len, self.action_max_len,
                                        ctypes.c_uint64(int(seed)), ctypes.c_uint32(int(env_id_base)),
                                        int(max_episode_steps), ctypes.byref(h)), "sgrl_engine_create")
        self._h = h
        self._L = L
        self.stride = L.sgrl_record_stride(h)
        self.lds_bytes = L.sgrl_lds_bytes(h)
        self.launch_groups = L.sgrl_launch_groups(h)
        self.fixed_dim_groups = L.sgrl_fixed_dim_groups(h)     # launch groups on a fixed-dimension kernel (csrc/step_spec.hip)
        self.paired_envs = L.sgrl_paired_envs(h)               # environments that step two to a wavefront (csrc/wave_half.h)
        dev = self.device
        # the step's outputs are views of ONE device buffer (sections 256-byte aligned): the NumPy surface fetches them with a
        # single device-to-host copy per step (step_wait)
        secs = [("obs", 4 * n * self.obs_max_len), ("rew", 4 * n), ("dist", 4 * n), ("done", n), ("trunc", n)]
        self._out_off, off = {}, 0
        for name, nbytes in secs:
            self._out_off[name] = (off, nbytes)
            off += _round_up(nbytes)
        self._out = torch.zeros(off, dtype=torch.uint8, device=dev)
        sec = lambda name: self._out[self._out_off[name][0]:self._out_off[name][0] + self._out_off[name][1]]
        self.obs = sec("obs").view(torch.float32).view(n, self.obs_max_len)
        self.rew = sec("rew").view(torch.float32)
        self.dist = sec("dist").view(torch.float32)
        self.done = sec("done")
        self.trunc = sec("trunc")
        self._act = torch.zeros((n, self.action_max_len), dtype=torch.float32, device=dev)
        self._act_host = None            # pinned staging for the NumPy surface's actions (made on first use)
        self._fetch_event = torch.cuda.Event()
        self.obs64 = None
        self.rew64 = None
        self._overflow_seen = np.zeros(n, dtype=np.int64)
        self.overflow_check_every, self._steps_since_overflow_check = 16, 0

    # ---- device surface ---------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def enable_f64_outputs(self):
        """Allocate double-precision observation/reward mirrors (parity tests)."""
        t = self.torch
        self.obs64 = t.zeros((self.num_envs, self.obs_max_len), dtype=t.float64, device=self.device)
        self.rew64 = t.zeros(self.num_envs, dtype=t.float64, device=self.device)

    def _p(self, t):
        return ctypes.c_void_p(0 if t is None else t.data_ptr())

    def reset_device(self):
        _lib.check(self._L.sgrl_reset(self._h, self._p(self.obs), self._p(self.obs64), self._stream()), "sgrl_reset")
        return self.obs

    def step_device(self, actions, auto_reset=True):
        """actions: float32 CUDA tensor [n, action_max_len] (contiguous).  Returns (obs, rew, done, dist) tensors
        that are overwritten by the next call."""
        assert actions.is_cuda and actions.dtype == self.torch.float32 and actions.is_contiguous()
        assert tuple(actions.shape) == (self.num_envs, self.action_max_len)
        _lib.check(self._L.sgrl_step(self._h, self._p(actions), self._p(self.obs), self._p(self.rew), self._p(self.done),
                                     self._p(self.dist), self._p(self.trunc), self._p(self.obs64), self._p(self.rew64),
                                     int(bool(auto_reset)), self._stream()), "sgrl_step")
        return self.obs, self.rew, self.done, self.dist

    def time_steps(self, actions, reps):
        ms = ctypes.c_float(0)
        _lib.check(self._L.sgrl_time_steps(self._h, self._p(actions), self._p(self.obs), self._p(self.rew),
                                           self._p(self.done), int(reps), self._stream(), ctypes.byref(ms)),
                   "sgrl_time_steps")
        return float(ms.value)

    # ---- raw state (teacher forcing / checkpoints) ------------------------------------------------------
    def get_records(self):
        rec = np.zeros((self.num_envs, self.stride))
        cnt = np.zeros((self.num_envs, 4), dtype=np.int32)
        _lib.check(self._L.sgrl_get_records(self._h, ctypes.c_void_p(rec.ctypes.data), ctypes.c_void_p(cnt.ctypes.data)),
                   "sgrl_get_records")
        return rec, cnt

    def set_records(self, rec, cnt=None):
        rec = np.ascontiguousarray(rec, dtype=np.float64)
        assert rec.shape == (self.num_envs, self.stride)
        cp = None
        if cnt is not None:
            cnt = np.ascontiguousarray(cnt, dtype=np.int32)
            cp = ctypes.c_void_p(cnt.ctypes.data)
        _lib.check(self._L.sgrl_set_records(self._h, ctypes.c_void_p(rec.ctypes.data), cp), "sgrl_set_records")

    def get_counters(self):
        """int32 [n, 4] per env: step count, episode, constraint evaluations that hit the row cap (rows dropped), solver
        diagnostics of the last step."""
        cnt = np.zeros((self.num_envs, 4), dtype=np.int32)
        _lib.check(self._L.sgrl_get_records(self._h, None, ctypes.c_void_p(cnt.ctypes.data)), "sgrl_get_records")
        return cnt

    def row_overflow_envs(self):
        """Number of environments in which at least one dynamics evaluation wanted more constraint rows than `max_rows`
        (the extra contacts were dropped for that evaluation -- MuJoCo's nconmax/njmax analogue).  Zero on every
        shipped configuration at the default caps; check it after a run with custom caps or new morphologies."""
        return int((self.get_counters()[:, 2] > 0).sum())

    def refresh_device(self):
        _lib.check(self._L.sgrl_refresh(self._h, self._p(self.obs), self._p(self.obs64), self._stream()), "sgrl_refresh")
        return self.obs

    def state_of(self, rec, i):
        """(qpos, qvel, torso_xy_stale, target) views of env i in a get_records() array."""
        m = self.models[self.env_morph[i]]
        r = rec[i]
        return r[:m.nq], r[m.nq:m.nq + m.nv], r[m.nq + m.nv:m.nq + m.nv + 2], r[m.nq + m.nv + 2:m.nq + m.nv + 4]

    # ---- NumPy / Gym surface (reference subproc_vec_env.py:54-90) -------------------------------------------
    def reset(self):
        self.reset_device()
        return self.obs.cpu().numpy()       # a fresh host array (the device buffer is overwritten by the next step)

    def _host_actions(self, actions):
        """[n, action_max_len] float32 from what a trainer hands to step(): an ndarray, or the reference's list of n arrays
        (reference trainer.py:191-200).  A list of float32 arrays is flattened through the buffer protocol (bytes.join: 0.5 ms for
        8192 arrays, where np.asarray walks them as nested sequences: 1.4 ms)."""
        n, amax = self.num_envs, self.action_max_len
        a = None
        if not isinstance(actions, np.ndarray) and len(actions) == n and isinstance(actions[0], np.ndarray) \
                and actions[0].dtype == np.float32:
            try:
                buf = b"".join(actions)
                if len(buf) == 4 * n * amax:          # any float64 / longer / shorter row changes the total
                    a = np.frombuffer(buf, dtype=np.float32).reshape(n, amax)
            except (TypeError, BufferError, ValueError):
                a = None
        if a is None:
            a = np.asarray(actions, dtype=np.float32)
        if a.shape != (n, amax):
            raise ValueError("actions must be %d arrays of length action_max_len=%d (reference trainer.py:191-195)" % (n, amax))
        return a

    def step_async(self, actions):
        t = self.torch
        if t.is_tensor(actions) and actions.is_cuda:          # already on the device: no staging
            self.step_device(actions.to(t.float32).contiguous(), True)
            self.waiting = True
            return
        a = self._host_actions(actions)
        if self._act_host is None:
            self._act_host = t.empty((self.num_envs, self.action_max_len), dtype=t.float32).pin_memory()
        self._act_host.numpy()[...] = a
        self._act.copy_(self._act_host, non_blocking=True)
        self.step_device(self._act, True)
        self.waiting = True

    def step_wait(self):
        """(obs f32 [n, obs_max_len], rews f32 [n], dones bool [n], infos) as fresh host arrays: ONE device-to-host copy of the
        output buffer into a pinned block of PyTorch's caching host allocator (the arrays are views of it and own it: nothing
        is overwritten by the next step), infos made when read (StepInfos).  A caller that KEEPS a slice of these arrays keeps the whole
        block (9.5 MB of pinned memory at 8 192 walker environments) alive: copy what outlives the step, as the reference's replay
        buffer does (reference common/buffer.py:75-84 assigns rows into its own arrays)."""
        t = self.torch
        n = self.num_envs
        host = t.empty(self._out.shape, dtype=t.uint8, pin_memory=True)
        host.copy_(self._out, non_blocking=True)
        self._fetch_event.record(t.cuda.current_stream(self.device))
        self._fetch_event.synchronize()
        h = host.numpy()
        sec = lambda name: h[self._out_off[name][0]:self._out_off[name][0] + self._out_off[name][1]]
        obs = sec("obs").view(np.float32).reshape(n, self.obs_max_len)
        rews = sec("rew").view(np.float32)
        dones = sec("done").view(np.bool_)
        dist = sec("dist").view(np.float32)
        trunc = sec("trunc")
        extra = None
        # truncated contact sets must not go unnoticed: newly dropped constraint rows are reported in the info dict + a warning.  The
        # counters live in engine memory (a synchronous 128 KB copy of their own): looked at every 16th step -- a report comes at most
        # 15 steps late, the counts themselves lose nothing (row_overflow_envs() reads them at any time)
        self._steps_since_overflow_check += 1
        over = None
        if self._steps_since_overflow_check >= self.overflow_check_every:
            self._steps_since_overflow_check = 0
            over = self.get_counters()[:, 2]
        if over is not None and (over > self._overflow_seen).any():
            import warnings
            over = over.astype(np.int64)
            new = np.nonzero(over > self._overflow_seen)[0]
            extra = {int(i): {"constraint_rows_dropped": int(over[i] - self._overflow_seen[i])} for i in new}
            warnings.warn("%d environment(s) exceeded max_rows in this step: contacts were dropped (raise max_rows)" % new.size,
                          RuntimeWarning)
            self._overflow_seen = np.maximum(self._overflow_seen, over)
        self.waiting = False
        return obs, rews, dones, StepInfos(dist, trunc, extra)

    def reset_task(self):
        raise NotImplementedError("reset_task is not defined by the reference ModularEnv either (would raise in the worker)")

    def get_images(self, env_ids=None, width=256, height=256):
        """`SubprocVecEnv.get_images()` (reference src/subproc_vec_env.py:70-73): one RGB frame per environment, uint8
        [n, height, width, 3] (NumPy), through this repository's own ray caster (sgrl_amd/render.py: MuJoCo's renderer is a
        third-party dependency; pixel parity is not claimed).  env_ids: which environments (default: all -- at thousands of
        environments pass a subset); the reference's frames are 500 x 500, here the size is an argument."""
        from . import render
        ids = list(range(self.num_envs)) if env_ids is None else [int(i) for i in env_ids]
        rec, _ = self.get_records()
        scenes = []
        for i in ids:
            m = self.models[self.env_morph[i]]
            scenes.append(render.scene_of(m, rec[i, :m.nq]))
        return render.render(scenes, width=width, height=height, device=self.device).cpu().numpy()

    def close(self):
        if self.closed:
            return
        if getattr(self, "_h", None):
            self._L.sgrl_engine_destroy(self._h)
            self._h = None
        self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
