"""Loader for the HIP extension (libsgrl_hip.so, C ABI in include/sgrl.h).

There is deliberately NO CPU fallback: if the shared library is missing, cannot be loaded, or no MI355X is
visible, every entry point raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from here.)
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGRL_HIP_LIB", os.path.join(_HERE, "libsgrl_hip.so"))   # override: A/B benchmarking of builds
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["engine.hip", "set_actor.hip", "train_gemm.hip", "render.hip"]

_lib = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_f64p = ctypes.POINTER(ctypes.c_double)


class SgrlError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into sgrl_amd/libsgrl_hip.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(_HERE, "..", "include", f) for f in ("sgrl.h", "sgrl_model.h", "sgrl_set.h", "sgrl_train.h", "sgrl_render.h")
             if os.path.exists(os.path.join(_HERE, "..", "include", f))]
    if os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
        return LIB_PATH
    # -fno-slp-vectorize: the SLP vectoriser turns independent f32 chains into packed v_pk_mul_f32 / v_pk_fma_f32 pairs; on
    # the MI355X a v_mul_f32 that overwrites the LOW half of a register pair still being written by a preceding v_pk_mul_f32
    # was observed to lose against it in lanes 48..63 now and then (timing dependent, no hazard wait inserted by the
    # compiler; found through the fn prologue of the Gram GEMM, tools/diag/fn_probe.py).  Without SLP no
    # kernel of this library contains packed-f32 arithmetic; measured cost of the flag: none (SET forward and k_env_step).
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-o", LIB_PATH] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgrlError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(no CPU fallback exists)" % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64.so.7; this library's DT_NEEDED entry has the same soname.  Import
    # torch FIRST so that both resolve to the one HIP/HSA runtime torch initialises (two runtimes in one process
    # see "no ROCm-capable device").  torch is the device-memory / stream plumbing of this package anyway.
    import torch  # noqa: F401
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise SgrlError("cannot load %s: %s" % (LIB_PATH, e))
    L.sgrl_last_error.restype = ctypes.c_char_p
    L.sgrl_version.restype = ctypes.c_char_p
    L.sgrl_engine_create.restype = ctypes.c_int
    L.sgrl_engine_create.argtypes = [ctypes.c_int, ctypes.POINTER(_i32p), _i32p, ctypes.POINTER(_f64p), _i32p, _i32p,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_void_p)]
    L.sgrl_engine_destroy.argtypes = [ctypes.c_void_p]
    L.sgrl_engine_destroy.restype = None
    for name in ("sgrl_num_envs", "sgrl_record_stride", "sgrl_lds_bytes", "sgrl_launch_groups"):
        getattr(L, name).argtypes = [ctypes.c_void_p]
        getattr(L, name).restype = ctypes.c_int
    vp = ctypes.c_void_p
    L.sgrl_reset.argtypes = [vp, vp, vp, vp]
    L.sgrl_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp]
    L.sgrl_refresh.argtypes = [vp, vp, vp, vp]
    L.sgrl_get_records.argtypes = [vp, vp, vp]
    L.sgrl_set_records.argtypes = [vp, vp, vp]
    L.sgrl_time_steps.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
    ci = ctypes.c_int
    L.sgrl_pack_transitions.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
    L.sgrl_ingest_rows.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp]
    _lib = L
    return L


EXPORTS = ["sgrl_engine_create", "sgrl_engine_destroy", "sgrl_num_envs", "sgrl_record_stride", "sgrl_lds_bytes", "sgrl_launch_groups",
           "sgrl_reset", "sgrl_step", "sgrl_get_records", "sgrl_set_records", "sgrl_refresh", "sgrl_time_steps", "sgrl_pack_transitions", "sgrl_ingest_rows",
           "sgrl_last_error", "sgrl_version"]


def check(rc, what):
    if rc != 0:
        raise SgrlError("%s failed (%d): %s" % (what, rc, lib().sgrl_last_error().decode()))
