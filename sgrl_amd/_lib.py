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


# ---- fixed-dimension instances of the step kernel (csrc/step_spec.hip) -----------------------------------------------------
HDR_DIM_FIELDS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 18)   # NBODY NJNT NQ NV NU NGEOM NPAIR INTEGRATOR FRAME_SKIP MAX_ROWS SOLVER
LIGHT_NV = 12                                           # dimension sets up to this many dofs form the "light" kernels (4 waves per SIMD)
# Families whose step kernel reads the int tables where they lie (constant address space) instead of from an LDS copy
# (csrc/engine_kernel.h SGRL_ITAB_GLOBAL): the slab loses the tables' 2.2-3.4 KB, keeps its contact frames as 6 doubles and takes the row
# cut that gives the MOST resident workgroups (step_body.h make_layout, floor 19).  Round 6 measured it (profiles/r6_slab_diet.json):
# humanoid++ gains a seventh resident (humanoid_9 26 672 -> 23 008 B) and 2 % (k_env_step 2.53 -> 2.48 ms, step + forward 4.47 -> 4.38);
# the cheetah family gains nothing (cheetah_14 stays at four residents: 39 208 B against the 32 000 five would need) and loses 4 % to
# the global reads -- so: the humanoid family only.  SGRL_BUILD_ITAB_GLOBAL=<family,...> (or "none") at build time overrides.
ITAB_GLOBAL = set(x for x in os.environ.get("SGRL_BUILD_ITAB_GLOBAL", "humanoid").split(",") if x and x != "none")
PAIR_NV = {"walker": 15, "hopper": 15}                              # sets up to this many dofs get the two-environments-per-wavefront instance (default LIGHT_NV)


def default_max_rows(model):
    """Row cap the vec-env packs a morphology with unless told otherwise (multi-geom bodies touch the floor in many places)."""
    return 256      # pack_model lowers it to the geometric worst case (walker_7 74, humanoid_9 140, cheetah_14 211): never bites


def spec_families():
    """[(family id, waves per SIMD, [dims, ...], int tables read from global memory 0 / 1)]: the fixed-dimension kernels of the step (csrc/step_spec.hip).  Per morphology
    family (walker / hopper / humanoid / cheetah) at the default row caps:
      * ONE kernel holding every dimension set of the family at two waves per SIMD -- a mixed batch of one family stays ONE
        launch (concurrent launches of kernels with different register budgets fragment the SIMDs' register files and LDS:
        six per-set launches of the walker mix measured 8-10 % SLOWER than the generic kernel, two or three class launches
        0-4 % faster, profiles/r3_variant_probe_*.log);
      * a LIGHT kernel for its sets with nv <= 12 at the register budget of four waves per SIMD (with the dimensions constant
        they need no more than 128 registers + ~30 spilled), used when the whole batch is light: 8192 x walker_3 1.73-1.83 ->
        1.40-1.49 ms.
    Deterministic order; the light kernels follow the full ones."""
    from . import mjcf, model_pack
    from .env_spec import env_spec_for
    fam = {}
    for n in mjcf.list_assets():
        m = mjcf.load_asset(n)
        ib, _ = model_pack.pack_model(m, spec=env_spec_for(n), max_rows=default_max_rows(m))
        d = tuple(int(ib[i]) for i in HDR_DIM_FIELDS)
        fam.setdefault(n.split("_")[1], set()).add(d)
    # the 12th entry of a member: 1 = the kernel also holds the instance that steps TWO environments of this set per wavefront
    # (csrc/wave_half.h; the engine pairs only where the slab pair keeps eight workgroups per CU: walker_2 / walker_3 / walker_4 /
    # hopper_3 / hopper_4 -- the two 15-dof sets on the pair-only diet of the layout, step_body.h make_layout_rows)
    out = []
    for key in sorted(fam):
        itab = 1 if key in ITAB_GLOBAL else 0
        out.append((len(out), 2, [d + (0 if itab else (1 if d[3] <= PAIR_NV.get(key, LIGHT_NV) else 0),) for d in sorted(fam[key])], itab))
    for key in sorted(fam):
        light = sorted(d for d in fam[key] if d[3] <= LIGHT_NV)
        if light:
            out.append((len(out), 4, [d + (0,) for d in light], 0))
    return out


def hash_text(t):
    import zlib
    return zlib.crc32(t.encode())


def _write_if_changed(path, text):
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)


def build(verbose=False):
    """Compile every HIP source for gfx950 into sgrl_amd/libsgrl_hip.so (hipcc cross-compiles without a GPU): one object per
    source + one per fixed-dimension instance of the step kernel, in parallel, then one link."""
    from concurrent.futures import ThreadPoolExecutor
    fams = spec_families()
    table = ("// generated by sgrl_amd/_lib.py build() from sgrl_amd/assets/models\n"
             "// SGRL_FAMILY(id, waves per SIMD, members, int tables in global memory)   SGRL_MEMBER(family, slot, nbody, njnt, nq, nv, nu, ngeom, npair, integrator, frame_skip, max_rows, solver, pair)\n")
    for i, w, dims, itab in fams:
        table += "SGRL_FAMILY(%d, %d, %d, %d)\n" % (i, w, len(dims), itab)
        table += "".join("SGRL_MEMBER(%d, %d, %s)\n" % (i, k, ", ".join(str(x) for x in d)) for k, d in enumerate(dims))
    _write_if_changed(os.path.join(CSRC, "spec_table.inc"), table)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs += [os.path.join(_HERE, "..", "include", f) for f in ("sgrl.h", "sgrl_model.h", "sgrl_set.h", "sgrl_train.h", "sgrl_render.h")
             if os.path.exists(os.path.join(_HERE, "..", "include", f))]
    newest_hdr = max(os.path.getmtime(d) for d in hdrs)
    objdir = os.path.join(_HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    # -fno-slp-vectorize: the SLP vectoriser turns independent f32 chains into packed v_pk_mul_f32 / v_pk_fma_f32 pairs; on
    # the MI355X a v_mul_f32 that overwrites the LOW half of a register pair still being written by a preceding v_pk_mul_f32
    # was observed to lose against it in lanes 48..63 now and then (timing dependent, no hazard wait inserted by the
    # compiler; found through the fn prologue of the Gram GEMM, tools/diag/fn_probe.py).  Without SLP no
    # kernel of this library contains packed-f32 arithmetic; measured cost of the flag: none (SET forward and k_env_step).
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]
    jobs = []
    for sname in SOURCES:
        src = os.path.join(CSRC, sname)
        if os.path.exists(src):
            jobs.append((os.path.join(objdir, sname.replace(".hip", ".o")), src, []))
    spec_src = os.path.join(CSRC, "step_spec.hip")
    for i, w, dims, itab in fams:
        members = ",".join("DimsFixed<%s>" % ",".join(str(x) for x in d) for d in dims)
        flags = ["-DSGRL_SPEC_ID=%d" % i, "-DSGRL_SPEC_WAVES=%d" % w, "-DSGRL_SPEC_FAMILY=" + members]
        if itab:
            flags.append("-DSGRL_ITAB_GLOBAL=1")
        tag = "%08x" % (hash_text(members + str(w) + ("g" if itab else "")) & 0xFFFFFFFF)
        jobs.append((os.path.join(objdir, "family_%d_%s.o" % (i, tag)), spec_src, flags))

    def compile_one(job):
        obj, src, flags = job
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(newest_hdr, os.path.getmtime(src)):
            return False
        cmd = base + flags + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return True

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        rebuilt = list(pool.map(compile_one, jobs))
    objs = [j[0] for j in jobs]
    for f in os.listdir(objdir):            # objects of earlier instance lists: they would travel to the GPU box for nothing
        if f.endswith(".o") and os.path.join(objdir, f) not in objs:
            os.remove(os.path.join(objdir, f))
    if any(rebuilt) or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd[:8]), "... (%d objects)" % len(objs), flush=True)
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
    for name in ("sgrl_num_envs", "sgrl_record_stride", "sgrl_lds_bytes", "sgrl_launch_groups", "sgrl_fixed_dim_groups", "sgrl_paired_envs"):
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
    L.sgrl_ingest_block.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp]
    L.sgrl_round_record.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    L.sgrl_ingest_ws_words.argtypes = [ci]
    L.sgrl_ingest_ws_words.restype = ctypes.c_int64
    _lib = L
    return L


EXPORTS = ["sgrl_engine_create", "sgrl_engine_destroy", "sgrl_num_envs", "sgrl_record_stride", "sgrl_lds_bytes", "sgrl_launch_groups", "sgrl_fixed_dim_groups", "sgrl_paired_envs",
           "sgrl_reset", "sgrl_step", "sgrl_get_records", "sgrl_set_records", "sgrl_refresh", "sgrl_time_steps", "sgrl_pack_transitions", "sgrl_round_record", "sgrl_ingest_rows", "sgrl_ingest_block", "sgrl_ingest_ws_words",
           "sgrl_last_error", "sgrl_version"]


def check(rc, what):
    if rc != 0:
        raise SgrlError("%s failed (%d): %s" % (what, rc, lib().sgrl_last_error().decode()))
