"""C-ABI surface: the shared library loads without a GPU, exports every symbol include/sgrl*.h declares, and the
host-side entry points fail loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from sgrl_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgrl_[a-z0-9_]+)\s*\(", text)) - {"sgrl_model_view"})


@pytest.fixture(scope="module")
def so():
    return ctypes.CDLL(_lib.build())


def test_exports_every_declared_symbol(so):
    names = _declared("sgrl.h")
    for extra in ("sgrl_set.h", "sgrl_train.h", "sgrl_render.h"):
        if os.path.exists(os.path.join(REPO, "include", extra)):
            names += _declared(extra)
    assert len(names) >= 13
    for n in names:
        assert hasattr(so, n), n
    assert sorted(n for n in _declared("sgrl.h")) == sorted(_lib.EXPORTS)


def test_version_and_error_strings(so):
    so.sgrl_version.restype = ctypes.c_char_p
    assert b"gfx950" in so.sgrl_version()


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sgrl_amd.vec_env import BatchedModularVecEnv
    with pytest.raises(_lib.SgrlError):
        BatchedModularVecEnv(["3d_hopper_3_shin"], 1)
    # and the raw ABI refuses too (returns an error code, never computes on the host)
    L = _lib.lib()
    from helpers import packed
    m, ib, fb = packed("3d_hopper_3_shin")
    ibp = (ctypes.POINTER(ctypes.c_int32) * 1)(ib.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    fbp = (ctypes.POINTER(ctypes.c_double) * 1)(fb.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    h = ctypes.c_void_p()
    rc = L.sgrl_engine_create(1, ibp, (ctypes.c_int32 * 1)(len(ib)), fbp, (ctypes.c_int32 * 1)(len(fb)),
                              (ctypes.c_int32 * 1)(1), 123, 9, ctypes.c_uint64(0), ctypes.c_uint32(0), 1000,
                              ctypes.byref(h))
    assert rc == -3 and not h.value
    assert b"no CPU fallback" in L.sgrl_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "sgrl_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libsgrl_oracle" not in text and "libsgrl_emu" not in text, f


def test_blob_size_formula_in_the_header_matches_the_packer():
    """include/sgrl_model.h sgrl_model_blob_sizes() (used by the kernel prologue) restates the packer's table sizes."""
    import re
    from sgrl_amd import mjcf, model_pack
    from sgrl_amd.env_spec import env_spec_for
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sgrl_model.h")).read()
    fi = re.search(r"\*n_int = SGRL_NHDR \+ ([^;]+);", text).group(1)
    ff = re.search(r"\*n_f64 = SGRL_NFHDR \+ ([^;]+);", text).group(1)
    for n in mjcf.list_assets():
        m = mjcf.load_asset(n)
        ib, fb = model_pack.pack_model(m, spec=env_spec_for(n))
        env = dict(nb=int(ib[1]), nj=int(ib[2]), nq=int(ib[3]), nv=int(ib[4]), nu=int(ib[5]), ng=int(ib[6]), np=int(ib[7]))
        assert 24 + eval(fi, {}, env) == len(ib), n
        assert 16 + eval(ff, {}, env) == len(fb), n
