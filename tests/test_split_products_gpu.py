"""The SET forward's tile products against float64, on the GPU, through the C ABI (VERDICT r2 item 3 i).

Every `k_gemm3` instantiation the forward launches -- plain, ReLU, row division, Gram-generated operand, equivariant
epilogue, stacked projections, residual + LayerNorm -- is run by `sgrl_set_debug_product` (include/sgrl_set.h) on the
production (N, K) shapes with a ragged row count, in both split forms (two f16 pieces x 3 products = the default, three bf16
pieces x 6), and its error against a float64 evaluation of the same operands is compared with the error the EXACT-f32 matrix
instruction (`k_gemm2`, the reference's arithmetic) commits on those operands.  The contract of DESIGN.md 4.2: a split product
is a float32 product -- error relative to sum_k |a_k w_k| no larger than the exact-f32 chain's (a few 1e-7) -- over float32's
RANGE: operands at 1e-20, 6e-5, 1, 6e4, 1e8, rows of very different size in one tile, rows whose first k-tile says nothing about
the rest, and heavy cancellation (the two-piece form scales every operand row by a power of two before it splits it:
csrc/gemm_f32.h pow2_scale; nothing is clamped, there is no counter to watch).  The fused back-to-back products of
csrc/chain_f16.h are held against float64 and against the single products they replace through `sgrl_set_debug_chain`."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F16X3, BF16X6, EXACT = 2, 3, 1
M = 2 * 128 + 77            # three row tiles, the last one ragged


@pytest.fixture(scope="module")
def handle():
    import torch
    from sgrl_amd.set_hip import HipSetActor
    from sgrl_amd.set_policy import make_policy
    assert torch.cuda.is_available()
    return HipSetActor(make_policy(device="cuda:0").eval())


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _run(h, kind, form, A, W, bias, C, N, K, rowdiv=None, aux_in=None, aux_out=None, lda=None):
    import torch
    from sgrl_amd.set_hip import _check
    _check(h.L, h.L.sgrl_set_debug_product(h.h, kind, form, _p(A), int(lda if lda else A.stride(0)), _p(W), int(W.stride(0)), _p(bias),
                                           _p(C), int(C.stride(0)), A.shape[0], N, K, _p(rowdiv), _p(aux_in), _p(aux_out),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "sgrl_set_debug_product")
    torch.cuda.synchronize()


def _operands(dist, N, K, seed):
    """A [M, K], W [N, K], bias [N] of one of four regimes."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(s, device="cuda", generator=g, dtype=torch.float32)
    if dist == "normal":
        A, W, b = r(M, K), r(N, K) / K ** 0.5, r(N)
    elif dist == "tiny":                  # down at the edge of f16's normal range
        A, W, b = r(M, K) * 6e-5, r(N, K), r(N) * 1e-5
    elif dist == "large":                 # up to 6e4: the edge of f16's range
        A = (torch.rand((M, K), device="cuda", generator=g) * 2 - 1) * 6.0e4
        W, b = r(N, K) * 1e-3, r(N)
    elif dist == "huge":                  # far beyond f16: activations of a diverging network, weights at 1e3
        A, W, b = r(M, K) * 1e8, r(N, K) * 1e3, r(N) * 1e10
    elif dist == "minute":                # far below f16's normal range on both sides
        A, W, b = r(M, K) * 1e-20, r(N, K) * 1e-6, r(N) * 1e-26
    elif dist == "ragged":                # every row its own magnitude (1e-12 .. 1e12); rows whose first k-tile is zero or 1e-6 of the rest
        A = r(M, K) * torch.pow(10.0, (torch.rand((M, 1), device="cuda", generator=g) * 24 - 12))
        A[::5, :16] = 0.0
        A[2::7, :16] *= 1e-6
        W = r(N, K) * torch.pow(10.0, (torch.rand((N, 1), device="cuda", generator=g) * 8 - 4))
        b = torch.zeros(N, device="cuda")
    elif dist == "sparse":                # rows whose SAMPLED entries are all zero (first k-tile, first four values of every later k-tile):
        # the row-scale estimate sees nothing; the rest of the row sits at 1e-9, 1e-3 or 1e5 -- must be split on the exact maxima
        A = r(M, K) * torch.pow(10.0, torch.tensor([-9.0, -3.0, 5.0], device="cuda")[torch.arange(M, device="cuda") % 3])[:, None]
        cols = torch.arange(K, device="cuda")
        A[:, (cols < 32) | (cols % 16 < 4)] = 0.0
        A[::11] = 0.0                      # and some rows that ARE all zero (nothing to repeat for)
        W, b = r(N, K) / K ** 0.5, torch.zeros(N, device="cuda")
    else:                                 # heavy cancellation: the second half of every row nearly undoes the first
        A = r(M, K)
        A[:, K // 2:] = -A[:, :K // 2] + 1e-4 * r(M, K // 2)
        W = r(N, K // 2).repeat(1, 2).contiguous()
        b = torch.zeros(N, device="cuda")
    return A.contiguous(), W.contiguous(), b.contiguous()


def _rel_err(C, ref64, scale64):
    return float(((C.double() - ref64).abs() / scale64).max())


@pytest.mark.parametrize("dist", ["normal", "tiny", "large", "huge", "minute", "ragged", "sparse", "cancel"])
@pytest.mark.parametrize("shape", [(128, 256), (256, 128), (768, 256), (256, 256)])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_plain_products_are_float32_products(handle, kind, shape, dist):
    import torch
    N, K = shape
    A, W, b = _operands(dist, N, K, seed=17 * kind + N + K)
    rd = (torch.rand(M, device="cuda") * 3 + 1).contiguous() if kind == 2 else None
    ref = A.double() @ W.double().t() + b.double()
    scale = A.double().abs() @ W.double().abs().t() + b.double().abs() + 1e-300
    if kind == 1:
        ref = ref.clamp_min(0)
    if kind == 2:
        ref, scale = ref / rd.double()[:, None], scale / rd.double()[:, None]
    errs = {}
    for form in (EXACT, F16X3, BF16X6):
        C = torch.full((M, N), float("nan"), device="cuda")
        handle.scale_redos(reset=True)
        _run(handle, kind, form, A, W, b, C, N, K, rowdiv=rd)
        assert torch.isfinite(C).all()
        errs[form] = _rel_err(C, ref, scale)
        if form == F16X3 and dist == "sparse":
            assert handle.scale_redos(reset=True) > 0      # a zero ESTIMATE is "unknown": the tiles were repeated on exact maxima
        if form == F16X3 and dist == "normal":
            assert handle.scale_redos(reset=True) == 0
    # the exact-f32 chain itself: a few 1e-7 of sum |a w| (K <= 256)
    assert errs[EXACT] < 6e-7, errs
    # the split forms are float32 products: no worse than the exact chain on the same operands (their accumulation error is
    # smaller: the hh products and the corrections accumulate separately), with a floor for operands whose pieces go subnormal
    floor = 1.5e-7
    assert errs[F16X3] <= max(errs[EXACT] * 1.02, floor), (errs, dist)
    assert errs[BF16X6] <= max(errs[EXACT] * 1.02, floor), (errs, dist)


def test_outliers_inside_a_row_cost_nothing(handle):
    """One element 1e6 x the rest of its row, late in the row (the first-tile estimate cannot know): the workgroup repeats its
    tile with the exact row maximum; the other rows of the tile are bit-identical to what they are without the outlier (a row's
    scale is its own).  In the outlier rows ONE term carries the sum, so the representation error of that single operand shows
    undiluted: two f16 pieces hold 22 bits (2^-22 = 2.4e-7 per operand), float32 holds 24 -- the one place where the split
    form is visibly, if harmlessly, coarser than an f32 product."""
    import torch
    N, K = 128, 256
    A, W, b = _operands("normal", N, K, seed=5)
    C0 = torch.empty((M, N), device="cuda")
    _run(handle, 0, F16X3, A, W, b, C0, N, K)
    A2 = A.clone()
    A2[3, 200] = 1.0e6
    A2[130, 255] = -3.0e9
    C = torch.empty((M, N), device="cuda")
    _run(handle, 0, F16X3, A2, W, b, C, N, K)
    ref = A2.double() @ W.double().t() + b.double()
    scale = A2.double().abs() @ W.double().abs().t() + b.double().abs()
    assert torch.isfinite(C).all() and _rel_err(C, ref, scale) < 5e-7
    keep = torch.ones(M, dtype=torch.bool, device="cuda")
    keep[3] = keep[130] = False
    assert torch.equal(C[keep], C0[keep])


def _chain(h, kind, A, K, W1, b1, hid, W2, b2, C, rowdiv=None, ln=None, Wp=None, zc=None, z2=None, fn=None, M_=None):
    import torch
    from sgrl_amd.set_hip import _check
    _check(h.L, h.L.sgrl_set_debug_chain(h.h, kind, _p(A), int(A.stride(0)), int(K), _p(Wp), _p(W1), _p(b1), int(hid), _p(W2), _p(b2), _p(C),
                                         int(C.stride(0)), int(M_ if M_ else A.shape[0]), _p(rowdiv), _p(ln), _p(zc), _p(z2), _p(fn),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "sgrl_set_debug_chain")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dist", ["normal", "huge", "minute", "ragged", "sparse"])
@pytest.mark.parametrize("hid,K", [(256, 256), (128, 160)])
def test_fused_pair_is_the_two_single_products(handle, hid, K, dist):
    """chain kind 0: relu(A W1' + b1) W2' + b2 in one kernel = the ReLU product followed by the plain product (the intermediate is
    split once, in registers, with its exact row scale) -- held against float64 and against the two launches."""
    import torch
    A, W1, b1 = _operands(dist, hid, K, seed=3 * hid + K)
    g = torch.Generator(device="cuda").manual_seed(hid)
    W2 = (torch.randn((128, hid), device="cuda", generator=g) / hid ** 0.5).contiguous()
    b2 = torch.zeros(128, device="cuda")
    H = (A.double() @ W1.double().t() + b1.double()).clamp_min(0)
    ref = H @ W2.double().t()
    scale = H.abs() @ W2.double().abs().t() + 1e-300
    C = torch.full((M, 128), float("nan"), device="cuda")
    _chain(handle, 0, A, K, W1, b1, hid, W2, b2, C)
    Hs = torch.empty((M, hid), device="cuda")
    _run(handle, 1, F16X3, A, W1, b1, Hs, hid, K)
    Cs = torch.empty((M, 128), device="cuda")
    _run(handle, 0, F16X3, Hs, W2, b2, Cs, 128, hid)
    assert torch.isfinite(C).all()
    # error of the pair relative to sum |h w2|: the first product's error (a few 1e-7 of sum |a w1|, which cancellation inside
    # relu(.) can leave larger than |h|) passes through W2 -- the two-launch path carries exactly the same
    e_f, e_s = _rel_err(C, ref, scale), _rel_err(Cs, ref, scale)
    assert e_f <= max(1.05 * e_s, 3e-7), (e_f, e_s, dist)


def test_fused_pair_with_residual_layernorm(handle):
    """chain kind 1: ln_io <- LayerNorm(ln_io + (relu(A W1' + b1) W2' + b2) / rowdiv) * w + b, hidden width 256."""
    import torch
    A, W1, b1 = _operands("normal", 256, 256, seed=77)
    g = torch.Generator(device="cuda").manual_seed(6)
    W2 = (torch.randn((128, 256), device="cuda", generator=g) / 16.0).contiguous()
    b2 = torch.randn(128, device="cuda", generator=g)
    rd = (torch.rand(M, device="cuda", generator=g) * 3 + 1).contiguous()
    res = torch.randn((M, 128), device="cuda", generator=g).contiguous()
    lnwb = torch.cat([torch.rand(128, device="cuda", generator=g) + 0.5, torch.randn(128, device="cuda", generator=g)]).contiguous()
    H = (A.double() @ W1.double().t() + b1.double()).clamp_min(0)
    x = res.double() + (H @ W2.double().t() + b2.double()) / rd.double()[:, None]
    ref = torch.nn.functional.layer_norm(x, (128,), lnwb[:128].double(), lnwb[128:].double(), 1e-5)
    io = res.clone()
    _chain(handle, 1, A, 256, W1, b1, 256, W2, b2, io, rowdiv=rd, ln=lnwb)
    # the two launches it replaces
    Hs = torch.empty((M, 256), device="cuda")
    _run(handle, 1, F16X3, A, W1, b1, Hs, 256, 256)
    io2 = res.clone()
    _run(handle, 6, F16X3, Hs, W2, b2, io2, 128, 256, rowdiv=rd, aux_in=lnwb)
    e_f, e_s = float((io.double() - ref).abs().max()), float((io2.double() - ref).abs().max())
    assert e_f <= max(1.5 * e_s, 2e-6), (e_f, e_s)


@pytest.mark.parametrize("zscale", [1.0, 1e4, 1e-9])
@pytest.mark.parametrize("hid,K,two", [(256, 128, False), (256, 128, True), (128, 144, True), (128, 144, False)])
def test_fused_projection_site(handle, hid, K, two, zscale):
    """chain kind 2: X -> Z (, Z2) -> fn -> relu(G(Z) W1' + b1) W2' + b2 in one kernel, against float64 and against the three
    launches it replaces (stacked projections, Gram-operand product, plain product)."""
    import torch
    from sgrl_amd.set_hip import gram_order
    g = torch.Generator(device="cuda").manual_seed(hid + K + two)
    r = lambda *s: torch.randn(s, device="cuda", generator=g, dtype=torch.float32)
    X = (r(3 * M, K) * zscale).contiguous()
    Wp = (r(64, K) / K ** 0.5).contiguous()
    Wp[30:32] = 0; Wp[62:64] = 0
    ia, ib, ok = [t.cuda() for t in gram_order()]
    W1 = (r(hid, 576) / 24.0).contiguous()
    W1[:, ~ok] = 0
    b1 = r(hid) * zscale * zscale
    W2 = (r(128, hid) / hid ** 0.5).contiguous()
    b2 = torch.zeros(128, device="cuda")
    gd = (r(3 * M, 2) * zscale).contiguous()                 # the gravity / direction columns 30, 31 (written by k_embed in the forward)
    def fresh():
        z = torch.zeros((3 * M, 32), device="cuda")
        z[:, 30:] = gd
        return z
    zc, z2, fn = fresh(), fresh() if two else None, torch.empty(M, device="cuda")
    C = torch.full((M, 128), float("nan"), device="cuda")
    _chain(handle, 2, X, K, W1, b1, hid, W2, b2, C, Wp=Wp, zc=zc, z2=z2, fn=fn, M_=M)
    # float64
    Z = (X.double() @ Wp.double().t())
    Zc = torch.cat([Z[:, :30], gd.double()], 1).view(M, 3, 32)
    G = torch.einsum("msa,msb->mab", Zc, Zc)
    Ag = G[:, ia, ib]
    H = (Ag @ W1.double().t() + b1.double()).clamp_min(0)
    ref = H @ W2.double().t()
    scale = H.abs() @ W2.double().abs().t() + 1e-300
    zscale64 = X.double().abs() @ Wp.double().abs().t() + 1e-300
    assert float(((zc[:, :30].double() - Z[:, :30]).abs() / zscale64[:, :30]).max()) < 3e-7
    assert torch.equal(zc[:, 30:], gd)
    if two:
        assert float(((z2[:, :30].double() - Z[:, 32:62]).abs() / zscale64[:, 32:62]).max()) < 3e-7
        assert torch.equal(z2[:, 30:], gd)
    # the three launches
    zc_s, z2_s, fn_s = fresh(), fresh(), torch.empty(M, device="cuda")
    _run(handle, 5, F16X3, X, Wp, None, zc_s, 64, K, aux_out=z2_s)
    Hs = torch.empty((M, hid), device="cuda")
    _run(handle, 3, F16X3, zc_s.view(M, 96), W1, b1, Hs, hid, 576, aux_out=fn_s, lda=96)
    Cs = torch.empty((M, 128), device="cuda")
    _run(handle, 0, F16X3, Hs, W2, b2, Cs, 128, hid)
    assert torch.equal(zc, zc_s) and torch.equal(fn, fn_s)
    fn_ref = torch.linalg.matrix_norm(torch.einsum("msa,msb->mab", zc.view(M, 3, 32).double(), zc.view(M, 3, 32).double())) + 1.0
    assert float(((fn.double() - fn_ref).abs() / fn_ref).max()) < 2e-6
    # the Gram entries are products of the float32 Z the kernels hold, not of the float64 one: compare like with like
    Zf = zc.view(M, 3, 32).double()
    Hf = (torch.einsum("msa,msb->mab", Zf, Zf)[:, ia, ib] @ W1.double().t() + b1.double()).clamp_min(0)
    ref_f = Hf @ W2.double().t()
    scale_f = Hf.abs() @ W2.double().abs().t() + 1e-300
    e_f, e_s = _rel_err(C, ref_f, scale_f), _rel_err(Cs, ref_f, scale_f)
    assert torch.isfinite(C).all() and e_f <= max(1.05 * e_s, 4e-7), (e_f, e_s)


@pytest.mark.parametrize("form", [F16X3, BF16X6])
@pytest.mark.parametrize("N", [128, 256])
def test_gram_operand_product(handle, form, N):
    """kind 3: A[m][k] = (Z'Z)[a_k][b_k] generated inside the kernel from Z [M, 3, 32]; fn = ||Z'Z||_F + 1."""
    import torch
    from sgrl_amd.set_hip import gram_order
    g = torch.Generator(device="cuda").manual_seed(N)
    for zscale in (1.0, 30.0):           # Gram entries are squares: 30 -> entries up to ~1e4 x 3
        Z = (torch.randn((M, 3, 32), device="cuda", generator=g) * zscale).contiguous()
        ia, ib, ok = [t.cuda() for t in gram_order()]
        W = torch.randn((N, 576), device="cuda", generator=g) / 24.0
        W[:, ~ok] = 0
        b = torch.randn(N, device="cuda", generator=g)
        G = torch.einsum("msa,msb->mab", Z.double(), Z.double())
        Ag = G[:, ia, ib]                                               # [M, 576] float64
        ref = (Ag @ W.double().t() + b.double()).clamp_min(0)
        scale = Ag.abs() @ W.double().abs().t() + b.double().abs() + 1e-300
        C = torch.empty((M, N), device="cuda")
        fn = torch.empty(M, device="cuda")
        _run(handle, 3, form, Z.view(M, 96), W, b, C, N, 576, aux_out=fn, lda=96)
        # exact-f32 counterpart: the same product with the Gram operand materialised in float32
        Ce = torch.empty((M, N), device="cuda")
        Af = torch.einsum("msa,msb->mab", Z, Z)[:, ia, ib].contiguous()
        _run(handle, 1, EXACT, Af, W, b, Ce, N, 576)
        e_split, e_exact = _rel_err(C, ref, scale), _rel_err(Ce, ref, scale)
        assert e_split <= max(1.05 * e_exact, 2.5e-7), (e_split, e_exact, zscale)
        fn_ref = torch.linalg.matrix_norm(G) + 1.0
        assert float(((fn.double() - fn_ref).abs() / fn_ref).max()) < 2e-6


@pytest.mark.parametrize("form", [F16X3, BF16X6])
def test_equivariant_epilogue_product(handle, form):
    """kind 4: tout[m][s][c] = sum_q zq[m][s][q] (A W' + b)[m][c * 32 + q] / rowdiv[m]; the [M, 1024] matrix is never stored."""
    import torch
    N, K = 1024, 256
    A, W, b = _operands("normal", N, K, seed=41)
    g = torch.Generator(device="cuda").manual_seed(3)
    zq = torch.randn((M, 3, 32), device="cuda", generator=g).contiguous()
    rd = (torch.rand(M, device="cuda", generator=g) * 3 + 1).contiguous()
    mat = (A.double() @ W.double().t() + b.double()).view(M, 32, 32)        # [m][c][q]
    ref = torch.einsum("msq,mcq->msc", zq.double(), mat) / rd.double()[:, None, None]
    smat = (A.double().abs() @ W.double().abs().t() + b.double().abs()).view(M, 32, 32)
    scale = torch.einsum("msq,mcq->msc", zq.double().abs(), smat) / rd.double()[:, None, None]
    T = torch.empty((M, 96), device="cuda")
    _run(handle, 4, form, A, W, b, T, N, K, rowdiv=rd, aux_in=zq.view(M, 96))
    Cm = torch.empty((M, N), device="cuda")
    _run(handle, 0, EXACT, A, W, b, Cm, N, K)
    Te = torch.einsum("msq,mcq->msc", zq, Cm.view(M, 32, 32)) / rd[:, None, None]
    e_split, e_exact = _rel_err(T.view(M, 3, 32), ref, scale), _rel_err(Te, ref, scale)
    assert e_split <= max(1.05 * e_exact, 2.5e-7), (e_split, e_exact)


@pytest.mark.parametrize("form", [F16X3, BF16X6])
@pytest.mark.parametrize("K", [128, 144])
def test_stacked_projection_product(handle, form, K):
    """kind 5: the 64 stacked projection columns, 0..29 -> Z rows, 32..61 -> Z2 rows (30 / 31 / 62 / 63 are not stored)."""
    import torch
    A, W, _ = _operands("normal", 64, K, seed=K)
    ref = A.double() @ W.double().t()
    scale = A.double().abs() @ W.double().abs().t() + 1e-300
    Z1 = torch.full((M, 32), 7.0, device="cuda")
    Z2 = torch.full((M, 32), 7.0, device="cuda")
    _run(handle, 5, form, A, W, None, Z1, 64, K, aux_out=Z2)
    assert (Z1[:, 30:] == 7.0).all() and (Z2[:, 30:] == 7.0).all()       # the gravity / direction columns are left alone
    e1 = _rel_err(Z1[:, :30], ref[:, :30], scale[:, :30])
    e2 = _rel_err(Z2[:, :30], ref[:, 32:62], scale[:, 32:62])
    assert max(e1, e2) < 3e-7, (e1, e2)


@pytest.mark.parametrize("form", [F16X3, BF16X6])
def test_residual_layernorm_epilogue_product(handle, form):
    """kind 6: ln_io <- LayerNorm(ln_io + (A W' + b) / rowdiv) * w + b over the 128 columns."""
    import torch
    N, K = 128, 256
    A, W, b = _operands("normal", N, K, seed=9)
    g = torch.Generator(device="cuda").manual_seed(4)
    rd = (torch.rand(M, device="cuda", generator=g) * 3 + 1).contiguous()
    res = torch.randn((M, N), device="cuda", generator=g).contiguous()
    lnwb = torch.cat([torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)]).contiguous()
    x = res.double() + (A.double() @ W.double().t() + b.double()) / rd.double()[:, None]
    ref = torch.nn.functional.layer_norm(x, (N,), lnwb[:N].double(), lnwb[N:].double(), 1e-5)
    io = res.clone()
    _run(handle, 6, form, A, W, b, io, N, K, rowdiv=rd, aux_in=lnwb)
    Cm = torch.empty((M, N), device="cuda")
    _run(handle, 2, EXACT, A, W, b, Cm, N, K, rowdiv=rd)
    exact = torch.nn.functional.layer_norm(res + Cm, (N,), lnwb[:N], lnwb[N:], 1e-5)
    e_split = float((io.double() - ref).abs().max())
    e_exact = float((exact.double() - ref).abs().max())
    assert e_split <= max(1.5 * e_exact, 2e-6), (e_split, e_exact)


@pytest.mark.parametrize("dist", ["normal", "huge", "ragged"])
def test_fused_equivariant_pair(handle, dist):
    """chain kind 3: linear3 -> ReLU -> linear4 -> contraction with z in one kernel (the [M, 1024] matrix and the 256-wide
    intermediate never reach memory), against float64 and against the two launches it replaces (ReLU product, equivariant-epilogue
    product)."""
    import torch
    K = 256
    A, W1, b1 = _operands(dist, 256, K, seed=91)
    g = torch.Generator(device="cuda").manual_seed(8)
    W2 = (torch.randn((1024, 256), device="cuda", generator=g) / 16.0).contiguous()
    b2 = torch.randn(1024, device="cuda", generator=g)
    zq = torch.randn((M, 3, 32), device="cuda", generator=g).contiguous()
    rd = (torch.rand(M, device="cuda", generator=g) * 3 + 1).contiguous()
    H = (A.double() @ W1.double().t() + b1.double()).clamp_min(0)
    mat = (H @ W2.double().t() + b2.double()).view(M, 32, 32)                       # [m][c][q]
    ref = torch.einsum("msq,mcq->msc", zq.double(), mat) / rd.double()[:, None, None]
    smat = (H.abs() @ W2.double().abs().t() + b2.double().abs()).view(M, 32, 32)
    scale = torch.einsum("msq,mcq->msc", zq.double().abs(), smat) / rd.double()[:, None, None] + 1e-300
    T = torch.full((M, 96), float("nan"), device="cuda")
    _chain(handle, 3, A, K, W1, b1, 256, W2, b2, T, rowdiv=rd, ln=zq.view(M, 96))
    Hs = torch.empty((M, 256), device="cuda")
    _run(handle, 1, F16X3, A, W1, b1, Hs, 256, K)
    Ts = torch.empty((M, 96), device="cuda")
    _run(handle, 4, F16X3, Hs, W2, b2, Ts, 1024, 256, rowdiv=rd, aux_in=zq.view(M, 96))
    assert torch.isfinite(T).all()
    e_f, e_s = _rel_err(T.view(M, 3, 32), ref, scale), _rel_err(Ts.view(M, 3, 32), ref, scale)
    assert e_f <= max(1.05 * e_s, 3e-7), (e_f, e_s, dist)
