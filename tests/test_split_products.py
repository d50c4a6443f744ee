"""The arithmetic of the SET tile products' two-piece form (sgrl_amd/csrc/gemm_f32.h: split2h / enc_word),
restated in NumPy and held to its stated bounds on the CPU.  The reference computes these products as plain float32
`F.linear` (subequivariant_attentions.py:90-151, SEActor.py:82-125): what is checked here is that cutting every operand into
two float16 pieces (h = f16(x), l' = f16((x - h) * 2^11), round to nearest) and rebuilding a.b from hh + (h l' + l' h) / 2^11 with
float32 accumulation loses nothing against a float32 FMA chain.  The GPU side of the same statement is tests/test_split_products_gpu.py (and tools/chain_lab.hip mode r; the round-3 lab tools/gemm_lab.hip is an archive)
(error of the kernels against float64) and tests/test_set_gpu.py (both product forms against the reference fixtures)."""
import numpy as np

LIM = 65000.0
SCALE = 2048.0


def split(x):
    """x (float32) -> (h, l') as float16, the kernel's split2h after its clamp."""
    c = np.clip(x.astype(np.float32), -LIM, LIM).astype(np.float32)
    h = c.astype(np.float16)
    r = (c - h.astype(np.float32)).astype(np.float32)           # exact in float32
    l = (r * np.float32(SCALE)).astype(np.float16)
    return h, l


def enc_word(x):
    h, l = split(x)
    return h.view(np.uint16).astype(np.uint32) | (l.view(np.uint16).astype(np.uint32) << 16)


def dec_word(w):
    h = (w & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
    l = (w >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    return h + l / np.float32(SCALE)


def test_two_pieces_represent_a_float32_to_2_pow_minus_22_in_the_normal_range():
    rng = np.random.RandomState(0)
    x = (rng.uniform(-1, 1, 200000) * 10.0 ** rng.uniform(-4.2, 4.8, 200000)).astype(np.float32)
    x = x[(np.abs(x) >= 6.2e-5) & (np.abs(x) <= LIM)]
    h, l = split(x)
    rec = h.astype(np.float64) + l.astype(np.float64) / SCALE
    rel = np.abs(rec - x.astype(np.float64)) / np.abs(x.astype(np.float64))
    assert rel.max() <= 2.0 ** -22 * 1.0001, rel.max()
    # the remainder x - h is exactly representable in float32 (what the kernel relies on)
    r64 = x.astype(np.float64) - h.astype(np.float64)
    assert (r64.astype(np.float32).astype(np.float64) == r64).all()
    # the scaled small piece never overflows float16
    assert np.isfinite(l.astype(np.float32)).all()


def test_below_the_normal_range_the_error_stays_absolutely_tiny():
    rng = np.random.RandomState(1)
    x = (rng.uniform(-1, 1, 100000) * 10.0 ** rng.uniform(-12, -4.3, 100000)).astype(np.float32)
    h, l = split(x)
    rec = h.astype(np.float64) + l.astype(np.float64) / SCALE
    assert np.abs(rec - x.astype(np.float64)).max() <= 2.0 ** -36


def test_out_of_range_operands_are_clamped_not_turned_into_infinities():
    x = np.array([7e4, -3e9, 65000.0, 64999.0, np.float32(3.0e38)], dtype=np.float32)
    h, l = split(x)
    assert np.isfinite(h.astype(np.float32)).all() and np.isfinite(l.astype(np.float32)).all()
    rec = h.astype(np.float64) + l.astype(np.float64) / SCALE
    assert np.allclose(rec, np.clip(x, -LIM, LIM), rtol=2.0 ** -21)


def test_words_round_trip():
    rng = np.random.RandomState(2)
    x = (rng.uniform(-1, 1, 50000) * 10.0 ** rng.uniform(-3, 4, 50000)).astype(np.float32)
    x = x[np.abs(x) >= 6.2e-5]
    w = enc_word(x)
    assert w.dtype == np.uint32
    y = dec_word(w)
    assert (np.abs(y.astype(np.float64) - x.astype(np.float64)) <= 2.0 ** -21 * np.abs(x.astype(np.float64))).all()


def _three_products(A, W):
    """What the kernel accumulates: hh in one float32 accumulator, (l'h + h l') in another, joined once at the end.
    A float16 x float16 product is exact in float32; the matrix core accumulates in float32 (modelled here k by k)."""
    ah, al = split(A)
    wh, wl = split(W)
    ah32, al32, wh32, wl32 = (v.astype(np.float32) for v in (ah, al, wh, wl))
    acc = np.zeros((A.shape[0], W.shape[0]), dtype=np.float32)
    cor = np.zeros_like(acc)
    for k in range(A.shape[1]):
        acc += np.outer(ah32[:, k], wh32[:, k])
        cor += np.outer(al32[:, k], wh32[:, k])
        cor += np.outer(ah32[:, k], wl32[:, k])
    return acc + cor * np.float32(1.0 / SCALE)


def test_three_products_match_float64_as_well_as_a_float32_chain_does():
    rng = np.random.RandomState(3)
    M, N, K = 48, 40, 256
    A = (rng.uniform(-0.5, 0.5, (M, K)) * 10.0 ** rng.uniform(-3, 3, (M, K))).astype(np.float32)     # six decades, as gemm_lab
    W = (rng.uniform(-0.5, 0.5, (N, K)) * 0.2).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    mag = np.abs(A.astype(np.float64)) @ np.abs(W.astype(np.float64)).T
    got = _three_products(A, W)
    chain = np.zeros((M, N), dtype=np.float32)
    for k in range(K):                      # the float32 FMA chain of the reference arithmetic
        chain += np.outer(A[:, k], W[:, k]).astype(np.float32)
    e_split = (np.abs(got.astype(np.float64) - ref) / mag).max()
    e_chain = (np.abs(chain.astype(np.float64) - ref) / mag).max()
    print("max |err| / sum |a w|: two-piece products %.2e, float32 chain %.2e" % (e_split, e_chain))
    # (this model adds the products one k at a time in float32 -- the matrix core adds sixteen per instruction with less
    # rounding: the kernels measure 1.3-1.8e-7 on these operands, profiles/r2_gemm_lab_forms.log)
    assert e_split < 5e-7, e_split
    assert e_split <= 1.25 * e_chain + 1e-8, (e_split, e_chain)
    # dropping the cross products would NOT do: the leading products alone are a float16-grade result
    ah, _ = split(A)
    wh, _ = split(W)
    lead = ah.astype(np.float64) @ wh.astype(np.float64).T
    assert (np.abs(lead - ref) / mag).max() > 1e-5
