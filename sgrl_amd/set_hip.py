"""Python front-end of the HIP SET-actor forward (C ABI: include/sgrl_set.h).

`HipSetActor` binds the parameters of an `SEPolicy` (set_policy.py, reference-compatible state_dict) to a handle:
it hands the library the DEVICE ADDRESSES of the parameters and a packing plan (`plan_segments`), and the library
rebuilds its flat weight buffer from the live storage at the top of every forward -- so optimizer steps, the
reference's `target_param.data.copy_(...)` soft updates (common/functional.py:7-10, invisible to version counters)
and `load_state_dict` are all seen by the next call.  It also describes the batch structure (morphologies x env
counts) to the engine and runs `actions = max_action * tanh(actor(obs))` for all environments in one call.
No CPU fallback: a missing extension or device raises `SgrlError`.
"""
import ctypes

import numpy as np
import torch

from . import _lib

NGLOBAL, NLAYER, LAYERS = 25, 31, 3
NW = NGLOBAL + LAYERS * NLAYER


def _bind(L):
    if getattr(L, "_set_bound", False):
        return
    vp = ctypes.c_void_p
    L.sgrl_set_create.argtypes = [ctypes.POINTER(vp)]
    L.sgrl_set_destroy.argtypes = [vp]
    L.sgrl_set_destroy.restype = None
    L.sgrl_set_weights.argtypes = [vp, vp, vp, ctypes.c_int]
    L.sgrl_set_bind_params.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int64]
    L.sgrl_set_graph.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp]
    L.sgrl_set_forward.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_float, vp]
    L.sgrl_set_forward_q.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp]
    L.sgrl_set_time_forward.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, vp,
                                        ctypes.POINTER(ctypes.c_float)]
    L.sgrl_set_num_nodes.argtypes = [vp]
    L.sgrl_set_workspace_bytes.argtypes = [vp]
    L.sgrl_set_workspace_bytes.restype = ctypes.c_int64
    L.sgrl_set_generation.argtypes = [vp]
    L.sgrl_set_generation.restype = ctypes.c_int64
    L.sgrl_set_peek.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int64]
    L.sgrl_set_debug_stop_after.argtypes = [vp, ctypes.c_int]
    L.sgrl_set_debug_small_nodes.argtypes = [vp, ctypes.c_int]
    L.sgrl_set_gemm_form.argtypes = [vp, ctypes.c_int]
    L.sgrl_set_last_error.restype = ctypes.c_char_p
    L.sgrl_set_hold_weights.argtypes = [vp, ctypes.c_int]
    L.sgrl_set_debug_redos.argtypes = [ctypes.c_int]
    L.sgrl_set_debug_redos.restype = ctypes.c_longlong
    ci = ctypes.c_int
    L.sgrl_set_debug_product.argtypes = [vp, ci, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp]
    L.sgrl_set_debug_chain.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]
    L._set_bound = True


def _check(L, rc, what):
    if rc != 0:
        raise _lib.SgrlError("%s failed (%d): %s" % (what, rc, L.sgrl_set_last_error().decode()))


GRAM_K = 576


def gram_order():
    """(a, b, valid) of the blocked lower triangle the Gram GEMMs run over (csrc/gemm_f32.h GRAM): the 36 4x4 blocks
    (A, B), B <= A, one per 16-wide k-tile: k = 16 (A (A + 1) / 2 + B) + 4 i + j <-> a = 4 A + i, b = 4 B + j; the
    entries above the diagonal inside the diagonal blocks are not used (zero weight)."""
    a, b, ok = [], [], []
    for A in range(8):
        for B in range(A + 1):
            for i in range(4):
                for j in range(4):
                    a.append(4 * A + i); b.append(4 * B + j); ok.append(4 * A + i >= 4 * B + j)
    return torch.tensor(a), torch.tensor(b), torch.tensor(ok)


def fold_gram_weight(w):
    """[out, 1024] weight acting on vec(G) of a symmetric 32x32 G -> [out, 576] acting on its blocked lower triangle."""
    out_f = w.shape[0]
    w3 = w.reshape(out_f, 32, 32)
    sym = w3 + w3.transpose(1, 2)
    idx_a, idx_b, ok = gram_order()
    f = sym[:, idx_a, idx_b].clone()
    diag = idx_a == idx_b
    f[:, diag] = w3[:, idx_a[diag], idx_b[diag]]
    f[:, ~ok] = 0
    return f.contiguous()


def perm32(t):
    """rows q * 32 + c -> c * 32 + q of a [1024, ...] tensor (include/sgrl_set.h PERM32)"""
    return t.reshape(32, 32, *t.shape[1:]).transpose(0, 1).reshape(t.shape).contiguous()


def pack_tensors(sd, prefix="actor.", critic=False):
    """[(tensor float32, ...)] in slot order (include/sgrl_set.h) from a state_dict-like mapping of torch tensors.
    critic=True packs a critic TransformerModel (scalar head: decoder_ng in the DECG / L1M_B slots, see sgrl_set.h)."""
    g = lambda k: sd[prefix + k].detach().float()
    out = [None] * NW
    out[0:3] = [g("pos_encoder.embeddings.%d.weight" % i) for i in range(3)]
    out[3], out[4] = g("transformer_encoder.rel_encoder.weight"), g("transformer_encoder.rel_encoder.bias")
    out[5], out[6] = g("transformer_encoder.norm.weight"), g("transformer_encoder.norm.bias")
    out[7], out[8], out[9] = g("g_encoder.weight"), g("encoder.weight"), g("encoder.bias")
    out[10] = g("gg_proj.weight")
    out[11], out[12], out[13], out[14] = fold_gram_weight(g("linear1_g.weight")), g("linear1_g.bias"), g("linear2_g.weight"), g("linear2_g.bias")
    w = g("linear1_ng.weight")
    out[15] = torch.cat([w, w.new_zeros(w.shape[0], 160 - w.shape[1])], dim=1)
    out[16], out[17], out[18] = g("linear1_ng.bias"), g("linear2_ng.weight"), g("linear2_ng.bias")
    if critic:
        z = w.new_zeros(1)
        out[19], out[21] = g("decoder_ng.weight").reshape(-1), g("decoder_ng.bias").reshape(-1)
        out[20], out[22], out[23], out[24] = z, z, z, z
    else:
        out[19] = g("decoder_g.weight").reshape(-1)
        out[20], out[21], out[22], out[23] = g("linear1_m.weight"), g("linear1_m.bias"), perm32(g("linear2_m.weight")), perm32(g("linear2_m.bias"))
        out[24] = g("g_proj.weight")
    scaling = float(128) ** -0.5   # (2 * head_dim)^-0.5, reference subequivariant_attentions.py:88
    for l in range(LAYERS):
        p = "transformer_encoder.layers.%d." % l
        a = p + "self_attn."
        b = NGLOBAL + l * NLAYER
        vg, wgo, wng = g(a + "vg_proj.weight"), g(a + "g_out.weight"), g(a + "ng_out.weight")
        wv, bv = g(a + "v_proj.weight"), g(a + "v_proj.bias")
        # attention output folds (include/sgrl_set.h): value projection . output projection, per head
        vfold = torch.cat([wng[:, 128 * h:128 * h + 128] @ wv[128 * h:128 * h + 128] for h in range(2)], 0)          # [256, 256]
        bfold = torch.cat([wng[:, 128 * h:128 * h + 128] @ bv[128 * h:128 * h + 128] for h in range(2)], 0)          # [256]
        ufold = torch.cat([wgo[:, 128 * h:128 * h + 126] @ vg[126 * h:126 * h + 126] for h in range(2)], 0)          # [256, 128]
        gdcol = torch.stack([wgo[:, 128 * h + 126:128 * h + 128] for h in range(2)], 0).contiguous()                 # [2, 128, 2]
        out[b:b + NLAYER] = [
            g(a + "g_proj.weight"), fold_gram_weight(g(a + "linear_g1.weight")), g(a + "linear_g1.bias"), g(a + "linear_g2.weight"),
            g(a + "linear_g2.bias"),
            torch.cat([g(a + "q_proj.weight") * scaling, g(a + "k_proj.weight"), vfold], 0),
            torch.cat([g(a + "q_proj.bias") * scaling, g(a + "k_proj.bias"), bfold], 0),
            ufold,
            wng, g(a + "ng_out.bias"), wgo,
            g(p + "g_proj2.weight"), g(p + "g_proj3.weight"), fold_gram_weight(g(p + "linear_g1.weight")), g(p + "linear_g1.bias"),
            g(p + "linear_g2.weight"), g(p + "linear_g2.bias"), g(p + "linear3.weight"), g(p + "linear3.bias"),
            perm32(g(p + "linear4.weight")), perm32(g(p + "linear4.bias")), g(p + "linear5.weight"), g(p + "linear1.weight"),
            g(p + "linear1.bias"), g(p + "linear2.weight"), g(p + "linear2.bias"), g(p + "norm1.weight"),
            g(p + "norm1.bias"), g(p + "norm2.weight"), g(p + "norm2.bias"), gdcol]
    return out


NSITES = 7
NEXTRA = 2          # folded head: decoder_g . linear2_m  [32, 256] and its bias [32] (include/sgrl_set.h)
PACK_COPY, PACK_PADCOL, PACK_FOLD, PACK_STACK, PACK_MATMUL, PACK_SUBMAT, PACK_PERM32 = 0, 1, 2, 3, 4, 5, 6
# struct sgrl_pack_seg (include/sgrl_set.h)
SEG_DTYPE = np.dtype([("dst", "<i8"), ("src0", "<u8"), ("src1", "<u8"), ("n", "<i4"), ("kind", "<i4"), ("a", "<i4"),
                      ("b", "<i4"), ("scale", "<f4"), ("lda", "<i4"), ("ldb", "<i4"), ("reserved", "<i4")])
assert SEG_DTYPE.itemsize == 56


def plan_segments(net, critic=False):
    """The packing plan of one SET network: (segments, offsets[NW + NSITES], total_floats, sources).

    Same layout as `pack_tensors` (slot order of include/sgrl_set.h, every slot 256-byte aligned) followed by the seven
    stacked projection operands, expressed as runs whose SOURCE is the parameter's own storage (`data_ptr()`), so the
    library can rebuild the buffer on the device whenever it wants."""
    sd = dict(net.named_parameters())
    segs, srcs, offs, pos = [], [], np.zeros(NW + NSITES + NEXTRA, dtype=np.int64), [0]

    def p(name):
        t = sd[name]
        assert t.dtype == torch.float32 and t.is_contiguous(), "SET parameters must be contiguous float32: " + name
        return t

    def emit(kind, t0, n, a=0, b=0, scale=1.0, t1=None, off0=0, off1=0, lda=0, ldb=0):
        """off0 / off1: element offsets into the source tensors (column / row blocks of a weight)."""
        segs.append((pos[0], t0.data_ptr() + 4 * off0, 0 if t1 is None else t1.data_ptr() + 4 * off1, n, kind, a, b, scale,
                     lda, ldb, 0))
        srcs.append((t0, t1, off0, off1))
        pos[0] += n

    anchor = p("g_encoder.weight")

    def align():
        fill = (-pos[0]) % 64
        if fill:
            emit(PACK_COPY, anchor, fill, a=0)

    def copy(name, scale=1.0, n=None):
        t = p(name)
        emit(PACK_COPY, t, t.numel() if n is None else n, a=t.numel(), scale=scale)

    def fold(name):
        t = p(name)
        assert t.shape[1] == 1024
        emit(PACK_FOLD, t, t.shape[0] * GRAM_K)

    def padcol(name, cols):
        t = p(name)
        emit(PACK_PADCOL, t, t.shape[0] * cols, a=t.shape[1], b=cols)

    def zero():
        emit(PACK_COPY, anchor, 64, a=0)

    def perm32(name):
        """[1024, K] (or [1024]) with the rows regrouped c * 32 + q <- q * 32 + c (include/sgrl_set.h PERM32)"""
        t = p(name)
        assert t.shape[0] == 1024
        emit(PACK_PERM32, t, t.numel(), a=t.numel() // 1024)

    def slot(i, fn, *args, **kw):
        align()
        offs[i] = pos[0]
        fn(*args, **kw)

    for i in range(3):
        slot(i, copy, "pos_encoder.embeddings.%d.weight" % i)
    slot(3, copy, "transformer_encoder.rel_encoder.weight"); slot(4, copy, "transformer_encoder.rel_encoder.bias")
    slot(5, copy, "transformer_encoder.norm.weight"); slot(6, copy, "transformer_encoder.norm.bias")
    slot(7, copy, "g_encoder.weight"); slot(8, copy, "encoder.weight"); slot(9, copy, "encoder.bias")
    slot(10, copy, "gg_proj.weight")
    slot(11, fold, "linear1_g.weight"); slot(12, copy, "linear1_g.bias")
    slot(13, copy, "linear2_g.weight"); slot(14, copy, "linear2_g.bias")
    slot(15, padcol, "linear1_ng.weight", 160); slot(16, copy, "linear1_ng.bias")
    slot(17, copy, "linear2_ng.weight"); slot(18, copy, "linear2_ng.bias")
    if critic:
        slot(19, copy, "decoder_ng.weight"); slot(20, zero); slot(21, copy, "decoder_ng.bias")
        slot(22, zero); slot(23, zero); slot(24, zero)
    else:
        slot(19, copy, "decoder_g.weight")
        slot(20, copy, "linear1_m.weight"); slot(21, copy, "linear1_m.bias")
        slot(22, perm32, "linear2_m.weight"); slot(23, perm32, "linear2_m.bias")
        slot(24, copy, "g_proj.weight")
    scaling = float(128) ** -0.5   # (2 * head_dim)^-0.5, reference subequivariant_attentions.py:88
    for l in range(LAYERS):
        lp = "transformer_encoder.layers.%d." % l
        at = lp + "self_attn."
        b0 = NGLOBAL + l * NLAYER

        wng, wgo = p(at + "ng_out.weight"), p(at + "g_out.weight")

        def qkv(kind):
            """q (scaled) | k | v folded through ng_out, head by head (include/sgrl_set.h)"""
            copy(at + "q_proj." + kind, scale=scaling); copy(at + "k_proj." + kind)
            v = p(at + "v_proj." + kind)
            for h in range(2):
                if kind == "weight":      # [128, 256] = Wng[:, 128h:128h+128] . Wv[128h:128h+128, :]
                    emit(PACK_MATMUL, wng, 128 * 256, a=128, b=256, t1=v, off0=128 * h, off1=128 * h * 256, lda=256, ldb=256)
                else:                     # [128] = Wng[:, 128h:128h+128] . bv[128h:128h+128]
                    emit(PACK_MATMUL, wng, 128, a=128, b=1, t1=v, off0=128 * h, off1=128 * h, lda=256, ldb=1)

        def ufold():
            """[256, 128]: rows 128 h + r = Wgo[r, 128h:128h+126] . Wvg[126h:126h+126, :]"""
            vgw = p(at + "vg_proj.weight")
            for h in range(2):
                emit(PACK_MATMUL, wgo, 128 * 128, a=126, b=128, t1=vgw, off0=128 * h, off1=126 * h * 128, lda=256, ldb=128)

        def gdcols():
            """[2, 128, 2]: the gravity / direction columns of g_out, per head"""
            for h in range(2):
                emit(PACK_SUBMAT, wgo, 128 * 2, b=2, off0=128 * h + 126, lda=256)
        slot(b0 + 0, copy, at + "g_proj.weight")
        slot(b0 + 1, fold, at + "linear_g1.weight"); slot(b0 + 2, copy, at + "linear_g1.bias")
        slot(b0 + 3, copy, at + "linear_g2.weight"); slot(b0 + 4, copy, at + "linear_g2.bias")
        slot(b0 + 5, qkv, "weight"); slot(b0 + 6, qkv, "bias")
        slot(b0 + 7, ufold)
        slot(b0 + 8, copy, at + "ng_out.weight"); slot(b0 + 9, copy, at + "ng_out.bias")
        slot(b0 + 10, copy, at + "g_out.weight")
        slot(b0 + 11, copy, lp + "g_proj2.weight"); slot(b0 + 12, copy, lp + "g_proj3.weight")
        slot(b0 + 13, fold, lp + "linear_g1.weight"); slot(b0 + 14, copy, lp + "linear_g1.bias")
        slot(b0 + 15, copy, lp + "linear_g2.weight"); slot(b0 + 16, copy, lp + "linear_g2.bias")
        slot(b0 + 17, copy, lp + "linear3.weight"); slot(b0 + 18, copy, lp + "linear3.bias")
        slot(b0 + 19, perm32, lp + "linear4.weight"); slot(b0 + 20, perm32, lp + "linear4.bias")
        slot(b0 + 21, copy, lp + "linear5.weight")
        slot(b0 + 22, copy, lp + "linear1.weight"); slot(b0 + 23, copy, lp + "linear1.bias")
        slot(b0 + 24, copy, lp + "linear2.weight"); slot(b0 + 25, copy, lp + "linear2.bias")
        slot(b0 + 26, copy, lp + "norm1.weight"); slot(b0 + 27, copy, lp + "norm1.bias")
        slot(b0 + 28, copy, lp + "norm2.weight"); slot(b0 + 29, copy, lp + "norm2.bias")
        slot(b0 + 30, gdcols)

    def stack(w0, w1, cols, cpad):
        t0 = p(w0)
        assert tuple(t0.shape) == (30, cols)
        emit(PACK_STACK, t0, 64 * cpad, a=cols, b=cpad, t1=None if w1 is None else p(w1))

    for l in range(LAYERS):
        lp = "transformer_encoder.layers.%d." % l
        slot(NW + 2 * l, stack, lp + "self_attn.g_proj.weight", None, 128, 128)
        slot(NW + 2 * l + 1, stack, lp + "g_proj2.weight", lp + "g_proj3.weight", 128, 128)
    slot(NW + 6, stack, "gg_proj.weight", None if critic else "g_proj.weight", 136, 144)

    def head_fold(kind):
        """decoder_g folded through linear2_m (exact algebra, reference SEActor.py:272-279: decoder_g(z . mat) = z . (mat . wdec)):
        row q = sum_c wdec[c] * linear2_m[q * 32 + c] -- the head's 1024-wide product becomes a 32-wide one."""
        wd, t = p("decoder_g.weight"), p("linear2_m." + kind)
        for q in range(32):
            if kind == "weight":
                emit(PACK_MATMUL, wd, 256, a=32, b=256, t1=t, off1=q * 32 * 256, lda=32, ldb=256)
            else:
                emit(PACK_MATMUL, wd, 1, a=32, b=1, t1=t, off1=q * 32, lda=32, ldb=1)
    if critic:
        slot(NW + 7, zero); slot(NW + 8, zero)
    else:
        slot(NW + 7, head_fold, "weight"); slot(NW + 8, head_fold, "bias")
    align()
    return np.array(segs, dtype=SEG_DTYPE), offs, pos[0], srcs


def graph_key(graph):
    """Content key of a graph dict (the parents vector determines traversals and relation: sgrl_amd/graph.py)."""
    return tuple(int(v) for v in graph["parents"])


class HipSetActor(object):
    """HIP forward of one SET network: the actor of an `SEPolicy` (default) or, with `net=` / `critic=True`, one critic
    `TransformerModel` of an `SECritic` (see `HipSetCritic`)."""

    def __init__(self, policy, device=None, net=None, critic=False):
        if not torch.cuda.is_available():
            raise _lib.SgrlError("HipSetActor needs an MI355X (no CPU fallback)")
        self.L = _lib.lib()
        _bind(self.L)
        self.policy = policy
        self.net = net if net is not None else policy.actor
        self.critic = bool(critic)
        self.device = torch.device(device) if device is not None else next(self.net.parameters()).device
        if self.device.type != "cuda":
            raise _lib.SgrlError("the SEPolicy must live on the GPU for the HIP path")
        h = ctypes.c_void_p()
        _check(self.L, self.L.sgrl_set_create(ctypes.byref(h)), "sgrl_set_create")
        self.h = h
        self._bound = None
        self._cfg_key = None
        self._cfg_info = {}
        self.n_env = 0
        self.act_ld = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.sgrl_set_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------------
    def sync_weights(self, force=False):
        """Bind the handle to the parameters' storage.  The VALUES are re-read by the library on every forward; this only
        has to run again when a parameter's storage moves (module.to(), re-created tensors)."""
        ptrs = tuple(p.data_ptr() for p in self.net.parameters())
        if not force and ptrs == self._bound:
            return
        segs, offs, total, srcs = plan_segments(self.net, critic=self.critic)
        if not all(src[0].is_cuda for src in srcs):
            raise _lib.SgrlError("the SET network must live on the GPU for the HIP path")
        _check(self.L, self.L.sgrl_set_bind_params(self.h, ctypes.c_void_p(segs.ctypes.data), len(segs),
                                                   ctypes.c_void_p(offs.ctypes.data), len(offs), ctypes.c_int64(total)),
               "sgrl_set_bind_params")
        self._bound = ptrs

    # ---- batch structure ------------------------------------------------------------------------------
    def configure(self, graphs, counts):
        """graphs: per-morphology dicts with 'parents', 'traversals' (3 index vectors) and 'relation' [L,L,3]; counts:
        envs each.  Structures seen before are switched to without any device work (the library caches them by content)."""
        key = (tuple(graph_key(g) for g in graphs), tuple(int(c) for c in counts))
        if key == self._cfg_key:
            return
        args = self._cfg_info.get(key)
        if args is None:
            Ls, trav, rel = [], [], []
            for g in graphs:
                t = [np.asarray(v.cpu() if torch.is_tensor(v) else v, dtype=np.int32) for v in g["traversals"]]
                Ls.append(len(t[0]))
                trav.append(np.concatenate(t))
                r = g["relation"]
                rel.append(np.asarray(r.detach().cpu() if torch.is_tensor(r) else r, dtype=np.float32).reshape(-1))
            args = (np.asarray(Ls, dtype=np.int32), np.asarray(counts, dtype=np.int32),
                    np.ascontiguousarray(np.concatenate(trav), dtype=np.int32),
                    np.ascontiguousarray(np.concatenate(rel), dtype=np.float32))
            if len(self._cfg_info) >= 128:
                self._cfg_info.clear()
            self._cfg_info[key] = args
        Ls, cnt, trav, rel = args
        vp = lambda a: ctypes.c_void_p(a.ctypes.data)
        _check(self.L, self.L.sgrl_set_graph(self.h, len(Ls), vp(Ls), vp(cnt), vp(trav), vp(rel)), "sgrl_set_graph")
        self._cfg_key = key
        self.n_env = int(cnt.sum())
        self.max_limbs = int(Ls.max())
        self.num_nodes = self.L.sgrl_set_num_nodes(self.h)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ld(t):
        """Row stride in floats; a one-row tensor may report any stride (NumPy's [None, :] gives 0): use its width."""
        return int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])

    def forward_batch(self, obs, out=None, act_ld=None):
        """obs: float32 CUDA [n_env, obs_ld] (rows zero padded beyond 41*L) -> actions float32 [n_env, act_ld]."""
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.dim() == 2 and obs.stride(1) == 1
        assert obs.shape[0] == self.n_env
        assert obs.shape[1] >= 41 * self.max_limbs, "observation rows narrower than 41 * max_limbs"
        self.sync_weights()
        act_ld = act_ld or 3 * self.max_limbs
        assert act_ld >= 3 * self.max_limbs, "action rows narrower than 3 * max_limbs"
        if out is None:
            out = torch.empty((self.n_env, act_ld), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.shape == (self.n_env, act_ld)
        _check(self.L, self.L.sgrl_set_forward(self.h, ctypes.c_void_p(obs.data_ptr()), self._ld(obs),
                                               ctypes.c_void_p(out.data_ptr()), int(act_ld),
                                               ctypes.c_float(float(self.policy.max_action)), self._stream()),
               "sgrl_set_forward")
        return out

    def forward_q(self, obs, action, out=None, q_ld=None):
        """critic network: obs [n_env, obs_ld], action [n_env, act_ld] (3 slots per limb) -> per-limb Q [n_env, q_ld]."""
        assert self.critic, "forward_q needs a handle created with critic=True"
        for t in (obs, action):
            assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == self.n_env
        assert obs.shape[1] >= 41 * self.max_limbs and action.shape[1] >= 3 * self.max_limbs
        self.sync_weights()
        q_ld = q_ld or self.max_limbs
        assert q_ld >= self.max_limbs
        if out is None:
            out = torch.empty((self.n_env, q_ld), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.shape == (self.n_env, q_ld)
        _check(self.L, self.L.sgrl_set_forward_q(self.h, ctypes.c_void_p(obs.data_ptr()), self._ld(obs),
                                                 ctypes.c_void_p(action.data_ptr()), self._ld(action),
                                                 ctypes.c_void_p(out.data_ptr()), int(q_ld), self._stream()),
               "sgrl_set_forward_q")
        return out

    def time_forward(self, obs, out, reps):
        self.sync_weights()
        ms = ctypes.c_float(0)
        _check(self.L, self.L.sgrl_set_time_forward(self.h, ctypes.c_void_p(obs.data_ptr()), self._ld(obs),
                                                    ctypes.c_void_p(out.data_ptr()), self._ld(out),
                                                    ctypes.c_float(float(self.policy.max_action)), int(reps),
                                                    self._stream(), ctypes.byref(ms)), "sgrl_set_time_forward")
        return float(ms.value)

    def forward_single(self, state, graph):
        """SEPolicy.forward(state [B, 41*L]) for the current morphology (reference agent.py:197)."""
        B = state.shape[0]
        self.configure([graph], [B])
        return self.forward_batch(state.contiguous().float(), act_ld=3 * len(graph["parents"]))

    def debug_stop_after(self, stage):
        """Parity probes: the following forwards return after stage 2l (attention of layer l) / 2l+1 (layer l); -1 = full."""
        _check(self.L, self.L.sgrl_set_debug_stop_after(self.h, int(stage)), "sgrl_set_debug_stop_after")

    def debug_small_nodes(self, nodes):
        """Batches of at most `nodes` nodes take the small-batch products (include/sgrl_set.h); 0 = never, -1 = default."""
        _check(self.L, self.L.sgrl_set_debug_small_nodes(self.h, int(nodes)), "sgrl_set_debug_small_nodes")

    FORM_F16X3, FORM_BF16X6 = 2, 3

    def gemm_form(self, form):
        """Form of the tile products (include/sgrl_set.h): FORM_F16X3 (default: two f16 pieces of every row-scaled operand,
        float32's range), FORM_BF16X6 (three bf16 pieces, slower; A/B comparisons), 0 = default."""
        _check(self.L, self.L.sgrl_set_gemm_form(self.h, int(form)), "sgrl_set_gemm_form")

    def hold_weights(self, hold=True):
        """Promise that the bound parameters do not change until the next call (include/sgrl_set.h sgrl_set_hold_weights): the next
        forward packs the weights, the following ones reuse them.  Call again (True) right after anything that changes the
        parameters -- an optimizer step, a soft update, a broadcast -- or with False to go back to packing on every forward."""
        self.sync_weights()
        _check(self.L, self.L.sgrl_set_hold_weights(self.h, 1 if hold else 0), "sgrl_set_hold_weights")

    def scale_redos(self, reset=True):
        """Workgroups that repeated a tile with exact row maxima since the last reset (include/sgrl_set.h sgrl_set_debug_redos;
        process-wide, synchronises): 0 unless an operand row's sampled estimate was more than 512 x below its maximum -- or zero
        (every sampled entry of a sparse row 0: "unknown", the tile is repeated on the exact maxima unless the whole row is zero)."""
        return int(self.L.sgrl_set_debug_redos(1 if reset else 0))

    def peek(self, which, per_node):
        out = np.zeros((self.num_nodes, per_node), dtype=np.float32)
        _check(self.L, self.L.sgrl_set_peek(self.h, which, ctypes.c_void_p(out.ctypes.data), out.size), "sgrl_set_peek")
        return out


class HipSetCritic(object):
    """Twin critics of an `SECritic` on the HIP path (inference only: the TD3 target values, reference agent.py:136-148):
    two handles, one per TransformerModel, sharing the batch structure."""

    def __init__(self, critic_module, device=None):
        self.q1 = HipSetActor(critic_module, device=device, net=critic_module.critic1, critic=True)
        self.q2 = HipSetActor(critic_module, device=device, net=critic_module.critic2, critic=True)

    def configure(self, graphs, counts):
        self.q1.configure(graphs, counts)
        self.q2.configure(graphs, counts)

    def forward_batch(self, obs, action, q_ld=None, which=(1, 2)):
        out = []
        if 1 in which:
            out.append(self.q1.forward_q(obs, action, q_ld=q_ld))
        if 2 in which:
            out.append(self.q2.forward_q(obs, action, q_ld=q_ld))
        return tuple(out)

    def forward_single(self, state, action, graph, which=(1, 2)):
        """SECritic.forward(state [B, 41 L], action [B, 3 L]) for one morphology -> per-limb Q [B, L] each."""
        B, L = state.shape[0], len(graph["parents"])
        self.configure([graph], [B])
        return self.forward_batch(state.contiguous().float(), action.contiguous().float(), q_ld=L, which=which)
