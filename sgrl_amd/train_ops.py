"""Differentiable linear layers of the TD3 update on the HIP kernels of csrc/train_gemm.hip (C ABI: include/sgrl_train.h).

`linear(x, weight, bias, relu)` is `torch.nn.functional.linear` (+ ReLU) as a `torch.autograd.Function` whose forward,
input gradient and weight / bias gradient run on this library's own small-product kernels instead of the vendor GEMMs
(which pick single-workgroup 256 x 256 tilings for the update's 700-row problems: DESIGN.md section 5).  Used by the SET
modules (set_policy.py) whenever autograd is recording on the GPU -- i.e. inside `Agent.update` (reference agent.py:117-183);
the no-grad rollout path is the fused forward of csrc/set_actor.hip.  No CPU fallback: on the CPU the modules use
`F.linear`; on the GPU a missing extension raises.
"""
import contextlib
import ctypes
import os

import numpy as np

import torch
import torch.nn.functional as F

from . import _lib

_bound = False
_ws = {}
ENABLED = os.environ.get("SGRL_TRAIN_GEMM", "1") != "0"      # 0: vendor GEMMs (A/B comparisons)


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.sgrl_linear_forward.argtypes = [vp, ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]
        L.sgrl_linear_backward.argtypes = [vp, ci, vp, ci, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, vp, vp]
        L.sgrl_linear_backward_xrelu.argtypes = [vp, ci, vp, ci, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp]
        L.sgrl_linear_dgrad_twin_xrelu.argtypes = [vp, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, vp]
        L.sgrl_linear_backward_acc.argtypes = [vp, ci, vp, ci, ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp, vp]
        L.sgrl_linear_dgrad_twin_acc.argtypes = [vp, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
        L.sgrl_linear_forward_twin.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
        L.sgrl_linear_dgrad_twin.argtypes = [vp, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, vp]
        L.sgrl_gram_forward.argtypes = [vp, vp, vp, ci, vp]
        L.sgrl_gram_backward.argtypes = [vp, vp, vp, vp, vp, ci, vp]
        L.sgrl_gram_tri_forward.argtypes = [vp, vp, vp, ci, vp]
        L.sgrl_gram_tri_backward.argtypes = [vp, vp, vp, vp, vp, ci, vp]
        L.sgrl_sym_fold.argtypes = [ci, vp, vp, vp, ci, vp]
        L.sgrl_linear_wgrad_group.argtypes = [ci, vp, vp, vp]
        L.sgrl_zmat_forward.argtypes = [vp, vp, vp, ci, vp]
        L.sgrl_zmat_backward.argtypes = [vp, vp, vp, vp, vp, ci, vp]
        L.sgrl_attention_forward.argtypes = [vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, ci, ci, vp]
        L.sgrl_attention_backward.argtypes = [vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
        L.sgrl_linear_forward_fused.argtypes = [vp, ci, vp, ci, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]
        L.sgrl_linear_forward_twin_fused.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp]
        L.sgrl_embed3_forward.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, ci, vp]
        L.sgrl_embed3_backward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
        L.sgrl_add_ln_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ctypes.c_float, vp]
        L.sgrl_add_ln_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
        L.sgrl_train_ws_floats.restype = ctypes.c_int64
        L.sgrl_train_last_error.restype = ctypes.c_char_p
        _bound = True
    return L


def _check(L, rc, what):
    if rc != 0:
        raise _lib.SgrlError("%s failed (%d): %s" % (what, rc, L.sgrl_train_last_error().decode()))


def _scratch(device):
    """Scratch of the split weight-gradient contractions (zero filled once; every call leaves its counters zero): one buffer
    per (device, stream) -- calls on one stream run in order (include/sgrl_train.h)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws.get(key)
    if ws is None:
        ws = torch.zeros(int(_L().sgrl_train_ws_floats()), dtype=torch.float32, device=device)
        _ws[key] = ws
    return ws


# struct sgrl_wgrad_desc (include/sgrl_train.h)
_DESC = np.dtype([("dy", "<u8"), ("y", "<u8"), ("rowdiv", "<u8"), ("x", "<u8"), ("dw", "<u8"), ("db", "<u8"), ("lddy", "<i4"),
                  ("ldy", "<i4"), ("ldx", "<i4"), ("lddw", "<i4"), ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("relu", "<i4")])
assert _DESC.itemsize == 80
_pending = None        # None: weight gradients are computed inside backward(); a list: they are postponed (deferred_wgrads)


@contextlib.contextmanager
def deferred_wgrads(enabled=True):
    """Inside this context the backward() of a linear layer whose weight (and bias) are leaf parameters launches only the
    INPUT-gradient product; its weight / bias gradients -- which nothing needs before the optimizer steps -- are collected and
    issued together (12 layers per launch) when the context exits, and only then stored into (or added to) the parameters'
    `.grad`: autograd itself never sees a gradient tensor that has not been computed yet."""
    global _pending
    if not enabled or _pending is not None:
        yield
        return
    _pending = []
    try:
        yield
    finally:
        todo, _pending = _pending, None
        flush_wgrads(todo)


def flush_wgrads(todo):
    if not todo:
        return
    L = _L()
    by_stream = {}
    for rec in todo:
        by_stream.setdefault((rec["dev"], rec["stream"]), []).append(rec)
    for (dev, stream), recs in by_stream.items():
        d = np.zeros(len(recs), dtype=_DESC)
        for i, r in enumerate(recs):
            d[i] = (r["dy"].data_ptr(), 0 if r["y"] is None else r["y"].data_ptr(), 0 if r["rowdiv"] is None else r["rowdiv"].data_ptr(),
                    r["x"].data_ptr(), r["dw"].data_ptr(), 0 if r["db"] is None else r["db"].data_ptr(), r["dy"].stride(0),
                    r.get("ldy", r["N"]), r["x"].stride(0), r["K"], r["M"], r["N"], r["K"], 1 if r["relu"] else 0)
        _check(L, L.sgrl_linear_wgrad_group(len(recs), ctypes.c_void_p(d.ctypes.data), _p(_scratch(dev)), ctypes.c_void_p(stream)),
               "sgrl_linear_wgrad_group")
        with torch.no_grad():
            # gradients of FOLDED weights (tri_weights): spread onto both mirror columns of their leaf parameters, one launch for all
            folded = [r for r in recs if r.get("fold_param") is not None]
            for i0 in range(0, len(folded), 16):
                part = folded[i0:i0 + 16]
                full = [torch.empty_like(r["fold_param"]) for r in part]
                _sym_fold_call(full, [r["dw"] for r in part], True, dev)
                for r, f in zip(part, full):
                    r["w_param"], r["dw"] = r["fold_param"], f
            for r in recs:
                for prm, g in ((r["w_param"], r["dw"]), (r["b_param"], r["db"])):
                    if prm is None:
                        continue
                    g = g.view_as(prm)
                    if prm.grad is None:
                        prm.grad = g
                    else:
                        prm.grad.add_(g)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu, rowdiv, addend=None, tail=None, x_relu=False, premasked=False, slot=None):
        # slot: x is one output of fan_out() -- this layer's input gradient is written into / accumulated onto the fan-out's shared
        # buffer (sgrl_linear_backward_acc) instead of travelling through autograd's own additions.
        # x_relu: x is the output of a ReLU layer whose backward is told `premasked` -- this layer's input gradient comes out masked
        # by x > 0 (the dgrad kernel's epilogue), and THAT layer's two backward products read no mask (include/sgrl_train.h
        # sgrl_linear_backward_xrelu).  Valid when nothing else consumes the ReLU layer's output (the SET feed-forward pairs).
        L = _L()
        N, K = weight.shape
        assert not premasked or relu
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) < K:
            x2 = x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        M = x2.shape[0]
        rd = None
        if rowdiv is not None:
            rd = rowdiv.reshape(-1)
            rd = rd if rd.is_contiguous() else rd.contiguous()
            assert rd.shape[0] == M and not relu
        # fused followers (include/sgrl_train.h sgrl_linear_forward_fused): a residual added to the result, columns appended to it
        nt = 0 if tail is None else tail.shape[-1]
        ad2 = tl2 = None
        if addend is not None:
            assert not relu and rd is None and tail is None
            ad2 = addend.reshape(M, N)
            ad2 = ad2 if (ad2.stride(1) == 1 and ad2.stride(0) >= N) else ad2.contiguous()
        if tail is not None:
            tl2 = tail.reshape(M, nt)
            tl2 = tl2 if tl2.is_contiguous() else tl2.contiguous()
        y = torch.empty((M, N + nt), dtype=torch.float32, device=x.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if ad2 is None and tl2 is None:
            _check(L, L.sgrl_linear_forward(_p(x2), x2.stride(0), _p(w), K, _p(bias), _p(rd), _p(y), N, M, N, K, 1 if relu else 0, st),
                   "sgrl_linear_forward")
        else:
            _check(L, L.sgrl_linear_forward_fused(_p(x2), x2.stride(0), _p(w), K, _p(bias), _p(rd), _p(ad2), 0 if ad2 is None else ad2.stride(0),
                                                  _p(tl2), nt, _p(y), N + nt, M, N, K, 1 if relu else 0, st), "sgrl_linear_forward_fused")
        ctx.ntail = nt
        ctx.ad_shape = None if addend is None else addend.shape
        ctx.tail_shape = None if tail is None else tail.shape
        ctx.save_for_backward(x2, w, y if (relu or rd is not None) else None, rd)
        ctx.has_bias, ctx.relu = bias is not None, bool(relu)
        ctx.x_relu, ctx.mask = bool(x_relu), bool(relu) and not premasked       # mask: dy still has to be masked by y > 0 here
        # leaf parameters (what deferred_wgrads may postpone): kept by reference so that their .grad can be set at the flush
        ctx.leaf = (weight, bias) if (weight.is_leaf and (bias is None or bias.is_leaf)) else None
        # a folded weight (tri_weights) whose leaf is known: its gradient may be postponed too -- it is unfolded into the leaf's .grad
        # at the flush, and autograd sees no gradient for the folded tensor
        ctx.fold_leaf = getattr(weight, "_sgrl_fold_leaf", None) if (ctx.leaf is None and (bias is None or bias.is_leaf)) else None
        ctx.bias_leaf = bias if (bias is not None and bias.is_leaf) else None
        ctx.x_shape = x.shape
        ctx.rd_shape = None if rowdiv is None else rowdiv.shape
        ctx.slot = slot
        return y.view(*x.shape[:-1], N + nt)

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        x2, w, yo, rd = ctx.saved_tensors
        N, K = w.shape
        M = x2.shape[0]
        nt = ctx.ntail
        dyf = dy.reshape(M, N + nt)
        if dyf.stride(1) != 1 or dyf.stride(0) < N + nt:
            dyf = dyf.contiguous()
        dy2 = dyf[:, :N] if nt else dyf             # the product's own columns: a row-strided view, read as is
        ldyo = N + nt                               # row stride of the saved output (ReLU mask / row-divisor gradient)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        need_rd = rd is not None and ctx.needs_input_grad[4]
        if need_b and not need_w:
            need_w = True                       # the bias gradient rides on the weight-gradient kernel
        acc_dx = False
        slot = ctx.slot if need_x else None
        if slot is not None and slot.buf is not None:          # a sibling consumer was here first: add onto its gradient
            dx, acc_dx = slot.buf.view(M, K), True
        else:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device) if need_x else None
            if slot is not None:
                slot.buf = dx
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device) if need_w else None
        db = torch.empty((N,), dtype=torch.float32, device=dy.device) if need_b else None
        drd = torch.empty((M,), dtype=torch.float32, device=dy.device) if need_rd else None
        stream = torch.cuda.current_stream(dy.device).cuda_stream
        st = ctypes.c_void_p(stream)
        now_w, now_b = dw, db
        deferred = _pending is not None and need_w and (ctx.leaf is not None or ctx.fold_leaf is not None)
        if deferred:                              # postponed: computed and stored into .grad when the deferred_wgrads context exits
            folded = ctx.leaf is None
            _pending.append({"dy": dy2, "y": yo if ctx.mask else None, "rowdiv": rd, "x": x2, "dw": dw, "db": db, "M": M, "N": N,
                             "K": K, "relu": ctx.mask, "dev": dy.device, "stream": stream, "ldy": ldyo,
                             "w_param": ctx.leaf[0] if (not folded and ctx.needs_input_grad[1]) else None,
                             "fold_param": ctx.fold_leaf if (folded and ctx.needs_input_grad[1]) else None,
                             "b_param": (ctx.bias_leaf if need_b else None)})
            now_w = now_b = None
        if dx is not None or now_w is not None or now_b is not None or drd is not None:
            _check(L, L.sgrl_linear_backward_acc(_p(dy2), dy2.stride(0), _p(yo), ldyo, 1 if ctx.mask else 0, _p(rd), _p(x2), x2.stride(0),
                                                 _p(w), K, _p(dx), K, _p(now_w), K, _p(now_b), _p(drd), M, N, K, 1 if ctx.x_relu else 0,
                                                 1 if acc_dx else 0, _p(_scratch(dy.device)), st), "sgrl_linear_backward_acc")
        dadd = dy.reshape(ctx.ad_shape) if (ctx.ad_shape is not None and ctx.needs_input_grad[5]) else None
        dtail = dyf[:, N:].reshape(ctx.tail_shape) if (nt and ctx.needs_input_grad[6]) else None
        # (a slot consumer hands its gradient to the fan-out through the slot: the first one also returns it so that the fan-out's
        # backward is certain to run; the others return nothing)
        return (dx.view(ctx.x_shape) if (need_x and not acc_dx) else None), (None if deferred else (dw if ctx.needs_input_grad[1] else None)), \
               (None if deferred else db), None, (drd.view(ctx.rd_shape) if need_rd else None), dadd, dtail, None, None, None


class _Linear2Fn(torch.autograd.Function):
    """The same layer of two networks of identical shape in one launch (csrc/train_gemm.hip k_sgemm_twin; include/sgrl_train.h):
    x is either ONE input both share, [..., K], or their two inputs stacked, [2, ..., K]; the result is stacked, [2, ..., N]."""

    @staticmethod
    def forward(ctx, x, w0, w1, b0, b1, relu, rowdiv, shared, addend=None, tail=None, x_relu=False, premasked=False, slot=None):
        L = _L()
        assert slot is None or not shared
        N, K = w0.shape
        assert w1.shape == w0.shape and (b0 is None) == (b1 is None)
        assert (not premasked or relu) and not (x_relu and shared)        # x_relu / premasked: see _LinearFn
        lead = x.shape[:-1] if shared else x.shape[1:-1]
        xs = x.reshape(-1, K) if shared else x.reshape(2, -1, K)
        if xs.stride(-1) != 1 or xs.stride(-2) < K or (not shared and xs.stride(0) < 0):
            xs = xs.contiguous()
        M = xs.shape[-2]
        x0, x1 = (xs, xs) if shared else (xs[0], xs[1])
        params = ((w0, b0), (w1, b1))               # the tensors autograd knows (leaf parameters, or e.g. a concatenation of some)
        w0 = w0 if w0.is_contiguous() else w0.contiguous()
        w1 = w1 if w1.is_contiguous() else w1.contiguous()
        rd = None
        if rowdiv is not None:
            rd = rowdiv.reshape(2, -1)
            rd = rd if rd.is_contiguous() else rd.contiguous()
            assert rd.shape[1] == M and not relu
        # fused followers: addend [2, ..., N] (a residual per network), tail [..., nt] or [2, ..., nt] (columns appended to each result)
        nt = 0 if tail is None else tail.shape[-1]
        ad2 = tl = None
        if addend is not None:
            assert not relu and rd is None and tail is None
            ad2 = addend.reshape(2, M, N)
            ad2 = ad2 if ad2.is_contiguous() else ad2.contiguous()
        if tail is not None:
            tl = tail.reshape(-1, M, nt)             # one tail both networks share, or one each
            tl = tl if tl.is_contiguous() else tl.contiguous()
            tl = (tl[0], tl[0]) if tl.shape[0] == 1 else (tl[0], tl[1])
        y = torch.empty((2, M, N + nt), dtype=torch.float32, device=x.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if ad2 is None and tl is None:
            _check(L, L.sgrl_linear_forward_twin(_p(x0), _p(x1), xs.stride(-2), _p(w0), _p(w1), K, _p(b0), _p(b1),
                                                 _p(None if rd is None else rd[0]), _p(None if rd is None else rd[1]), _p(y[0]), _p(y[1]),
                                                 N, M, N, K, 1 if relu else 0, st), "sgrl_linear_forward_twin")
        else:
            _check(L, L.sgrl_linear_forward_twin_fused(_p(x0), _p(x1), xs.stride(-2), _p(w0), _p(w1), K, _p(b0), _p(b1),
                                                       _p(None if rd is None else rd[0]), _p(None if rd is None else rd[1]),
                                                       _p(None if ad2 is None else ad2[0]), _p(None if ad2 is None else ad2[1]), N,
                                                       _p(None if tl is None else tl[0]), _p(None if tl is None else tl[1]), nt,
                                                       _p(y[0]), _p(y[1]), N + nt, M, N, K, 1 if relu else 0, st),
                   "sgrl_linear_forward_twin_fused")
        ctx.ntail = nt
        ctx.ad_shape = None if addend is None else addend.shape
        ctx.tail_shape = None if tail is None else tail.shape
        ctx.save_for_backward(xs, w0, w1, y if (relu or rd is not None) else None, rd)
        ctx.has_bias, ctx.relu, ctx.shared = b0 is not None, bool(relu), bool(shared)
        ctx.x_relu, ctx.mask = bool(x_relu), bool(relu) and not premasked
        ctx.leaf = [((w, b) if (w.is_leaf and (b is None or b.is_leaf)) else None) for w, b in params]
        ctx.fold_leaf = [(getattr(w, "_sgrl_fold_leaf", None) if (lf is None and (b is None or b.is_leaf)) else None)
                         for (w, b), lf in zip(params, ctx.leaf)]
        ctx.params = params
        ctx.x_shape = x.shape
        ctx.rd_shape = None if rowdiv is None else rowdiv.shape
        ctx.slot = slot
        return y.view(2, *lead, N + nt)

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        xs, w0, w1, yo, rd = ctx.saved_tensors
        N, K = w0.shape
        M = xs.shape[-2]
        nt = ctx.ntail
        dyf = dy.reshape(2, M, N + nt)               # a row-strided view (the gradient of one part of a concatenation) is read as is
        if dyf.stride(2) != 1 or dyf.stride(1) < N + nt or dyf.stride(0) < 0:
            dyf = dyf.contiguous()
        dy2 = dyf[:, :, :N] if nt else dyf
        lddy = dy2.stride(1)
        ldyo = N + nt
        need_x = ctx.needs_input_grad[0]
        need_w = [ctx.needs_input_grad[1], ctx.needs_input_grad[2]]
        need_b = [ctx.has_bias and ctx.needs_input_grad[3], ctx.has_bias and ctx.needs_input_grad[4]]
        need_rd = rd is not None and ctx.needs_input_grad[6]
        dev = dy.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        st = ctypes.c_void_p(stream)
        acc_dx = False
        slot = ctx.slot if need_x else None
        if slot is not None and slot.buf is not None:
            dx, acc_dx = slot.buf.view(2, M, K), True
        else:
            dx = torch.empty((2, M, K), dtype=torch.float32, device=dev) if need_x else None
            if slot is not None:
                slot.buf = dx
        drd = torch.empty((2, M), dtype=torch.float32, device=dev) if need_rd else None
        if need_x:
            xm = (xs[0], xs[1]) if ctx.x_relu else (None, None)
            _check(L, L.sgrl_linear_dgrad_twin_acc(_p(dy2[0]), _p(dy2[1]), lddy, _p(None if yo is None else yo[0]),
                                                   _p(None if yo is None else yo[1]), ldyo, 1 if ctx.mask else 0,
                                                   _p(None if rd is None else rd[0]), _p(None if rd is None else rd[1]),
                                                   _p(w0), _p(w1), K, _p(dx[0]), _p(dx[1]), K, _p(None if drd is None else drd[0]),
                                                   _p(None if drd is None else drd[1]), _p(xm[0]), _p(xm[1]), xs.stride(-2), M, N, K,
                                                   1 if acc_dx else 0, st), "sgrl_linear_dgrad_twin_acc")
        elif need_rd:                               # the row divisor's gradient without an input gradient: the single-network kernel twice
            for i in range(2):
                _check(L, L.sgrl_linear_backward(_p(dy2[i]), lddy, _p(yo[i]), ldyo, 0, _p(rd[i]), _p(None), 0, _p(None), 0, _p(None), 0,
                                                 _p(None), 0, _p(None), _p(drd[i]), M, N, K, _p(_scratch(dev)), st), "sgrl_linear_backward")
        # weight / bias gradients: one descriptor per network -- postponed (deferred_wgrads) or issued together now
        grads_w, grads_b, recs = [None, None], [None, None], []
        for i, (w, b) in enumerate(ctx.params):
            if not (need_w[i] or need_b[i]):
                continue
            x_i = xs if ctx.shared else xs[i]
            dw = torch.empty((N, K), dtype=torch.float32, device=dev)
            db = torch.empty((N,), dtype=torch.float32, device=dev) if need_b[i] else None
            folded = ctx.leaf[i] is None and ctx.fold_leaf[i] is not None
            deferred = _pending is not None and (ctx.leaf[i] is not None or folded)
            rec = {"dy": dy2[i], "y": yo[i] if ctx.mask else None, "rowdiv": None if rd is None else rd[i], "x": x_i, "dw": dw, "db": db,
                   "M": M, "N": N, "K": K, "relu": ctx.mask, "dev": dev, "stream": stream, "ldy": ldyo,
                   "w_param": (w if need_w[i] else None) if (deferred and not folded) else None,
                   "fold_param": (ctx.fold_leaf[i] if need_w[i] else None) if (deferred and folded) else None,
                   "b_param": (b if need_b[i] else None) if deferred else None}
            if deferred:
                _pending.append(rec)
            else:
                recs.append(rec)
                grads_w[i], grads_b[i] = (dw if need_w[i] else None), db
        if recs:
            flush_wgrads(recs)                      # w_param / b_param are None: nothing is stored, the gradients are returned below
        dx_out = None
        if need_x and not acc_dx:
            dx_out = (dx[0] + dx[1]).view(ctx.x_shape) if ctx.shared else dx.view(ctx.x_shape)
        dadd = dy.reshape(ctx.ad_shape) if (ctx.ad_shape is not None and ctx.needs_input_grad[8]) else None
        dtail = None
        if nt and ctx.needs_input_grad[9]:
            dtail = dyf[:, :, N:].reshape(2, *ctx.x_shape[(0 if ctx.shared else 1):-1], nt).sum_to_size(ctx.tail_shape)
        return dx_out, grads_w[0], grads_w[1], grads_b[0], grads_b[1], None, (drd.view(ctx.rd_shape) if need_rd else None), None, dadd, dtail, \
            None, None, None


class _GramFn(torch.autograd.Function):
    """z [..., 3, 32] -> (vec(Z'Z) [..., 1024], ||Z'Z||_F + 1 [..., 1])."""

    @staticmethod
    def forward(ctx, z):
        L = _L()
        z2 = z.reshape(-1, 96)
        z2 = z2 if z2.is_contiguous() else z2.contiguous()
        M = z2.shape[0]
        gram = torch.empty((M, 1024), dtype=torch.float32, device=z.device)
        fn = torch.empty((M,), dtype=torch.float32, device=z.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        _check(L, L.sgrl_gram_forward(_p(z2), _p(gram), _p(fn), M, st), "sgrl_gram_forward")
        ctx.save_for_backward(z2, fn)
        ctx.z_shape = z.shape
        lead = z.shape[:-2]
        return gram.view(*lead, 1024), fn.view(*lead, 1)

    @staticmethod
    def backward(ctx, dgram, dfn):
        L = _L()
        z2, fn = ctx.saved_tensors
        M = z2.shape[0]
        dg = None if dgram is None else dgram.reshape(M, 1024)
        if dg is not None and not dg.is_contiguous():
            dg = dg.contiguous()
        df = None if dfn is None else dfn.reshape(M)
        if df is not None and not df.is_contiguous():
            df = df.contiguous()
        dz = torch.empty((M, 96), dtype=torch.float32, device=z2.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(z2.device).cuda_stream)
        _check(L, L.sgrl_gram_backward(_p(z2), _p(dg), _p(df), _p(fn), _p(dz), M, st), "sgrl_gram_backward")
        return dz.view(ctx.z_shape)


TRI = 528      # lower triangle of a symmetric 32 x 32 matrix, packed k = a (a + 1) / 2 + b (include/sgrl_train.h)


class _GramTriFn(torch.autograd.Function):
    """z [..., 3, 32] -> (tri(Z'Z) [..., 528], ||Z'Z||_F + 1 [..., 1]): the invariants without their mirror copies."""

    @staticmethod
    def forward(ctx, z):
        L = _L()
        z2 = z.reshape(-1, 96)
        z2 = z2 if z2.is_contiguous() else z2.contiguous()
        M = z2.shape[0]
        tri = torch.empty((M, TRI), dtype=torch.float32, device=z.device)
        fn = torch.empty((M,), dtype=torch.float32, device=z.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        _check(L, L.sgrl_gram_tri_forward(_p(z2), _p(tri), _p(fn), M, st), "sgrl_gram_tri_forward")
        ctx.save_for_backward(z2, fn)
        ctx.z_shape = z.shape
        lead = z.shape[:-2]
        return tri.view(*lead, TRI), fn.view(*lead, 1)

    @staticmethod
    def backward(ctx, dtri, dfn):
        L = _L()
        z2, fn = ctx.saved_tensors
        M = z2.shape[0]
        dg = None if dtri is None else dtri.reshape(M, TRI)
        if dg is not None and not dg.is_contiguous():
            dg = dg.contiguous()
        df = None if dfn is None else dfn.reshape(M)
        if df is not None and not df.is_contiguous():
            df = df.contiguous()
        dz = torch.empty((M, 96), dtype=torch.float32, device=z2.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(z2.device).cuda_stream)
        _check(L, L.sgrl_gram_tri_backward(_p(z2), _p(dg), _p(df), _p(fn), _p(dz), M, st), "sgrl_gram_tri_backward")
        return dz.view(ctx.z_shape)


def _sym_fold_call(full, tri, unfold, device):
    n = len(full)
    pw = (ctypes.c_void_p * n)(*[t.data_ptr() for t in full])
    pt = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tri])
    rows = (ctypes.c_int * n)(*[t.shape[0] for t in full])
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _check(_L(), _L().sgrl_sym_fold(n, pw, pt, rows, 1 if unfold else 0, st), "sgrl_sym_fold")


class _FoldFn(torch.autograd.Function):
    """Weights [rows, 1024] acting on vec(G), G symmetric -> [rows, 528] acting on tri(G) (mirror columns added), up to 16 matrices in
    one launch; the backward spreads a folded gradient onto both mirror columns, one launch too."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.set_materialize_grads(False)       # a folded weight whose gradient was postponed (deferred_wgrads) gets None here, not zeros to unfold
        ws = [w if w.is_contiguous() else w.contiguous() for w in ws]
        out = [torch.empty((w.shape[0], TRI), dtype=torch.float32, device=w.device) for w in ws]
        _sym_fold_call(ws, out, False, ws[0].device)
        ctx.shapes = [w.shape for w in ws]
        return tuple(out)

    @staticmethod
    def backward(ctx, *ds):
        idx = [i for i, d in enumerate(ds) if d is not None and ctx.needs_input_grad[i]]
        res = [None] * len(ds)
        if idx:
            src = [ds[i] if ds[i].is_contiguous() else ds[i].contiguous() for i in idx]
            full = [torch.empty(tuple(ctx.shapes[i]), dtype=torch.float32, device=src[0].device) for i in idx]
            _sym_fold_call(full, src, True, src[0].device)
            for i, f in zip(idx, full):
                res[i] = f
        return tuple(res)


class _ZmatFn(torch.autograd.Function):
    """z [..., 3, 32], mat [..., 32, 32] -> z . mat [..., 3, 32] per node."""

    @staticmethod
    def forward(ctx, z, mat):
        L = _L()
        z2, m2 = z.reshape(-1, 96), mat.reshape(-1, 1024)
        z2 = z2 if z2.is_contiguous() else z2.contiguous()
        m2 = m2 if m2.is_contiguous() else m2.contiguous()
        M = z2.shape[0]
        t = torch.empty((M, 96), dtype=torch.float32, device=z.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        _check(L, L.sgrl_zmat_forward(_p(z2), _p(m2), _p(t), M, st), "sgrl_zmat_forward")
        ctx.save_for_backward(z2, m2)
        ctx.shapes = (z.shape, mat.shape)
        return t.view(z.shape)

    @staticmethod
    def backward(ctx, dt):
        L = _L()
        z2, m2 = ctx.saved_tensors
        M = z2.shape[0]
        d = dt.reshape(M, 96)
        d = d if d.is_contiguous() else d.contiguous()
        dz, dm = torch.empty_like(z2), torch.empty_like(m2)
        st = ctypes.c_void_p(torch.cuda.current_stream(z2.device).cuda_stream)
        _check(L, L.sgrl_zmat_backward(_p(z2), _p(m2), _p(d), _p(dz), _p(dm), M, st), "sgrl_zmat_backward")
        return dz.view(ctx.shapes[0]), dm.view(ctx.shapes[1])


class _AttnFn(torch.autograd.Function):
    """qkv [B, L, 768], vgp [B, L, 3, 252], gdir [B, L, 3, 2], bias [2, L, L] or None -> (o [B, L, 256], og [B, L, 3, 256])."""

    @staticmethod
    def forward(ctx, qkv, vgp, gdir, bias, scale):
        L = _L()
        B, Ln = qkv.shape[0], qkv.shape[1]
        qkv, vgp, gdir = (t if t.is_contiguous() else t.contiguous() for t in (qkv, vgp, gdir))
        bz = None if bias is None else (bias if bias.is_contiguous() else bias.contiguous())
        w = torch.empty((B, 2, Ln, Ln), dtype=torch.float32, device=qkv.device)
        o = torch.empty((B, Ln, 256), dtype=torch.float32, device=qkv.device)
        og = torch.empty((B, Ln, 3, 256), dtype=torch.float32, device=qkv.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
        _check(L, L.sgrl_attention_forward(_p(qkv), _p(vgp), _p(gdir), _p(bz), ctypes.c_float(scale), _p(w), _p(o), _p(og), B, Ln, st),
               "sgrl_attention_forward")
        ctx.save_for_backward(qkv, vgp, gdir, w)
        ctx.has_bias, ctx.scale = bias is not None, float(scale)
        return o, og

    @staticmethod
    def backward(ctx, d_o, d_og):
        L = _L()
        qkv, vgp, gdir, w = ctx.saved_tensors
        B, Ln = qkv.shape[0], qkv.shape[1]
        d_o = torch.zeros((B, Ln, 256), dtype=torch.float32, device=qkv.device) if d_o is None else (d_o if d_o.is_contiguous() else d_o.contiguous())
        d_og = torch.zeros((B, Ln, 3, 256), dtype=torch.float32, device=qkv.device) if d_og is None else (d_og if d_og.is_contiguous() else d_og.contiguous())
        dqkv, dvgp, ds = torch.empty_like(qkv), torch.empty_like(vgp), torch.empty_like(w)
        dgdh = torch.empty((B, Ln, 3, 2, 2), dtype=torch.float32, device=qkv.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
        _check(L, L.sgrl_attention_backward(_p(qkv), _p(vgp), _p(gdir), ctypes.c_float(ctx.scale), _p(w), _p(d_o), _p(d_og), _p(dqkv),
                                            _p(dvgp), _p(dgdh), _p(ds), B, Ln, st), "sgrl_attention_backward")
        dgdir = dgdh.sum(3) if ctx.needs_input_grad[2] else None
        dbias = ds.sum(0) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        return dqkv, dvgp, dgdir, dbias, None


class _AddLNFn(torch.autograd.Function):
    """y = LayerNorm(x + res) * w + b over 128 columns, one launch forward and one backward (csrc/train_gemm.hip k_add_ln_*);
    w1 / b1 given: x stacks TWO networks along dim 0, each with its own affine pair."""

    @staticmethod
    def forward(ctx, x, res, w0, b0, w1, b1, eps):
        L = _L()
        nets = 2 if w1 is not None else 1
        x2 = x.reshape(-1, 128)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        r2 = None
        if res is not None:
            r2 = res.expand_as(x).reshape(-1, 128)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        total = x2.shape[0]
        assert total % nets == 0
        y = torch.empty_like(x2)
        xhat = torch.empty_like(x2)
        rstd = torch.empty((total,), dtype=torch.float32, device=x.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _check(L, L.sgrl_add_ln_forward(_p(x2), _p(r2), _p(w0), _p(b0), _p(w1), _p(b1), _p(y), _p(xhat), _p(rstd), total // nets, nets,
                                        ctypes.c_float(eps), st), "sgrl_add_ln_forward")
        ctx.save_for_backward(xhat, rstd, w0, w1)
        ctx.nets, ctx.x_shape, ctx.res_shape = nets, x.shape, None if res is None else res.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        xhat, rstd, w0, w1 = ctx.saved_tensors
        total, nets = xhat.shape[0], ctx.nets
        dy2 = dy.reshape(total, 128)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        need = ctx.needs_input_grad
        dx = torch.empty_like(xhat)
        new = lambda want: torch.empty((128,), dtype=torch.float32, device=dy.device) if want else None
        dw0, db0 = new(need[2]), new(need[3])
        dw1, db1 = new(nets == 2 and need[4]), new(nets == 2 and need[5])
        st = ctypes.c_void_p(torch.cuda.current_stream(dy.device).cuda_stream)
        _check(L, L.sgrl_add_ln_backward(_p(dy2), _p(xhat), _p(rstd), _p(w0), _p(w1), _p(dx), _p(dw0), _p(db0), _p(dw1), _p(db1),
                                         total // nets, nets, st), "sgrl_add_ln_backward")
        dxv = dx.view(ctx.x_shape)
        dres = None
        if ctx.res_shape is not None and need[1]:
            dres = dxv if tuple(ctx.res_shape) == tuple(ctx.x_shape) else dxv.sum_to_size(ctx.res_shape)
        return (dxv if need[0] else None), dres, dw0, db0, dw1, db1, None


class _Embed3Fn(torch.autograd.Function):
    """cat([w0[idx[0]], w1[idx[1]], w2[idx[2]]], dim=1) -- the three traversal embeddings -- one launch forward, one backward."""

    @staticmethod
    def forward(ctx, idx, w0, w1, w2):
        L = _L()
        n = (w0.shape[1], w1.shape[1], w2.shape[1])
        ws = [w if w.is_contiguous() else w.contiguous() for w in (w0, w1, w2)]
        Ln = idx.shape[1]
        out = torch.empty((Ln, sum(n)), dtype=torch.float32, device=w0.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(w0.device).cuda_stream)
        _check(L, L.sgrl_embed3_forward(_p(idx), _p(ws[0]), _p(ws[1]), _p(ws[2]), n[0], n[1], n[2], _p(out), Ln, st), "sgrl_embed3_forward")
        ctx.save_for_backward(idx)
        ctx.n, ctx.rows = n, w0.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _L()
        (idx,) = ctx.saved_tensors
        d = dout if dout.is_contiguous() else dout.contiguous()
        n, rows = ctx.n, ctx.rows
        dws = [torch.empty((rows, n[t]), dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[1 + t] else None for t in range(3)]
        st = ctypes.c_void_p(torch.cuda.current_stream(dout.device).cuda_stream)
        _check(L, L.sgrl_embed3_backward(_p(idx), _p(d), _p(dws[0]), _p(dws[1]), _p(dws[2]), n[0], n[1], n[2], idx.shape[1], rows, st),
               "sgrl_embed3_backward")
        return None, dws[0], dws[1], dws[2]


_NO_GRAD_KERNELS = False      # no_grad_kernels(): the own kernels also where nothing is differentiated


@contextlib.contextmanager
def no_grad_kernels(enabled=True):
    """Inside this context (meant for torch.no_grad() passes) the operations of this module launch their own kernels although nothing
    requires a gradient -- the twin target critics of a TD3 update walk both networks in one pass that way (set_policy.SECritic)."""
    global _NO_GRAD_KERNELS
    old, _NO_GRAD_KERNELS = _NO_GRAD_KERNELS, bool(enabled) or _NO_GRAD_KERNELS
    try:
        yield
    finally:
        _NO_GRAD_KERNELS = old


def _on_device_with_grad(*ts):
    if not (ENABLED and ts[0].is_cuda and ts[0].dtype == torch.float32):
        return False
    if _NO_GRAD_KERNELS and not torch.is_grad_enabled():
        return True
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


class _Slot(object):
    """What the consumers of one fan_out() share during a backward pass: the buffer their input gradients are summed in."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class _FanOutFn(torch.autograd.Function):
    """x -> n aliases of x, one per consumer.  Consumers that are this module's linear layers are handed the slot and sum their
    input gradients IN the slot's buffer, product by product (the input-gradient kernel's accumulate epilogue); whatever the other
    consumers return is added here.  Without it autograd launches one element-wise addition per extra consumer: eight per SET layer
    and pass (the vector stream feeds two projections and the residual, the invariant features two feed-forward pairs, ...)."""

    @staticmethod
    def forward(ctx, x, slot, n):
        ctx.slot, ctx.x_shape = slot, x.shape
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        slot = ctx.slot
        buf, slot.buf = slot.buf, None
        total, owned = buf, buf is not None          # the slot's buffer is ours to add onto; a consumer's own gradient is not
        for g in gs:
            if g is None or (buf is not None and g.data_ptr() == buf.data_ptr() and g.numel() == buf.numel()):
                continue                    # nothing, or the slot's own buffer as returned by its first consumer
            g = g.reshape(ctx.x_shape)
            if total is None:
                total = g
            elif owned:
                total = total.view(ctx.x_shape).add_(g)
            else:
                total, owned = total + g, True
        return (None if total is None else total.view(ctx.x_shape)), None, None


_FAN_OUT = True      # False (tests, A/B): every consumer's input gradient through autograd's own additions


def fan_out(x, n):
    """(slot, [n aliases of x]) where the own kernels differentiate x on the GPU, (None, [x] * n) otherwise.  Pass the slot to the
    linear layers among x's consumers (linear / linear2 `slot=`), each with its OWN alias; any other consumer takes an alias too."""
    if n > 1 and _FAN_OUT and _on_device_with_grad(x) and x.requires_grad:
        slot = _Slot()
        return slot, list(_FanOutFn.apply(x, slot, n))
    return None, [x] * n


def linear(x, weight, bias=None, relu=False, rowdiv=None, addend=None, tail=None, x_relu=False, premasked=False, slot=None):
    """act(x @ weight.T + bias) / rowdiv, differentiable in x, weight, bias and rowdiv (act = ReLU if relu; rowdiv: one value
    per row, broadcast over the output features -- the `/ F_norm` of the reference's SET layers).  Two followers can ride on
    the product's launch: `addend` (same shape as the result) is added to it -- a residual --, `tail` [..., t] is appended to it
    along the last dimension (torch.cat([result, tail], -1)).  x_relu / premasked: a ReLU layer feeding ONLY this one -- the pair
    linear(linear(x, w1, relu=True, premasked=True), w2, x_relu=True) applies the ReLU's mask once, in the epilogue of the second
    layer's input gradient (_LinearFn); they change nothing in the forward and nothing outside the HIP path."""
    if _on_device_with_grad(x, weight, bias, rowdiv, addend, tail):
        if addend is None or (not relu and rowdiv is None and tail is None):
            return _LinearFn.apply(x, weight, bias, bool(relu), rowdiv, addend, tail, bool(x_relu), bool(premasked), slot)
        y = _LinearFn.apply(x, weight, bias, bool(relu), rowdiv, None, None, bool(x_relu), bool(premasked), slot)     # a follower the kernel does not take: own launches
    else:
        y = F.linear(x, weight, bias)
        y = F.relu(y) if relu else y
        y = y if rowdiv is None else y / rowdiv
    y = y if addend is None else addend + y
    return y if tail is None else torch.cat([y, tail], dim=-1)


class _Adjacent3Fn(torch.autograd.Function):
    """Three parameters that lie back to back in one allocation, seen as the one stacked tensor they already are: no launch in
    the forward, views of the incoming gradient in the backward (what `torch.cat` + its backward produce with two copies)."""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.n = a.shape[0]
        return a.detach().as_strided((3 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())

    @staticmethod
    def backward(ctx, d):
        n = ctx.n
        return d[:n], d[n:2 * n], d[2 * n:]


def adjacent3(a, b, c):
    """True if b starts where a ends and c where b ends (same shape, dtype, contiguous): SubequivariantAttention lays its q / k / v
    projections out that way (set_policy.py `_adjoin`)."""
    if not (a.shape == b.shape == c.shape and a.dtype == b.dtype == c.dtype and a.is_contiguous() and b.is_contiguous()
            and c.is_contiguous() and a.device == b.device == c.device):
        return False
    step = a.numel() * a.element_size()
    return b.data_ptr() == a.data_ptr() + step and c.data_ptr() == b.data_ptr() + step and \
        a.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()


def stacked3(a, b, c):
    """torch.cat([a, b, c], 0) -- without the copy when the three already lie back to back."""
    if adjacent3(a, b, c):
        return _Adjacent3Fn.apply(a, b, c)
    return torch.cat([a, b, c], dim=0)


def linear2(x, w0, w1, b0=None, b1=None, relu=False, rowdiv=None, shared=False, addend=None, tail=None, x_relu=False, premasked=False,
            slot=None):
    """`linear` for the same layer of two networks at once: returns [2, ..., N]; x = the input both share ([..., K], shared=True) or
    their inputs stacked ([2, ..., K]); rowdiv stacked [2, ..., 1]; addend stacked [2, ..., N]; tail [2, ..., t] (or one both share,
    [..., t] / expanded)."""
    if _on_device_with_grad(x, w0, w1, b0, b1, rowdiv, addend, tail) and (addend is None or (not relu and rowdiv is None and tail is None)):
        if tail is not None and tail.dim() == x.dim() + (1 if shared else 0) and tail.stride(0) == 0:
            tail = tail[0]                          # an expanded pair: the one tensor both networks share
        return _Linear2Fn.apply(x, w0, w1, b0, b1, bool(relu), rowdiv, bool(shared), addend, tail, bool(x_relu), bool(premasked), slot)
    xs = (x, x) if shared else (x[0], x[1])
    ts = (None, None) if tail is None else ((tail[0], tail[1]) if tail.dim() == xs[0].dim() + 1 else (tail, tail))
    return torch.stack([linear(xs[0], w0, b0, relu, None if rowdiv is None else rowdiv[0], None if addend is None else addend[0], ts[0]),
                        linear(xs[1], w1, b1, relu, None if rowdiv is None else rowdiv[1], None if addend is None else addend[1], ts[1])])


def gram_fn(z):
    """z [..., 3, 32] -> (vec(Z'Z) [..., 1024], ||Z'Z||_F + 1 [..., 1])  (reference SEActor.py:94-98)."""
    if _on_device_with_grad(z):
        return _GramFn.apply(z)
    gram = torch.einsum("...sa,...sc->...ac", z, z).flatten(-2)
    return gram, gram.norm(dim=-1, keepdim=True) + 1.0


TRI_GRAM = True      # False (tests): the invariant layers on all 1 024 entries of Z'Z (rounds 2-4)


def tri_weights(weights, like):
    """The invariant layers' weights [rows, 1024] folded onto the lower triangle, [rows, 528] each (one launch for up to 16), or None
    where the own kernels do not run for a pass over `like` (the pass's input): then `gram_fn` and the unfolded weights are used
    (gram_tri_fn pairs with the folded ones)."""
    if not (TRI_GRAM and _on_device_with_grad(like, *weights) and len(weights) <= 16 and all(w.dim() == 2 and w.shape[1] == 1024 for w in weights)):
        return None
    out = _FoldFn.apply(*weights)
    for o, w in zip(out, weights):
        if w.is_leaf and w.requires_grad:
            o._sgrl_fold_leaf = w          # deferred_wgrads may postpone the folded weight's gradient and unfold it into w.grad (flush_wgrads)
    return out


def gram_tri_fn(z):
    """z [..., 3, 32] -> (tri(Z'Z) [..., 528], ||Z'Z||_F + 1 [..., 1]); only on the own kernels (callers hold folded weights)."""
    return _GramTriFn.apply(z)


def set_attention(qkv, vgp, gdir, bias, scale):
    """Limb attention of the SET layers (reference subequivariant_attentions.py:90-151 between the projections), 2 heads x 128
    channels: qkv [B, L, 768] = q | k | v (q is multiplied by `scale` here), vector values given in parts -- vgp [B, L, 3, 252]
    (126 projected channels per head) and gdir [B, L, 3, 2] (channels 126, 127 of BOTH heads) --, bias [2, L, L] or None ->
    (o [B, L, 256], og [B, L, 3, 256]) with w = softmax_j(scale q_i . k_j + bias) per head."""
    B, Ln = qkv.shape[:2]
    if _on_device_with_grad(qkv, vgp, gdir, bias) and qkv.shape[-1] == 768 and vgp.shape[-1] == 252 and Ln <= 14:
        return _AttnFn.apply(qkv, vgp, gdir, bias, float(scale))
    qh = (qkv[..., :256] * scale).view(B, Ln, 2, 128)
    kh, vh = qkv[..., 256:512].view(B, Ln, 2, 128), qkv[..., 512:].view(B, Ln, 2, 128)
    vg = torch.cat([vgp.view(B, Ln, 3, 2, 126), gdir.unsqueeze(3).expand(B, Ln, 3, 2, 2)], dim=-1)
    s = torch.einsum("bihd,bjhd->bhij", qh, kh)
    if bias is not None:
        s = s + bias.unsqueeze(0)
    w = F.softmax(s, dim=-1)
    o = torch.einsum("bhij,bjhd->bihd", w, vh).reshape(B, Ln, 256)
    og = torch.einsum("bhij,bjshd->bishd", w, vg).reshape(B, Ln, 3, 256)
    return o, og


def zmat(z, mat):
    """Per node: z [..., 3, 32] . mat [..., 32, 32] -> [..., 3, 32]  (reference SEActor.py:108-110: torch.bmm(g_src3, mat3))."""
    if _on_device_with_grad(z, mat) and z.shape[-2:] == (3, 32) and mat.shape[-2:] == (32, 32):
        return _ZmatFn.apply(z, mat)
    return torch.einsum("...sa,...ac->...sc", z, mat)


def add_layer_norm(x, res, norm):
    """norm(x + res) for an nn.LayerNorm over the last dimension (res may be None): one launch forward, one backward when autograd is
    recording on the GPU and the width is 128; the module itself otherwise."""
    if x.shape[-1] == 128 and norm.elementwise_affine and norm.bias is not None and _on_device_with_grad(x, res, norm.weight, norm.bias):
        return _AddLNFn.apply(x, res, norm.weight, norm.bias, None, None, float(norm.eps))
    return norm(x if res is None else x + res)


def add_layer_norm2(x, res, n0, n1):
    """The same for the stacked activations of two networks: x [2, ..., 128] -> stack(n0(x[0] + res[0]), n1(x[1] + res[1]))."""
    if x.shape[-1] == 128 and x.shape[0] == 2 and n0.elementwise_affine and n1.elementwise_affine and n0.bias is not None and \
            n1.bias is not None and n0.eps == n1.eps and _on_device_with_grad(x, res, n0.weight, n0.bias, n1.weight, n1.bias):
        return _AddLNFn.apply(x, res, n0.weight, n0.bias, n1.weight, n1.bias, float(n0.eps))
    s = x if res is None else x + res
    a, b = s.unbind(0)          # (not s[0], s[1]: each index costs a zero-filled gradient, a slice copy and an add going back)
    return torch.stack([n0(a), n1(b)])


_idx3_cache = {}


def embed3(embeddings, positional_indices):
    """torch.cat([emb(idx) for emb, idx in zip(embeddings, positional_indices)], dim=1) for three nn.Embedding tables of equal row
    count (the traversal embeddings of the SET models): one launch forward and one backward when autograd records on the GPU."""
    ws = [e.weight for e in embeddings]
    if len(ws) == 3 and _on_device_with_grad(*ws) and all(w.shape[0] == ws[0].shape[0] for w in ws) and sum(w.shape[1] for w in ws) <= 128 \
            and all(e.padding_idx is None and e.max_norm is None for e in embeddings) and all(i.dtype == torch.int64 and i.dim() == 1 for i in positional_indices):
        key = tuple((i.data_ptr(), i.shape[0], i._version) for i in positional_indices)
        ent = _idx3_cache.get(key)
        if ent is None:
            if len(_idx3_cache) > 256:
                _idx3_cache.clear()
            # the index tensors are kept alive with the entry, so their addresses cannot be reused by other tensors
            ent = _idx3_cache[key] = (torch.stack(list(positional_indices)).contiguous(), list(positional_indices))
        return _Embed3Fn.apply(ent[0], ws[0], ws[1], ws[2])
    return torch.cat([emb(idx) for emb, idx in zip(embeddings, positional_indices)], dim=1)
