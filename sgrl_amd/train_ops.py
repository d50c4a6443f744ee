"""Differentiable linear layers of the TD3 update on the HIP kernels of csrc/train_gemm.hip (C ABI: include/sgrl_train.h).

`linear(x, weight, bias, relu)` is `torch.nn.functional.linear` (+ ReLU) as a `torch.autograd.Function` whose forward,
input gradient and weight / bias gradient run on this library's own small-product kernels instead of the vendor GEMMs
(which pick single-workgroup 256 x 256 tilings for the update's 700-row problems: DESIGN.md section 5).  Used by the SET
modules (set_policy.py) whenever autograd is recording on the GPU -- i.e. inside `Agent.update` (reference agent.py:117-183);
the no-grad rollout path is the fused forward of csrc/set_actor.hip.  No CPU fallback: on the CPU the modules use
`F.linear`; on the GPU a missing extension raises.
"""
import ctypes
import os

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
        L.sgrl_linear_forward.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp]
        L.sgrl_linear_backward.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, vp, vp]
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


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        L = _L()
        N, K = weight.shape
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) < K:
            x2 = x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        M = x2.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _check(L, L.sgrl_linear_forward(_p(x2), x2.stride(0), _p(w), K, _p(bias), _p(y), N, M, N, K, 1 if relu else 0, st), "sgrl_linear_forward")
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        x2, w, yr = ctx.saved_tensors
        N, K = w.shape
        M = x2.shape[0]
        dy2 = dy.reshape(M, N)
        if dy2.stride(1) != 1 or dy2.stride(0) < N:
            dy2 = dy2.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty((M, K), dtype=torch.float32, device=dy.device) if need_x else None
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device) if need_w else None
        db = torch.empty((N,), dtype=torch.float32, device=dy.device) if need_b else None
        st = ctypes.c_void_p(torch.cuda.current_stream(dy.device).cuda_stream)
        _check(L, L.sgrl_linear_backward(_p(dy2), dy2.stride(0), _p(yr), N, _p(x2), x2.stride(0), _p(w), K, _p(dx), K, _p(dw), K,
                                         _p(db), M, N, K, _p(_scratch(dy.device)), st), "sgrl_linear_backward")
        return (dx.view(ctx.x_shape) if need_x else None), dw, db, None


def linear(x, weight, bias=None, relu=False):
    """relu(x @ weight.T + bias) if relu else x @ weight.T + bias, differentiable."""
    if ENABLED and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and \
            (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _LinearFn.apply(x, weight, bias, bool(relu))
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y
