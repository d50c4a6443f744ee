"""The live weight-packing plan (sgrl_amd/set_hip.plan_segments, executed on the device by k_pack in set_actor.hip)
must reproduce the flat buffer `pack_tensors` builds on the host: same offsets, same values, no gaps.  The kernel's
arithmetic per segment kind is restated here in NumPy (it is pure index work plus one multiply)."""
import numpy as np
import pytest
import torch

from sgrl_amd import set_hip
from sgrl_amd.set_policy import make_critic, make_policy


def _run_plan(segs, srcs, total):
    out = np.full(total, np.nan, dtype=np.float32)
    for sg, (t0, t1, off0, off1) in zip(segs, srcs):
        n, kind, a, b = int(sg["n"]), int(sg["kind"]), int(sg["a"]), int(sg["b"])
        lda, ldb = int(sg["lda"]), int(sg["ldb"])
        s0 = t0.detach().numpy().reshape(-1)[off0:]
        s1 = None if t1 is None else t1.detach().numpy().reshape(-1)[off1:]
        v = np.zeros(n, dtype=np.float32)
        if kind == set_hip.PACK_COPY:
            v[:a] = s0[:a] * np.float32(sg["scale"])
        elif kind == set_hip.PACK_PADCOL:
            v = v.reshape(-1, b)
            v[:, :a] = s0.reshape(-1, a)
        elif kind == set_hip.PACK_FOLD:
            # blocked lower triangle (csrc/gemm_f32.h GRAM): written out independently of set_hip.gram_order()
            w3 = s0.reshape(-1, 32, 32)
            v = v.reshape(-1, 576)
            k = 0
            for A in range(8):
                for B in range(A + 1):
                    for i in range(4):
                        for j in range(4):
                            aa, bb = 4 * A + i, 4 * B + j
                            if aa > bb:
                                v[:, k] = w3[:, aa, bb] + w3[:, bb, aa]
                            elif aa == bb:
                                v[:, k] = w3[:, aa, aa]
                            k += 1
            assert k == 576
        elif kind == set_hip.PACK_STACK:
            v = v.reshape(64, b)
            v[:30, :a] = s0.reshape(30, a)
            if t1 is not None:
                v[32:62, :a] = s1.reshape(30, a)
        elif kind == set_hip.PACK_MATMUL:      # dst [rows, b] = A [rows, a] (stride lda) . B [a, b] (stride ldb)
            rows = n // b
            A = np.stack([s0[r * lda:r * lda + a] for r in range(rows)])
            Bm = np.stack([s1[k * ldb:k * ldb + b] for k in range(a)])
            v = (A.astype(np.float64) @ Bm.astype(np.float64) * float(sg["scale"])).astype(np.float32)
        elif kind == set_hip.PACK_PERM32:
            k = n // 1024
            v = s0[:n].reshape(32, 32, k).transpose(1, 0, 2).reshape(-1)
        elif kind == set_hip.PACK_SUBMAT:
            rows = n // b
            v = np.stack([s0[r * lda:r * lda + b] for r in range(rows)])
        else:
            raise AssertionError(kind)
        d = int(sg["dst"])
        assert np.isnan(out[d:d + n]).all(), "segments overlap"
        out[d:d + n] = v.reshape(-1)
    assert not np.isnan(out).any(), "segments leave gaps"
    return out


@pytest.mark.parametrize("critic", [False, True])
def test_plan_reproduces_the_host_pack(critic):
    torch.manual_seed(3)
    if critic:
        net = make_critic(use_hip=False).critic2
    else:
        net = make_policy(use_hip=False).actor
    segs, offs, total, srcs = set_hip.plan_segments(net, critic=critic)
    assert segs.dtype.itemsize == 56 and total % 64 == 0
    flat = _run_plan(segs, srcs, total)
    sd = {"actor." + k: v for k, v in net.state_dict().items()}
    tens = set_hip.pack_tensors(sd, critic=critic)
    pos = 0
    for i, t in enumerate(tens):
        assert offs[i] == pos, i
        ref = t.reshape(-1).numpy()
        if np.array_equal(flat[pos:pos + ref.size], ref):
            pass
        else:       # folded slots (matrix products): float32 torch.matmul vs the float64 restatement above
            np.testing.assert_allclose(flat[pos:pos + ref.size], ref, rtol=1e-5, atol=1e-6, err_msg="slot %d" % i)
        pad = (-ref.size) % 64
        assert (flat[pos + ref.size:pos + ref.size + pad] == 0).all()
        pos += ref.size + pad
    # the seven stacked projection operands follow the slot table
    for k in range(set_hip.NSITES):
        cpad = 144 if k == 6 else 128
        blk = flat[offs[set_hip.NW + k]:offs[set_hip.NW + k] + 64 * cpad].reshape(64, cpad)
        assert (blk[30:32] == 0).all() and (blk[62:] == 0).all()
        if k == 6:
            np.testing.assert_array_equal(blk[:30, :136], net.gg_proj.weight.detach().numpy())
            assert (blk[:, 136:] == 0).all()
            if critic:
                assert (blk[32:] == 0).all()
            else:
                np.testing.assert_array_equal(blk[32:62, :136], net.g_proj.weight.detach().numpy())
        else:
            layer = net.transformer_encoder.layers[k // 2]
            if k % 2 == 0:
                np.testing.assert_array_equal(blk[:30], layer.self_attn.g_proj.weight.detach().numpy())
                assert (blk[32:] == 0).all()
            else:
                np.testing.assert_array_equal(blk[:30], layer.g_proj2.weight.detach().numpy())
                np.testing.assert_array_equal(blk[32:62], layer.g_proj3.weight.detach().numpy())
    # the folded head (decoder_g through linear2_m): [32, 256] and [32] behind the sites; a zero filler for the critic
    o_w, o_b = offs[set_hip.NW + set_hip.NSITES], offs[set_hip.NW + set_hip.NSITES + 1]
    if critic:
        assert (flat[o_w:o_w + 64] == 0).all() and (flat[o_b:o_b + 64] == 0).all()
    else:
        wd = net.decoder_g.weight.detach().double().numpy().reshape(32)
        w2 = net.linear2_m.weight.detach().double().numpy().reshape(32, 32, 256)       # [q][c][k]
        b2 = net.linear2_m.bias.detach().double().numpy().reshape(32, 32)
        np.testing.assert_allclose(flat[o_w:o_w + 32 * 256].reshape(32, 256), np.einsum("c,qck->qk", wd, w2), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(flat[o_b:o_b + 32], b2 @ wd, rtol=1e-5, atol=1e-7)


def test_plan_sources_are_the_live_parameter_storage():
    net = make_policy(use_hip=False).actor
    segs, offs, total, srcs = set_hip.plan_segments(net)
    ranges = [(p.data_ptr(), p.data_ptr() + 4 * p.numel()) for p in net.parameters()]
    for col in ("src0", "src1"):
        for ptr in segs[col]:
            assert int(ptr) == 0 or any(lo <= int(ptr) < hi for lo, hi in ranges)      # inside some parameter's own storage
    # a reference-style soft update writes through .data: storage (and therefore the plan) is unchanged
    before = [p.data_ptr() for p in net.parameters()]
    for p in net.parameters():
        p.data.copy_(0.5 * p.data)
    assert before == [p.data_ptr() for p in net.parameters()]
