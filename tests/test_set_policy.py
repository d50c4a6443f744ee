"""The nn.Module surface: identical state_dict keys/shapes, forward parity with the reference fixtures
through the differentiable PyTorch path, gradient flow, checkpoint round trip."""
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle.formula import apply_formula_
from sgrl_amd import graph as G
from sgrl_amd.set_policy import make_policy

TRAV = ["pre", "inlcrs", "postlcrs"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "set_state_dict_keys.json")) as f:
        keys = json.load(f)
    with open(os.path.join(golden_dir, "graphs.json")) as f:
        graphs = json.load(f)
    return keys, graphs, np.load(os.path.join(golden_dir, "set_forward.npz"))


def test_state_dict_keys_and_shapes_match_the_reference(gold):
    keys, _, _ = gold
    pol = make_policy(use_hip=False)
    sd = pol.state_dict()
    assert sorted(sd.keys()) == sorted(keys.keys())
    for k, shp in keys.items():
        assert list(sd[k].shape) == shp, k
    assert sum(p.numel() for p in pol.parameters()) == 4712712


def test_forward_matches_reference_fixtures(gold):
    keys, graphs, z = gold
    pol = make_policy(use_hip=False).eval()
    apply_formula_(pol)
    pol64 = make_policy(use_hip=False).double().eval()
    apply_formula_(pol64)
    for name, g in graphs.items():
        gd = G.getGraphDict(g["parents"], TRAV, [], device=torch.device("cpu"))
        pol.change_morphology(gd)
        gd64 = dict(gd)
        gd64["relation"] = gd["relation"].double()
        pol64.change_morphology(gd64)
        for B in (1, 5):
            obs = torch.from_numpy(z["%s/B%d/obs" % (name, B)])
            with torch.no_grad():
                a32 = pol(obs).numpy()
                a64 = pol64(obs.double()).numpy()
            assert a32.shape == (B, 3 * len(g["parents"]))
            assert np.abs(a32 - z["%s/B%d/act_f32" % (name, B)]).max() < 1e-5, name
            # relation is float32 in both (getGraphDict), so the f64 run agrees to double rounding
            assert np.abs(a64 - z["%s/B%d/act_f64" % (name, B)]).max() < 1e-7, name


def test_initialisation_conventions():
    torch.manual_seed(0)
    pol = make_policy(use_hip=False)
    a = pol.actor
    assert float(a.encoder.weight.abs().max()) <= 0.1 and float(a.g_encoder.weight.abs().max()) <= 0.1
    l0, l1 = a.transformer_encoder.layers[0], a.transformer_encoder.layers[2]
    for (n0, p0), (n1, p1) in zip(l0.named_parameters(), l1.named_parameters()):
        assert torch.equal(p0, p1), n0   # deepcopy clones start identical (reference SEActor.py:14-15,131)
        assert p0.data_ptr() != p1.data_ptr()


def test_differentiable_and_checkpoint_round_trip(gold):
    keys, graphs, z = gold
    pol = make_policy(use_hip=False)
    apply_formula_(pol)
    g = graphs["3d_walker_7_full"]
    pol.change_morphology(G.getGraphDict(g["parents"], TRAV, [], device=torch.device("cpu")))
    obs = torch.from_numpy(z["3d_walker_7_full/B5/obs"])
    out = pol(obs)
    out.pow(2).sum().backward()
    used = [n for n, p in pol.named_parameters() if p.grad is not None and p.grad.abs().sum() > 0]
    unused = [n for n, p in pol.named_parameters() if p.grad is None]
    assert len(used) > 100
    assert all(("in_proj" in n or "out_proj" in n) for n in unused), unused   # present-but-unused tensors
    buf = io.BytesIO()
    torch.save({"agent": pol.state_dict()}, buf)
    buf.seek(0)
    pol2 = make_policy(use_hip=False)
    pol2.load_state_dict(torch.load(buf)["agent"])
    pol2.change_morphology(pol.graph)
    with torch.no_grad():
        assert torch.equal(pol2(obs), pol(obs))


def test_linear_followers_and_fused_norm_fall_back_to_the_plain_operations_on_the_cpu():
    """train_ops.linear(tail= / addend=), linear2 and add_layer_norm(2) off the GPU are exactly cat / add / the LayerNorm module
    (the HIP forms are held against float64 by tests/test_train_ops_gpu.py)."""
    import torch
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(5)
    x, gd = torch.randn(4, 5, 3, 16, generator=g), torch.randn(4, 5, 3, 2, generator=g)
    w0, w1 = torch.randn(30, 16, generator=g), torch.randn(30, 16, generator=g)
    z = train_ops.linear(x, w0, tail=gd)
    assert z.shape == (4, 5, 3, 32) and torch.equal(z, torch.cat([x @ w0.T, gd], -1))
    r = torch.randn(4, 5, 3, 30, generator=g)
    assert torch.equal(train_ops.linear(x, w0, addend=r), r + x @ w0.T)
    z2 = train_ops.linear2(x, w0, w1, shared=True, tail=gd.unsqueeze(0).expand(2, *gd.shape))
    assert torch.equal(z2[1], torch.cat([x @ w1.T, gd], -1)) and z2.shape == (2, 4, 5, 3, 32)
    r2 = torch.randn(2, 4, 5, 3, 30, generator=g)
    assert torch.equal(train_ops.linear2(x, w0, w1, shared=True, addend=r2)[0], r2[0] + x @ w0.T)
    n0, n1 = torch.nn.LayerNorm(128), torch.nn.LayerNorm(128)
    with torch.no_grad():
        n1.weight.mul_(0.5); n1.bias.add_(0.25)
    a, b = torch.randn(2, 6, 128, generator=g), torch.randn(2, 6, 128, generator=g)
    assert torch.equal(train_ops.add_layer_norm(a[0], b[0], n0), n0(a[0] + b[0]))
    assert torch.equal(train_ops.add_layer_norm(a[0], None, n0), n0(a[0]))
    y = train_ops.add_layer_norm2(a, b, n0, n1)
    assert torch.equal(y[0], n0(a[0] + b[0])) and torch.equal(y[1], n1(a[1] + b[1]))


def test_qkv_projections_lie_back_to_back_and_stack_without_a_copy():
    """The training path multiplies by [Wq; Wk; Wv] (one product, SEActor.py:34-46 shares the input): the three parameters are
    views of one allocation, the stack is that allocation, gradients come back as its thirds; construction, deepcopy, dtype / device
    moves and load_state_dict keep the layout, separated tensors fall back to a concatenation with the same values."""
    import copy
    from sgrl_amd import train_ops
    pol = make_policy(device="cpu")
    attn = [m for m in pol.modules() if hasattr(m, "qkv_stacked")]
    assert len(attn) == 3
    for net in (pol, copy.deepcopy(pol), copy.deepcopy(pol).double().float()):
        for m in (x for x in net.modules() if hasattr(x, "qkv_stacked")):
            assert train_ops.adjacent3(m.q_proj.weight, m.k_proj.weight, m.v_proj.weight)
            assert train_ops.adjacent3(m.q_proj.bias, m.k_proj.bias, m.v_proj.bias)
    m = attn[0]
    w, b = m.qkv_stacked()
    assert w.data_ptr() == m.q_proj.weight.data_ptr() and w.shape == (768, 256) and b.shape == (768,)
    assert torch.equal(w, torch.cat([m.q_proj.weight, m.k_proj.weight, m.v_proj.weight]))
    coef = torch.arange(768, dtype=torch.float32)[:, None]
    ((w * coef).sum() + (b * coef[:, 0]).sum()).backward()
    for i, p in enumerate((m.q_proj, m.k_proj, m.v_proj)):
        assert torch.equal(p.weight.grad, coef[256 * i:256 * (i + 1)].expand(256, 256))
        assert torch.equal(p.bias.grad, coef[256 * i:256 * (i + 1), 0])
    other = copy.deepcopy(pol)
    sd = {k: v + 1.0 for k, v in pol.state_dict().items()}
    other.load_state_dict(sd)
    m2 = [x for x in other.modules() if hasattr(x, "qkv_stacked")][0]
    assert train_ops.adjacent3(m2.q_proj.weight, m2.k_proj.weight, m2.v_proj.weight)
    assert torch.equal(m2.qkv_stacked()[0], w.detach() + 1.0)
    # separate tensors: the plain concatenation
    a, b_, c = (torch.randn(4, 5, requires_grad=True) for _ in range(3))
    assert not train_ops.adjacent3(a, b_, c)
    assert torch.equal(train_ops.stacked3(a, b_, c), torch.cat([a, b_, c]))
