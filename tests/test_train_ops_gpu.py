"""csrc/train_gemm.hip through sgrl_amd/train_ops.linear: forward, input / weight / bias gradients against float64 PyTorch on
the shapes of a batch-100 TD3 update (ragged sizes, unaligned rows, split contractions, ReLU masks), bit-reproducible."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (M rows, K in, N out, bias, relu)
    (700, 256, 256, True, True), (700, 1024, 256, True, True), (700, 256, 1024, True, False), (2100, 128, 30, False, False),
    (2100, 128, 252, False, False), (2100, 136, 30, False, False), (700, 17, 128, True, False), (700, 145, 128, True, True),
    (1400, 256, 768, True, False), (49, 3, 2, True, False), (4200, 32, 128, False, False), (1, 256, 128, True, True),
    (65, 20, 128, True, False), (300, 256, 1, True, False),
]


@pytest.mark.parametrize("M,K,N,has_bias,relu", SHAPES)
def test_linear_matches_float64(M, K, N, has_bias, relu):
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g) if has_bias else None
    dy = torch.randn(M, N, generator=g)
    # float64 reference on the CPU
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    br = b.double().requires_grad_() if has_bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy.double())
    xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    bd = b.cuda().requires_grad_() if has_bias else None
    y = train_ops.linear(xd, wd, bd, relu)
    assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith("_LinearFn")
    y.backward(dy.cuda())
    scale = float(yr.abs().max()) + 1.0
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 3e-6 * scale
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) < 3e-6 * (float(xr.grad.abs().max()) + 1.0)
    assert float((wd.grad.cpu().double() - wr.grad).abs().max()) < 3e-6 * (float(wr.grad.abs().max()) + 1.0) * np.sqrt(M / 64 + 1)
    if has_bias:
        assert float((bd.grad.cpu().double() - br.grad).abs().max()) < 3e-6 * (float(br.grad.abs().max()) + 1.0) * np.sqrt(M / 64 + 1)
    # reproducible to the bit (split contractions are reduced in a fixed order)
    xd2, wd2 = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    bd2 = b.cuda().requires_grad_() if has_bias else None
    y2 = train_ops.linear(xd2, wd2, bd2, relu)
    y2.backward(dy.cuda())
    assert torch.equal(y, y2) and torch.equal(xd.grad, xd2.grad) and torch.equal(wd.grad, wd2.grad)


def test_strided_inputs_frozen_inputs_and_no_grad_fallback():
    from sgrl_amd import train_ops
    torch.manual_seed(3)
    base = torch.randn(50, 7, 8, 3, device="cuda")
    x = base.transpose(-1, -2)                      # [50, 7, 3, 8], non-contiguous (the g_encoder input of the SET model)
    w = torch.randn(128, 8, device="cuda", requires_grad=True)
    y = train_ops.linear(x, w)                      # x does not require grad: no input gradient is computed
    assert y.shape == (50, 7, 3, 128)
    y.sum().backward()
    ref = torch.nn.functional.linear(x.double().cpu(), w.detach().double().cpu())
    assert float((y.detach().cpu().double() - ref).abs().max()) < 1e-5
    assert float((w.grad.cpu().double() - x.double().cpu().reshape(-1, 8).sum(0)[None].expand(128, 8)).abs().max()) < 1e-3
    with torch.no_grad():
        y0 = train_ops.linear(x, w)
    assert y0.grad_fn is None and float((y0 - y.detach()).abs().max()) < 1e-4


def test_set_policy_gradients_agree_with_the_vendor_gemm_path():
    """The whole differentiable SET actor + critic: gradients through train_ops.linear vs through F.linear."""
    from oracle.formula import synth_obs
    from sgrl_amd import graph as G, mjcf, train_ops
    from sgrl_amd.rollout import TRAV
    from sgrl_amd.td3 import Agent, default_train_args
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    agent = Agent(default_train_args(), device=dev, use_hip=False)
    m = mjcf.load_asset("3d_humanoid_9_full")
    agent.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=dev))
    L = m.num_limbs
    obs = torch.from_numpy(synth_obs(L, 20, 4).astype(np.float32)).to(dev)
    grads = []
    for enabled in (True, False):
        train_ops.ENABLED = enabled
        try:
            agent.actor.zero_grad(); agent.critic.zero_grad()
            q1 = agent.critic.Q1(obs, agent.actor(obs))
            (-q1.mean()).backward()
            grads.append([p.grad.detach().clone() for p in list(agent.actor.parameters()) + list(agent.critic.parameters())
                          if p.grad is not None])
        finally:
            train_ops.ENABLED = True
    assert len(grads[0]) == len(grads[1]) and len(grads[0]) > 100
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= 2e-4 * (float(b.abs().max()) + 1e-6) + 1e-7


@pytest.mark.parametrize("M,K,N", [(700, 256, 768), (700, 256, 1024), (900, 256, 128), (700, 256, 1), (33, 20, 7)])
def test_linear_with_row_divisor_matches_float64(M, K, N):
    """y = (x w^T + b) / fn: forward, and the gradients wrt x, w, b AND fn."""
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / np.sqrt(K), torch.randn(N, generator=g)
    fn = torch.rand(M, 1, generator=g) * 3 + 1
    dy = torch.randn(M, N, generator=g)
    ref = [t.double().requires_grad_() for t in (x, w, b, fn)]
    yr = torch.nn.functional.linear(ref[0], ref[1], ref[2]) / ref[3]
    yr.backward(dy.double())
    dev = [t.cuda().requires_grad_() for t in (x, w, b, fn)]
    y = train_ops.linear(dev[0], dev[1], dev[2], rowdiv=dev[3])
    assert type(y.grad_fn).__name__.startswith("_LinearFn")
    y.backward(dy.cuda())
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 3e-6 * (float(yr.abs().max()) + 1)
    for d, r, name in zip(dev, ref, "x w b fn".split()):
        scale = (float(r.grad.abs().max()) + 1.0) * (np.sqrt(M / 64 + 1) if name in ("w", "b") else 1.0)
        assert d.grad.shape == r.grad.shape
        assert float((d.grad.cpu().double() - r.grad).abs().max()) < 4e-6 * scale, name


def test_gram_fn_matches_float64():
    from sgrl_amd import train_ops
    torch.manual_seed(11)
    z = torch.randn(50, 9, 3, 32) * 0.7
    z[3, 2] = 0.0                                        # a node whose Gram matrix vanishes: no norm term, no NaN
    dgram, dfn = torch.randn(50, 9, 1024), torch.randn(50, 9, 1)
    zr = z.double().requires_grad_()
    gr = torch.einsum("blsa,blsc->blac", zr, zr).flatten(-2)
    eps_safe = gr.detach().norm(dim=-1, keepdim=True) > 0
    fr = gr.norm(dim=-1, keepdim=True) + 1.0
    (gr * dgram.double()).sum().backward(retain_graph=True)
    g1 = zr.grad.clone(); zr.grad = None
    (torch.where(eps_safe, fr, fr.detach()) * dfn.double()).sum().backward()
    ref_grad = g1 + torch.nan_to_num(zr.grad)
    zd = z.cuda().requires_grad_()
    gram, fn = train_ops.gram_fn(zd)
    assert gram.shape == (50, 9, 1024) and fn.shape == (50, 9, 1)
    assert float((gram.detach().cpu().double() - gr.detach()).abs().max()) < 1e-5
    assert float((fn.detach().cpu().double() - fr.detach()).abs().max()) < 1e-5 * float(fr.max())
    ((gram * dgram.cuda()).sum() + (fn * dfn.cuda()).sum()).backward()
    assert torch.isfinite(zd.grad).all()
    assert float((zd.grad.cpu().double() - ref_grad).abs().max()) < 2e-5 * (float(ref_grad.abs().max()) + 1)


@pytest.mark.parametrize("L,with_bias", [(7, True), (14, False), (2, True), (9, True)])
def test_set_attention_matches_float64(L, with_bias):
    from sgrl_amd import train_ops
    torch.manual_seed(L)
    B, scale = 23, 128 ** -0.5
    qkv, vgp, gdir = torch.randn(B, L, 768) * 0.6, torch.randn(B, L, 3, 252), torch.randn(B, L, 3, 2)
    bias = torch.randn(2, L, L) if with_bias else None
    d_o, d_og = torch.randn(B, L, 256), torch.randn(B, L, 3, 256)
    ref = [t.double().requires_grad_() for t in (qkv, vgp, gdir)] + ([bias.double().requires_grad_()] if with_bias else [None])
    enabled = train_ops.ENABLED
    train_ops.ENABLED = False                                    # the einsum formulation, in float64 on the CPU
    try:
        o_r, og_r = train_ops.set_attention(*ref, scale)
    finally:
        train_ops.ENABLED = enabled
    ((o_r * d_o.double()).sum() + (og_r * d_og.double()).sum()).backward()
    dev = [t.cuda().requires_grad_() for t in (qkv, vgp, gdir)] + ([bias.cuda().requires_grad_()] if with_bias else [None])
    o, og = train_ops.set_attention(*dev, scale)
    assert type(o.grad_fn).__name__.startswith("_AttnFn")
    ((o * d_o.cuda()).sum() + (og * d_og.cuda()).sum()).backward()
    assert float((o.detach().cpu().double() - o_r.detach()).abs().max()) < 1e-5
    assert float((og.detach().cpu().double() - og_r.detach()).abs().max()) < 1e-5
    for d, r, name in zip(dev, ref, "qkv vgp gdir bias".split()):
        if r is None:
            continue
        assert d.grad.shape == r.grad.shape
        assert float((d.grad.cpu().double() - r.grad).abs().max()) < 2e-5 * (float(r.grad.abs().max()) + 1), name


def test_deferred_weight_gradients_equal_the_immediate_ones():
    """train_ops.deferred_wgrads: the weight / bias gradients of a whole backward pass issued together at the end (grouped
    launches) equal those computed layer by layer inside backward() to float32 rounding -- the same products; a grouped launch
    sums a small gradient's long contraction in one run where the lone launch splits it (fixed orders both, so each mode is
    reproducible bit for bit by itself)."""
    from oracle.formula import synth_obs
    from sgrl_amd import graph as G, mjcf, train_ops
    from sgrl_amd.rollout import TRAV
    from sgrl_amd.td3 import Agent, default_train_args
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    agent = Agent(default_train_args(), device=dev, use_hip=False)
    m = mjcf.load_asset("3d_cheetah_12_rightbknee")
    agent.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=dev))
    L = m.num_limbs
    obs = torch.from_numpy(synth_obs(L, 100, 4).astype(np.float32)).to(dev)
    act = torch.rand(100, 3 * L, device=dev) * 2 - 1
    grads = []
    for deferred in (False, True):
        agent.critic.zero_grad()
        q1, q2 = agent.critic(obs, act)
        loss = (q1 ** 2).mean() + (q2 ** 2).mean()
        with train_ops.deferred_wgrads(enabled=deferred):
            loss.backward()
        torch.cuda.synchronize()
        grads.append([p.grad.detach().clone() for p in agent.critic.parameters() if p.grad is not None])
    assert len(grads[0]) == len(grads[1]) > 100
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= 2e-6 * (float(a.abs().max()) + 1e-12) + 1e-12
    # each mode by itself is deterministic
    agent.critic.zero_grad()
    q1, q2 = agent.critic(obs, act)
    with train_ops.deferred_wgrads(enabled=True):
        ((q1 ** 2).mean() + (q2 ** 2).mean()).backward()
    torch.cuda.synchronize()
    again = [p.grad.detach().clone() for p in agent.critic.parameters() if p.grad is not None]
    for a, b in zip(grads[1], again):
        assert torch.equal(a, b)


def test_zmat_matches_float64():
    from sgrl_amd import train_ops
    torch.manual_seed(21)
    z, mat, dt = torch.randn(31, 5, 3, 32), torch.randn(31, 5, 32, 32) * 0.2, torch.randn(31, 5, 3, 32)
    zr, mr = z.double().requires_grad_(), mat.double().requires_grad_()
    tr = torch.einsum("blsa,blac->blsc", zr, mr)
    tr.backward(dt.double())
    zd, md = z.cuda().requires_grad_(), mat.cuda().requires_grad_()
    t = train_ops.zmat(zd, md)
    assert type(t.grad_fn).__name__.startswith("_ZmatFn")
    t.backward(dt.cuda())
    assert float((t.detach().cpu().double() - tr.detach()).abs().max()) < 1e-5
    assert float((zd.grad.cpu().double() - zr.grad).abs().max()) < 1e-5
    assert float((md.grad.cpu().double() - mr.grad).abs().max()) < 1e-5


def test_twin_critic_pass_equals_the_two_separate_passes():
    """set_policy.twin_forward (both critics of SECritic walked at once, every linear layer one launch for the pair:
    train_ops.linear2 / k_sgemm_twin) against the two TransformerModels run one after the other: Q values bit for bit (the
    same product kernels on the same operands), parameter gradients to float32 rounding (the attention of layers 1, 2 and
    the weight-free operations see the pair as one batch, so a few sums run in another order), with and without deferred
    weight gradients, and through the geo_grad = False path of the actor loss."""
    from oracle.formula import synth_obs
    from sgrl_amd import graph as G, mjcf, set_policy, train_ops
    from sgrl_amd.rollout import TRAV
    from sgrl_amd.td3 import Agent, default_train_args
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    agent = Agent(default_train_args(), device=dev, use_hip=False)
    for name in ("3d_walker_7_full", "3d_hopper_3_shin"):
        m = mjcf.load_asset(name)
        agent.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=dev))
        L = m.num_limbs
        obs = torch.from_numpy(synth_obs(L, 100, 4).astype(np.float32)).to(dev)
        act = torch.rand(100, 3 * L, device=dev) * 2 - 1
        out = {}
        for twin in (False, True):
            for deferred in (False, True):
                set_policy.TWIN_CRITICS = twin
                agent.critic.zero_grad()
                q1, q2 = agent.critic(obs, act)
                loss = (q1 ** 2).mean() + 0.5 * (q2 ** 3).mean()
                with train_ops.deferred_wgrads(enabled=deferred):
                    loss.backward()
                torch.cuda.synchronize()
                out[(twin, deferred)] = (q1.detach().clone(), q2.detach().clone(),
                                         {k: p.grad.detach().clone() for k, p in agent.critic.named_parameters() if p.grad is not None})
        set_policy.TWIN_CRITICS = True
        ref = out[(False, False)]
        for key in ((True, False), (True, True)):
            got = out[key]
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (name, key)
            assert got[2].keys() == ref[2].keys() and len(ref[2]) > 100
            for k in ref[2]:
                a, b = ref[2][k], got[2][k]
                assert float((a - b).abs().max()) <= 3e-6 * (float(a.abs().max()) + 1e-12) + 1e-12, (name, key, k)
        # the critic inside the actor loss takes gradients w.r.t. the action only (geo_grad = False): twin path, d/d action
        a_req = act.clone().requires_grad_(True)
        grads = []
        for twin in (False, True):
            set_policy.TWIN_CRITICS = twin
            q1, q2 = agent.critic(obs, a_req)
            g, = torch.autograd.grad(q1.mean() + q2.mean(), a_req)
            grads.append(g)
        set_policy.TWIN_CRITICS = True
        assert float((grads[0] - grads[1]).abs().max()) <= 3e-6 * float(grads[0].abs().max()) + 1e-12


@pytest.mark.parametrize("M,K,N,has_bias,relu,rowdiv,shared", [
    (700, 256, 256, True, True, False, False), (700, 256, 1024, True, False, True, False), (13, 126, 30, False, False, False, False),
    (2100, 128, 252, False, False, False, False), (700, 20, 128, True, False, False, True), (65, 256, 1, True, False, True, False),
    (300, 576, 256, True, True, False, False)])
def test_linear2_matches_float64(M, K, N, has_bias, relu, rowdiv, shared):
    """train_ops.linear2 (the same layer of two networks in one launch: k_sgemm_twin, include/sgrl_train.h
    sgrl_linear_forward_twin / sgrl_linear_dgrad_twin): values and every gradient against float64 PyTorch for both networks,
    stacked and shared inputs, ReLU masks and row divisors, ragged sizes, a row-strided incoming gradient."""
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(M + 5 * K + 11 * N)
    x = torch.randn((M, K) if shared else (2, M, K), generator=g)
    w = [torch.randn(N, K, generator=g) / np.sqrt(K) for _ in range(2)]
    b = [torch.randn(N, generator=g) for _ in range(2)] if has_bias else [None, None]
    rd = (torch.rand(2, M, 1, generator=g) + 0.5) if rowdiv else None
    dy_wide = torch.randn(2, M, N + 8, generator=g)           # the gradient arrives as a slice of a wider tensor
    xd = x.cuda().requires_grad_()
    wd = [t.cuda().requires_grad_() for t in w]
    bd = [t.cuda().requires_grad_() if t is not None else None for t in b]
    rdd = rd.cuda().requires_grad_() if rowdiv else None
    y = train_ops.linear2(xd, wd[0], wd[1], bd[0], bd[1], relu, rdd, shared)
    assert y.shape == (2, M, N) and type(y.grad_fn).__name__.startswith("_Linear2Fn")
    y.backward(dy_wide.cuda()[..., :N])
    # float64 reference; the ReLU's mask is the DEVICE's (an output within float32 rounding of zero may fall on the other side
    # in float64, which would move a gradient by a whole term without either result being wrong)
    mask = (y.detach().cpu() > 0).double()
    xr = x.double().requires_grad_()
    wr = [t.double().requires_grad_() for t in w]
    br = [t.double().requires_grad_() if t is not None else None for t in b]
    rdr = rd.double().requires_grad_() if rowdiv else None
    ys = []
    for i in range(2):
        yi = torch.nn.functional.linear(xr if shared else xr[i], wr[i], br[i])
        yi = yi * mask[i] if relu else yi
        ys.append(yi / rdr[i] if rowdiv else yi)
    yr = torch.stack(ys)
    yr.backward(dy_wide[..., :N].double())
    tol = lambda ref: 3e-6 * (float(ref.abs().max()) + 1.0)
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < tol(yr)
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) < tol(xr.grad)
    for i in range(2):
        assert float((wd[i].grad.cpu().double() - wr[i].grad).abs().max()) < tol(wr[i].grad) * np.sqrt(M / 64 + 1), i
        if has_bias:
            assert float((bd[i].grad.cpu().double() - br[i].grad).abs().max()) < tol(br[i].grad) * np.sqrt(M / 64 + 1), i
    if rowdiv:
        assert float((rdd.grad.cpu().double() - rdr.grad).abs().max()) < tol(rdr.grad)


@pytest.mark.parametrize("rows,twin,with_res", [(700, False, True), (700, True, True), (4200, True, False), (13, False, False),
                                                (1, True, True), (2100, False, True)])
def test_add_layer_norm_matches_float64(rows, twin, with_res):
    """train_ops.add_layer_norm / add_layer_norm2 (one launch forward, one backward) against float64 LayerNorm(x + res) on the CPU:
    value, gradients of x, res and both affine pairs; bit-reproducible; frozen affine parameters (the critic inside the actor loss)."""
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(rows * 3 + twin + 2 * with_res)
    shape = (2, rows, 128) if twin else (rows, 128)
    x = torch.randn(shape, generator=g) * 3 + 0.5
    res = torch.randn(shape, generator=g) if with_res else None
    dy = torch.randn(shape, generator=g)
    norms = [torch.nn.LayerNorm(128) for _ in range(2 if twin else 1)]
    for n in norms:
        with torch.no_grad():
            n.weight.copy_(torch.randn(128, generator=g)); n.bias.copy_(torch.randn(128, generator=g))
    # float64 reference
    ref_norms = [torch.nn.LayerNorm(128).double() for _ in norms]
    for rn, n in zip(ref_norms, norms):
        rn.load_state_dict({k: v.double() for k, v in n.state_dict().items()})
    xr = x.double().requires_grad_()
    rr = res.double().requires_grad_() if with_res else None
    sr = xr + rr if with_res else xr
    yr = torch.stack([ref_norms[0](sr[0]), ref_norms[1](sr[1])]) if twin else ref_norms[0](sr)
    yr.backward(dy.double())

    def run():
        dn = [torch.nn.LayerNorm(128).cuda() for _ in norms]
        for d, n in zip(dn, norms):
            d.load_state_dict(n.state_dict())
        xd = x.cuda().requires_grad_()
        rd = res.cuda().requires_grad_() if with_res else None
        y = train_ops.add_layer_norm2(xd, rd, dn[0], dn[1]) if twin else train_ops.add_layer_norm(xd, rd, dn[0])
        assert type(y.grad_fn).__name__.startswith("_AddLNFn")
        y.backward(dy.cuda())
        return y, xd, rd, dn
    y, xd, rd, dn = run()
    tol = lambda ref: 2e-5 * (float(ref.abs().max()) + 1.0)
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < tol(yr)
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) < tol(xr.grad)
    if with_res:
        assert float((rd.grad.cpu().double() - rr.grad).abs().max()) < tol(rr.grad)
    for d, rn in zip(dn, ref_norms):
        assert float((d.weight.grad.cpu().double() - rn.weight.grad).abs().max()) < tol(rn.weight.grad) * np.sqrt(rows / 64 + 1)
        assert float((d.bias.grad.cpu().double() - rn.bias.grad).abs().max()) < tol(rn.bias.grad) * np.sqrt(rows / 64 + 1)
    y2, xd2, _, dn2 = run()
    assert torch.equal(y, y2) and torch.equal(xd.grad, xd2.grad) and all(torch.equal(a.weight.grad, b.weight.grad) for a, b in zip(dn, dn2))
    # frozen affine parameters: only the input gradient is produced
    fz = torch.nn.LayerNorm(128).cuda()
    fz.load_state_dict(norms[0].state_dict())
    for p in fz.parameters():
        p.requires_grad_(False)
    xs = (x[0] if twin else x).cuda().requires_grad_()
    train_ops.add_layer_norm(xs, None, fz).backward((dy[0] if twin else dy).cuda())
    assert fz.weight.grad is None and xs.grad is not None
    # no autograd: the module itself
    with torch.no_grad():
        yn = train_ops.add_layer_norm(xs, None, fz)
    assert yn.grad_fn is None and float((yn - fz(xs)).abs().max()) == 0.0


@pytest.mark.parametrize("twin", [False, True])
def test_linear_followers_match_float64(twin):
    """The two followers that ride on a product's launch (sgrl_linear_forward_fused): `tail` appends columns (z = [proj(x) | gdir],
    30 + 2 of a 32-wide row), `addend` adds a residual (g + linear5(.)); values and every gradient against float64 torch.cat / add."""
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(11 + twin)
    lead = (2,) if twin else ()
    x = torch.randn(*lead, 100, 7, 3, 128, generator=g)
    gdir = torch.randn(100, 7, 3, 2, generator=g)
    ws = [torch.randn(30, 128, generator=g) / 11 for _ in range(2)]
    w5 = [torch.randn(128, 32, generator=g) / 6 for _ in range(2)]
    res = torch.randn(*lead, 100, 7, 3, 128, generator=g)
    dz = torch.randn(*lead, 100, 7, 3, 32, generator=g)
    dg = torch.randn(*lead, 100, 7, 3, 128, generator=g)

    def run(dev, dt):
        xs, gd, rs = x.to(dev, dt).requires_grad_(), gdir.to(dev, dt).requires_grad_(), res.to(dev, dt).requires_grad_()
        w = [t.to(dev, dt).requires_grad_() for t in ws]
        v = [t.to(dev, dt).requires_grad_() for t in w5]
        if dev == "cpu":
            if twin:
                z = torch.stack([torch.cat([xs[i] @ w[i].T, gd], -1) for i in range(2)])
                out = torch.stack([rs[i] + z[i] @ v[i].T for i in range(2)])
            else:
                z = torch.cat([xs @ w[0].T, gd], -1)
                out = rs + z @ v[0].T
        elif twin:
            z = train_ops.linear2(xs, w[0], w[1], tail=gd.unsqueeze(0).expand(2, *gd.shape))
            out = train_ops.linear2(z, v[0], v[1], addend=rs)
            assert type(z.grad_fn).__name__.startswith("_Linear2Fn")
        else:
            z = train_ops.linear(xs, w[0], tail=gd)
            out = train_ops.linear(z, v[0], addend=rs)
            assert type(z.grad_fn).__name__.startswith("_LinearFn")
        (z * dz.to(dev, dt)).sum().backward(retain_graph=True)
        (out * dg.to(dev, dt)).sum().backward()
        grads = [xs.grad, gd.grad, rs.grad] + [t.grad for t in (w if twin else w[:1])] + [t.grad for t in (v if twin else v[:1])]
        return [z.detach(), out.detach()] + grads
    ref = run("cpu", torch.float64)
    got = run("cuda", torch.float32)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert float((a.cpu().double() - b).abs().max()) < 2e-5 * (float(b.abs().max()) + 1.0) * 6
    got2 = run("cuda", torch.float32)
    assert all(torch.equal(a, b) for a, b in zip(got, got2))


@pytest.mark.parametrize("N,nt", [(128, 128), (30, 2), (100, 60), (32, 1)])
def test_wide_appended_blocks_match_cat(N, nt):
    """tail blocks of any width ([inv | ng]: 128 + 128): the last product tile's spare columns, then column tiles of their own."""
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(N + nt)
    x = torch.randn(2, 70, 5, 64, generator=g).cuda().requires_grad_()
    t = torch.randn(2, 70, 5, nt, generator=g).cuda().requires_grad_()
    w = [(torch.randn(N, 64, generator=g) / 8).cuda().requires_grad_() for _ in range(2)]
    b = [torch.randn(N, generator=g).cuda().requires_grad_() for _ in range(2)]
    dy = torch.randn(2, 70, 5, N + nt, generator=g).cuda()
    y1 = train_ops.linear(x[0], w[0], b[0], tail=t[0])
    y2 = train_ops.linear2(x, w[0], w[1], b[0], b[1], tail=t)
    ref = [torch.cat([torch.nn.functional.linear(x[i].double(), w[i].double(), b[i].double()), t[i].double()], -1) for i in range(2)]
    assert y1.shape == (70, 5, N + nt) and y2.shape == (2, 70, 5, N + nt)
    assert float((y1.double() - ref[0]).abs().max()) < 1e-5 and float((y2.double() - torch.stack(ref)).abs().max()) < 1e-5
    assert torch.equal(y2[..., N:], t) and torch.equal(y1[..., N:], t[0])
    (y2 * dy).sum().backward()
    assert torch.equal(t.grad, dy[..., N:])
    gx = torch.stack([dy[i, ..., :N].double() @ w[i].double() for i in range(2)])
    assert float((x.grad.double() - gx).abs().max()) < 1e-4


def test_embed3_matches_the_three_embeddings():
    """train_ops.embed3 (one launch each way) against torch's three nn.Embedding lookups + cat: values bit-equal, weight
    gradients equal (repeated indices included), rows no limb points at receive zero."""
    from sgrl_amd import train_ops
    torch.manual_seed(2)
    embs = torch.nn.ModuleList([torch.nn.Embedding(15, s) for s in (42, 42, 44)]).cuda()
    ref = torch.nn.ModuleList([torch.nn.Embedding(15, s) for s in (42, 42, 44)]).cuda()
    ref.load_state_dict(embs.state_dict())
    idx = [torch.tensor(v, device="cuda") for v in ([0, 1, 2, 3, 4, 5, 6], [3, 3, 0, 6, 2, 14, 1], [6, 5, 4, 3, 2, 1, 0])]
    dy = torch.randn(7, 128, device="cuda")
    out = train_ops.embed3(embs, idx)
    assert type(out.grad_fn).__name__.startswith("_Embed3Fn")
    want = torch.cat([e(i) for e, i in zip(ref, idx)], dim=1)
    assert torch.equal(out, want)
    out.backward(dy)
    want.backward(dy)
    for a, b in zip(embs, ref):
        assert float((a.weight.grad - b.weight.grad).abs().max()) < 1e-6
    assert float(embs[0].weight.grad[7:].abs().max()) == 0.0
    out2 = train_ops.embed3(embs, idx)          # cached index block
    assert torch.equal(out2, want)


@pytest.mark.parametrize("max_norm", [0.1, 0.0, 1e9])
def test_table_optimizer_steps_match_torch(max_norm):
    """td3.clip_and_step (gradient clipping + Adam over a device table of tensor addresses, three launches) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on ragged tensors, three steps: parameters, both moments, step counters
    and the clipped gradients left behind; and the soft target update against the reference's formula."""
    from sgrl_amd import td3
    g = torch.Generator().manual_seed(int(max_norm * 10) % 97 + 3)
    shapes = [(256, 256), (128,), (30, 128), (1,), (15, 42), (1024, 256), (5000,), (4097,), (3, 3, 3)]
    base = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * (0.01 if k else 3.0) for s in shapes] for k in range(3)]
    outs = []
    for table in (True, False):
        td3._TABLE_OPT = table
        ps = [torch.nn.Parameter(b.clone().cuda()) for b in base]
        opt = torch.optim.Adam(ps, lr=1e-4, capturable=True)
        for k in range(3):
            for p, gr in zip(ps, grads[k]):
                p.grad = gr.clone().cuda()
            td3.clip_and_step(opt, max_norm)
        outs.append(([p.detach().clone() for p in ps], [opt.state[p]["exp_avg"].clone() for p in ps],
                     [opt.state[p]["exp_avg_sq"].clone() for p in ps], [float(opt.state[p]["step"]) for p in ps], [p.grad.clone() for p in ps]))
    td3._TABLE_OPT = True
    (pa, ma, va, sa, ga), (pb, mb, vb, sb, gb) = outs
    assert sa == sb == [3.0] * len(shapes)
    for a, b, b0 in zip(pa, pb, base):
        assert float((a - b).abs().max()) <= 5e-7                          # one float32 ulp of parameters of magnitude ~4
        assert float((a.cpu() - b0).abs().max()) > 1e-5                    # and they did move
    for a, b in zip(ma + va + ga, mb + vb + gb):
        assert float((a - b).abs().max()) <= 2e-6 * (float(b.abs().max()) + 1e-12)
    # soft update
    src = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    dst = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    want = [0.005 * a.detach().double() + 0.995 * b.detach().double() for a, b in zip(src, dst)]
    class _N(torch.nn.Module):
        def __init__(self, ps):
            super().__init__()
            self.ps = torch.nn.ParameterList(ps)
    td3.soft_update_network(_N(src), _N(dst), 0.005)
    for b, w in zip(dst, want):
        assert float((b.detach().double() - w).abs().max()) < 1e-6


@pytest.mark.parametrize("M,K,H,N,rowdiv,tail,deferred", [(1792, 256, 1024, 256, True, False, False), (700, 1024, 256, 128, False, True, True),
                                                          (1792, 256, 256, 1024, True, False, True), (65, 145, 128, 20, False, False, False)])
def test_feed_forward_pair_with_the_mask_in_the_second_layers_epilogue(M, K, H, N, rowdiv, tail, deferred):
    """lin2(relu(lin1(x))) with the ReLU mask applied ONCE, in the epilogue of lin2's input gradient (x_relu / premasked:
    include/sgrl_train.h sgrl_linear_backward_xrelu) against float64, and against the same pair with the mask read by lin1's two
    backward products (the form of rounds 2-4)."""
    from sgrl_amd import train_ops
    g = torch.Generator().manual_seed(M + K + H + N)
    x = torch.randn(M, K, generator=g)
    w1, b1 = torch.randn(H, K, generator=g) / np.sqrt(K), torch.randn(H, generator=g)
    w2, b2 = torch.randn(N, H, generator=g) / np.sqrt(H), torch.randn(N, generator=g)
    rd = (torch.rand(M, 1, generator=g) + 0.5) if rowdiv else None
    tl = torch.randn(M, 5, generator=g) if tail else None
    dy = torch.randn(M, N + (5 if tail else 0), generator=g)
    ref = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2)]
    rdr = rd.double().requires_grad_() if rowdiv else None
    yr = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
    yr = yr / rdr if rowdiv else yr
    yr = torch.cat([yr, tl.double()], -1) if tail else yr
    yr.backward(dy.double())

    def run(fused):
        leaves = [torch.nn.Parameter(t.cuda()) for t in (x, w1, b1, w2, b2)]
        rdd = rd.cuda().requires_grad_() if rowdiv else None
        tld = tl.cuda() if tail else None
        with train_ops.deferred_wgrads(deferred):
            h = train_ops.linear(leaves[0], leaves[1], leaves[2], relu=True, premasked=fused)
            y = train_ops.linear(h, leaves[3], leaves[4], rowdiv=rdd, tail=tld, x_relu=fused)
            y.backward(dy.cuda())
        return y, leaves, rdd

    y, lv, rdd = run(True)
    y0, lv0, rdd0 = run(False)
    assert torch.equal(y, y0)
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 3e-6 * (float(yr.abs().max()) + 1.0)
    for a, a0, r in zip(lv, lv0, ref):
        tol = 3e-6 * (float(r.grad.abs().max()) + 1.0) * np.sqrt(M / 64 + 1)
        assert float((a.grad.cpu().double() - r.grad).abs().max()) < tol
        assert float((a.grad - a0.grad).abs().max()) < tol
    if rowdiv:
        assert float((rdd.grad.cpu().double() - rdr.grad).abs().max()) < 3e-6 * (float(rdr.grad.abs().max()) + 1.0) * np.sqrt(N / 64 + 1)


def test_twin_feed_forward_pair_with_the_mask_in_the_second_layers_epilogue():
    """The same for the two critics' layers in one launch (linear2: k_sgemm_twin): both forms against each other and float64."""
    from sgrl_amd import train_ops
    M, K, H, N = 1792, 256, 1024, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, M, K, generator=g)
    ws = [torch.randn(H, K, generator=g) / np.sqrt(K) for _ in range(2)] + [torch.randn(N, H, generator=g) / np.sqrt(H) for _ in range(2)]
    bs = [torch.randn(H, generator=g) for _ in range(2)] + [torch.randn(N, generator=g) for _ in range(2)]
    rd = torch.rand(2, M, 1, generator=g) + 0.5
    dy = torch.randn(2, M, N, generator=g)
    xr = x.double().requires_grad_()
    wr, br = [w.double().requires_grad_() for w in ws], [b.double().requires_grad_() for b in bs]
    yr = torch.stack([torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(xr[i], wr[i], br[i])), wr[2 + i], br[2 + i]) / rd[i].double()
                      for i in range(2)])
    yr.backward(dy.double())

    def run(fused):
        xd = x.cuda().requires_grad_()
        wd, bd = [torch.nn.Parameter(w.cuda()) for w in ws], [torch.nn.Parameter(b.cuda()) for b in bs]
        with train_ops.deferred_wgrads(True):
            h = train_ops.linear2(xd, wd[0], wd[1], bd[0], bd[1], relu=True, premasked=fused)
            y = train_ops.linear2(h, wd[2], wd[3], bd[2], bd[3], rowdiv=rd.cuda(), x_relu=fused)
            assert type(y.grad_fn).__name__.startswith("_Linear2Fn")
            y.backward(dy.cuda())
        return y, [xd] + wd + bd

    y, lv = run(True)
    y0, lv0 = run(False)
    assert torch.equal(y, y0)
    for a, a0, r in zip(lv, lv0, [xr] + wr + br):
        tol = 3e-6 * (float(r.grad.abs().max()) + 1.0) * np.sqrt(M / 64 + 1)
        assert float((a.grad.cpu().double() - r.grad).abs().max()) < tol
        assert float((a.grad - a0.grad).abs().max()) < tol


@pytest.mark.parametrize("deferred", [False, True])
def test_triangular_gram_and_folded_weights_match_the_full_form(deferred):
    """tri(Z'Z) [528] with the invariant layer's weight folded onto the lower triangle (sgrl_gram_tri_*, sgrl_sym_fold) against the
    full vec(Z'Z) [1024] form in float64: values, dz, the weight gradient (unfolded onto both mirror columns), the norm's gradient."""
    from sgrl_amd import train_ops
    M = 300
    g = torch.Generator().manual_seed(11)
    z = torch.randn(M, 3, 32, generator=g)
    w = torch.randn(256, 1024, generator=g) / 32
    w2 = torch.randn(64, 1024, generator=g) / 32
    b = torch.randn(256, generator=g)
    dy, dfn = torch.randn(M, 256, generator=g), torch.randn(M, 1, generator=g)
    dy2 = torch.randn(M, 64, generator=g)
    zr, wr, w2r, br = z.double().requires_grad_(), w.double().requires_grad_(), w2.double().requires_grad_(), b.double().requires_grad_()
    gram = torch.einsum("msa,msc->mac", zr, zr).flatten(-2)
    fnr = gram.norm(dim=-1, keepdim=True) + 1.0
    yr = torch.relu(torch.nn.functional.linear(gram, wr, br))
    y2r = torch.nn.functional.linear(gram, w2r)
    ((yr * dy.double()).sum() + (fnr * dfn.double()).sum() + (y2r * dy2.double()).sum()).backward()
    zd, wd, w2d, bd = z.cuda().requires_grad_(), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(w2.cuda()), torch.nn.Parameter(b.cuda())
    with train_ops.deferred_wgrads(deferred):       # deferred: the folded weights' gradients are grouped and unfolded at the flush
        wt = train_ops.tri_weights([wd, w2d], zd)
        assert wt is not None and wt[0].shape == (256, 528) and wt[1].shape == (64, 528)
        tri, fn = train_ops.gram_tri_fn(zd)
        y = train_ops.linear(tri, wt[0], bd, relu=True)
        y2 = train_ops.linear(tri, wt[1])
        ((y * dy.cuda()).sum() + (fn * dfn.cuda()).sum() + (y2 * dy2.cuda()).sum()).backward()
    assert float((fn.detach().cpu().double() - fnr.detach()).abs().max()) < 1e-5 * float(fnr.abs().max())
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 1e-5 * (float(yr.abs().max()) + 1)
    assert float((y2.detach().cpu().double() - y2r.detach()).abs().max()) < 1e-5 * (float(y2r.abs().max()) + 1)
    for got, ref in ((zd.grad, zr.grad), (wd.grad, wr.grad), (w2d.grad, w2r.grad), (bd.grad, br.grad)):
        assert float((got.cpu().double() - ref).abs().max()) < 2e-5 * (float(ref.abs().max()) + 1), (got.shape,)
    # the folded weight's gradient lands on both mirror columns
    gw = wd.grad.view(256, 32, 32)
    assert torch.equal(gw, gw.transpose(1, 2))


@pytest.mark.parametrize("twin", [False, True], ids=["one_network", "twin"])
def test_fan_out_sums_the_consumers_input_gradients_in_one_buffer(twin):
    """train_ops.fan_out: a tensor feeding two linear layers (one behind a ReLU pair, one with appended columns) and a residual --
    the SET layer's pattern (reference SEActor.py:93-121) -- gets the gradient autograd's own additions give it (float64 reference),
    with the two products accumulating in one buffer (sgrl_linear_backward_acc / sgrl_linear_dgrad_twin_acc)."""
    from sgrl_amd import train_ops
    torch.manual_seed(11)
    lead = (2,) if twin else ()
    M, K = 333, 128
    x0 = torch.randn(*lead, M, K)
    wa, wb, wc = torch.randn(2, 64, K) / 11, torch.randn(2, 30, K) / 11, torch.randn(2, 40, 64) / 8
    tail = torch.randn(*lead, M, 2)
    da, db, dres = torch.randn(*lead, M, 40), torch.randn(*lead, M, 32), torch.randn(*lead, M, K)

    def ref():
        x = x0.double().requires_grad_()
        tot = 0.0
        for i in range(2 if twin else 1):
            xi = x[i] if twin else x
            ya = torch.relu(xi @ wa[i].double().T) @ wc[i].double().T
            yb = torch.cat([xi @ wb[i].double().T, (tail[i] if twin else tail).double()], -1)
            tot = tot + (ya * (da[i] if twin else da).double()).sum() + (yb * (db[i] if twin else db).double()).sum()
        tot = tot + ((x * 3.0) * dres.double()).sum()
        tot.backward()
        return x.grad

    def run(use_fan_out):
        x = x0.cuda().requires_grad_()
        h = x * 1.0                                           # a non-leaf, as inside a network
        if use_fan_out:
            slot, (h1, h2, h3) = train_ops.fan_out(h, 3)
            assert slot is not None
        else:
            slot, (h1, h2, h3) = None, (h, h, h)
        W = [[w[i].cuda().requires_grad_() for i in range(2)] for w in (wa, wb, wc)]
        if twin:
            hid = train_ops.linear2(h1, W[0][0], W[0][1], relu=True, premasked=True, slot=slot)
            ya = train_ops.linear2(hid, W[2][0], W[2][1], x_relu=True)
            yb = train_ops.linear2(h2, W[1][0], W[1][1], tail=tail.cuda(), slot=slot)
        else:
            hid = train_ops.linear(h1, W[0][0], relu=True, premasked=True, slot=slot)
            ya = train_ops.linear(hid, W[2][0], x_relu=True)
            yb = train_ops.linear(h2, W[1][0], tail=tail.cuda(), slot=slot)
        ((ya * da.cuda()).sum() + (yb * db.cuda()).sum() + ((h3 * 3.0) * dres.cuda()).sum()).backward()
        return x.grad, [w.grad for ws in W for w in ws if w.grad is not None]

    gr = ref()
    g1, w1 = run(True)
    g0, w0 = run(False)
    scale = float(gr.abs().max()) + 1.0
    assert float((g1.cpu().double() - gr).abs().max()) < 3e-6 * scale
    assert float((g0.cpu().double() - gr).abs().max()) < 3e-6 * scale
    for a, b in zip(w1, w0):
        assert torch.equal(a, b)                              # the weight gradients do not know about the fan-out
    g1b, _ = run(True)
    assert torch.equal(g1, g1b)                               # same order of accumulation every time: bit-reproducible
