"""GPU parity of the HIP SET-actor forward (C ABI of include/sgrl_set.h) against the reference-generated golden
vectors and the NumPy oracle.  float32 arithmetic on the f32 matrix cores: tolerance 2e-5 absolute on tanh-squashed
actions of magnitude <= 1 (the reference's own f32-vs-f64 gap on these vectors is ~1e-6)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TRAV = ["pre", "inlcrs", "postlcrs"]
TOL = 2e-5


@pytest.fixture(scope="module")
def ctx(golden_dir):
    import torch
    assert torch.cuda.is_available()
    from oracle.formula import apply_formula_
    from sgrl_amd.set_policy import make_policy
    with open(os.path.join(golden_dir, "graphs.json")) as f:
        graphs = json.load(f)
    with open(os.path.join(golden_dir, "set_state_dict_keys.json")) as f:
        keys = json.load(f)
    z = np.load(os.path.join(golden_dir, "set_forward.npz"))
    pol = make_policy(device="cuda:0", use_hip=True).eval()
    apply_formula_(pol)
    return torch, pol, graphs, keys, z


def _gd(torch, g, dev="cuda:0"):
    from sgrl_amd import graph as G
    return G.getGraphDict(g["parents"], TRAV, [], device=torch.device(dev))


def test_module_fast_path_matches_reference_fixtures(ctx):
    torch, pol, graphs, keys, z = ctx
    worst = 0.0
    for name, g in graphs.items():
        pol.change_morphology(_gd(torch, g))
        for B in (1, 5):
            obs = torch.from_numpy(z["%s/B%d/obs" % (name, B)]).cuda()
            with torch.no_grad():
                a = pol(obs)
            assert a.shape == (B, 3 * len(g["parents"])) and a.is_cuda
            a = a.cpu().numpy()
            e64 = np.abs(a - z["%s/B%d/act_f64" % (name, B)]).max()
            e32 = np.abs(a - z["%s/B%d/act_f32" % (name, B)]).max()
            worst = max(worst, e64)
            assert e64 < TOL and e32 < TOL, (name, B, e64, e32)
    assert pol._hip is not None   # the HIP path ran
    print("worst |hip - ref_f64| = %.3e" % worst)


def test_grad_mode_uses_the_differentiable_path_and_agrees(ctx):
    torch, pol, graphs, keys, z = ctx
    g = graphs["3d_walker_7_full"]
    pol.change_morphology(_gd(torch, g))
    obs = torch.from_numpy(z["3d_walker_7_full/B5/obs"]).cuda()
    a_grad = pol(obs)
    assert a_grad.requires_grad
    with torch.no_grad():
        a_hip = pol(obs)
    assert float((a_grad - a_hip).abs().max()) < TOL


def test_mixed_morphology_batch_against_numpy_oracle(ctx):
    torch, pol, graphs, keys, z = ctx
    from oracle import set_ref
    from oracle.formula import synth_obs
    from sgrl_amd.set_hip import HipSetActor
    names = sorted(n for n in graphs if "walker" in n)
    assert len(names) == 8
    counts = [3, 1, 2, 5, 1, 4, 2, 3]
    gds = [_gd(torch, graphs[n]) for n in names]
    act = HipSetActor(pol)
    act.configure(gds, counts)
    Lmax = max(len(graphs[n]["parents"]) for n in names)
    obs = np.zeros((sum(counts), 41 * Lmax), dtype=np.float32)
    rows = []
    r = 0
    for k, n in enumerate(names):
        L = len(graphs[n]["parents"])
        o = synth_obs(L, counts[k], 500 + k).astype(np.float32)
        obs[r:r + counts[k], :41 * L] = o
        rows.append((r, counts[k], L, o))
        r += counts[k]
    out = act.forward_batch(torch.from_numpy(obs).cuda()).cpu().numpy()
    assert out.shape == (sum(counts), 3 * Lmax)
    sd = set_ref.formula_state_dict(keys, np.float64)
    for (r0, c, L, o), n in zip(rows, names):
        ref = set_ref.set_actor_forward(sd, o.astype(np.float64), graphs[n]["traversals"],
                                        np.array(graphs[n]["relation"], dtype=np.float32).astype(np.float64))
        assert np.abs(out[r0:r0 + c, :3 * L] - ref).max() < TOL, n
        assert (out[r0:r0 + c, 3 * L:] == 0).all()


def test_full_size_batch_properties(ctx):
    """8 walkers x 1024 envs: rows equal the single-env forward; gravity-axis rotation invariance; determinism."""
    torch, pol, graphs, keys, z = ctx
    from oracle.formula import synth_obs
    from sgrl_amd.set_hip import HipSetActor
    names = sorted(n for n in graphs if "walker" in n)
    gds = [_gd(torch, graphs[n]) for n in names]
    act = HipSetActor(pol)
    act.configure(gds, [1024] * 8)
    obs = np.zeros((8192, 287), dtype=np.float32)
    for k, n in enumerate(names):
        L = len(graphs[n]["parents"])
        obs[1024 * k:1024 * (k + 1), :41 * L] = synth_obs(L, 1024, 900 + k)
    x = torch.from_numpy(obs).cuda()
    a0 = act.forward_batch(x).clone()
    a1 = act.forward_batch(x).clone()
    assert torch.equal(a0, a1)
    assert torch.isfinite(a0).all() and float(a0.abs().max()) <= 1.0
    # rotate every 3-vector about z
    th = 0.77
    rz = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=torch.float32).cuda()
    xr = x.clone().view(8192, 7, 41)
    xr[..., :24] = (xr[..., :24].reshape(8192, 7, 8, 3) @ rz.T).reshape(8192, 7, 24)
    a2 = act.forward_batch(xr.view(8192, 287))
    assert float((a2 - a0).abs().max()) < 5e-5
    # single rows through the module surface
    for i in (0, 1023, 4096, 8191):
        k = i // 1024
        L = len(graphs[names[k]]["parents"])
        pol.change_morphology(gds[k])
        with torch.no_grad():
            one = pol(x[i:i + 1, :41 * L])
        assert float((one[0] - a0[i, :3 * L]).abs().max()) < 1e-6


def test_weights_are_repacked_after_an_in_place_update(ctx):
    torch, pol, graphs, keys, z = ctx
    import copy
    p2 = copy.deepcopy(pol)
    p2._hip = None
    g = graphs["3d_hopper_3_shin"]
    p2.change_morphology(_gd(torch, g))
    obs = torch.from_numpy(z["3d_hopper_3_shin/B5/obs"]).cuda()
    with torch.no_grad():
        a = p2(obs).clone()
        p2.actor.decoder_g.weight.mul_(0.5)
        b = p2(obs).clone()
    p2.use_hip = False
    with torch.no_grad():
        c = p2(obs)
    assert float((a - b).abs().max()) > 1e-3
    assert float((b - c).abs().max()) < TOL


def test_critic_hip_path_matches_reference_fixtures_and_torch_path():
    """SECritic under no_grad on the GPU runs the HIP kernels (sgrl_set_forward_q); values against the fixtures produced
    by executing the reference's SECritic (tests/golden/critic_forward.npz) and against the module's PyTorch path."""
    import torch
    from oracle.formula import apply_formula_
    from sgrl_amd import graph as G, mjcf
    from sgrl_amd import set_policy
    from sgrl_amd.set_policy import make_critic
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "critic_forward.npz"))
    crit = make_critic(device="cuda:0").eval()
    apply_formula_(crit)
    assert set_policy.TWIN_TARGETS           # the default since round 6 (set_policy.py); the values of both paths are checked here
    for name in sorted({k.split("/")[0] for k in z.files}):
        m = mjcf.load_asset(name)
        crit.change_morphology(G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cuda:0")))
        obs, act = torch.from_numpy(z[name + "/obs"]).cuda(), torch.from_numpy(z[name + "/act"]).cuda()
        with torch.no_grad():
            q1, q2 = crit(obs, act)                       # HIP path: both networks in one pass of the training kernels (TWIN_TARGETS)
            q1b = crit.Q1(obs, act)                       # HIP path: the rollout kernels (sgrl_set_forward_q)
            set_policy.TWIN_TARGETS = False
            try:
                r1, r2 = crit(obs, act)                   # both networks on the rollout kernels, one after the other
            finally:
                set_policy.TWIN_TARGETS = True
        assert crit._hip is not None and q1.shape == (4, m.num_limbs)
        assert float((q1 - r1).abs().max()) < 2e-5 * np.abs(z[name + "/q1"]).max() and float((q2 - r2).abs().max()) < 2e-5 * np.abs(z[name + "/q2"]).max()
        crit.use_hip = False
        with torch.no_grad():
            t1, t2 = crit(obs, act)                       # PyTorch path, same weights
        crit.use_hip = True
        scale = np.abs(z[name + "/q1"]).max()
        for got, ref in ((q1, z[name + "/q1"]), (q2, z[name + "/q2"]), (q1b, z[name + "/q1"])):
            assert np.abs(got.cpu().numpy() - ref).max() < 2e-4 * scale, name
        assert float((q1 - t1).abs().max()) < 2e-4 * scale and float((q2 - t2).abs().max()) < 2e-4 * scale
    # in-place parameter update -> re-pack
    with torch.no_grad():
        crit.critic1.decoder_ng.bias.add_(0.5)
        q1c = crit.Q1(obs, act)
    crit.use_hip = False
    with torch.no_grad():
        t1c = crit.Q1(obs, act)
    assert float((q1c - t1c).abs().max()) < 2e-4 * max(1.0, float(t1c.abs().max()))
    assert float((q1c - q1).abs().max()) > 1e-3          # the bias reaches the output through 1 / fn
    # grad mode keeps the differentiable path
    crit.use_hip = True
    q = crit.Q1(obs, act.clone().requires_grad_(True))
    assert q.requires_grad
    assert set_policy.TWIN_TARGETS


def _soft_update_like_the_reference(source, target, tau):
    """common/functional.py:7-10 of the reference, verbatim semantics: writes through `.data` (no version bump)."""
    for target_param, local_param in zip(target.parameters(), source.parameters()):
        target_param.data.copy_(tau * local_param.data + (1 - tau) * target_param.data)


def test_reference_style_soft_update_reaches_the_hip_path(ctx):
    """agent.py:104-105,185-187: actor_target / critic_target are only ever written by soft_update_network, then read
    under no_grad (agent.py:136-148) -- the HIP path.  Neither `_version` nor `data_ptr()` changes; the values must."""
    torch, pol, graphs, keys, z = ctx
    import copy
    from sgrl_amd.set_policy import make_critic
    from oracle.formula import apply_formula_
    g = graphs["3d_walker_5_foot"]
    gd = _gd(torch, g)
    obs = torch.from_numpy(z["3d_walker_5_foot/B5/obs"]).cuda()
    actor = copy.deepcopy(pol); actor._hip = None
    target = copy.deepcopy(pol); target._hip = None
    with torch.no_grad():
        for p in actor.parameters():
            p.mul_(1.25)
    actor.change_morphology(gd); target.change_morphology(gd)
    with torch.no_grad():
        t0 = target(obs).clone()                     # first HIP forward binds the handle
    vers = [p._version for p in target.parameters()]
    for it in range(3):
        _soft_update_like_the_reference(actor, target, 0.3)
        with torch.no_grad():
            hip = target(obs).clone()
        target.use_hip = False
        with torch.no_grad():
            ref = target(obs)
        target.use_hip = True
        assert float((hip - ref).abs().max()) < TOL, it
    assert vers == [p._version for p in target.parameters()]      # the update really was invisible to version counters
    assert float((hip - t0).abs().max()) > 1e-3
    # same for the twin critics
    crit = make_critic(device="cuda:0").eval(); apply_formula_(crit)
    crit_t = make_critic(device="cuda:0").eval(); apply_formula_(crit_t)
    with torch.no_grad():
        for p in crit.parameters():
            p.mul_(0.8)
    crit_t.change_morphology(gd)
    act = (torch.rand((5, 3 * len(g["parents"])), device="cuda") * 2 - 1)
    with torch.no_grad():
        q0 = crit_t(obs, act)[0].clone()
    _soft_update_like_the_reference(crit, crit_t, 0.5)
    with torch.no_grad():
        q1, q2 = crit_t(obs, act)
    crit_t.use_hip = False
    with torch.no_grad():
        r1, r2 = crit_t(obs, act)
    scale = max(1.0, float(r1.abs().max()))
    assert float((q1 - r1).abs().max()) < 2e-4 * scale and float((q2 - r2).abs().max()) < 2e-4 * scale
    assert float((q1 - q0).abs().max()) > 1e-4


def test_morphology_and_batch_switching_is_cached_by_content(ctx):
    """trainer.py:173-200: change_morphology + select_action per env per step.  Switching between structures the handle
    has seen must give the same numbers as a fresh handle, also when the graph dicts are NEW objects (content key)."""
    torch, pol, graphs, keys, z = ctx
    names = ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full", "3d_cheetah_14_full", "3d_walker_2_right_leg_left_knee"]
    first = {}
    for rnd in range(3):
        for n in names:
            for B in (1, 5):
                pol.change_morphology(_gd(torch, graphs[n]))       # a new dict object every time
                obs = torch.from_numpy(z["%s/B%d/obs" % (n, B)]).cuda()
                with torch.no_grad():
                    a = pol(obs).cpu().numpy()
                assert np.abs(a - z["%s/B%d/act_f64" % (n, B)]).max() < TOL, (rnd, n, B)
                if rnd == 0:
                    first[(n, B)] = a
                else:
                    assert np.array_equal(a, first[(n, B)]), (rnd, n, B)


def test_too_narrow_rows_are_rejected(ctx):
    torch, pol, graphs, keys, z = ctx
    from sgrl_amd.set_hip import HipSetActor
    from sgrl_amd._lib import SgrlError
    act = HipSetActor(pol)
    act.configure([_gd(torch, graphs["3d_walker_7_full"])], [2])
    obs = torch.zeros((2, 287), device="cuda")
    with pytest.raises(AssertionError):
        act.forward_batch(obs[:, :200].contiguous())
    with pytest.raises(AssertionError):
        act.forward_batch(obs, act_ld=12)
    # and the raw ABI refuses as well
    import ctypes
    out = torch.zeros((2, 21), device="cuda")
    act.sync_weights()
    rc = act.L.sgrl_set_forward(act.h, ctypes.c_void_p(obs.data_ptr()), 200, ctypes.c_void_p(out.data_ptr()), 21,
                                ctypes.c_float(1.0), act._stream())
    assert rc != 0 and b"Lmax" in act.L.sgrl_set_last_error()


@pytest.mark.parametrize("small_nodes", [-1, 0], ids=["small_batch_products", "tile128_products"])
def test_layer_probes_on_the_gpu(ctx, golden_dir, small_nodes):
    """Walk the HIP forward stage by stage against the reference's own forward hooks on `layers.i.self_attn`, `layers.i`
    and `transformer_encoder` (tests/golden/set_probes_walker7.npz, captured by tools/capture_golden.py): a compensating
    error pair inside a layer cannot hide behind a correct final action."""
    torch, pol, graphs, keys, z = ctx
    from sgrl_amd.set_hip import HipSetActor
    p = np.load(os.path.join(golden_dir, "set_probes_walker7.npz"))
    g = graphs["3d_walker_7_full"]
    L, B = 7, 2
    act = HipSetActor(pol)
    act.debug_small_nodes(small_nodes)      # 14 nodes: the small-batch products by default; 0 forces the 128 x 128 tile kernels
    act.configure([_gd(torch, g)], [B])
    obs = torch.from_numpy(p["obs"].astype(np.float32)).cuda()

    def nodes(a):           # fixture [L, B, ...] -> node-major [B * L, ...] (limbs of one env contiguous)
        return np.swapaxes(a, 0, 1).reshape((B * L,) + a.shape[2:])

    def close(got, ref, what):
        ref = nodes(ref)
        err = np.abs(got.reshape(ref.shape) - ref).max()
        assert err < 2e-5 * (1 + np.abs(ref).max()), (what, err, np.abs(ref).max())

    try:
        for l in range(3):
            act.debug_stop_after(2 * l)
            act.forward_batch(obs)
            close(act.peek(8, 384), p["layer%d/attn/out0" % l], "layer%d attention vector output" % l)
            close(act.peek(9, 128), p["layer%d/attn/out1" % l], "layer%d attention scalar output" % l)
            act.debug_stop_after(2 * l + 1)
            act.forward_batch(obs)
            close(act.peek(0, 384), p["layer%d/out0" % l], "layer%d g" % l)
            close(act.peek(1, 256)[:, 128:], p["layer%d/out1" % l], "layer%d ng" % l)
    finally:
        act.debug_stop_after(-1)
    out = act.forward_batch(obs).cpu().numpy()
    close(act.peek(0, 384), p["encoder/out0"], "encoder g")
    close(act.peek(10, 160)[:, 17:145], p["encoder/out1"], "encoder ng (final norm)")
    assert np.abs(out - p["act_f64"]).max() < TOL


@pytest.mark.gpu
def test_small_batch_products_match_the_tile_kernels(ctx):
    """One input through both product paths of the forward (include/sgrl_set.h sgrl_set_debug_small_nodes): the 32 x 32 tile
    kernels batches below 2048 nodes take, and the 128 x 128 tile kernels of the collection step -- actor and critic, a
    mixed-morphology batch of 1 000 nodes."""
    torch, pol, graphs, keys, z = ctx
    from sgrl_amd.set_hip import HipSetActor
    from oracle.formula import synth_obs
    names = ["3d_walker_7_full", "3d_hopper_3_shin", "3d_cheetah_14_full"]
    names = [n for n in names if n in graphs] or list(graphs)[:2]
    counts = [40] * len(names)
    gds = [_gd(torch, graphs[n]) for n in names]
    act = HipSetActor(pol)
    act.configure(gds, counts)
    amax = 3 * max(len(graphs[n]["parents"]) for n in names)
    omax = 41 * max(len(graphs[n]["parents"]) for n in names)
    obs = torch.zeros((sum(counts), omax), device="cuda")
    r = 0
    for n, c in zip(names, counts):
        L = len(graphs[n]["parents"])
        obs[r:r + c, :41 * L] = torch.from_numpy(synth_obs(L, c, 3).astype(np.float32)).cuda()
        r += c
    outs = []
    for sn in (100000, 0):
        act.debug_small_nodes(sn)
        outs.append(act.forward_batch(obs).cpu().numpy().copy())
    act.debug_small_nodes(-1)
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 1e-3
    assert np.abs(outs[0] - outs[1]).max() < TOL


def test_tile_product_forms_agree_and_the_default_form_has_f32_range(ctx):
    """The two split forms of the 128 x 128 tile products (include/sgrl_set.h sgrl_set_gemm_form): f16 x 3 (default) and
    bf16 x 6 both reproduce the reference fixture at the suite's tolerance and agree far below it.  Absurd inputs (observations
    x 1e8, x 1e-12) go through the default form as they go through the module's own float32 forward: the operand rows are scaled
    into f16's range by powers of two before they are split (csrc/gemm_f32.h), nothing is clamped -- there is no range contract
    (reference SEActor.py:334-347: plain f32)."""
    torch, pol, graphs, keys, z = ctx
    from sgrl_amd.set_hip import HipSetActor
    name = "3d_walker_7_full"
    g = graphs[name]
    act = HipSetActor(pol)
    act.configure([_gd(torch, g)], [5])
    act.debug_small_nodes(0)                      # the tile kernels, whatever the batch size
    obs = torch.from_numpy(z["%s/B5/obs" % name]).cuda()
    ref = z["%s/B5/act_f64" % name]
    outs = {}
    for form in (HipSetActor.FORM_F16X3, HipSetActor.FORM_BF16X6):
        act.gemm_form(form)
        outs[form] = act.forward_batch(obs).cpu().numpy().copy()
        assert np.abs(outs[form] - ref).max() < TOL, form
    assert np.abs(outs[HipSetActor.FORM_F16X3] - outs[HipSetActor.FORM_BF16X6]).max() < 5e-6
    act.gemm_form(0)                              # default = f16 x 3
    pol.change_morphology(_gd(torch, g))
    for factor in (1e8, 1e-12):
        x = obs * factor
        with torch.enable_grad():                 # the module's own float32 PyTorch forward (the grad path never takes the HIP kernels)
            want = pol(x).detach().cpu().numpy()
        got = act.forward_batch(x).cpu().numpy()
        act.gemm_form(HipSetActor.FORM_BF16X6)
        other = act.forward_batch(x).cpu().numpy()
        act.gemm_form(0)
        assert np.isfinite(got).all()
        assert np.abs(got - other).max() < 2e-4, factor       # float32 evaluations of an ill-scaled input
        assert np.abs(got[:, :want.shape[1]] - want).max() < 2e-4, factor


def test_product_forms_agree_on_engine_observations_at_size(ctx):
    """Realistic operands: 8 walker variants x 256 environments stepped by the engine under random actions (falls, resets,
    velocity spikes included), the resulting observations through the tile kernels in both product forms -- the two-piece
    f16 form agrees with the bf16 x 6 form well below the suite's tolerance; both are deterministic."""
    torch, pol, graphs, keys, z = ctx
    from sgrl_amd.rollout import Rollout
    from sgrl_amd.set_hip import HipSetActor
    names = sorted(n for n in graphs if "walker" in n)
    ro = Rollout(names, 256, policy=pol, seed=5, device="cuda:0")
    ro.reset()
    worst = 0.0
    for t in range(60):
        obs, rew, done, _ = ro.step(ro.random_actions())
        if t % 20 != 19:
            continue
        outs = {}
        for form in (HipSetActor.FORM_F16X3, HipSetActor.FORM_BF16X6):
            ro.actor.gemm_form(form)
            a1 = ro.actor.forward_batch(obs, act_ld=ro.env.action_max_len).clone()
            a2 = ro.actor.forward_batch(obs, act_ld=ro.env.action_max_len)
            assert torch.equal(a1, a2), form                       # deterministic
            assert bool(torch.isfinite(a1).all())
            outs[form] = a1
        worst = max(worst, float((outs[HipSetActor.FORM_F16X3] - outs[HipSetActor.FORM_BF16X6]).abs().max()))
    ro.actor.gemm_form(0)
    assert ro.actor.num_nodes > 2048                               # the tile kernels, not the small-batch products
    assert worst < 1e-5, worst
    print("two-piece vs three-piece products on engine observations: max |action diff| = %.2e" % worst)


def test_weight_hold_packs_once_and_again_when_told(ctx):
    """sgrl_set_hold_weights (include/sgrl_set.h): a rollout loop promises that the parameters do not change between two calls -- the
    forwards in between reuse the packed weights (so an in-place change is NOT seen: the contract), the call after the change makes
    the next forward pack again; without the hold every forward reads the live parameters (the default, tested above)."""
    torch, pol, graphs, keys, z = ctx
    from sgrl_amd.set_hip import HipSetActor
    from sgrl_amd.set_policy import make_policy
    name = "3d_walker_7_full"
    pol2 = make_policy(device="cuda:0").eval()
    pol2.load_state_dict(pol.state_dict())
    act = HipSetActor(pol2)
    act.configure([_gd(torch, graphs[name])], [5])
    act.debug_small_nodes(0)
    obs = torch.from_numpy(z["%s/B5/obs" % name]).cuda()
    base = act.forward_batch(obs).clone()
    act.hold_weights(True)
    held = act.forward_batch(obs).clone()
    assert torch.equal(held, base)
    with torch.no_grad():
        pol2.actor.linear1_m.weight.mul_(1.5)                 # an in-place update the holder has not been told about
    assert torch.equal(act.forward_batch(obs), base)              # still the packed weights: that is the promise
    act.hold_weights(True)                                        # "the parameters just changed"
    changed = act.forward_batch(obs).clone()
    assert not torch.equal(changed, base)
    assert torch.equal(act.forward_batch(obs), changed)
    act.hold_weights(False)                                       # back to the default: live reads on every forward
    with torch.no_grad():
        pol2.actor.linear1_m.weight.div_(1.5)
    assert float((act.forward_batch(obs) - base).abs().max()) < 1e-6


def test_rollout_notices_parameter_changes_under_a_weight_hold():
    """a Rollout that holds its actor's packed weights compares the parameters' (storage, version) fingerprint before
    every policy forward -- load_state_dict, an optimizer step or a manual in-place edit between two rounds is picked up without
    anybody calling weights_changed(); unchanged parameters keep the pack (same actions, bit for bit)."""
    import torch
    from sgrl_amd.rollout import Rollout
    from sgrl_amd.set_policy import make_policy
    pol = make_policy(device="cuda:0").eval()
    ro = Rollout(["3d_walker_7_full", "3d_hopper_3_shin"], 3, policy=pol, seed=1, device="cuda:0", hold_weights=True)
    ro.reset()
    a0 = ro.policy_forward().clone()
    assert torch.equal(ro.policy_forward(), a0)
    with torch.no_grad():
        pol.actor.linear1_m.weight.mul_(1.5)                      # nobody tells the rollout
    a1 = ro.policy_forward().clone()
    assert not torch.equal(a1, a0)
    sd = {k: v.clone() for k, v in pol.state_dict().items()}
    with torch.no_grad():
        pol.actor.linear1_m.weight.div_(1.5)
    assert not torch.equal(ro.policy_forward(), a1)
    pol.load_state_dict(sd)                                       # e.g. a snapshot restore
    assert torch.equal(ro.policy_forward(), a1)


def test_rollout_notices_the_librarys_own_raw_pointer_updates():
    """td3.clip_and_step (table Adam), the table soft update and a GraphedUpdates replay write parameters through raw
    pointers; they bump the tensors' version counters (td3._touched), so a Rollout holding the actor's packed weights re-packs
    without weights_changed()."""
    import torch
    from sgrl_amd import td3
    from sgrl_amd.rollout import Rollout
    from sgrl_amd.set_policy import make_policy
    pol = make_policy(device="cuda:0").eval()
    tgt = make_policy(device="cuda:0").eval()
    ro = Rollout(["3d_walker_7_full", "3d_hopper_3_shin"], 3, policy=pol, seed=1, device="cuda:0", hold_weights=True)
    ro.reset()
    a0 = ro.policy_forward().clone()
    opt = torch.optim.Adam(pol.parameters(), lr=1e-2)
    for g in opt.param_groups:
        g["capturable"] = True                                   # the table path's precondition (device-side step counters)
    for p in pol.parameters():
        p.grad = torch.ones_like(p)
    v0 = [p._version for p in pol.parameters()]
    td3.clip_and_step(opt, 0.0)
    assert all(p._version > v for p, v in zip(pol.parameters(), v0))
    a1 = ro.policy_forward().clone()
    assert not torch.equal(a1, a0)                                # the held pack was rebuilt from the stepped parameters
    v1 = [p._version for p in tgt.parameters()]
    td3.soft_update_network(pol, tgt, 0.5)
    assert all(p._version > v for p, v in zip(tgt.parameters(), v1))


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
def test_the_ab_product_forms_of_the_environment_are_forwards_too(mode):
    """SGRL_SET_GEMM=f32 / bf16x6 (read once per process: a child process each) on both sides of the small-batch threshold, with and
    without a weight hold, before and after a parameter change: the HIP forward equals the PyTorch path.  Round 6 found the f32 mode
    returning garbage from 2 048 nodes on (its generated-operand products read weight words nobody had encoded) -- the arm the
    learning A/B had used as its control."""
    import re
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "diag", "stale_pack_probe.py")], env=dict(os.environ, SGRL_SET_GEMM=mode),
                       capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("SGRL_SET_GEMM=" + mode)]
    assert len(lines) == 3, r.stdout + r.stderr
    for l in lines:
        before, after = (float(x) for x in re.findall(r"(?:before the change|weights_changed\(\)) ([0-9.e+-]+|nan)", l))
        assert before < 2e-6 and after < 2e-6, l
