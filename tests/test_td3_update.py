"""TD3 update (sgrl_amd/td3.py) against tests/golden/td3_update.npz -- numbers produced by executing the reference's own
`Agent.update` (reference src/agent.py:117-183) on formula weights and scripted batches (tools/capture_golden_update.py).

float32 autograd on both sides; the two implementations order their reductions differently (node-major einsum vs the
reference's limb-major bmm), so losses are compared at 1e-4 relative, clipped gradients at 2e-3 of the gradient norm bound
and parameters through their per-tensor sums at 1e-3 (CPU; 2e-2 on the GPU) of the largest step a tensor can take
(numel x lr; measured 5e-5)."""
import os

import numpy as np
import pytest
import torch

from sgrl_amd import graph as G, mjcf
from sgrl_amd.td3 import Agent, default_train_args, soft_update_network

TRAV = ["pre", "inlcrs", "postlcrs"]


def _sums(module, grads=False):
    out = []
    for _, p in module.named_parameters():
        t = p.grad if grads else p
        out.append(0.0 if t is None else float(t.detach().double().sum()))
    return np.array(out)


def run_script(z, device, use_hip):
    """Replays the capture script of tools/capture_golden_update.py on the build's Agent; returns per-update records."""
    from oracle.formula import apply_formula_
    hyper = dict(zip([str(k) for k in z["hyper_keys"]], z["hyper_vals"]))
    args = default_train_args(lr=hyper["lr"], policy_noise=hyper["policy_noise"], noise_clip=hyper["noise_clip"],
                              discount=hyper["discount"], policy_freq=int(hyper["policy_freq"]),
                              grad_clipping_value=hyper["grad_clipping_value"], max_action=hyper["max_action"])
    args.agent.target_smoothing_tau, args.agent.reward_scale = hyper["target_smoothing_tau"], hyper["reward_scale"]
    torch.manual_seed(0)
    agent = Agent(args, device=device, use_hip=use_hip)
    assert [n for n, _ in agent.actor.named_parameters()] == [str(s) for s in z["actor_param_names"]]
    assert [n for n, _ in agent.critic.named_parameters()] == [str(s) for s in z["critic_param_names"]]
    apply_formula_(agent.actor)
    apply_formula_(agent.critic)
    with torch.no_grad():
        for tgt, src in ((agent.actor_target, agent.actor), (agent.critic_target, agent.critic)):
            for tp, sp in zip(tgt.parameters(), src.parameters()):
                tp.copy_(0.97 * sp)
    agent.models2train()
    out = []
    for it in range(3):
        tag = "it%d/" % it
        m = mjcf.load_asset(str(z[tag + "name"]))
        agent.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=torch.device(device)))
        if tag + "obs" in z.files:
            rows = {k: z[tag + k] for k in ("obs", "action", "next_obs", "reward", "done")}
        else:       # the 256-row fixture stores seeds, not rows (tools/capture_golden_update.py)
            from oracle.formula import scripted_batch
            rows = scripted_batch(m.num_limbs, int(hyper["batch"]), int(z[tag + "batch_seed"]))
        batch = {k: torch.from_numpy(rows[k]).to(device) for k in ("obs", "action", "next_obs", "reward", "done")}
        before = {nm: _sums(getattr(agent, nm)) for nm in ("actor", "critic", "actor_target", "critic_target")}
        loss = agent.update(batch, it, noise=torch.from_numpy(z[tag + "noise"]).to(device))
        rec = {"critic_loss": float(loss["loss/critic_loss"]),
               "actor_loss": float(loss["loss/actor_loss"]) if "loss/actor_loss" in loss else float("nan"),
               "train_reward_mean": loss["misc/train_reward_mean"],
               "critic_grad_sums": _sums(agent.critic, grads=True), "actor_grad_sums": _sums(agent.actor, grads=True),
               "before": before}
        for nm in ("actor", "critic", "actor_target", "critic_target"):
            rec[nm + "_param_sums"] = _sums(getattr(agent, nm))
        out.append(rec)
    agent.models2eval()
    out.append(agent.select_action(z["select_action/obs"]))
    return agent, hyper, out


def check_against_golden(z, agent, hyper, out, loss_rtol=1e-4, grad_tol=2e-3, step_tol=1e-3):
    numel = {nm: np.array([p.numel() for p in getattr(agent, nm).parameters()]) for nm in ("actor", "critic")}
    numel["actor_target"], numel["critic_target"] = numel["actor"], numel["critic"]
    for it in range(3):
        tag, rec = "it%d/" % it, out[it]
        assert abs(rec["critic_loss"] - float(z[tag + "critic_loss"])) < loss_rtol * abs(float(z[tag + "critic_loss"])), it
        ref_al = float(z[tag + "actor_loss"])
        assert np.isnan(ref_al) == np.isnan(rec["actor_loss"]), "policy_freq: actor updated at the wrong iterations"
        if not np.isnan(ref_al):
            assert abs(rec["actor_loss"] - ref_al) < loss_rtol * max(abs(ref_al), 1e-2), it
        assert abs(rec["train_reward_mean"] - float(z[tag + "train_reward_mean"])) < 1e-6
        # gradients after clip_grad_norm_: total norm <= grad_clipping_value, so per-tensor sums are compared against it
        clip = hyper["grad_clipping_value"]
        for nm in ("critic", "actor"):
            d = np.abs(rec[nm + "_grad_sums"] - z[tag + nm + "_grad_sums"])
            assert d.max() < grad_tol * clip * np.sqrt(numel[nm].max()), (it, nm, d.max())
        # parameters: per-tensor sums; an Adam step moves an element by at most ~lr
        for nm in ("actor", "critic", "actor_target", "critic_target"):
            d = np.abs(rec[nm + "_param_sums"] - z[tag + nm + "_param_sums"])
            bound = step_tol * numel[nm] * hyper["lr"] + 1e-5 * np.abs(z[tag + nm + "_param_sums"]) + 1e-6
            assert (d < bound).all(), (it, nm, int(np.argmax(d / bound)), float((d / bound).max()))
        # ... and the step itself went the reference's way (not just "stayed close to the start")
        for nm in ("actor", "critic"):
            moved_ref = z[tag + nm + "_param_sums"] - (z["it%d/" % (it - 1) + nm + "_param_sums"] if it else rec["before"][nm])
            moved = rec[nm + "_param_sums"] - rec["before"][nm]
            big = np.abs(moved_ref) > 0.05 * numel[nm] * hyper["lr"]
            if nm == "actor" and np.isnan(ref_al):
                assert np.abs(moved).max() == 0.0          # no actor step on odd iterations
                continue
            assert big.sum() > 5
            assert (np.sign(moved[big]) == np.sign(moved_ref[big])).all(), (it, nm)
            np.testing.assert_allclose(moved[big], moved_ref[big], rtol=0.05)
    np.testing.assert_allclose(out[3], z["select_action/action"], atol=5e-5)


@pytest.fixture(scope="module", params=["td3_update.npz", "td3_update_b256.npz"], ids=["batch6", "batch256"])
def golden(golden_dir, request):
    """batch6: rows stored in the fixture; batch256: the reference's own agent_batch_size (configs/default.py:61,
    trainer.py:289-291), rows regenerated from their seeds."""
    return np.load(os.path.join(golden_dir, request.param))


def test_update_matches_the_reference_on_cpu(golden):
    agent, hyper, out = run_script(golden, "cpu", use_hip=False)
    check_against_golden(golden, agent, hyper, out)
    # targets moved by exactly tau towards the online networks at it = 0 and 2, not at it = 1
    tau = hyper["target_smoothing_tau"]
    for nm in ("actor", "critic"):
        b, a = out[1]["before"][nm + "_target"], out[1][nm + "_target_param_sums"]
        assert np.array_equal(a, b)
        b, a, online = out[2]["before"][nm + "_target"], out[2][nm + "_target_param_sums"], out[2][nm + "_param_sums"]
        np.testing.assert_allclose(a, tau * online + (1 - tau) * b, rtol=1e-5, atol=1e-6)


def test_agent_surface_and_state_dict_keys():
    agent = Agent(default_train_args())
    keys = list(agent.state_dict().keys())
    assert any(k.startswith("actor.actor.transformer_encoder.layers.0.self_attn.q_proj") for k in keys)
    assert any(k.startswith("actor_target.actor.") for k in keys)
    assert any(k.startswith("critic.critic1.") for k in keys) and any(k.startswith("critic_target.critic2.") for k in keys)
    for a, b in ((agent.actor, agent.actor_target), (agent.critic, agent.critic_target)):
        for p, q in zip(a.parameters(), b.parameters()):
            assert torch.equal(p, q)                   # tau = 1.0 sync at construction (agent.py:104-105)
    assert not agent.actor.training
    agent.models2train()
    assert agent.critic_target.training
    with pytest.raises(NotImplementedError):
        Agent(default_train_args(actor_type="smp"))
    soft_update_network(agent.actor, agent.actor_target, 0.5)


@pytest.mark.gpu
def test_update_matches_the_reference_on_the_gpu_with_hip_targets(golden):
    """Same script on cuda:0: the no-grad halves (target action, twin target Q, select_action) run on the HIP kernels and
    must follow the in-place optimizer and Polyak updates between iterations."""
    agent, hyper, out = run_script(golden, "cuda:0", use_hip=True)
    from sgrl_amd import set_policy, train_ops
    assert agent.actor_target._hip is not None and agent.actor._hip is not None
    # the twin target critics: one pass of the training kernels for both networks (set_policy.TWIN_TARGETS), or the rollout kernels
    assert (set_policy.TWIN_TARGETS and set_policy.TWIN_CRITICS and train_ops.ENABLED) or agent.critic_target._hip is not None
    check_against_golden(golden, agent, hyper, out, loss_rtol=3e-4, step_tol=2e-2)


def test_swat_agent_updates():
    """actor_type = critic_type = 'swat' (reference agent.py:28-29,66-67): the same TD3 update over the SWAT modules."""
    from oracle.formula import synth_obs
    torch.manual_seed(1)
    agent = Agent(default_train_args(actor_type="swat", critic_type="swat"))
    m = mjcf.load_asset("3d_walker_5_foot")
    agent.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=torch.device("cpu")))
    B, L = 6, m.num_limbs
    batch = {"obs": torch.from_numpy(synth_obs(L, B, 1).astype(np.float32)), "next_obs": torch.from_numpy(synth_obs(L, B, 2).astype(np.float32)),
             "action": torch.rand(B, 3 * L) * 2 - 1, "reward": torch.randn(B, 1), "done": torch.zeros(B, 1)}
    before = [p.detach().clone() for p in agent.actor.parameters()]
    agent.models2train()
    out = agent.update(batch, 0)
    assert np.isfinite(float(out["loss/critic_loss"])) and "loss/actor_loss" in out
    assert any(float((p - q).abs().max()) > 0 for p, q in zip(agent.actor.parameters(), before))
    assert agent.select_action(batch["obs"][0].numpy()).shape == (1, 3 * L)


def test_smp_agent_updates():
    """actor_type = critic_type = 'smp' (reference agent.py:30-31,67-68), both-way message passing: the critic returns the
    limb-summed twin values [B, 1] and the same TD3 update runs over them."""
    from oracle.formula import synth_obs
    torch.manual_seed(2)
    agent = Agent(default_train_args(actor_type="smp", critic_type="smp", td=True, bu=True, max_children=5))
    m = mjcf.load_asset("3d_humanoid_9_full")
    agent.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=torch.device("cpu")))
    B, L = 6, m.num_limbs
    batch = {"obs": torch.from_numpy(synth_obs(L, B, 1).astype(np.float32)), "next_obs": torch.from_numpy(synth_obs(L, B, 2).astype(np.float32)),
             "action": torch.rand(B, 3 * L) * 2 - 1, "reward": torch.randn(B, 1), "done": torch.zeros(B, 1)}
    before = [p.detach().clone() for p in agent.actor.parameters()]
    n_params = len(list(agent.actor.parameters()))
    agent.models2train()
    out = agent.update(batch, 0)
    assert np.isfinite(float(out["loss/critic_loss"])) and "loss/actor_loss" in out
    assert len(list(agent.actor.parameters())) == n_params          # the per-limb module list shares ONE module's parameters
    assert any(float((p - q).abs().max()) > 0 for p, q in zip(agent.actor.parameters(), before))
    assert agent.select_action(batch["obs"][0].numpy()).shape == (1, 3 * L)


def test_adam_step_follows_torch_adam():
    """td3.adam_step (shared bias corrections, a dozen multi-tensor launches) against torch.optim.Adam, on the optimizer's own
    state: same parameters to rounding, same step counters, state_dict layout untouched."""
    from sgrl_amd.td3 import adam_step
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 2))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    o1, o2 = torch.optim.Adam(ps, lr=1e-3), torch.optim.Adam(qs, lr=1e-3)
    for g in o2.param_groups:
        g["capturable"] = True
    for it in range(6):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), (g.clone() if not (it == 0 and q is qs[2]) else None)     # a parameter without a gradient is skipped
            if q.grad is None:
                p.grad = None
        o1.step()
        adam_step(o2)
    assert max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(ps, qs)) < 5e-7
    assert [float(o2.state[q]["step"]) for q in qs] == [6.0, 6.0, 5.0]
    assert set(o2.state_dict()["state"][0].keys()) == set(o1.state_dict()["state"][0].keys())
    o3 = torch.optim.Adam([torch.nn.Parameter(torch.randn(3))], lr=1e-3)     # not capturable: plain opt.step()
    o3.param_groups[0]["params"][0].grad = torch.ones(3)
    adam_step(o3)
    assert float(o3.state[o3.param_groups[0]["params"][0]]["step"]) == 1.0


def test_skipping_the_unused_critic_gradients_changes_nothing_that_is_used():
    """skip_unused_critic_grads: the actor pass does not compute the critic's parameter gradients (nobody steps on them).
    Parameters of all four networks and the losses are identical to the default path; only the stale critic .grad differs."""
    import copy
    from oracle.formula import synth_obs
    torch.manual_seed(4)
    a1 = Agent(default_train_args())
    a2 = copy.deepcopy(a1)
    a2.actor_optimizer = torch.optim.Adam(a2.actor.parameters(), lr=a2.args.lr)
    a2.critic_optimizer = torch.optim.Adam(a2.critic.parameters(), lr=a2.args.lr)
    m = mjcf.load_asset("3d_walker_4_right_knee_left_foot")
    gd = G.getGraphDict(m.parents, TRAV, [], device=torch.device("cpu"))
    B, L = 5, m.num_limbs
    for ag in (a1, a2):
        ag.change_morphology(gd)
        ag.models2train()
    for it in range(3):
        batch = {"obs": torch.from_numpy(synth_obs(L, B, 10 + it).astype(np.float32)), "next_obs": torch.from_numpy(synth_obs(L, B, 20 + it).astype(np.float32)),
                 "action": torch.rand(B, 3 * L) * 2 - 1, "reward": torch.randn(B, 1), "done": torch.zeros(B, 1)}
        noise = torch.randn(B, 3 * L) * 0.2
        o1 = a1.update(batch, it, noise=noise.clone())
        o2 = a2.update(batch, it, noise=noise.clone(), skip_unused_critic_grads=True)
        # (not bit-equal: with frozen weights the linear layers take a different GEMM entry point of the BLAS)
        assert abs(float(o1["loss/critic_loss"]) - float(o2["loss/critic_loss"])) < 1e-5 * abs(float(o1["loss/critic_loss"]))
        if "loss/actor_loss" in o1:
            assert abs(float(o1["loss/actor_loss"]) - float(o2["loss/actor_loss"])) < 1e-5 * max(abs(float(o1["loss/actor_loss"])), 1e-2)
    for nm in ("actor", "critic", "actor_target", "critic_target"):
        for p, q in zip(getattr(a1, nm).parameters(), getattr(a2, nm).parameters()):
            assert float((p - q).abs().max()) < 3e-6, nm          # three Adam steps of 1e-4 each: agreement to a few % of one step
    assert all(p.requires_grad for p in a2.critic.parameters())


def test_captured_optimizer_tables_are_released_with_their_graphs():
    """td3._table keeps an entry a hipGraph points at for as long as an owner (a captured graph) lives; GraphedUpdates names the
    owner while it captures and releases it when it drops the graph: bookkeeping only, checked without a device."""
    from sgrl_amd import td3
    saved = dict(td3._tables)
    try:
        td3._tables.clear()
        a, b = ("g", 0, 0), ("g", 0, 1)
        td3._tables["shared"] = {"owners": {a, b}, "captured": True}
        td3._tables["only_a"] = {"owners": {a}, "captured": True}
        td3._tables["eager"] = {"owners": set(), "captured": False}
        td3.release_tables(a)
        assert td3._tables["shared"]["captured"] is True and td3._tables["shared"]["owners"] == {b}
        assert td3._tables["only_a"]["captured"] is False and not td3._tables["only_a"]["owners"]
        assert td3._tables["eager"]["captured"] is False
        td3.release_tables(b)
        assert td3._tables["shared"]["captured"] is False
    finally:
        td3._tables.clear()
        td3._tables.update(saved)
