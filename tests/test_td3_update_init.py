"""TD3 update in the DEFAULT-INITIALISATION regime against tests/golden/td3_update_default_init.npz -- numbers produced by executing
the reference's own `Agent.update` (reference src/agent.py:117-183) on default-like weights regenerated from seeds
(tools/capture_golden_update_init.py; oracle/formula.py `apply_default_like_`, rule table checked there against the reference's
freshly constructed networks).

This is the regime that decides whether BASELINE config 5 takes off: the critic does not depend on the action yet, the actor's
gradient through `critic.Q1` is ~1e-6 in total norm, 1e-16 .. 1e-6 per tensor, and most of its elements sit below Adam's eps.  The
formula-weight fixtures (test_td3_update.py) compare per-tensor SUMS against bounds scaled by the clip value and would pass a
gradient of this size that was wrong by 10x; here every tensor's gradient -- captured where the reference clips, i.e. raw -- is
compared through its L2 norm and eight sampled elements with a tolerance RELATIVE TO THE TENSOR'S OWN NORM: 1e-3 (2e-2 for tensors whose
norm is below 1e-6 of the network's largest), plus ten times
the discrepancy the reference itself shows between its float32 and float64 runs of the same script (two tensors, the relative-
position encoder's biases, have a structurally zero gradient -- softmax shift invariance -- and hold rounding noise only)."""
import os
import warnings

import numpy as np
import pytest
import torch

from sgrl_amd import graph as G, mjcf
from sgrl_amd import td3
from sgrl_amd.td3 import Agent, default_train_args

TRAV = ["pre", "inlcrs", "postlcrs"]
NS = 8


def _sample_idx(numel):
    return np.unique(np.linspace(0, numel - 1, NS).astype(np.int64)) if numel >= NS else np.arange(numel)


def _grad_record(module):
    norms, samples = [], []
    for _, p in module.named_parameters():
        if p.grad is None:
            norms.append(np.nan)
            samples.append(np.full(NS, np.nan))
            continue
        g = p.grad.detach().double().reshape(-1).cpu()
        norms.append(float(g.norm()))
        s = g[torch.from_numpy(_sample_idx(g.numel()))].numpy()
        samples.append(np.pad(s, (0, NS - s.size), constant_values=np.nan))
    return np.array(norms), np.stack(samples)


def run_script(z, device, use_hip, trainer_path=False):
    """The capture script on the build's Agent; the raw gradients are grabbed where the build clips (td3.clip_and_step).
    trainer_path: the keyword arguments GraphedUpdates / DeviceTrainer run an update with (device-side statistics, weight gradients
    collected and issued in groups, the critic's own gradients of the actor pass -- which nothing reads -- not computed)."""
    from oracle.formula import apply_default_like_, scripted_batch
    hyper = dict(zip([str(k) for k in z["hyper_keys"]], z["hyper_vals"]))
    args = default_train_args(lr=hyper["lr"], policy_noise=hyper["policy_noise"], noise_clip=hyper["noise_clip"],
                              discount=hyper["discount"], policy_freq=int(hyper["policy_freq"]),
                              grad_clipping_value=hyper["grad_clipping_value"], max_action=hyper["max_action"])
    args.agent.target_smoothing_tau, args.agent.reward_scale = hyper["target_smoothing_tau"], hyper["reward_scale"]
    torch.manual_seed(0)
    agent = Agent(args, device=device, use_hip=use_hip)
    assert [n for n, _ in agent.actor.named_parameters()] == [str(s) for s in z["actor_param_names"]]
    assert [n for n, _ in agent.critic.named_parameters()] == [str(s) for s in z["critic_param_names"]]
    for mod in (agent.actor, agent.critic):
        apply_default_like_(mod, int(z["seed"]))
    with torch.no_grad():
        for tgt, src in ((agent.actor_target, agent.actor), (agent.critic_target, agent.critic)):
            for tp, sp in zip(tgt.parameters(), src.parameters()):
                tp.copy_(sp)
    if trainer_path == "table_optimizer":       # what GraphedUpdates switches on: device-side step counters -> the table optimizer
        for opt in (agent.actor_optimizer, agent.critic_optimizer):
            for g in opt.param_groups:
                g["capturable"] = True
    agent.models2train()
    grabbed = {}
    real = td3.clip_and_step

    def spy(opt, max_norm):
        which = "critic" if opt is agent.critic_optimizer else "actor"
        grabbed[which] = _grad_record(getattr(agent, which))
        return real(opt, max_norm)

    td3.clip_and_step = spy
    out = []
    try:
        for it in range(3):
            tag = "it%d/" % it
            m = mjcf.load_asset(str(z["plan_names"][it]))
            agent.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=torch.device(device)))
            rows = scripted_batch(m.num_limbs, int(z["batch"]), int(z["plan_seeds"][it]))
            batch = {k: torch.from_numpy(rows[k]).to(device) for k in ("obs", "action", "next_obs", "reward", "done")}
            before = {nm: [p.detach().double().clone() for p in getattr(agent, nm).parameters()] for nm in ("actor", "critic")}
            grabbed.clear()
            kw = dict(lazy_stats=True, skip_unused_critic_grads=True) if trainer_path else {}
            loss = agent.update(batch, it, noise=torch.from_numpy(z[tag + "noise"]).to(device), **kw)
            rec = {"critic_loss": float(loss["loss/critic_loss"]),
                   "actor_loss": float(loss["loss/actor_loss"]) if "loss/actor_loss" in loss else float("nan")}
            for nm in ("critic", "actor"):
                if nm in grabbed:
                    rec[nm + "_grad_norms"], rec[nm + "_grad_samples"] = grabbed[nm]
                rec[nm + "_step_norms"] = np.array([float((p.detach().double() - q).norm()) for p, q in
                                                    zip(getattr(agent, nm).parameters(), before[nm])])
                rec[nm + "_param_absmax"] = np.array([float(q.abs().max()) for q in before[nm]])
            out.append(rec)
    finally:
        td3.clip_and_step = real
    return agent, hyper, out


def check(z, out, rel=1e-3, loss_rtol=1e-4, report=None):
    worst = {}
    warnings.filterwarnings("ignore", "All-NaN slice encountered")
    for it in range(3):
        tag, rec = "it%d/" % it, out[it]
        assert abs(rec["critic_loss"] - float(z[tag + "critic_loss"])) < loss_rtol * abs(float(z[tag + "critic_loss"])), it
        ref_al = float(z[tag + "actor_loss"])
        assert np.isnan(ref_al) == np.isnan(rec["actor_loss"])
        if not np.isnan(ref_al):
            # the actor loss is a mean of Q values ~1e-4 .. 1e-2 whose float32 evaluation the reference itself holds to ~1e-6 absolute
            assert abs(rec["actor_loss"] - ref_al) < 1e-4 * abs(ref_al) + 2e-6, (it, rec["actor_loss"], ref_al)
        for nm in ("critic", "actor"):
            k = tag + nm + "_grad_norms"
            if k not in z.files:
                assert nm + "_grad_norms" not in rec, "policy_freq: actor stepped at the wrong iteration"
                continue
            n32, n64 = z[k], z[k + "_f64"]
            s32, s64 = z[tag + nm + "_grad_samples"], z[tag + nm + "_grad_samples_f64"]
            got_n, got_s = rec[nm + "_grad_norms"], rec[nm + "_grad_samples"]
            assert np.array_equal(np.isnan(got_n), np.isnan(n64)), (it, nm, "a different set of parameters received gradients")
            live = ~np.isnan(n64)
            # what the reference's own float32 run leaves unresolved, per tensor
            ref_noise_n = np.abs(n32 - n64)
            ref_noise_s = np.nanmax(np.abs(s32 - s64), axis=1)
            # (tensors whose gradient is ten orders of magnitude below the network's largest -- attention biases that softmax all but
            # cancels -- are sums of cancelling terms: 2e-2 there, measured 1.1e-3 on the device)
            rel_t = np.where(n64 >= 1e-6 * np.nanmax(n64), rel, 20.0 * rel)
            tol_n = rel_t * n64 + 10.0 * ref_noise_n
            tol_s = rel_t * n64 + 10.0 * ref_noise_s
            dn = np.abs(got_n - n64)
            ds = np.nanmax(np.abs(got_s - s64), axis=1)
            # tensors whose gradient is structurally zero hold rounding noise on both sides: theirs need only be as small
            noise_only = live & (n64 < 1e-6 * np.maximum(n32, 1e-300))
            assert noise_only.sum() <= 2
            assert (got_n[noise_only] <= 1e3 * n32[noise_only] + 1e-30).all(), (it, nm, got_n[noise_only], n32[noise_only])
            cmp = live & ~noise_only
            bad = cmp & ((dn > tol_n) | (ds > tol_s))
            names = z[nm + "_param_names"]
            assert not bad.any(), (it, nm, [(str(names[i]), got_n[i], n64[i], dn[i] / max(n64[i], 1e-300), ds[i] / max(n64[i], 1e-300))
                                            for i in np.nonzero(bad)[0][:6]])
            worst[(it, nm, "norm")] = float(np.max(dn[cmp] / n64[cmp]))
            worst[(it, nm, "sample")] = float(np.max(ds[cmp] / n64[cmp]))
            worst[(it, nm, "ref_f32_norm")] = float(np.max(ref_noise_n[cmp] / n64[cmp]))
            # the total gradient norm (what the clip acts on)
            tot, tot64 = np.sqrt(np.nansum(got_n ** 2)), np.sqrt(np.nansum(n64 ** 2))
            assert abs(tot - tot64) < 3e-4 * tot64, (it, nm, tot, tot64)      # measured: CPU 1e-6, device 5e-5
            worst[(it, nm, "total")] = abs(tot - tot64) / tot64
        # the steps: clip + Adam (eps 1e-8: elements with |g| << eps move by lr g / eps, often below the parameter's float32 spacing)
        for nm in ("critic", "actor"):
            st, st64 = rec[nm + "_step_norms"], z[tag + nm + "_step_norms_f64"]
            numel = z[nm + "_numel"]
            ulp_floor = np.sqrt(numel) * rec[nm + "_param_absmax"] * 1.2e-7
            assert (np.abs(st - st64) <= 2e-3 * st64 + ulp_floor).all(), (it, nm, int(np.argmax(np.abs(st - st64) - 2e-3 * st64 - ulp_floor)))
            if tag + nm + "_grad_norms" not in z.files:
                assert st.max() == 0.0
    if report is not None:
        report.update(worst)
    return worst


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "td3_update_default_init.npz"))


def test_default_init_regime_is_what_the_fixture_says(golden):
    """The facts of the regime, from the reference's own float64 run: a critic gradient of order 1, an actor gradient five
    orders of magnitude below it, most actor tensors below Adam's eps per element."""
    z = golden
    gc = np.sqrt(np.nansum(z["it0/critic_grad_norms_f64"] ** 2))
    ga = np.sqrt(np.nansum(z["it0/actor_grad_norms_f64"] ** 2))
    assert 0.1 < gc < 2.0 and 1e-7 < ga < 1e-4
    per_elem = z["it0/actor_grad_norms_f64"] / np.sqrt(z["actor_numel"])
    live = ~np.isnan(per_elem)
    assert (per_elem[live] < 1e-8).mean() > 0.8          # below Adam's eps: the step is lr g / eps, not lr sign(g)


def test_update_matches_the_reference_at_default_init_on_cpu(golden):
    agent, hyper, out = run_script(golden, "cpu", use_hip=False)
    worst = check(golden, out)
    print({"%d/%s/%s" % k: "%.2e" % v for k, v in worst.items()})


@pytest.mark.gpu
@pytest.mark.parametrize("trainer_path", [False, True, "table_optimizer"], ids=["plain", "trainer_path", "trainer_path_table_optimizer"])
def test_update_matches_the_reference_at_default_init_on_the_device(golden, trainer_path):
    """The shipped arithmetic: own exact-f32 training products, HIP target networks, table optimizer."""
    agent, hyper, out = run_script(golden, "cuda:0", use_hip=True, trainer_path=trainer_path)
    worst = check(golden, out)
    print({"%d/%s/%s" % k: "%.2e" % v for k, v in worst.items()})
