#!/usr/bin/env python3
"""Golden vectors for the TD3 update IN THE DEFAULT-INITIALISATION REGIME (VERDICT r5 "next round" item 1a): EXECUTES the
reference's `Agent.update` (reference src/agent.py:117-183) for three iterations on networks whose weights follow torch's default
initialisation rules, regenerated from seeds (oracle/formula.py `apply_default_like_`), targets = copies of the online networks
(agent.py:100-101), scripted 256-row batches (the reference's own agent_batch_size).  Build container only; writes
tests/golden/td3_update_default_init.npz (numbers only).

Why: at default init the actor's gradient through a critic that does not yet depend on the action is ~1e-11 per element; the
formula-weight fixtures (td3_update*.npz) compare per-tensor SUMS against bounds scaled by the clip value and cannot see a
gradient of that size.  Stored per update, captured at the moment the reference clips (so BEFORE clipping, and before the actor
pass adds its own gradients to the critic's): per-tensor L2 norms and eight sampled elements of every gradient, the L2 norm of
every parameter's step, the losses -- float32 as the reference runs, and the same script in float64 (the reference's modules
`.double()`, same noise) as the yardstick for what float32 can resolve.

The script first CHECKS the rule table against three freshly constructed reference agents: constant tensors equal, no element
beyond its bound, sample spread within 5 sigma of the rule's."""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import numpy as np
import torch

import refstub
refstub.install()
_shim = types.ModuleType("numpy.lib.arraysetops")
_shim.isin = np.isin
sys.modules["numpy.lib.arraysetops"] = _shim

import utils as ref_utils  # noqa: E402
from agent import Agent  # noqa: E402
from capture_golden_update import make_args, HYPER  # noqa: E402
from oracle.formula import apply_default_like_, default_like_rule, scripted_batch  # noqa: E402

SEED = 6
BATCH = 256
PLAN = [("3d_walker_7_full", 11), ("3d_walker_7_full", 23), ("3d_hopper_3_shin", 35)]
NS = 8          # sampled elements per tensor


def sample_idx(numel):
    return np.unique(np.linspace(0, numel - 1, NS).astype(np.int64)) if numel >= NS else np.arange(numel)


def grad_record(module):
    norms, samples = [], []
    for _, p in module.named_parameters():
        g = p.grad
        if g is None:
            norms.append(np.nan)
            samples.append(np.full(NS, np.nan))
            continue
        g = g.detach().double().reshape(-1)
        norms.append(float(g.norm()))
        s = g[torch.from_numpy(sample_idx(g.numel()))].numpy()
        samples.append(np.pad(s, (0, NS - s.size), constant_values=np.nan))
    return np.array(norms), np.stack(samples)


def check_rules():
    """The rule table against the reference's own default initialisation, three seeds."""
    worst = 0.0
    for s in (1, 2, 3):
        torch.manual_seed(s)
        a = Agent(make_args())
        for mod in (a.actor, a.critic):
            sd = dict(mod.named_parameters())
            for name, p in sd.items():
                kind, val = default_like_rule(name, tuple(p.shape))
                x = p.detach().double().reshape(-1)
                if kind == "c":
                    assert bool((x == val).all()), (name, "constant rule")
                    continue
                if kind == "u" and val is None:
                    w = sd.get(name[:-len("bias")] + "weight") if name.endswith(".bias") else None
                    val = 1.0 / np.sqrt(p.shape[1] if p.dim() == 2 else w.shape[1])
                n = x.numel()
                if kind == "u":
                    assert float(x.abs().max()) <= val * (1 + 1e-6), (name, float(x.abs().max()), val)
                    sd_rule, kurt = val / np.sqrt(3.0), 1.8
                else:
                    sd_rule, kurt = val, 3.0
                # sample variance of n draws: relative standard error sqrt((kurt - 1) / n)
                z = abs(float(x.var(unbiased=False)) / sd_rule ** 2 - 1.0) / np.sqrt((kurt - 1.0) / n)
                worst = max(worst, z)
                assert z < 5.0, (name, z)
            # the three layers are clones
            for name, p in sd.items():
                if ".layers.0." in name:
                    assert torch.equal(p, sd[name.replace(".layers.0.", ".layers.1.")]) and torch.equal(p, sd[name.replace(".layers.0.", ".layers.2.")])
    print("rule table holds on three default-initialised reference agents (largest spread deviation %.2f sigma)" % worst)


def run(dtype):
    torch.set_default_dtype(dtype)
    xm = refstub.all_xmls()
    torch.manual_seed(0)
    agent = Agent(make_args())
    for mod in (agent.actor, agent.critic):
        apply_default_like_(mod, SEED)
    with torch.no_grad():       # agent.py:100-101: the targets start as copies
        for tgt, src in ((agent.actor_target, agent.actor), (agent.critic_target, agent.critic)):
            for tp, sp in zip(tgt.parameters(), src.parameters()):
                tp.copy_(sp)
    agent.models2train()
    res = {}
    grabbed = {}
    real_clip = torch.nn.utils.clip_grad_norm_

    def clip_spy(params, max_norm, *a, **k):
        params = list(params)
        which = "critic" if params[0] is next(agent.critic.parameters()) else "actor"
        grabbed[which] = grad_record(agent.critic if which == "critic" else agent.actor)
        return real_clip(params, max_norm, *a, **k)

    real_normal = torch.Tensor.normal_

    def normal32(self, mean=0, std=1, *, generator=None):      # the float64 run consumes the float32 run's noise
        if self.dtype == torch.float64:
            tmp = torch.empty(self.shape, dtype=torch.float32)
            real_normal(tmp, mean, std)
            return self.copy_(tmp)
        return real_normal(self, mean, std)

    torch.nn.utils.clip_grad_norm_ = clip_spy
    torch.Tensor.normal_ = normal32
    try:
        for it, (name, seed) in enumerate(PLAN):
            parents = ref_utils.getGraphStructure(xm[name])
            gd = ref_utils.getGraphDict(parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
            agent.change_morphology(gd)
            L = len(parents)
            b = scripted_batch(L, BATCH, seed)
            torch.manual_seed(1000 + it)
            noise = torch.zeros(BATCH, 3 * L, dtype=torch.float32).normal_(0, HYPER["policy_noise"]).numpy().copy()
            before = {nm: [p.detach().double().clone() for p in getattr(agent, nm).parameters()] for nm in ("actor", "critic")}
            grabbed.clear()
            torch.manual_seed(1000 + it)
            loss = agent.update({k: torch.from_numpy(v).to(dtype) for k, v in b.items()}, it)
            tag = "it%d/" % it
            res[tag + "noise"] = noise
            res[tag + "critic_loss"] = np.array(float(loss["loss/critic_loss"]))
            res[tag + "actor_loss"] = np.array(float(loss["loss/actor_loss"]) if "loss/actor_loss" in loss else np.nan)
            for nm in ("critic", "actor"):
                if nm in grabbed:
                    res[tag + nm + "_grad_norms"], res[tag + nm + "_grad_samples"] = grabbed[nm]
                res[tag + nm + "_step_norms"] = np.array([float((p.detach().double() - q).norm())
                                                          for p, q in zip(getattr(agent, nm).parameters(), before[nm])])
            print(dtype, it, name, "critic_loss %.6f" % res[tag + "critic_loss"], "actor_loss", res[tag + "actor_loss"],
                  "|g_critic| %.3e" % np.sqrt(np.nansum(res[tag + "critic_grad_norms"] ** 2)),
                  "|g_actor| %.3e" % (np.sqrt(np.nansum(res[tag + "actor_grad_norms"] ** 2)) if tag + "actor_grad_norms" in res else np.nan))
    finally:
        torch.nn.utils.clip_grad_norm_ = real_clip
        torch.Tensor.normal_ = real_normal
        torch.set_default_dtype(torch.float32)
    names = {"actor_param_names": np.array([n for n, _ in agent.actor.named_parameters()]),
             "critic_param_names": np.array([n for n, _ in agent.critic.named_parameters()]),
             "actor_numel": np.array([p.numel() for p in agent.actor.parameters()]),
             "critic_numel": np.array([p.numel() for p in agent.critic.parameters()])}
    return res, names


def main():
    check_rules()
    r32, names = run(torch.float32)
    r64, _ = run(torch.float64)
    out = dict(names)
    out["seed"], out["batch"] = np.array(SEED), np.array(BATCH)
    out["plan_names"], out["plan_seeds"] = np.array([p[0] for p in PLAN]), np.array([p[1] for p in PLAN])
    out["hyper_keys"] = np.array(sorted(HYPER))
    out["hyper_vals"] = np.array([float(BATCH if k == "batch" else HYPER[k]) for k in sorted(HYPER)])
    for k, v in r32.items():
        out[k] = v
    for k, v in r64.items():
        if not k.endswith("noise"):
            out[k + "_f64"] = v
    # what float32 resolves on the reference's own arithmetic: f32 vs f64, per tensor, relative to the tensor's norm
    for it in range(3):
        for nm in ("critic", "actor"):
            k = "it%d/%s_grad_norms" % (it, nm)
            if k in r32:
                rel = np.abs(r32[k] - r64[k]) / np.maximum(r64[k], 1e-300)
                print("it %d %s: reference f32 vs f64 gradient norms: median rel %.2e, max rel %.2e (tensor %d)" % (
                    it, nm, np.nanmedian(rel), np.nanmax(rel), int(np.nanargmax(rel))))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "td3_update_default_init.npz"), **out)
    print("td3_update_default_init.npz written")


if __name__ == "__main__":
    main()
