#!/usr/bin/env python3
"""Golden vectors for the TD3 update: EXECUTES the reference's `Agent.update` (reference src/agent.py:117-183, with
`try_update_target_network` :185-187 and `functional.soft_update_network`, common/functional.py:7-10) on formula
weights and scripted batches.  Build container only; writes tests/golden/td3_update.npz (numbers only).

Script: actor / critic = formula weights (oracle/formula.py); the target networks start as 0.97 x the online ones so that
the Polyak step is visible; three updates it = 0, 1, 2 (policy_freq = 2: actor + targets move at it = 0 and 2) -- the
first two on a walker_7 batch, the third after change_morphology to hopper_3 (the reference cycles through morphologies,
trainer.py:245-250).  The clipped-noise draw of agent.py:128-129 is the first consumer of torch's global RNG inside
update(): it is regenerated here from the same seed and stored, so the build's Agent can be fed the identical noise.
Stored per update: losses, per-tensor sums of the clipped gradients and of every parameter of the four networks."""
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
# numpy 2.x removed numpy.lib.arraysetops; agent.py:1 imports `isin` from it and never uses it
_shim = types.ModuleType("numpy.lib.arraysetops")
_shim.isin = np.isin
sys.modules["numpy.lib.arraysetops"] = _shim

import utils as ref_utils  # noqa: E402
from agent import Agent  # noqa: E402
from capture_golden import _args_ns  # noqa: E402
from oracle.formula import apply_formula_, scripted_batch  # noqa: E402

HYPER = dict(lr=1e-4, policy_noise=0.2, noise_clip=0.5, discount=0.99, policy_freq=2, grad_clipping_value=0.1,
             max_action=1.0, target_smoothing_tau=0.005, reward_scale=1.0, batch=6)


def make_args():
    a = _args_ns()
    a.actor_type = a.critic_type = "set"
    a.limb_obs_size, a.limb_action_size, a.msg_dim, a.batch_size = 41, 3, 32, 100
    a.max_action, a.max_children, a.disable_fold, a.td, a.bu = HYPER["max_action"], 3, True, False, False
    for k in ("lr", "policy_noise", "noise_clip", "discount", "policy_freq", "grad_clipping_value"):
        setattr(a, k, HYPER[k])
    a.agent = types.SimpleNamespace(target_smoothing_tau=HYPER["target_smoothing_tau"], reward_scale=HYPER["reward_scale"])
    return a


def tensor_sums(module, grads=False):
    out = []
    for _, p in module.named_parameters():
        t = p.grad if grads else p
        out.append(0.0 if t is None else float(t.detach().double().sum()))
    return np.array(out)


def main(batch=None, out_name="td3_update.npz"):
    """batch: rows per update (default HYPER["batch"] = 6, batches stored in the fixture).  `python tools/capture_golden_update.py 256`
    writes tests/golden/td3_update_b256.npz at the reference's own agent_batch_size (configs/default.py:61): the batches are NOT
    stored (1.8 MB of synthetic rows), only their seeds -- the tests regenerate them with oracle.formula.scripted_batch."""
    if batch is not None:
        HYPER["batch"] = int(batch)
    store_batches = HYPER["batch"] <= 16
    xm = refstub.all_xmls()
    agent = Agent(make_args())
    apply_formula_(agent.actor)
    apply_formula_(agent.critic)
    with torch.no_grad():
        for tgt, src in ((agent.actor_target, agent.actor), (agent.critic_target, agent.critic)):
            for tp, sp in zip(tgt.parameters(), src.parameters()):
                tp.copy_(0.97 * sp)
    res = {"hyper_keys": np.array(sorted(HYPER)), "hyper_vals": np.array([float(HYPER[k]) for k in sorted(HYPER)]),
           "actor_param_names": np.array([n for n, _ in agent.actor.named_parameters()]),
           "critic_param_names": np.array([n for n, _ in agent.critic.named_parameters()])}
    plan = [("3d_walker_7_full", 11), ("3d_walker_7_full", 23), ("3d_hopper_3_shin", 35)]
    agent.models2train()
    for it, (name, seed) in enumerate(plan):
        parents = ref_utils.getGraphStructure(xm[name])
        gd = ref_utils.getGraphDict(parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
        agent.change_morphology(gd)
        L = len(parents)
        b = scripted_batch(L, HYPER["batch"], seed)
        torch.manual_seed(1000 + it)
        noise = torch.zeros(HYPER["batch"], 3 * L).normal_(0, HYPER["policy_noise"]).numpy().copy()
        torch.manual_seed(1000 + it)
        loss = agent.update({k: torch.from_numpy(v) for k, v in b.items()}, it)
        tag = "it%d/" % it
        res[tag + "name"] = np.array(name)
        res[tag + "batch_seed"] = np.array(seed)
        if store_batches:
            for k, v in b.items():
                res[tag + k] = v
        res[tag + "noise"] = noise
        res[tag + "critic_loss"] = np.array(float(loss["loss/critic_loss"]))
        res[tag + "actor_loss"] = np.array(float(loss["loss/actor_loss"]) if "loss/actor_loss" in loss else np.nan)
        res[tag + "train_reward_mean"] = np.array(loss["misc/train_reward_mean"])
        res[tag + "critic_grad_sums"] = tensor_sums(agent.critic, grads=True)
        res[tag + "actor_grad_sums"] = tensor_sums(agent.actor, grads=True)
        for nm, mod in (("actor", agent.actor), ("critic", agent.critic), ("actor_target", agent.actor_target),
                        ("critic_target", agent.critic_target)):
            res[tag + nm + "_param_sums"] = tensor_sums(mod)
        print(it, name, "critic_loss %.6f" % res[tag + "critic_loss"], "actor_loss", res[tag + "actor_loss"])
    # select_action (agent.py:189-198) after the updates, on the last morphology
    agent.models2eval()
    ob = scripted_batch(3, 1, 99)["obs"][0]
    res["select_action/obs"] = ob
    res["select_action/action"] = agent.select_action(ob)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", out_name), **res)
    print(out_name, "written")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(int(sys.argv[1]), "td3_update_b%d.npz" % int(sys.argv[1]))
    else:
        main()
