"""SECritic module surface: state_dict keys and forward values against the reference's own SECritic
(tests/golden/critic_*.{json,npz}, produced by tools/capture_golden_critic.py)."""
import json
import os

import numpy as np
import torch

from oracle.formula import apply_formula_
from sgrl_amd import graph as G, mjcf
from sgrl_amd.set_policy import make_critic


def test_state_dict_and_forward_match_reference(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "critic_state_dict_keys.json")))
    z = np.load(os.path.join(golden_dir, "critic_forward.npz"))
    crit = make_critic().eval()
    sd = crit.state_dict()
    assert sorted(sd) == sorted(keys)
    assert all(list(sd[k].shape) == keys[k] for k in keys)
    assert sum(p.numel() for p in crit.parameters()) == 8761330
    apply_formula_(crit)
    names = sorted({k.split("/")[0] for k in z.files})
    for name in names:
        m = mjcf.load_asset(name)
        crit.change_morphology(G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu")))
        obs, act = torch.from_numpy(z[name + "/obs"]), torch.from_numpy(z[name + "/act"])
        with torch.no_grad():
            q1, q2 = crit(obs, act)
            assert torch.equal(crit.Q1(obs, act), q1)
        assert q1.shape == (4, m.num_limbs)     # per-limb Q values (reference SECritic.py:87-91)
        scale = np.abs(z[name + "/q1"]).max()          # ~3e-3 with the formula weights
        assert np.abs(q1.numpy() - z[name + "/q1"]).max() < 1e-5 * scale
        assert np.abs(q2.numpy() - z[name + "/q2"]).max() < 1e-5 * scale
    # differentiable
    q1, q2 = crit(obs, act.requires_grad_(True))
    (q1.sum() + q2.sum()).backward()
    assert act.grad is not None and float(act.grad.abs().sum()) > 0


def test_twin_forward_walks_both_critics_exactly_like_the_two_separate_passes():
    """set_policy.twin_forward (what SECritic.forward runs on the GPU in grad mode: both TransformerModels at once on stacked
    activations) through its plain-PyTorch operations on the CPU: the same values as critic1(x) / critic2(x), and the same
    gradients, for a chain and a branched morphology.  (The device kernels behind it: tests/test_train_ops_gpu.py.)"""
    import torch
    from sgrl_amd import graph as G, mjcf, set_policy
    from sgrl_amd.set_policy import make_critic
    torch.manual_seed(3)
    crit = make_critic(device="cpu", use_hip=False)
    for name in ("3d_hopper_3_shin", "3d_walker_7_full"):
        m = mjcf.load_asset(name)
        crit.change_morphology(G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu")))
        L = m.num_limbs
        obs, act = torch.randn(6, 41 * L), torch.rand(6, 3 * L) * 2 - 1
        crit.zero_grad()
        q1, q2 = crit(obs, act)                       # CPU: the two networks one after the other
        ((q1 ** 2).mean() + (q2 ** 3).mean()).backward()
        ref = {k: p.grad.clone() for k, p in crit.named_parameters() if p.grad is not None}
        crit.zero_grad()
        q = set_policy.twin_forward(crit.critic1, crit.critic2, crit._input(obs, act), crit.graph, False)
        t1, t2 = q[0].reshape(6, -1), q[1].reshape(6, -1)
        assert torch.allclose(t1, q1, rtol=0, atol=1e-6) and torch.allclose(t2, q2, rtol=0, atol=1e-6)
        ((t1 ** 2).mean() + (t2 ** 3).mean()).backward()
        got = {k: p.grad for k, p in crit.named_parameters() if p.grad is not None}
        assert got.keys() == ref.keys() and len(ref) > 100
        for k in ref:
            assert float((got[k] - ref[k]).abs().max()) <= 1e-5 * (float(ref[k].abs().max()) + 1e-9), (name, k)
