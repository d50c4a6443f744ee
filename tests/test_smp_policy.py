"""SMP baseline modules (sgrl_amd/smp_policy.py) against fixtures produced by executing the reference's ActorGraphPolicy /
CriticGraphPolicy (tools/capture_golden_smp.py): state_dict keys and shapes identical, forward within f32 rounding."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.formula import apply_formula_
from sgrl_amd import mjcf
from sgrl_amd.smp_policy import ActorGraphPolicy, CriticGraphPolicy


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "smp_state_dict_keys.json")) as f:
        keys = json.load(f)
    return keys, np.load(os.path.join(golden_dir, "smp_forward.npz"))


@pytest.mark.parametrize("td,bu", [(True, True), (True, False)])
def test_smp_actor_and_critic_match_the_reference(gold, td, bu):
    keys, z = gold
    mode = "td%d_bu%d" % (td, bu)
    mc = keys["max_children"]
    pol = ActorGraphPolicy(41, 3, 32, 1, 1.0, mc, True, td, bu, None).eval()
    crit = CriticGraphPolicy(41, 3, 32, 1, mc, True, td, bu, None).eval()
    names = sorted({k.split("/")[1] for k in z.files if k.startswith(mode + "/")})
    assert len(names) == 5
    for name in names:
        m = mjcf.load_asset(name)
        gd = {"parents": list(m.parents)}
        pol.change_morphology(gd)
        crit.change_morphology(gd)
        if name == "3d_walker_7_full":      # the shared module is listed once per limb, as in the reference
            assert {k: list(v.shape) for k, v in pol.state_dict().items()} == keys["actor_" + mode]
            assert {k: list(v.shape) for k, v in crit.state_dict().items()} == keys["critic_" + mode]
        apply_formula_(pol)
        apply_formula_(crit)
        tag = "%s/%s/" % (mode, name)
        obs, act = torch.from_numpy(z[tag + "obs"]), torch.from_numpy(z[tag + "act_in"])
        with torch.no_grad():
            a = pol(obs)
            q1, q2 = crit(obs, act)
            q1b = crit.Q1(obs, act)
        assert a.shape == (4, 3 * m.num_limbs) and q1.shape == (4, 1) and q2.shape == (4, 1)
        np.testing.assert_allclose(a.numpy(), z[tag + "action"], atol=3e-6)
        scale = max(1.0, np.abs(z[tag + "q1"]).max())
        np.testing.assert_allclose(q1.numpy(), z[tag + "q1"], atol=1e-5 * scale)
        np.testing.assert_allclose(q2.numpy(), z[tag + "q2"], atol=1e-5 * scale)
        np.testing.assert_allclose(q1b.numpy(), q1.numpy(), atol=1e-6 * scale)


def test_smp_is_differentiable_and_rejects_what_the_reference_cannot_run():
    torch.manual_seed(0)
    pol = ActorGraphPolicy(41, 3, 32, 1, 1.0, 5, True, True, True, None)
    m = mjcf.load_asset("3d_humanoid_9_full")
    pol.change_morphology({"parents": list(m.parents)})
    x = torch.randn(3, 41 * m.num_limbs, requires_grad=True)
    pol(x).sum().backward()
    assert x.grad is not None and all(p.grad is not None for p in pol.parameters())
    with pytest.raises(NotImplementedError):
        ActorGraphPolicy(41, 3, 32, 1, 1.0, 5, True, False, True, None)      # no top-down: reference raises in forward
    with pytest.raises(NotImplementedError):
        ActorGraphPolicy(41, 3, 32, 1, 1.0, 5, False, True, True, None)      # torchfold path not rebuilt
    with pytest.raises(AssertionError):
        pol(torch.randn(3, 41 * 4))                                          # wrong width for the morphology
    with pytest.raises(AssertionError):
        ActorGraphPolicy(41, 3, 32, 1, 1.0, 1, True, True, True, None).change_morphology({"parents": list(m.parents)})
