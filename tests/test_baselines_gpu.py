"""f4 on the device under test: the SWAT and SMP baseline modules (PyTorch, no HIP kernels of their own) run on
cuda:0 against the fixtures produced by executing the reference's StructurePolicy / CriticStructurePolicy
(reference src/StructureActor.py:176-273) and ActorGraphPolicy / CriticGraphPolicy (reference src/ModularActor.py:99-384)."""
import os

import numpy as np
import pytest
import torch

from oracle.formula import apply_formula_
from sgrl_amd import graph as G, mjcf

pytestmark = pytest.mark.gpu


def test_swat_forward_on_the_gpu_matches_the_reference_fixture(golden_dir):
    from sgrl_amd.set_policy import default_args
    from sgrl_amd.swat_policy import CriticStructurePolicy, StructurePolicy
    z = np.load(os.path.join(golden_dir, "swat_forward.npz"))
    dev = torch.device("cuda:0")
    for cond in (0, 1):
        args = default_args(condition_decoder_on_features=cond)
        pol = StructurePolicy(41, 3, 32, 1, 1.0, 3, True, False, False, args).to(dev).eval()
        crit = CriticStructurePolicy(41, 3, 32, 1, 3, True, False, False, args).to(dev).eval()
        apply_formula_(pol)
        apply_formula_(crit)
        for name in sorted({k.split("/")[1] for k in z.files if k.startswith("cond%d/" % cond)}):
            m = mjcf.load_asset(name)
            gd = G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=dev)
            pol.change_morphology(gd)
            crit.change_morphology(gd)
            tag = "cond%d/%s/" % (cond, name)
            obs, act = torch.from_numpy(z[tag + "obs"]).to(dev), torch.from_numpy(z[tag + "act_in"]).to(dev)
            with torch.no_grad():
                a = pol(obs)
                q1, q2 = crit(obs, act)
            scale = max(1.0, np.abs(z[tag + "q1"]).max())
            np.testing.assert_allclose(a.cpu().numpy(), z[tag + "action"], atol=2e-5)
            np.testing.assert_allclose(q1.cpu().numpy(), z[tag + "q1"], atol=5e-5 * scale)
            np.testing.assert_allclose(q2.cpu().numpy(), z[tag + "q2"], atol=5e-5 * scale)


def test_smp_forward_on_the_gpu_matches_the_reference_fixture(golden_dir):
    import json
    from sgrl_amd.smp_policy import ActorGraphPolicy, CriticGraphPolicy
    z = np.load(os.path.join(golden_dir, "smp_forward.npz"))
    mc = json.load(open(os.path.join(golden_dir, "smp_state_dict_keys.json")))["max_children"]
    dev = torch.device("cuda:0")
    for td, bu in ((True, True), (True, False)):
        mode = "td%d_bu%d" % (td, bu)
        pol = ActorGraphPolicy(41, 3, 32, 1, 1.0, mc, True, td, bu, None).eval()
        crit = CriticGraphPolicy(41, 3, 32, 1, mc, True, td, bu, None).eval()
        for name in sorted({k.split("/")[1] for k in z.files if k.startswith(mode + "/")}):
            m = mjcf.load_asset(name)
            pol.change_morphology({"parents": list(m.parents)})
            crit.change_morphology({"parents": list(m.parents)})
            pol.to(dev)
            crit.to(dev)
            apply_formula_(pol)
            apply_formula_(crit)
            tag = "%s/%s/" % (mode, name)
            obs, act = torch.from_numpy(z[tag + "obs"]).to(dev), torch.from_numpy(z[tag + "act_in"]).to(dev)
            with torch.no_grad():
                a = pol(obs)
                q1, q2 = crit(obs, act)
            scale = max(1.0, np.abs(z[tag + "q1"]).max())
            np.testing.assert_allclose(a.cpu().numpy(), z[tag + "action"], atol=2e-5)
            np.testing.assert_allclose(q1.cpu().numpy(), z[tag + "q1"], atol=5e-5 * scale)
            np.testing.assert_allclose(q2.cpu().numpy(), z[tag + "q2"], atol=5e-5 * scale)
