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
