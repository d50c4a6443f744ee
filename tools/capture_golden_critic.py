#!/usr/bin/env python3
"""Golden vectors for the twin SET critics: executes the reference SECritic (reference src/SECritic.py:8-124) with the
formula weights of oracle/formula.py on synthetic (state, action) batches.  Build container only."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import numpy as np, torch
import refstub
refstub.install()
import utils as ref_utils
from SECritic import SECritic
from capture_golden import _args_ns
from oracle.formula import apply_formula_, synth_obs
crit = SECritic(41, 3, 32, 1, 3, True, False, False, _args_ns()).eval()
apply_formula_(crit)
keys = {k: list(v.shape) for k, v in crit.state_dict().items()}
json.dump(keys, open(os.path.join(REPO, "tests", "golden", "critic_state_dict_keys.json"), "w"), indent=0, sort_keys=True)
xm = refstub.all_xmls()
res = {}
for name in ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full", "3d_cheetah_14_full", "3d_walker_2_right_leg_left_knee"]:
    parents = ref_utils.getGraphStructure(xm[name])
    gd = ref_utils.getGraphDict(parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
    crit.change_morphology(gd)
    L = len(parents)
    obs = synth_obs(L, 4, 77 + L).astype(np.float32)
    act = np.random.RandomState(L).uniform(-1, 1, size=(4, 3 * L)).astype(np.float32)
    with torch.no_grad():
        q1, q2 = crit(torch.from_numpy(obs), torch.from_numpy(act))
        q1b = crit.Q1(torch.from_numpy(obs), torch.from_numpy(act))
    assert torch.equal(q1, q1b)
    res[name + "/obs"], res[name + "/act"], res[name + "/q1"], res[name + "/q2"] = obs, act, q1.numpy(), q2.numpy()
np.savez_compressed(os.path.join(REPO, "tests", "golden", "critic_forward.npz"), **res)
print("critic golden written; |q| mean", np.mean([np.abs(v).mean() for k, v in res.items() if k.endswith("q1")]))
