#!/usr/bin/env python3
"""Golden vectors for the SWAT baseline: executes the reference StructurePolicy / CriticStructurePolicy
(reference src/StructureActor.py:176-273, src/StructureCritic.py:8-125) with the formula weights of oracle/formula.py on
synthetic batches, with and without condition_decoder_on_features.  Build container only; numbers only."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import numpy as np, torch
import refstub
refstub.install()
import utils as ref_utils
from StructureActor import StructurePolicy
from StructureCritic import CriticStructurePolicy
from capture_golden import _args_ns
from oracle.formula import apply_formula_, synth_obs
xm = refstub.all_xmls()
res, keys = {}, {}
for cond in (0, 1):
    a = _args_ns()
    a.condition_decoder_on_features = cond
    pol = StructurePolicy(41, 3, 32, 1, 1.0, 3, True, False, False, a).eval()
    crit = CriticStructurePolicy(41, 3, 32, 1, 3, True, False, False, a).eval()
    apply_formula_(pol); apply_formula_(crit)
    keys["actor_cond%d" % cond] = {k: list(v.shape) for k, v in pol.state_dict().items()}
    keys["critic_cond%d" % cond] = {k: list(v.shape) for k, v in crit.state_dict().items()}
    for name in ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full", "3d_cheetah_14_full", "3d_walker_2_right_leg_left_knee"]:
        parents = ref_utils.getGraphStructure(xm[name])
        gd = ref_utils.getGraphDict(parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
        pol.change_morphology(gd); crit.change_morphology(gd)
        L = len(parents)
        obs = synth_obs(L, 4, 31 + L).astype(np.float32)
        act = np.random.RandomState(100 + L).uniform(-1, 1, size=(4, 3 * L)).astype(np.float32)
        with torch.no_grad():
            out = pol(torch.from_numpy(obs))
            q1, q2 = crit(torch.from_numpy(obs), torch.from_numpy(act))
            assert torch.equal(q1, crit.Q1(torch.from_numpy(obs), torch.from_numpy(act)))
        tag = "cond%d/%s/" % (cond, name)
        res[tag + "obs"], res[tag + "act_in"], res[tag + "action"] = obs, act, out.numpy()
        res[tag + "q1"], res[tag + "q2"] = q1.numpy(), q2.numpy()
json.dump(keys, open(os.path.join(REPO, "tests", "golden", "swat_state_dict_keys.json"), "w"), indent=0, sort_keys=True)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "swat_forward.npz"), **res)
print("swat golden written:", len(res), "arrays; |action| mean", np.mean([np.abs(v).mean() for k, v in res.items() if k.endswith("action")]))
