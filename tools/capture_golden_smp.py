#!/usr/bin/env python3
"""Golden vectors for the SMP baseline: executes the reference ActorGraphPolicy / CriticGraphPolicy (reference
src/ModularActor.py:99-384, src/ModularCritic.py:143-520; disable_fold path) with the formula weights of oracle/formula.py on
synthetic batches, for every message-passing mode the reference can run.  Build container only; numbers only."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import numpy as np, torch
import refstub
refstub.install()
import utils as ref_utils
from ModularActor import ActorGraphPolicy
from ModularCritic import CriticGraphPolicy
from oracle.formula import apply_formula_, synth_obs
xm = refstub.all_xmls()
NAMES = ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full", "3d_cheetah_14_full", "3d_walker_2_right_leg_left_knee"]
graphs = {n: ref_utils.getGraphStructure(xm[n]) for n in NAMES}
MAXC = max(max(p.count(i) for i in range(len(p))) for p in graphs.values())
res, keys = {}, {"max_children": MAXC}
for td, bu in ((1, 1), (1, 0)):     # without top-down messages the reference's disable_fold path raises (ModularActor.py:244)
    mode = "td%d_bu%d" % (td, bu)
    pol = ActorGraphPolicy(41, 3, 32, 1, 1.0, MAXC, True, bool(td), bool(bu), None).eval()
    crit = None if (bu and not td) else CriticGraphPolicy(41, 3, 32, 1, MAXC, True, bool(td), bool(bu), None).eval()
    for name in NAMES:
        parents = graphs[name]
        gd = {"parents": parents}
        pol.change_morphology(gd)
        apply_formula_(pol)
        L = len(parents)
        if name == NAMES[0]:
            keys["actor_" + mode] = {k: list(v.shape) for k, v in pol.state_dict().items()}
        obs = synth_obs(L, 4, 51 + L).astype(np.float32)
        act = np.random.RandomState(200 + L).uniform(-1, 1, size=(4, 3 * L)).astype(np.float32)
        tag = "%s/%s/" % (mode, name)
        with torch.no_grad():
            res[tag + "action"] = pol(torch.from_numpy(obs)).numpy()
        res[tag + "obs"], res[tag + "act_in"] = obs, act
        if crit is not None:
            crit.change_morphology(gd)
            apply_formula_(crit)
            if name == NAMES[0]:
                keys["critic_" + mode] = {k: list(v.shape) for k, v in crit.state_dict().items()}
            with torch.no_grad():
                q1, q2 = crit(torch.from_numpy(obs), torch.from_numpy(act))
                q1b = crit.Q1(torch.from_numpy(obs), torch.from_numpy(act))
            assert torch.allclose(q1, q1b)
            res[tag + "q1"], res[tag + "q2"] = q1.numpy(), q2.numpy()
json.dump(keys, open(os.path.join(REPO, "tests", "golden", "smp_state_dict_keys.json"), "w"), indent=0, sort_keys=True)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "smp_forward.npz"), **res)
print("smp golden written:", len(res), "arrays; max_children", MAXC, "|action| mean",
      np.mean([np.abs(v).mean() for k, v in res.items() if k.endswith("action")]),
      "q shapes", {k: v.shape for k, v in res.items() if k.endswith("q1")}.popitem())
