"""Pins oracle/set_ref.py (NumPy restatement of the SET actor) to the reference: golden vectors were produced by the
reference's own SEPolicy with formula weights (tools/capture_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import set_ref


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "set_state_dict_keys.json")) as f:
        keys = json.load(f)
    with open(os.path.join(golden_dir, "graphs.json")) as f:
        graphs = json.load(f)
    return keys, graphs, np.load(os.path.join(golden_dir, "set_forward.npz"))


def test_f64_matches_reference_on_all_29_morphologies(gold):
    keys, graphs, z = gold
    sd = set_ref.formula_state_dict(keys, np.float64)
    assert sum(int(np.prod(s)) for s in keys.values()) == 4712712   # parameter count of the reference actor
    n = 0
    for name, g in graphs.items():
        for B in (1, 5):
            obs = z["%s/B%d/obs" % (name, B)].astype(np.float64)
            act = set_ref.set_actor_forward(sd, obs, g["traversals"], np.array(g["relation"]))
            ref = z["%s/B%d/act_f64" % (name, B)]
            assert act.shape == ref.shape == (B, 3 * len(g["parents"]))
            assert np.abs(act - ref).max() < 1e-11, (name, B)
            n += 1
    assert n == 58


def test_f32_matches_reference_f32(gold):
    keys, graphs, z = gold
    sd = set_ref.formula_state_dict(keys, np.float32)
    for name in ("3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full", "3d_cheetah_14_full"):
        g = graphs[name]
        obs = z["%s/B5/obs" % name]
        act = set_ref.set_actor_forward(sd, obs, g["traversals"], np.array(g["relation"], dtype=np.float32))
        assert act.dtype == np.float32
        assert np.abs(act - z["%s/B5/act_f32" % name]).max() < 2e-5


def test_layer_probes(gold, golden_dir):
    keys, graphs, _ = gold
    p = np.load(os.path.join(golden_dir, "set_probes_walker7.npz"))
    sd = set_ref.formula_state_dict(keys, np.float64)
    g = graphs["3d_walker_7_full"]
    probes = {}
    act = set_ref.set_actor_forward(sd, p["obs"], g["traversals"], np.array(g["relation"]), probes=probes)
    assert np.abs(act - p["act_f64"]).max() < 1e-12
    for li in range(3):
        ref_g = np.transpose(p["layer%d/out0" % li], (1, 0, 2, 3))     # reference layout [L,B,3,128]
        ref_ng = np.transpose(p["layer%d/out1" % li], (1, 0, 2))
        assert np.abs(probes["layer%d/g" % li] - ref_g).max() < 1e-11
        assert np.abs(probes["layer%d/ng" % li] - ref_ng).max() < 1e-11


def test_subequivariance_and_batch_consistency(gold):
    """Rotating all eight 3-vectors of every limb about the gravity axis leaves the action unchanged."""
    keys, graphs, z = gold
    sd = set_ref.formula_state_dict(keys, np.float64)
    g = graphs["3d_walker_5_foot"]
    obs = z["3d_walker_5_foot/B5/obs"].astype(np.float64)
    a0 = set_ref.set_actor_forward(sd, obs, g["traversals"], np.array(g["relation"]))
    th = 1.234
    rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    o = obs.reshape(5, 5, 41).copy()
    v = o[..., :24].reshape(5, 5, 8, 3)
    o[..., :24] = (v @ rz.T).reshape(5, 5, 24)
    a1 = set_ref.set_actor_forward(sd, o.reshape(5, -1), g["traversals"], np.array(g["relation"]))
    assert np.abs(a0 - a1).max() < 1e-13
    one = set_ref.set_actor_forward(sd, obs[2:3], g["traversals"], np.array(g["relation"]))
    assert np.abs(one[0] - a0[2]).max() < 1e-13
