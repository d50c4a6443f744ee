"""Graph preprocessing against tests/golden/graphs.json (produced by the reference's own
utils.getGraphStructure/getGraphDict/getGraphJoints/getMotorJoints and wrappers.ModularEnvWrapper)."""
import json
import os

import numpy as np
import pytest
import torch

from sgrl_amd import graph, mjcf

TRAV = ["pre", "inlcrs", "postlcrs"]


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "graphs.json")) as f:
        return json.load(f)


def test_covers_all_29_morphologies(golden):
    assert len(golden) == 29
    assert sorted(golden) == mjcf.list_assets()


def test_integer_tables_bit_exact(golden):
    for name, g in golden.items():
        parents = g["parents"]
        trav = graph.getTraversal(parents, TRAV)
        assert trav == g["traversals"], name
        gd = graph.getGraphDict(parents, TRAV, [], device=torch.device("cpu"))
        assert [t.tolist() for t in gd["traversals"]] == g["traversals"]
        assert gd["traversals"][0].dtype == torch.int64
        assert gd["adjacency"].numpy().astype(int).tolist() == g["adjacency"], name
        mask = gd["mask"].numpy()
        assert np.isneginf(mask).astype(int).tolist() == g["mask_is_neg_inf"], name
        assert (mask == 0).astype(int).tolist() == g["mask_is_zero"], name
        assert graph.findMaxChildren([name], {name: parents}) == g["max_children"]


def test_float_tables(golden):
    for name, g in golden.items():
        gd = graph.getGraphDict(g["parents"], TRAV, [], device=torch.device("cpu"))
        for k in ("ppr", "sym_lap", "distance", "transition", "relation"):
            a = gd[k].numpy().astype(np.float64)
            b = np.array(g[k])
            assert a.shape == b.shape, (name, k)
            assert gd[k].dtype == torch.float32
            np.testing.assert_allclose(a, b, atol=1e-6, rtol=0, err_msg="%s %s" % (name, k))
        assert gd["relation"].shape == (len(g["parents"]), len(g["parents"]), 3)


def test_spot_values_from_survey():
    gd = graph.getGraphDict([-1, 0, 1], TRAV, [], device=torch.device("cpu"))
    assert [t.tolist() for t in gd["traversals"]] == [[0, 1, 2], [2, 1, 0], [2, 1, 0]]
    np.testing.assert_allclose(gd["ppr"][0].numpy(), [0.395257, 0.391304, 0.213439], atol=1e-6)
    assert abs(float(gd["sym_lap"][0, 1]) + 0.707107) < 1e-6
    assert abs(float(gd["distance"][0, 2]) - 0.666667) < 1e-6
    t = graph.getTraversal([-1, 0, 1, 2, 0, 4, 5], TRAV)
    assert t[1] == [6, 2, 1, 0, 5, 4, 3] and t[2] == [6, 5, 1, 0, 4, 3, 2]


def test_compiled_assets_carry_the_same_structure(golden):
    """parents / joint names / motor order / action_order re-derived from the compiled models."""
    for name, g in golden.items():
        m = mjcf.load_asset(name)
        assert m.parents == g["parents"], name
        joints = [[b] + [jn for jn, jb in zip(m.joint_names, m.jnt_body) if m.body_names[jb] == b]
                  for b in m.body_names[1:]]
        assert joints == g["joints"], name
        assert m.motor_joints == g["motors"], name
        order = graph.action_order_for(joints, m.motor_joints)
        assert order == g["action_order"], name
        # the engine's inverse table: actuator u is fed by policy slot act_slot[u]
        for u, s in enumerate(m.act_slot):
            assert order[s] == u
        assert m.num_limbs == g["num_limbs"]
        assert g["limb_obs_size"] == 41 and g["limb_action_size"] == 3


def test_single_limb_graph_dict():
    assert graph.getGraphDict([-1], TRAV, [], device=torch.device("cpu")) == {"parents": [-1]}


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference XMLs only exist in the build container")
def test_xml_readers_on_reference_files(golden):
    base = "/root/reference/src/environments"
    for sub in ["3d_hoppers", "3d_walkers", "3d_humanoids", "3d_cheetahs", "zero_shot"]:
        for f in sorted(os.listdir(os.path.join(base, sub))):
            if f.endswith(".xml"):
                p = os.path.join(base, sub, f)
                g = golden[f[:-4]]
                assert graph.getGraphStructure(p) == g["parents"]
                assert graph.getGraphJoints(p) == g["joints"]
                assert graph.getMotorJoints(p) == g["motors"]
                assert graph.action_order_for(graph.getGraphJoints(p), graph.getMotorJoints(p)) == g["action_order"]


def test_xml_readers_on_synthetic_mjcf(tmp_path):
    xml = """<mujoco><compiler angle="degree" coordinate="local" inertiafromgeom="true"/>
    <default><geom contype="1" conaffinity="0"/></default>
    <worldbody><geom type="plane" size="1 1 1" conaffinity="1"/>
      <body name="torso" pos="0 0 1"><joint name="root" type="free"/><geom type="sphere" size="0.1"/>
        <body name="a_thigh" pos="0 0 -0.2"><joint name="a_thigh_joint_x" axis="1 0 0" range="-1 1"/>
          <joint name="a_thigh_joint_y" axis="0 1 0" range="-1 1"/><joint name="a_thigh_joint_z" axis="0 0 1" range="-1 1"/>
          <geom type="capsule" fromto="0 0 0 0 0 -0.3" size="0.04"/>
          <body name="a_foot" pos="0 0 -0.3"><joint name="a_foot_joint_x" axis="1 0 0" range="-1 1"/>
            <joint name="a_foot_joint_y" axis="0 1 0" range="-1 1"/><joint name="a_foot_joint_z" axis="0 0 1" range="-1 1"/>
            <geom type="sphere" size="0.05"/></body></body>
        <body name="b_shin" pos="0 0.1 -0.2"><joint name="b_shin_joint_x" axis="1 0 0" range="-1 1"/>
          <joint name="b_shin_joint_y" axis="0 1 0" range="-1 1"/><joint name="b_shin_joint_z" axis="0 0 1" range="-1 1"/>
          <geom type="sphere" size="0.05"/></body>
      </body></worldbody>
    <actuator><motor joint="b_shin_joint_x"/><motor joint="b_shin_joint_y"/><motor joint="b_shin_joint_z"/>
      <motor joint="a_thigh_joint_x"/><motor joint="a_thigh_joint_y"/><motor joint="a_thigh_joint_z"/>
      <motor joint="a_foot_joint_x"/><motor joint="a_foot_joint_y"/><motor joint="a_foot_joint_z"/></actuator></mujoco>"""
    p = tmp_path / "3d_walker_test.xml"
    p.write_text(xml)
    assert graph.getGraphStructure(str(p)) == [-1, 0, 1, 0]
    assert graph.getGraphStructure(str(p), "line") == [-1, 0, 1, 2]
    assert graph.getGraphStructure(str(p), "tree") == [-1, 0, 0, 0]
    joints = graph.getGraphJoints(str(p))
    motors = graph.getMotorJoints(str(p))
    order = graph.action_order_for(joints, motors)
    assert order == [-1, -1, -1, 3, 4, 5, 6, 7, 8, 0, 1, 2]
    m = mjcf.compile_mjcf(str(p))
    assert m.parents == [-1, 0, 1, 0]
    assert list(m.act_slot) == [9, 10, 11, 3, 4, 5, 6, 7, 8]
