#!/usr/bin/env python3
"""Generate tests/golden/* by EXECUTING the reference (imported from /root/reference, never copied).

Run in the build container only:   python tools/capture_golden.py
Outputs (data only: inputs + the reference's outputs):
  tests/golden/graphs.json        per-XML parents / traversals / adjacency / mask / relation / joints /
                                  motors / action_order      (src/utils.py:236-484, src/wrappers.py:28-37)
  tests/golden/set_forward.npz    SEPolicy.forward on formula weights, f32 and f64 (src/SEActor.py:334-347)
  tests/golden/env_arith.npz      ModularEnv.step/_get_obs/reset_model on a fake simulator, one case set
                                  per distinct env family file (src/environments/<name>.py:15-164)
  tests/golden/wrapper_pad.npz    ModularEnvWrapper.step/reset padding + action scatter (src/wrappers.py:39-65)
"""
import hashlib
import importlib
import json
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import refstub  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def _args_ns():
    a = types.SimpleNamespace()
    a.attention_embedding_size = 128
    a.attention_heads = 2
    a.attention_hidden_size = 256
    a.attention_layers = 3
    a.dropout_rate = 0.0
    a.condition_decoder_on_features = 0
    a.transformer_norm = 1
    a.traversal_types = ["pre", "inlcrs", "postlcrs"]
    a.rel_size = 3
    return a


def capture_graphs(xmls):
    import torch
    import utils as ref_utils
    import wrappers as ref_wrappers
    out = {}
    for name, path in xmls.items():
        parents = ref_utils.getGraphStructure(path, "morphology")
        gd = ref_utils.getGraphDict(parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
        joints = ref_utils.getGraphJoints(path)
        motors = ref_utils.getMotorJoints(path)

        # run the reference wrapper's own constructor on a fake env to obtain action_order
        L = len(parents)
        fake = types.SimpleNamespace()
        fake.observation_space = types.SimpleNamespace(shape=(41 * L,))
        fake.action_space = types.SimpleNamespace(shape=(3 * (L - 1),), high=np.ones(3 * (L - 1)))
        fake.model = types.SimpleNamespace(body_names=["world"] + [j[0] for j in joints])
        fake.xml = path
        w = ref_wrappers.ModularEnvWrapper(fake, obs_max_len=41 * 14)
        mask = gd["mask"].numpy()
        out[name] = {
            "parents": [int(p) for p in parents],
            "traversals": [[int(v) for v in t.tolist()] for t in gd["traversals"]],
            "adjacency": gd["adjacency"].numpy().astype(int).tolist(),
            "mask_is_neg_inf": np.isneginf(mask).astype(int).tolist(),
            "mask_is_zero": (mask == 0).astype(int).tolist(),
            "ppr": gd["ppr"].numpy().astype(np.float64).tolist(),
            "sym_lap": gd["sym_lap"].numpy().astype(np.float64).tolist(),
            "distance": gd["distance"].numpy().astype(np.float64).tolist(),
            "transition": gd["transition"].numpy().astype(np.float64).tolist(),
            "relation": gd["relation"].numpy().astype(np.float64).tolist(),
            "joints": joints,
            "motors": motors,
            "action_order": [int(v) for v in w.action_order],
            "num_limbs": int(w.num_limbs),
            "limb_obs_size": int(w.limb_obs_size),
            "limb_action_size": int(w.limb_action_size),
            "max_children": int(ref_utils.findMaxChildren([name], {name: parents})),
        }
    with open(os.path.join(GOLD, "graphs.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("graphs.json:", len(out), "morphologies")
    return out


def capture_wrapper_pad(xmls):
    """ModularEnvWrapper.step: un-pad + scatter the action, zero-pad the observation."""
    import wrappers as ref_wrappers
    import utils as ref_utils
    res = {}
    rng = np.random.RandomState(7)
    for name in ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full", "3d_cheetah_14_full",
                 "3d_walker_2_right_leg_left_knee"]:
        path = xmls[name]
        joints = ref_utils.getGraphJoints(path)
        L = len(joints)

        class FakeEnv(object):
            pass
        fake = FakeEnv()
        fake.observation_space = types.SimpleNamespace(shape=(41 * L,))
        fake.action_space = types.SimpleNamespace(shape=(3 * (L - 1),), high=np.ones(3 * (L - 1)))
        fake.model = types.SimpleNamespace(body_names=["world"] + [j[0] for j in joints])
        fake.xml = path
        seen = {}

        def step(a, seen=seen, L=L):
            seen["a"] = [float(v) for v in a]
            return np.arange(41 * L, dtype=np.float64) + 0.5, 1.25, False, {"dist": 3.0}
        fake.step = step
        fake.reset = lambda L=L: np.arange(41 * L, dtype=np.float64) - 0.25
        obs_max_len = 41 * 14
        w = ref_wrappers.ModularEnvWrapper(fake, obs_max_len=obs_max_len)
        act = rng.uniform(-1, 1, size=3 * 14)
        ob, r, d, info = w.step(act)
        ob0 = w.reset()
        res[name + "/action_in"] = act
        res[name + "/env_action"] = np.array(seen["a"])
        res[name + "/obs_step"] = ob
        res[name + "/obs_reset"] = ob0
    np.savez_compressed(os.path.join(GOLD, "wrapper_pad.npz"), **res)
    print("wrapper_pad.npz:", len(res), "arrays")


def capture_set_forward(xmls, graphs):
    import torch
    import utils as ref_utils
    from SEActor import SEPolicy
    from oracle.formula import apply_formula_, synth_obs
    args = _args_ns()
    pol = SEPolicy(41, 3, 32, 1, 1.0, 3, True, False, False, args)
    pol.eval()
    apply_formula_(pol)
    pol64 = SEPolicy(41, 3, 32, 1, 1.0, 3, True, False, False, args).double()
    pol64.eval()
    apply_formula_(pol64)
    # record state_dict key -> shape for the nn.Module surface test
    keys = {k: list(v.shape) for k, v in pol.state_dict().items()}
    with open(os.path.join(GOLD, "set_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    res = {}
    seed = 100
    for name, path in xmls.items():
        parents = graphs[name]["parents"]
        gd = ref_utils.getGraphDict(parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
        pol.change_morphology(gd)
        gd64 = dict(gd)
        gd64["relation"] = gd["relation"].double()
        pol64.change_morphology(gd64)
        L = len(parents)
        for B in (1, 5):
            seed += 1
            obs = synth_obs(L, B, seed).astype(np.float32).astype(np.float64)  # f32-representable inputs
            with torch.no_grad():
                a32 = pol(torch.from_numpy(obs).float()).numpy()
                a64 = pol64(torch.from_numpy(obs)).numpy()
            res["%s/B%d/obs" % (name, B)] = obs.astype(np.float32)
            res["%s/B%d/act_f32" % (name, B)] = a32
            res["%s/B%d/act_f64" % (name, B)] = a64
    np.savez_compressed(os.path.join(GOLD, "set_forward.npz"), **res)
    sat = max(float(np.abs(v).max()) for k, v in res.items() if k.endswith("act_f32"))
    mean = np.mean([float(np.abs(v).mean()) for k, v in res.items() if k.endswith("act_f32")])
    print("set_forward.npz:", len(res), "arrays; max|a| = %.4f mean|a| = %.4f" % (sat, mean))

    # intermediate probes for one morphology (walker_7, B=2): lets the kernel tests localise a mismatch
    name = "3d_walker_7_full"
    parents = graphs[name]["parents"]
    gd = ref_utils.getGraphDict(parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu"))
    gd64 = dict(gd)
    gd64["relation"] = gd["relation"].double()
    pol64.change_morphology(gd64)
    obs = synth_obs(len(parents), 2, 999)
    probes = {"obs": obs}
    hooks = []
    enc = pol64.actor.transformer_encoder

    def mk(tag):
        def hook(mod, inp, out):
            if isinstance(out, tuple):
                for i, o in enumerate(out):
                    probes["%s/out%d" % (tag, i)] = o.detach().numpy().copy()
            else:
                probes[tag] = out.detach().numpy().copy()
        return hook
    for li, layer in enumerate(enc.layers):
        hooks.append(layer.register_forward_hook(mk("layer%d" % li)))
        hooks.append(layer.self_attn.register_forward_hook(mk("layer%d/attn" % li)))
    hooks.append(enc.register_forward_hook(mk("encoder")))
    with torch.no_grad():
        probes["act_f64"] = pol64(torch.from_numpy(obs)).numpy()
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(GOLD, "set_probes_walker7.npz"), **probes)
    print("set_probes_walker7.npz:", sorted(probes.keys()))


# ----------------------------------------------------------------------------------------------
# env arithmetic on a fake simulator
# ----------------------------------------------------------------------------------------------
class FakeData(object):
    def __init__(self, names, jnames, xpos, xquat, xvelp, xvelr, xaxis, qpos, qvel):
        self._n = {n: i for i, n in enumerate(names)}
        self._j = {n: i for i, n in enumerate(jnames)}
        self.xpos, self.xquat, self.xvelp, self.xvelr, self.xaxis = xpos, xquat, xvelp, xvelr, xaxis
        self.qpos, self.qvel = qpos, qvel

    def get_body_xpos(self, n):
        return self.xpos[self._n[n]]

    def get_body_xquat(self, n):
        return self.xquat[self._n[n]]

    def get_body_xvelp(self, n):
        return self.xvelp[self._n[n]]

    def get_body_xvelr(self, n):
        return self.xvelr[self._n[n]]

    def get_joint_xaxis(self, n):
        return self.xaxis[self._j[n]]


def _rand_quat(rng, tilt):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.uniform(-tilt, tilt)
    yaw = rng.uniform(-np.pi, np.pi)
    q1 = np.array([np.cos(ang / 2), *(np.sin(ang / 2) * ax)])
    q0 = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    w0, x0, y0, z0 = q0
    w1, x1, y1, z1 = q1
    return np.array([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1,
                     w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                     w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
                     w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1])


def _body_joint_info(path):
    """names in XML pre-order, joint names, joint ranges (radians, as MuJoCo stores them for angle=degree)."""
    root = ET.parse(path).getroot()
    wb = root.find("worldbody")
    names, jnames, jranges = [], [], []

    def rec(b):
        names.append(b.get("name"))
        for j in b.findall("joint"):
            if j.get("type", "hinge") == "free":
                jnames.append(j.get("name"))
                jranges.append([0.0, 0.0])
            else:
                jnames.append(j.get("name"))
                lo, hi = [float(v) for v in j.get("range").split()]
                jranges.append([np.radians(lo), np.radians(hi)])
        for c in b.findall("body"):
            rec(c)
    rec(wb.find("body"))
    opt = root.find("option")
    ts = float(opt.get("timestep", "0.002")) if opt is not None else 0.002
    return names, jnames, np.array(jranges), ts


def capture_env_arith(xmls):
    envdir = os.path.join(refstub.REF_SRC, "environments")
    # one representative per distinct env-file hash
    seen = {}
    for f in sorted(os.listdir(envdir)):
        if not f.endswith(".py") or f == "ModularEnv.py":
            continue
        h = hashlib.md5(open(os.path.join(envdir, f), "rb").read()).hexdigest()
        seen.setdefault(h, []).append(f[:-3])
    # map env name -> xml (v2 names use the v1 xml: identical files)
    res = {}
    meta = {}
    rng = np.random.RandomState(1234)
    ncase_total = 0
    for h, group in sorted(seen.items(), key=lambda kv: kv[1][0]):
        envname = group[0]
        xmlname = envname.replace("_v2_", "_")
        if xmlname not in xmls:
            print("  skip (no xml):", envname)
            continue
        path = xmls[xmlname]
        mod = importlib.import_module("environments." + envname)
        names, jnames, jranges, ts = _body_joint_info(path)
        L = len(names)
        nq, nv = 7 + 3 * (L - 1), 6 + 3 * (L - 1)
        env = mod.ModularEnv(path)  # fake base ctor: no physics, no step
        nb = L + 1
        model = types.SimpleNamespace()
        model.body_names = ["world"] + names
        model.nq, model.nv = nq, nv
        model.opt = types.SimpleNamespace(timestep=ts)
        model.jnt_qposadr = np.array([0] + [7 + i for i in range(3 * (L - 1))])
        model.jnt_range = jranges
        model.body_jntadr = np.array([-1, 0] + [1 + 3 * i for i in range(L - 1)])
        model.body_name2id = lambda n, _names=model.body_names: _names.index(n)
        env.model = model
        meta[envname] = {"xml": xmlname, "group": group, "names": names, "timestep": ts, "L": L}
        ncases = 24
        for c in range(ncases):
            def snap(height):
                xpos = rng.uniform(-1, 1, size=(nb, 3))
                xpos[:, 2] = rng.uniform(0.0, 1.6, size=nb)
                xpos[1, 2] = height
                xquat = np.stack([_rand_quat(rng, 0.5) for _ in range(nb)])
                xvelp = rng.normal(0, 6.0, size=(nb, 3))   # some beyond the +-10 clip
                xvelr = rng.normal(0, 5.0, size=(nb, 3))
                xaxis = rng.normal(size=(len(jnames), 3))
                xaxis /= np.linalg.norm(xaxis, axis=1, keepdims=True)
                qpos = np.zeros(nq)
                qpos[0:2] = xpos[1, 0:2] + rng.normal(0, 1e-3, size=2)
                qpos[2] = height
                tilt = [0.3, 0.95, 1.05, 1.4][c % 4]
                qpos[3:7] = _rand_quat(rng, tilt)
                qpos[7:] = rng.uniform(jranges[1:, 0] - 0.05, jranges[1:, 1] + 0.05)
                qvel = rng.normal(0, [1.0, 0.2, 30.0][c % 3], size=nv)
                return FakeData(model.body_names, jnames, xpos, xquat, xvelp, xvelr, xaxis, qpos, qvel)
            # heights straddling every family threshold (0.26, 0.45, 0.54, 0.6, 0.664, 0.8, 0.834625, 0.95, 1.74..2.0)
            hts = [0.2, 0.27, 0.44, 0.46, 0.53, 0.55, 0.59, 0.61, 0.66, 0.67, 0.79, 0.81, 0.83, 0.84, 0.94, 0.96,
                   1.2, 1.73, 1.75, 1.83, 1.84, 1.86, 1.99, 2.01]
            before = snap(hts[c] + 0.01)
            after = snap(hts[c])
            target = rng.uniform(-1, 1, size=2) * (10000 if c % 5 else 0.8)
            if c == 7:
                target = after.xpos[1, :2] + np.array([0.3, 0.2])  # dist_after < 1 and |target| > 1 -> resample
                target = target + np.sign(target) * 1.5
                after.xpos[1, :2] = target - np.array([0.3, 0.2])
            a = list(rng.uniform(-1, 1, size=3 * (L - 1)))
            env.sim = types.SimpleNamespace(data=before, model=model)
            env.data = before
            env._after = after
            env.target = target.copy()
            env.np_random = np.random.RandomState(1000 + c)
            ob, rew, done, info = env.step(a)
            key = "%s/c%02d/" % (envname, c)
            res[key + "before_torso_xpos"] = before.xpos[1].copy()
            res[key + "before_torso_quat"] = before.qpos[3:7].copy()
            res[key + "after_xpos"] = after.xpos
            res[key + "after_qpos"] = after.qpos
            res[key + "after_qvel"] = after.qvel
            res[key + "after_xvelp"] = after.xvelp
            res[key + "after_xvelr"] = after.xvelr
            res[key + "after_xaxis"] = after.xaxis
            res[key + "jnt_range"] = jranges
            res[key + "target_in"] = target
            res[key + "action"] = np.array(a)
            res[key + "obs"] = np.asarray(ob, dtype=np.float64)
            res[key + "reward"] = np.float64(rew)
            res[key + "done"] = np.bool_(done)
            res[key + "dist"] = np.float64(info["dist"])
            res[key + "target_out"] = np.array(env.target, dtype=np.float64)
            res[key + "resample_u"] = np.float64(np.random.RandomState(1000 + c).uniform(-np.pi, np.pi))
            ncase_total += 1
        # reset_model: capture the draws -> (qpos, qvel, target) mapping with a seeded RandomState
        env.sim = types.SimpleNamespace(data=before, model=model)
        env.data = before
        env.init_qpos = np.zeros(nq)
        env.init_qpos[2] = 1.3
        env.init_qpos[3] = 1.0
        env.init_qvel = np.zeros(nv)
        env.np_random = np.random.RandomState(4242)
        env.reset_model()
        qp, qv = env._set_state_args
        res[envname + "/reset/qpos"] = qp
        res[envname + "/reset/qvel"] = qv
        res[envname + "/reset/target"] = np.array(env.target, dtype=np.float64)
        res[envname + "/reset/init_qpos_after"] = env.init_qpos.copy()
        r2 = np.random.RandomState(4242)
        is_cheetah = "cheetah" in envname
        draws = [r2.uniform(-np.pi, np.pi)]
        draws += list(r2.uniform(-0.1, 0.1, size=nq) if is_cheetah else r2.uniform(-0.005, 0.005, size=nq))
        draws += list(r2.randn(nv) if is_cheetah else r2.uniform(-0.005, 0.005, size=nv))
        draws.append(r2.uniform(-np.pi, np.pi))
        if "_v2_" in envname:
            draws.append(r2.uniform(10, 20))
        res[envname + "/reset/draws"] = np.array(draws)
    np.savez_compressed(os.path.join(GOLD, "env_arith.npz"), **res)
    with open(os.path.join(GOLD, "env_arith_meta.json"), "w") as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    print("env_arith.npz: %d step cases over %d env files" % (ncase_total, len(meta)))


def main():
    refstub.install()
    os.makedirs(GOLD, exist_ok=True)
    xmls = refstub.all_xmls()
    graphs = capture_graphs(xmls)
    capture_wrapper_pad(xmls)
    capture_set_forward(xmls, graphs)
    capture_env_arith(xmls)


if __name__ == "__main__":
    main()
