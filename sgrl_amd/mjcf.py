"""MJCF -> flat rigid-body model (host side, init time).

The reference hands its XML morphologies (reference src/environments/*/*.xml) to MuJoCo 2.1.0's
compiler through mujoco-py (`MujocoEnv.__init__(self, xml, 4)`, reference
src/environments/ModularEnv.py:12).  MuJoCo is an un-vendored third-party dependency that is
absent from this image, so this module restates the parts of the MJCF compile step the shipped
morphologies use ([3P-knowledge]; parity against the real compiler is
UNPINNED):

  * <compiler angle coordinate inertiafromgeom>, one level of <default> (joint / geom / motor)
  * <option timestep integrator gravity>
  * bodies with pos/quat; free + hinge joints (axis, range, armature, damping, stiffness,
    solreflimit, solimplimit); capsule (fromto | pos+size+quat/axisangle), sphere and plane geoms
  * global -> local coordinate conversion, inertia from geoms (density * volume)
  * motors (joint transmission, scalar gear, ctrlrange)
  * contact pair list with contype/conaffinity + parent filter and per-pair parameter mixing
  * dof_invweight0 / body_invweight0 at qpos0

Output is a `Model` of NumPy arrays which `pack()` turns into the (int32, float64) blobs documented
in include/sgrl_model.h and consumed by both the HIP engine and the CPU oracle.
"""
import json
import math
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE = 0, 2, 3
JNT_FREE, JNT_HINGE = 0, 3
INT_EULER, INT_RK4 = 0, 1
MAGIC = 0x5347524C  # 'SGRL'

LIMB_NONE, LIMB_TORSO, LIMB_THIGH, LIMB_SHIN, LIMB_FOOT = 0, 1, 2, 3, 4


# ------------------------------------------------------------------------------------------------
# small math helpers (float64)
# ------------------------------------------------------------------------------------------------
def _num_prefix(tok):
    """MuJoCo reads numbers with a C++ stream; '0.13/2' yields 0.13 ([3P-knowledge])."""
    m = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", tok)
    if not m:
        raise ValueError("cannot parse number from %r" % tok)
    return float(m.group(0))


def _vec(text, n=None):
    vals = [_num_prefix(t) for t in text.split()]
    if n is not None and len(vals) > n:
        vals = vals[:n]
    return vals


def quat_mul(a, b):
    w0, x0, y0, z0 = a
    w1, x1, y1, z1 = b
    return np.array([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1,
                     w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                     w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
                     w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1])


def quat_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def axisangle_to_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if angle == 0.0 or n < 1e-15:
        return np.array([1.0, 0.0, 0.0, 0.0])
    axis = axis / n
    s = math.sin(angle / 2)
    return np.array([math.cos(angle / 2), axis[0] * s, axis[1] * s, axis[2] * s])


def z_to_quat(vec):
    """Quaternion rotating the z axis onto `vec`."""
    v = np.asarray(vec, dtype=np.float64)
    v = v / np.linalg.norm(v)
    axis = np.cross([0.0, 0.0, 1.0], v)
    a = np.linalg.norm(axis)
    if a < 1e-10:
        axis = np.array([1.0, 0.0, 0.0])
    else:
        axis = axis / a
    ang = math.atan2(a, v[2])
    return np.array([math.cos(ang / 2), *(axis * math.sin(ang / 2))])


def normalize_quat(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


# ------------------------------------------------------------------------------------------------
BUILTIN_JOINT = dict(type="hinge", pos=[0, 0, 0], axis=[0, 0, 1], limited=False, range=[0, 0], armature=0.0,
                     damping=0.0, stiffness=0.0, margin=0.0, solreflimit=[0.02, 1.0],
                     solimplimit=[0.9, 0.95, 0.001, 0.5, 2.0])
BUILTIN_GEOM = dict(type="sphere", size=[0, 0, 0], contype=1, conaffinity=1, condim=3,
                    friction=[1.0, 0.005, 0.0001], density=1000.0, margin=0.0, gap=0.0, solref=[0.02, 1.0],
                    solimp=[0.9, 0.95, 0.001, 0.5, 2.0], solmix=1.0)
BUILTIN_MOTOR = dict(ctrllimited=False, ctrlrange=[0.0, 0.0], gear=1.0)


def _merge_vec(text, base):
    """Partial vector attributes keep the trailing defaults."""
    vals = _vec(text)
    out = list(base)
    for i, v in enumerate(vals[:len(out)]):
        out[i] = v
    return out


def _bool(text):
    return text.strip().lower() == "true"


class Model(object):
    """Flat arrays; see include/sgrl_model.h for the packed layout."""

    INT_FIELDS = ["body_parent", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum", "body_limbtype",
                  "jnt_type", "jnt_body", "jnt_qposadr", "jnt_dofadr", "jnt_limited",
                  "dof_body", "dof_jnt", "dof_parent",
                  "geom_type", "geom_body",
                  "pair_g1", "pair_g2", "pair_condim",
                  "act_dof", "act_slot"]
    F64_FIELDS = ["qpos0",
                  "body_pos", "body_quat", "body_ipos", "body_inertia", "body_mass", "body_invweight0",
                  "jnt_pos", "jnt_axis", "jnt_range", "jnt_stiffness", "jnt_solref", "jnt_solimp", "jnt_margin",
                  "dof_armature", "dof_damping", "dof_invweight0",
                  "geom_pos", "geom_quat", "geom_size",
                  "pair_mu", "pair_margin", "pair_solref", "pair_solimp",
                  "act_gear", "act_ctrlrange"]

    def __init__(self):
        self.name = ""
        self.body_names = []
        self.joint_names = []
        self.motor_joints = []

    # ---- sizes
    @property
    def num_limbs(self):
        return self.nbody - 1

    def to_json(self):
        d = {"name": self.name, "body_names": self.body_names, "joint_names": self.joint_names,
             "motor_joints": self.motor_joints, "parents": self.parents,
             "header": {k: getattr(self, k) for k in ("nbody", "njnt", "nq", "nv", "nu", "ngeom", "npair",
                                                       "integrator")},
             "opt": {"timestep": self.timestep, "gravity": list(self.gravity)}}
        for k in self.INT_FIELDS:
            d[k] = np.asarray(getattr(self, k)).astype(int).ravel().tolist()
        for k in self.F64_FIELDS:
            d[k] = [float.hex(float(v)) for v in np.asarray(getattr(self, k), dtype=np.float64).ravel()]
        return d

    @staticmethod
    def from_json(d):
        m = Model()
        m.name = d["name"]
        m.body_names = d["body_names"]
        m.joint_names = d["joint_names"]
        m.motor_joints = d["motor_joints"]
        m.parents = d["parents"]
        for k, v in d["header"].items():
            setattr(m, k, int(v))
        m.timestep = float(d["opt"]["timestep"])
        m.gravity = np.array(d["opt"]["gravity"], dtype=np.float64)
        for k in Model.INT_FIELDS:
            setattr(m, k, np.array(d[k], dtype=np.int32))
        for k in Model.F64_FIELDS:
            setattr(m, k, np.array([float.fromhex(s) for s in d[k]], dtype=np.float64))
        m._reshape()
        return m

    def _reshape(self):
        self.body_pos = self.body_pos.reshape(-1, 3)
        self.body_quat = self.body_quat.reshape(-1, 4)
        self.body_ipos = self.body_ipos.reshape(-1, 3)
        self.body_inertia = self.body_inertia.reshape(-1, 6)
        self.body_invweight0 = self.body_invweight0.reshape(-1, 2)
        self.jnt_pos = self.jnt_pos.reshape(-1, 3)
        self.jnt_axis = self.jnt_axis.reshape(-1, 3)
        self.jnt_range = self.jnt_range.reshape(-1, 2)
        self.jnt_solref = self.jnt_solref.reshape(-1, 2)
        self.jnt_solimp = self.jnt_solimp.reshape(-1, 5)
        self.geom_pos = self.geom_pos.reshape(-1, 3)
        self.geom_quat = self.geom_quat.reshape(-1, 4)
        self.geom_size = self.geom_size.reshape(-1, 3)
        self.pair_solref = self.pair_solref.reshape(-1, 2)
        self.pair_solimp = self.pair_solimp.reshape(-1, 5)
        self.act_ctrlrange = self.act_ctrlrange.reshape(-1, 2)


def limb_type_of(name):
    """One-hot class by substring, reference src/environments/3d_walker_7_full.py:52-61."""
    if name == "torso":
        return LIMB_TORSO
    if "thigh" in name:
        return LIMB_THIGH
    if "shin" in name:
        return LIMB_SHIN
    if "foot" in name:
        return LIMB_FOOT
    return LIMB_NONE


# ------------------------------------------------------------------------------------------------
def _capsule_inertia(mass, r, h):
    """Solid capsule, axis z, cylinder half-length h: (Ixx, Iyy, Izz) about its centre."""
    height = 2.0 * h
    sphere_mass = mass * 4 * r / (4 * r + 3 * height)
    cyl_mass = mass - sphere_mass
    ixx = cyl_mass * (3 * r * r + height * height) / 12.0
    izz = cyl_mass * r * r / 2.0
    sph_i = 2.0 * sphere_mass * r * r / 5.0
    ixx += sph_i + sphere_mass * height * (3 * r + 2 * height) / 8.0
    izz += sph_i
    return ixx, ixx, izz


def capsule_volume(r, h, mode="mujoco210"):
    """Volume of a capsule of radius r and cylinder half-length h.

    mode="exact":     pi r^2 (2h) + 4/3 pi r^3.
    mode="mujoco210": pi (r^2 (2h) + r^3).  [3P-knowledge] The closed-source MuJoCo line (<= 2.1.0, the
        version mujoco-py 2.1.2.14 binds, reference requirements.txt:5) evaluated the hemisphere term
        with an integer 4/3; evidence: the body masses gym reports for Hopper (3.5343 = 1000 pi .05^2 (.4+.05))
        and Ant (0.036477 = 5 pi .08^2 (.28284+.08)) under that library.  Default, because the reference
        ran on that library; parity with it is unpinned either way.
    """
    if mode == "exact":
        return math.pi * r * r * 2 * h + 4.0 / 3.0 * math.pi * r ** 3
    return math.pi * (r * r * 2 * h + r ** 3)


# What compile_mjcf reads, per element (context "default/<tag>" = inside <default>).  Anything else in a model file is
# either on the no-dynamics list below or an error: an attribute this compiler does not understand must not be dropped
# silently (it could be a `ref`, a `springref`, a `frictionloss`, an <inertial> ... that MuJoCo would honour).
_JOINT_ATTRS = {"type", "pos", "axis", "limited", "range", "armature", "damping", "stiffness", "margin", "solreflimit",
                "solimplimit", "name"}
_GEOM_ATTRS = {"type", "size", "contype", "conaffinity", "condim", "friction", "density", "margin", "gap", "solmix",
               "solref", "solimp", "fromto", "pos", "quat", "axisangle", "name"}
_MOTOR_ATTRS = {"ctrllimited", "ctrlrange", "gear", "joint", "name"}
_CONSUMED = {
    "mujoco": {"model"},
    "compiler": {"angle", "coordinate", "inertiafromgeom"},
    "option": {"timestep", "integrator", "gravity"},
    "default": set(), "worldbody": set(), "actuator": set(),
    "default/joint": _JOINT_ATTRS - {"name"}, "default/geom": _GEOM_ATTRS - {"name", "fromto", "pos", "quat", "axisangle"},
    "default/motor": {"ctrllimited", "ctrlrange", "gear"},
    "body": {"name", "pos", "quat", "axisangle"},
    "joint": _JOINT_ATTRS, "geom": _GEOM_ATTRS, "motor": _MOTOR_ATTRS,
}
# rendering / memory settings with no effect on the dynamics
_NO_DYNAMICS_ATTRS = {"geom": {"material", "rgba"}, "default/geom": {"material", "rgba"}}
_NO_DYNAMICS_ELEMENTS = {"light", "camera", "map", "visual", "asset", "texture", "material", "skybox", "quality", "headlight",
                         "global", "rgba"}
_NO_DYNAMICS_SIZE = {"nstack", "nuser_geom", "nconmax", "njmax"}
_VISUAL_INCLUDES = {"skybox.xml", "visual.xml", "materials.xml"}


def audit_mjcf(xml_path):
    """Every element / attribute of the file that compile_mjcf neither consumes nor knows to be irrelevant to the
    dynamics, as 'element@attribute' (or 'element' for a whole unknown element).  Empty list = fully understood."""
    root = ET.parse(xml_path).getroot()
    unknown = []

    def walk(e, in_default):
        tag = e.tag
        if not isinstance(tag, str):
            return
        if tag in _NO_DYNAMICS_ELEMENTS:
            return
        if tag == "include":
            if os.path.basename(e.get("file", "")) not in _VISUAL_INCLUDES:
                unknown.append("include@file=%s" % e.get("file"))
            return
        if tag == "size":
            unknown.extend("size@" + k for k in e.attrib if k not in _NO_DYNAMICS_SIZE)
            return
        ctx = "default/" + tag if in_default and tag != "default" else tag
        if ctx not in _CONSUMED:
            unknown.append(ctx)
            return
        ok = _CONSUMED[ctx] | _NO_DYNAMICS_ATTRS.get(ctx, set())
        unknown.extend("%s@%s" % (ctx, k) for k in e.attrib if k not in ok)
        if tag == "compiler" and e.get("inertiafromgeom", "true") != "true":
            unknown.append("compiler@inertiafromgeom=%s" % e.get("inertiafromgeom"))
        if tag in ("joint",) and e.get("type", "hinge") not in ("hinge", "free"):
            unknown.append("joint@type=%s" % e.get("type"))
        if tag == "geom" and not in_default and e.get("type", "sphere") not in ("plane", "sphere", "capsule"):
            unknown.append("geom@type=%s" % e.get("type"))
        for c in e:
            walk(c, in_default or tag == "default")
    walk(root, False)
    return sorted(set(unknown))


def compile_mjcf(xml_path, name=None, capsule_volume_mode="mujoco210", strict=True):
    """strict: refuse a file that contains anything this compiler would silently drop (see audit_mjcf)."""
    if strict:
        unknown = audit_mjcf(xml_path)
        if unknown:
            raise ValueError("%s: MJCF elements / attributes not understood by sgrl_amd.mjcf (would be ignored): %s"
                             % (os.path.basename(xml_path), ", ".join(unknown)))
    root = ET.parse(xml_path).getroot()
    comp = root.find("compiler")
    angle_deg = (comp.get("angle", "degree") == "degree") if comp is not None else True
    coord_global = (comp.get("coordinate", "local") == "global") if comp is not None else False
    ang = (math.pi / 180.0) if angle_deg else 1.0

    dj, dg, dm = dict(BUILTIN_JOINT), dict(BUILTIN_GEOM), dict(BUILTIN_MOTOR)
    default = root.find("default")

    def read_joint_attrs(e, base):
        d = dict(base)
        if e is None:
            return d
        for k in ("type",):
            if e.get(k) is not None:
                d[k] = e.get(k)
        if e.get("pos") is not None:
            d["pos"] = _vec(e.get("pos"), 3)
        if e.get("axis") is not None:
            d["axis"] = _vec(e.get("axis"), 3)
        if e.get("limited") is not None:
            d["limited"] = _bool(e.get("limited"))
        if e.get("range") is not None:
            d["range"] = _vec(e.get("range"), 2)
        for k in ("armature", "damping", "stiffness", "margin"):
            if e.get(k) is not None:
                d[k] = _num_prefix(e.get(k))
        if e.get("solreflimit") is not None:
            d["solreflimit"] = _merge_vec(e.get("solreflimit"), base["solreflimit"])
        if e.get("solimplimit") is not None:
            d["solimplimit"] = _merge_vec(e.get("solimplimit"), base["solimplimit"])
        return d

    def read_geom_attrs(e, base):
        d = dict(base)
        if e is None:
            return d
        if e.get("type") is not None:
            d["type"] = e.get("type")
        if e.get("size") is not None:
            d["size"] = _merge_vec(e.get("size"), [0, 0, 0])
        for k in ("contype", "conaffinity", "condim"):
            if e.get(k) is not None:
                d[k] = int(e.get(k))
        if e.get("friction") is not None:
            d["friction"] = _merge_vec(e.get("friction"), base["friction"])
        for k in ("density", "margin", "gap", "solmix"):
            if e.get(k) is not None:
                d[k] = _num_prefix(e.get(k))
        if e.get("solref") is not None:
            d["solref"] = _merge_vec(e.get("solref"), base["solref"])
        if e.get("solimp") is not None:
            d["solimp"] = _merge_vec(e.get("solimp"), base["solimp"])
        return d

    def read_motor_attrs(e, base):
        d = dict(base)
        if e is None:
            return d
        if e.get("ctrllimited") is not None:
            d["ctrllimited"] = _bool(e.get("ctrllimited"))
        if e.get("ctrlrange") is not None:
            d["ctrlrange"] = _vec(e.get("ctrlrange"), 2)
        if e.get("gear") is not None:
            d["gear"] = _vec(e.get("gear"))[0]
        return d

    if default is not None:
        dj = read_joint_attrs(default.find("joint"), dj)
        dg = read_geom_attrs(default.find("geom"), dg)
        dm = read_motor_attrs(default.find("motor"), dm)

    opt = root.find("option")
    timestep = 0.002
    integrator = INT_EULER
    gravity = [0.0, 0.0, -9.81]
    if opt is not None:
        if opt.get("timestep") is not None:
            timestep = float(opt.get("timestep"))
        if opt.get("integrator") is not None:
            integrator = INT_RK4 if opt.get("integrator").upper() == "RK4" else INT_EULER
        if opt.get("gravity") is not None:
            gravity = _vec(opt.get("gravity"), 3)

    def elem_quat(e):
        """Orientation attributes (quat | axisangle); identity when absent."""
        if e.get("quat") is not None:
            return normalize_quat(_vec(e.get("quat"), 4))
        if e.get("axisangle") is not None:
            aa = _vec(e.get("axisangle"), 4)
            return axisangle_to_quat(aa[:3], aa[3] * ang)
        return np.array([1.0, 0.0, 0.0, 0.0])

    # ---- traversal -----------------------------------------------------------------------------
    bodies = [dict(name="world", parent=-1, gpos=np.zeros(3), gquat=np.array([1.0, 0, 0, 0]),
                   pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), joints=[], geoms=[])]
    joints, geoms = [], []
    wb = root.find("worldbody")

    def add_geom(e, bidx):
        a = read_geom_attrs(e, dg)
        b = bodies[bidx]
        gt = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE}[a["type"]]
        size = list(a["size"])
        if e.get("fromto") is not None:
            ft = np.array(_vec(e.get("fromto"), 6))
            p_from, p_to = ft[:3], ft[3:]
            vec = p_from - p_to
            size[1] = 0.5 * np.linalg.norm(vec)
            pos_spec = 0.5 * (p_from + p_to)
            quat_spec = z_to_quat(vec)
        else:
            pos_spec = np.array(_vec(e.get("pos"), 3)) if e.get("pos") is not None else np.zeros(3)
            quat_spec = elem_quat(e)
        if coord_global:
            r = quat_to_mat(b["gquat"])
            pos = r.T @ (pos_spec - b["gpos"])
            quat = quat_mul(quat_conj(b["gquat"]), quat_spec)
        else:
            pos, quat = pos_spec, quat_spec
        g = dict(name=e.get("name", ""), type=gt, body=bidx, pos=pos, quat=normalize_quat(quat), size=size, attr=a)
        geoms.append(g)
        b["geoms"].append(len(geoms) - 1)

    def add_body(e, parent):
        p = bodies[parent]
        pos_spec = np.array(_vec(e.get("pos"), 3)) if e.get("pos") is not None else np.zeros(3)
        quat_spec = elem_quat(e)
        if coord_global:
            gpos, gquat = pos_spec, quat_spec
            rp = quat_to_mat(p["gquat"])
            pos = rp.T @ (gpos - p["gpos"])
            quat = quat_mul(quat_conj(p["gquat"]), gquat)
        else:
            pos, quat = pos_spec, quat_spec
            rp = quat_to_mat(p["gquat"])
            gpos = p["gpos"] + rp @ pos
            gquat = quat_mul(p["gquat"], quat)
        b = dict(name=e.get("name"), parent=parent, gpos=gpos, gquat=normalize_quat(gquat), pos=pos,
                 quat=normalize_quat(quat), joints=[], geoms=[])
        bodies.append(b)
        bidx = len(bodies) - 1
        for je in e.findall("joint"):
            a = read_joint_attrs(je, dj)
            jt = JNT_FREE if a["type"] == "free" else JNT_HINGE
            if a["type"] not in ("free", "hinge"):
                raise NotImplementedError("joint type %s" % a["type"])
            jpos = np.array(a["pos"], dtype=np.float64)
            jaxis = np.array(a["axis"], dtype=np.float64)
            if coord_global:
                r = quat_to_mat(b["gquat"])
                jpos = r.T @ (jpos - b["gpos"])
                jaxis = r.T @ jaxis
            if jt == JNT_HINGE:
                jaxis = jaxis / np.linalg.norm(jaxis)
            else:
                jpos = np.zeros(3)
                jaxis = np.array([0.0, 0.0, 1.0])
            rng = [a["range"][0] * ang, a["range"][1] * ang] if jt == JNT_HINGE else [0.0, 0.0]
            joints.append(dict(name=je.get("name"), type=jt, body=bidx, pos=jpos, axis=jaxis, attr=a, range=rng))
            b["joints"].append(len(joints) - 1)
        for ge in e.findall("geom"):
            add_geom(ge, bidx)
        for ce in e.findall("body"):
            add_body(ce, bidx)

    for ge in wb.findall("geom"):
        add_geom(ge, 0)
    top = wb.findall("body")
    if len(top) != 1:
        raise ValueError("expected exactly one top-level body")
    add_body(top[0], 0)

    m = Model()
    m.name = name or os.path.basename(xml_path)[:-4]
    m.timestep = timestep
    m.integrator = integrator
    m.gravity = np.array(gravity, dtype=np.float64)
    nbody = len(bodies)
    m.nbody = nbody
    m.body_names = [b["name"] for b in bodies]
    m.parents = [-1] + [bodies[i]["parent"] - 1 for i in range(2, nbody)]  # limb-graph parents (torso=-1)
    m.body_parent = np.array([max(b["parent"], 0) for b in bodies], dtype=np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies])
    m.body_quat = np.array([b["quat"] for b in bodies])
    m.body_limbtype = np.array([LIMB_NONE] + [limb_type_of(b["name"]) for b in bodies[1:]], dtype=np.int32)

    # joints / dofs
    njnt = len(joints)
    m.njnt = njnt
    m.joint_names = [j["name"] for j in joints]
    jq, jd = [], []
    nq = nv = 0
    for j in joints:
        jq.append(nq)
        jd.append(nv)
        if j["type"] == JNT_FREE:
            nq += 7
            nv += 6
        else:
            nq += 1
            nv += 1
    m.nq, m.nv = nq, nv
    m.jnt_type = np.array([j["type"] for j in joints], dtype=np.int32)
    m.jnt_body = np.array([j["body"] for j in joints], dtype=np.int32)
    m.jnt_qposadr = np.array(jq, dtype=np.int32)
    m.jnt_dofadr = np.array(jd, dtype=np.int32)
    m.jnt_limited = np.array([1 if (j["type"] == JNT_HINGE and j["attr"]["limited"]) else 0 for j in joints],
                             dtype=np.int32)
    m.jnt_pos = np.array([j["pos"] for j in joints])
    m.jnt_axis = np.array([j["axis"] for j in joints])
    m.jnt_range = np.array([j["range"] for j in joints])
    m.jnt_stiffness = np.array([j["attr"]["stiffness"] for j in joints], dtype=np.float64)
    m.jnt_solref = np.array([j["attr"]["solreflimit"] for j in joints], dtype=np.float64)
    m.jnt_solimp = np.array([j["attr"]["solimplimit"] for j in joints], dtype=np.float64)
    m.jnt_margin = np.array([j["attr"]["margin"] for j in joints], dtype=np.float64)

    m.body_jntadr = np.array([b["joints"][0] if b["joints"] else -1 for b in bodies], dtype=np.int32)
    m.body_jntnum = np.array([len(b["joints"]) for b in bodies], dtype=np.int32)
    dof_body, dof_jnt, dof_arm, dof_damp = [], [], [], []
    for ji, j in enumerate(joints):
        n = 6 if j["type"] == JNT_FREE else 1
        for _ in range(n):
            dof_body.append(j["body"])
            dof_jnt.append(ji)
            dof_arm.append(j["attr"]["armature"])
            dof_damp.append(j["attr"]["damping"])
    m.dof_body = np.array(dof_body, dtype=np.int32)
    m.dof_jnt = np.array(dof_jnt, dtype=np.int32)
    m.dof_armature = np.array(dof_arm, dtype=np.float64)
    m.dof_damping = np.array(dof_damp, dtype=np.float64)
    m.body_dofadr = np.array([m.jnt_dofadr[b["joints"][0]] if b["joints"] else -1 for b in bodies], dtype=np.int32)
    m.body_dofnum = np.array([sum(6 if joints[j]["type"] == JNT_FREE else 1 for j in b["joints"]) for b in bodies],
                             dtype=np.int32)
    # dof parent chain: previous dof of the same body, else last dof of the nearest ancestor with dofs
    dof_parent = []
    for d in range(nv):
        b = dof_body[d]
        if d > m.body_dofadr[b]:
            dof_parent.append(d - 1)
        else:
            p = bodies[b]["parent"]
            while p > 0 and m.body_dofnum[p] == 0:
                p = bodies[p]["parent"]
            dof_parent.append(-1 if p <= 0 else int(m.body_dofadr[p] + m.body_dofnum[p] - 1))
    m.dof_parent = np.array(dof_parent, dtype=np.int32)

    qpos0 = np.zeros(nq)
    for ji, j in enumerate(joints):
        if j["type"] == JNT_FREE:
            b = bodies[j["body"]]
            qpos0[jq[ji]:jq[ji] + 3] = b["pos"]
            qpos0[jq[ji] + 3:jq[ji] + 7] = b["quat"]
    m.qpos0 = qpos0

    # geoms + inertia
    ngeom = len(geoms)
    m.ngeom = ngeom
    m.geom_type = np.array([g["type"] for g in geoms], dtype=np.int32)
    m.geom_body = np.array([g["body"] for g in geoms], dtype=np.int32)
    m.geom_pos = np.array([g["pos"] for g in geoms])
    m.geom_quat = np.array([g["quat"] for g in geoms])
    m.geom_size = np.array([g["size"] for g in geoms], dtype=np.float64)
    m.geom_names = [g["name"] for g in geoms]

    body_mass = np.zeros(nbody)
    body_ipos = np.zeros((nbody, 3))
    body_inertia = np.zeros((nbody, 6))
    for bi, b in enumerate(bodies):
        if bi == 0:
            continue
        parts = []
        for gi in b["geoms"]:
            g = geoms[gi]
            dens = g["attr"]["density"]
            r = g["size"][0]
            if g["type"] == GEOM_SPHERE:
                mass = dens * 4.0 / 3.0 * math.pi * r ** 3
                ii = 2.0 * mass * r * r / 5.0
                diag = (ii, ii, ii)
            elif g["type"] == GEOM_CAPSULE:
                h = g["size"][1]
                mass = dens * capsule_volume(r, h, capsule_volume_mode)
                diag = _capsule_inertia(mass, r, h)
            else:
                continue
            rg = quat_to_mat(g["quat"])
            parts.append((mass, g["pos"], rg @ np.diag(diag) @ rg.T))
        mtot = sum(p[0] for p in parts)
        if mtot <= 0:
            raise ValueError("body %s has no mass" % b["name"])
        com = sum(p[0] * p[1] for p in parts) / mtot
        inert = np.zeros((3, 3))
        for mass, pos, ig in parts:
            dvec = pos - com
            inert += ig + mass * (np.dot(dvec, dvec) * np.eye(3) - np.outer(dvec, dvec))
        body_mass[bi] = mtot
        body_ipos[bi] = com
        body_inertia[bi] = [inert[0, 0], inert[1, 1], inert[2, 2], inert[0, 1], inert[0, 2], inert[1, 2]]
    m.body_mass, m.body_ipos, m.body_inertia = body_mass, body_ipos, body_inertia

    # actuators
    act = root.find("actuator")
    motors = act.findall("motor") if act is not None else []
    m.nu = len(motors)
    m.motor_joints = [mo.get("joint") for mo in motors]
    act_dof, gear, crange = [], [], []
    for mo in motors:
        a = read_motor_attrs(mo, dm)
        ji = m.joint_names.index(mo.get("joint"))
        act_dof.append(int(m.jnt_dofadr[ji]))
        gear.append(a["gear"])
        crange.append(a["ctrlrange"] if a["ctrllimited"] else [-1e30, 1e30])
    m.act_dof = np.array(act_dof, dtype=np.int32)
    m.act_gear = np.array(gear, dtype=np.float64)
    m.act_ctrlrange = np.array(crange, dtype=np.float64).reshape(-1, 2)
    # policy slot feeding each actuator (reference wrappers.py:28-46): slot 3*i+k <- joint k of limb i
    limb_joints = {}
    for li, b in enumerate(bodies[1:]):
        limb_joints[li] = [joints[j]["name"] for j in b["joints"]]
    act_slot = np.full(m.nu, -1, dtype=np.int32)
    for li in range(1, nbody - 1):
        for k, jn in enumerate(limb_joints[li][:3]):
            if jn in m.motor_joints:
                act_slot[m.motor_joints.index(jn)] = 3 * li + k
    m.act_slot = act_slot

    # contact pairs
    pg1, pg2, pcd, pmu, pmg, psr, psi = [], [], [], [], [], [], []
    for g1 in range(ngeom):
        for g2 in range(g1 + 1, ngeom):
            a1, a2 = geoms[g1]["attr"], geoms[g2]["attr"]
            b1, b2 = geoms[g1]["body"], geoms[g2]["body"]
            if b1 == b2:
                continue
            if not ((a1["contype"] & a2["conaffinity"]) or (a2["contype"] & a1["conaffinity"])):
                continue
            if b1 != 0 and b2 != 0 and (bodies[b1]["parent"] == b2 or bodies[b2]["parent"] == b1):
                continue
            t1, t2 = geoms[g1]["type"], geoms[g2]["type"]
            ga, gb = (g1, g2) if t1 <= t2 else (g2, g1)
            if geoms[ga]["type"] == GEOM_PLANE and geoms[gb]["type"] == GEOM_PLANE:
                continue
            if geoms[ga]["type"] == GEOM_SPHERE:
                raise NotImplementedError("sphere-sphere / sphere-capsule contacts are not used by the shipped XMLs")
            mix = a1["solmix"] / (a1["solmix"] + a2["solmix"])
            pg1.append(ga)
            pg2.append(gb)
            pcd.append(max(a1["condim"], a2["condim"]))
            pmu.append(max(a1["friction"][0], a2["friction"][0]))
            pmg.append(max(a1["margin"], a2["margin"]) - max(a1["gap"], a2["gap"]))
            psr.append([mix * a1["solref"][i] + (1 - mix) * a2["solref"][i] for i in range(2)])
            psi.append([mix * a1["solimp"][i] + (1 - mix) * a2["solimp"][i] for i in range(5)])
    m.npair = len(pg1)
    m.pair_g1 = np.array(pg1, dtype=np.int32)
    m.pair_g2 = np.array(pg2, dtype=np.int32)
    m.pair_condim = np.array(pcd, dtype=np.int32)
    m.pair_mu = np.array(pmu, dtype=np.float64)
    m.pair_margin = np.array(pmg, dtype=np.float64)
    m.pair_solref = np.array(psr, dtype=np.float64).reshape(-1, 2)
    m.pair_solimp = np.array(psi, dtype=np.float64).reshape(-1, 5)

    _set_invweight0(m)
    return m


# ------------------------------------------------------------------------------------------------
# qpos0 kinematics / mass matrix by Jacobians (independent of the CRBA used in the engine)
# ------------------------------------------------------------------------------------------------
def kinematics_np(m, qpos):
    nb = m.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xquat[0] = [1, 0, 0, 0]
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for b in range(1, nb):
        p = m.body_parent[b]
        j0, jn = m.body_jntadr[b], m.body_jntnum[b]
        if jn == 1 and m.jnt_type[j0] == JNT_FREE:
            qa = m.jnt_qposadr[j0]
            pos = np.array(qpos[qa:qa + 3])
            quat = normalize_quat(qpos[qa + 3:qa + 7])
            xanchor[j0] = pos
            xaxis[j0] = [0, 0, 1]
        else:
            pos = xpos[p] + quat_to_mat(xquat[p]) @ m.body_pos[b]
            quat = quat_mul(xquat[p], m.body_quat[b])
            for j in range(j0, j0 + jn):
                r = quat_to_mat(quat)
                xanchor[j] = pos + r @ m.jnt_pos[j]
                xaxis[j] = r @ m.jnt_axis[j]
                qloc = axisangle_to_quat(m.jnt_axis[j], qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]])
                quat = quat_mul(quat, qloc)
                pos = xanchor[j] - quat_to_mat(quat) @ m.jnt_pos[j]
        xpos[b] = pos
        xquat[b] = normalize_quat(quat)
    return xpos, xquat, xanchor, xaxis


def mass_matrix_np(m, qpos):
    """M(q) = sum_b m_b Jp^T Jp + Jr^T I_b Jr (+ armature), Jacobians at body COMs."""
    xpos, xquat, xanchor, xaxis = kinematics_np(m, qpos)
    nv = m.nv
    M = np.diag(m.dof_armature.astype(np.float64)).copy()
    jacs = {}
    for b in range(1, m.nbody):
        r = quat_to_mat(xquat[b])
        com = xpos[b] + r @ m.body_ipos[b]
        jp = np.zeros((3, nv))
        jr = np.zeros((3, nv))
        a = b
        while a > 0:
            for j in range(m.body_jntadr[a], m.body_jntadr[a] + m.body_jntnum[a]):
                d = m.jnt_dofadr[j]
                if m.jnt_type[j] == JNT_FREE:
                    ra = quat_to_mat(xquat[a])
                    for k in range(3):
                        jp[k, d + k] = 1.0
                        jr[:, d + 3 + k] = ra[:, k]
                        jp[:, d + 3 + k] = np.cross(ra[:, k], com - xpos[a])
                else:
                    jr[:, d] = xaxis[j]
                    jp[:, d] = np.cross(xaxis[j], com - xanchor[j])
            a = m.body_parent[a]
        ib = m.body_inertia[b]
        inert = np.array([[ib[0], ib[3], ib[4]], [ib[3], ib[1], ib[5]], [ib[4], ib[5], ib[2]]])
        iw = r @ inert @ r.T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ iw @ jr
        jacs[b] = (jp, jr)
    return M, jacs


def _set_invweight0(m):
    M, jacs = mass_matrix_np(m, m.qpos0)
    Minv = np.linalg.inv(M)
    biw = np.zeros((m.nbody, 2))
    for b in range(1, m.nbody):
        jp, jr = jacs[b]
        biw[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
        biw[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
    diw = np.zeros(m.nv)
    for j in range(m.njnt):
        d = m.jnt_dofadr[j]
        if m.jnt_type[j] == JNT_FREE:
            diw[d:d + 3] = np.mean(np.diag(Minv)[d:d + 3])
            diw[d + 3:d + 6] = np.mean(np.diag(Minv)[d + 3:d + 6])
        else:
            diw[d] = Minv[d, d]
    m.body_invweight0 = biw
    m.dof_invweight0 = diw


# ------------------------------------------------------------------------------------------------
def save_model(m, path):
    with open(path, "w") as f:
        json.dump(m.to_json(), f, separators=(",", ":"))


def load_model(path):
    with open(path) as f:
        return Model.from_json(json.load(f))


ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "models")


def load_asset(name):
    """Load a pre-compiled morphology shipped with the package (compiled from the reference XMLs)."""
    p = os.path.join(ASSET_DIR, name + ".json")
    if not os.path.exists(p):
        raise FileNotFoundError("no compiled model %r under %s" % (name, ASSET_DIR))
    return load_model(p)


def list_assets():
    return sorted(f[:-5] for f in os.listdir(ASSET_DIR) if f.endswith(".json"))
