"""Limb-graph preprocessing for the SET actor (host side, init time).

Mirrors the *interface* of the reference's graph helpers so callers read the same:
  getGraphStructure / getGraphJoints / getMotorJoints   reference src/utils.py:236-330
  getChildrens / lcrs / getTraversal                    reference src/utils.py:348-409
  getAdjacency / getGraphTransition / PPR / getDistance reference src/utils.py:333-346,411-447
  getGraphDict                                          reference src/utils.py:449-484
  findMaxChildren                                       reference src/utils.py:26-32
  action_order_for                                      reference src/wrappers.py:28-37

Implementation is independent (ElementTree walk + NumPy float64, cast once to float32); integer
outputs (parents, traversals, adjacency, mask pattern, action_order) are bit-exact against
tests/golden/graphs.json, float outputs within 1e-6.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

try:  # torch is only needed for getGraphDict's tensor outputs
    import torch
except Exception:  # pragma: no cover
    torch = None


# ------------------------------------------------------------------------------------------------
# MJCF structure readers
# ------------------------------------------------------------------------------------------------
def _root_body(xml_file):
    tree = ET.parse(xml_file) if not isinstance(xml_file, ET.Element) else None
    root = tree.getroot() if tree is not None else xml_file
    wb = root.find("worldbody")
    if wb is None:
        raise Exception("The given xml file does not follow the standard MuJoCo format.")
    bodies = wb.findall("body")
    if len(bodies) != 1:
        raise Exception("worldbody can only contain one body (torso) for the current implementation, "
                        "but found {}".format(len(bodies)))
    return root, bodies[0]


def _preorder_bodies(body):
    """Yield (element, parent_index) in document pre-order."""
    out = []

    def rec(b, parent):
        idx = len(out)
        out.append((b, parent))
        for c in b.findall("body"):
            rec(c, idx)
    rec(body, -1)
    return out


def getGraphStructure(xml_file, graph_type="morphology"):
    """Parents list (pre-order, root = -1) of the limb tree.  reference utils.py:236-276"""
    _, torso = _root_body(xml_file)
    parents = [p for _, p in _preorder_bodies(torso)]
    base = os.path.basename(xml_file) if isinstance(xml_file, str) else ""
    if "walker" in base and "flipped" in base:
        parents[0] = -2
    if graph_type == "tree":
        parents[1:] = [0] * (len(parents) - 1)
    elif graph_type == "line":
        parents[1:] = list(range(len(parents) - 1))
    return parents


def getGraphJoints(xml_file):
    """[[body_name, joint_name...], ...] in pre-order, bodies without joints skipped.  utils.py:279-316"""
    _, torso = _root_body(xml_file)
    res = []
    for b, _ in _preorder_bodies(torso):
        js = b.findall("joint")
        if js:
            res.append([b.get("name")] + [j.get("name") for j in js])
    return res


def getMotorJoints(xml_file):
    """Joint names in actuator order.  utils.py:319-330"""
    root, _ = _root_body(xml_file)
    act = root.find("actuator")
    return [m.get("joint") for m in act.findall("motor")]


def getBodyNames(xml_file):
    _, torso = _root_body(xml_file)
    return [b.get("name") for b, _ in _preorder_bodies(torso)]


def action_order_for(joints, motors):
    """Policy slot -> actuator index (-1 for the 3 torso slots).  reference wrappers.py:28-37"""
    order = [-1, -1, -1] * len(joints)
    for i in range(1, len(joints)):
        for k in range(3):
            order[3 * i + k] = motors.index(joints[i][1 + k])
    return order


# ------------------------------------------------------------------------------------------------
# tree helpers
# ------------------------------------------------------------------------------------------------
def getChildrens(parents):
    n = len(parents)
    ch = [[] for _ in range(n)]
    for node in range(n):
        p = parents[node]
        if 0 <= p < node + 1 and p != node:
            ch[p].append(node)
    return ch


def lcrs(graph):
    """Left-child right-sibling binary form of a children-list tree."""
    out = [[] for _ in graph]
    for node, kids in enumerate(graph):
        if not kids:
            continue
        out[node].insert(0, kids[0])
        prev = kids[0]
        for sib in kids[1:]:
            out[prev].append(sib)
            prev = sib
    return out


def _inorder(children):
    order = []
    stack = [(0, 0)]
    # recursive definition kept explicit: left, node, right (right only when exactly two children)
    def visit(n):
        if children[n]:
            visit(children[n][0])
        order.append(n)
        if len(children[n]) == 2:
            visit(children[n][1])
    del stack
    visit(0)
    return order


def _postorder(children):
    order = []

    def visit(n):
        for c in children[n]:
            visit(c)
        order.append(n)
    visit(0)
    return order


def getTraversal(parents, traversal_types, device=None):
    """Position of every node in each requested traversal.  reference utils.py:368-409"""
    children = getChildrens(parents)
    n = len(children)
    res = []
    for t in traversal_types:
        if t == "pre":
            idx = list(range(n))
        else:
            if t == "inlcrs":
                trav = _inorder(lcrs(children))
            elif t == "postlcrs":
                trav = _postorder(lcrs(children))
            else:
                raise ValueError("unknown traversal type %r" % t)
            pos = {node: k for k, node in enumerate(trav)}
            idx = [pos[i] for i in range(n)]
        if device is not None:
            idx = torch.as_tensor(idx, dtype=torch.long, device=device)
        res.append(idx)
    return res


def _adjacency_np(parents):
    n = len(parents)
    a = np.zeros((n, n), dtype=np.float64)
    for i, p in enumerate(parents):
        if p >= 0:
            a[i, p] = 1.0
            a[p, i] = 1.0
    return a


def _transition_np(adj, self_loop=True):
    a = adj + np.eye(len(adj)) if self_loop else adj
    return (a / a.sum(1, keepdims=True)).T


def _ppr_np(transition, damping=0.9):
    n = len(transition)
    inv = np.linalg.inv(np.eye(n) - damping * transition)
    # column i of (1-d)*inv is PPR started at i; the reference stacks them and transposes
    return ((1.0 - damping) * inv).T


def _distance_np(adj):
    n = len(adj)
    dist = np.full((n, n), -1, dtype=np.int64)
    for s in range(n):
        dist[s, s] = 0
        frontier = [s]
        d = 0
        while frontier:
            d += 1
            nxt = []
            for v in frontier:
                for u in range(n):
                    if adj[v, u] and dist[s, u] < 0:
                        dist[s, u] = d
                        nxt.append(u)
            frontier = nxt
    return dist.astype(np.float64) / n


def getAdjacency(parents):
    return torch.from_numpy(_adjacency_np(parents)).float()


def getGraphTransition(adjacency, self_loop=True):
    a = adjacency.double().numpy() if torch is not None and isinstance(adjacency, torch.Tensor) else np.asarray(adjacency)
    return torch.from_numpy(_transition_np(a, self_loop)).float()


def PPR(transition, start=None, damping=0.9, max_iter=1000):
    t = transition.double().numpy() if isinstance(transition, torch.Tensor) else np.asarray(transition, dtype=np.float64)
    n = len(t)
    if damping == 1:
        s = np.full((n, 1), 1.0 / n) if start is None else np.eye(n)[start].reshape(n, 1)
        prev = np.full((n, 1), 1.0 / n)
        for _ in range(max_iter):
            cur = damping * t @ prev + (1 - damping) * s
            if (np.abs(cur - prev) < 1e-8).all():
                break
            prev = cur
        return torch.from_numpy(cur).float()
    full = _ppr_np(t, damping)
    if start is None:
        return torch.from_numpy(full.T.mean(1, keepdims=True)).float()
    return torch.from_numpy(full[start].reshape(n, 1)).float()


def getDistance(adjacency):
    a = adjacency.numpy() if torch is not None and isinstance(adjacency, torch.Tensor) else np.asarray(adjacency)
    return _distance_np(a)


def graph_arrays(parents, ppr_damping=0.9, self_loop=True):
    """NumPy-only version of getGraphDict's numeric content (used by the HIP weight/graph packer)."""
    adj = _adjacency_np(parents)
    n = len(parents)
    trans = _transition_np(adj, self_loop)
    reach = adj + np.eye(n)
    mask = np.where(reach == 0, -np.inf, 0.0)
    deg = adj.sum(1)
    with np.errstate(divide="ignore"):
        dm = np.diag(deg ** -0.5)
    sym_lap = dm @ (np.diag(deg) - adj) @ dm
    dist = _distance_np(adj)
    ppr = _ppr_np(trans, ppr_damping)
    rel = np.stack([ppr, sym_lap, dist], axis=2)
    return {
        "adjacency": adj.astype(np.float32), "transition": trans.astype(np.float32),
        "mask": mask.astype(np.float32), "sym_lap": sym_lap.astype(np.float32),
        "distance": dist.astype(np.float32), "ppr": ppr.astype(np.float32),
        "relation": rel.astype(np.float32),
    }


def getGraphDict(parents, trav_types=[], rel_types=[], self_loop=True, ppr_damping=0.9, device=None):
    """Graph dictionary consumed by SEPolicy.change_morphology.  reference utils.py:449-484"""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if len(parents) == 1:
        return {"parents": parents}
    arr = graph_arrays(parents, ppr_damping, self_loop)
    gd = {"parents": parents, "traversals": getTraversal(parents, trav_types, device)}
    for k in ("ppr", "transition", "adjacency", "distance", "sym_lap", "mask", "relation"):
        gd[k] = torch.from_numpy(arr[k]).to(device)
    return gd


def findMaxChildren(env_names, graphs):
    best = 0
    for name in env_names:
        g = list(graphs[name])
        most = max(g, key=g.count)
        best = max(best, g.count(most))
    return best


def quat2mat(q):
    """Rotation matrix of a unit quaternion (w, x, y, z).  reference utils.py:82-104"""
    w, x, y, z = q
    return np.array([
        [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
        [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
        [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y],
    ])
