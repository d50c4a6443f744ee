"""SMP (shared modular policies: message passing along the limb tree) actor / critic behind the reference's module
surfaces.

`ActorGraphPolicy` / `CriticGraphPolicy` keep the constructor signatures, `forward`, `Q1`, `change_morphology` and the
`state_dict()` keys of reference src/ModularActor.py:99-384 / src/ModularCritic.py:143-520 (`sNet.<i>.*`,
`actor.<i>.*` / `critic.<i>.*`: ONE shared module listed once per limb, exactly as the reference's
`nn.ModuleList([module] * num_limbs)` shows it).  Only the reference's `disable_fold` code path exists here (torchfold's
dynamic batching is replaced by what it was a workaround for: the limbs of one tree DEPTH are evaluated as one batched
call of the shared module -- bottom-up from the deepest level, top-down from the root).  Message passing modes: `bu and td`
(both ways, the published SMP) and `td` only -- the two the reference's `disable_fold` path can run (without top-down
messages its forward raises at ModularActor.py:244, `torch.stack` of a list of None).
Plain differentiable PyTorch -- this baseline has no HIP fast path (the SET model is the one the north star names); outputs
are pinned to fixtures produced by executing the reference's own modules (tests/golden/smp_forward.npz,
tools/capture_golden_smp.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MLPBase(nn.Module):
    """reference utils.MLPBase (utils.py:222-233): 400 / 300 hidden units, ReLU."""

    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.l1 = nn.Linear(num_inputs, 400)
        self.l2 = nn.Linear(400, 300)
        self.l3 = nn.Linear(300, num_outputs)

    def forward(self, x):
        return self.l3(F.relu(self.l2(F.relu(self.l1(x)))))


def _up_message(mod, x, m):
    """fc1 -> normalise -> [., m] -> tanh -> fc2 -> tanh -> fc3 -> normalise (ModularActor.py:35-47)."""
    h = F.normalize(mod.fc1(x), dim=-1)
    h = torch.tanh(torch.cat([h, m], dim=-1))
    h = torch.tanh(mod.fc2(h))
    return F.normalize(mod.fc3(h), dim=-1)


# ---- actor node modules (parameter names of ModularActor.py:12-96) ------------------------------------------------------
class ActorUp(nn.Module):
    def __init__(self, state_dim, msg_dim, max_children):
        super().__init__()
        self.fc1 = nn.Linear(state_dim, 64)
        self.fc2 = nn.Linear(64 + msg_dim * max_children, 64)
        self.fc3 = nn.Linear(64, msg_dim)

    def forward(self, x, m):
        return _up_message(self, x, m)


class ActorDownAction(nn.Module):
    def __init__(self, self_input_dim, action_dim, msg_dim, max_action, max_children):
        super().__init__()
        self.max_action = max_action
        self.action_base = MLPBase(self_input_dim + msg_dim, action_dim)
        self.msg_base = MLPBase(self_input_dim + msg_dim, msg_dim * max_children)

    def forward(self, x, m):
        xm = torch.tanh(torch.cat([x, m], dim=-1))
        return self.max_action * torch.tanh(self.action_base(xm)), F.normalize(self.msg_base(xm), dim=-1)


# ---- critic node modules (ModularCritic.py:11-140) ----------------------------------------------------------------------
class CriticUp(nn.Module):
    def __init__(self, state_dim, action_dim, msg_dim, max_children):
        super().__init__()
        self.fc1 = nn.Linear(state_dim + action_dim, 64)
        self.fc2 = nn.Linear(64 + msg_dim * max_children, 64)
        self.fc3 = nn.Linear(64, msg_dim)

    def forward(self, x, u, m):
        return _up_message(self, torch.cat([x, u], dim=-1), m)


class CriticDownAction(nn.Module):
    def __init__(self, self_input_dim, action_dim, msg_dim, max_children):
        super().__init__()
        self.baseQ1 = MLPBase(self_input_dim + action_dim + msg_dim, 1)
        self.baseQ2 = MLPBase(self_input_dim + action_dim + msg_dim, 1)
        self.msg_base = MLPBase(self_input_dim + msg_dim, msg_dim * max_children)

    def forward(self, x, u, m, twin=True):
        xum = torch.cat([x, u, m], dim=-1)
        msg_down = F.normalize(self.msg_base(torch.tanh(torch.cat([x, m], dim=-1))), dim=-1)
        return self.baseQ1(xum), (self.baseQ2(xum) if twin else None), msg_down


# ---- the tree schedule ----------------------------------------------------------------------------------------------------
class _Tree(object):
    """Evaluation order of one morphology: limbs grouped by depth, the children slots of every limb, and the slot a limb
    occupies in its parent's outgoing message (ModularActor.py:283-326)."""

    def __init__(self, parents, max_children):
        parents = [int(p) for p in parents]
        L = len(parents)
        self.L = L
        depth = [0] * L
        for i in range(1, L):
            depth[i] = depth[parents[i]] + 1 if parents[i] >= 0 else 0
        self.levels = [[i for i in range(L) if depth[i] == d] for d in range(max(depth) + 1)]
        self.children = []
        for i in range(L):
            ch = [j for j, p in enumerate(parents) if p == i]
            assert len(ch) <= max_children, "limb %d has %d children, max_children is %d" % (i, len(ch), max_children)
            self.children.append(ch + [-1] * (max_children - len(ch)))
        self.slot = []
        for i in range(L):
            k = parents[:i].count(parents[i])
            if parents[0] == -2 and i == 1:      # flipped structure: message order mirrored at the root
                k = (max_children - 1) - k
            self.slot.append(k)
        self.parents = parents


class _GraphModule(nn.Module):
    def _init_common(self, state_dim, action_dim, msg_dim, batch_size, max_children, disable_fold, td, bu):
        if not disable_fold:
            raise NotImplementedError("the torchfold path of the reference is not rebuilt: construct with disable_fold=True "
                                      "(limbs of one tree depth are batched instead)")
        self.num_limbs = 1
        self.msg_dim, self.batch_size, self.max_children, self.disable_fold = msg_dim, batch_size, max_children, disable_fold
        self.state_dim, self.action_dim = state_dim, action_dim
        if not td:
            raise NotImplementedError("SMP without top-down messages: the reference's disable_fold path cannot run it "
                                      "(ModularActor.py:244 stacks a list of None); modes built: td, td + bu")
        self.td, self.bu = td, bu
        self.parents = [-1]
        self._tree = _Tree(self.parents, max_children)

    def _relist(self, names):
        for n in names:
            if hasattr(self, n):
                setattr(self, n, nn.ModuleList([getattr(self, n)[0]] * self.num_limbs))

    def change_morphology(self, graph):
        self.graph = graph
        self.parents = [int(p) for p in graph["parents"]]
        self.num_limbs = len(self.parents)
        self._tree = _Tree(self.parents, self.max_children)
        self._relist(self._lists)

    def _split(self, t, dim):
        assert t.shape[1] == dim * self.num_limbs, \
            "state.shape[1] expects {} but got {} with num_limbs being {} and state_dim being {}".format(
                dim * self.num_limbs, t.shape[1], self.num_limbs, dim)
        return t.reshape(t.shape[0], self.num_limbs, dim).transpose(0, 1)       # [L, B, dim]

    def _bottom_up(self, fn):
        """fn(nodes, msg_in [n, B, msg_dim * max_children]) -> msg_up [n, B, msg_dim]; deepest level first."""
        tr = self._tree
        up = [None] * tr.L
        for level in reversed(tr.levels):
            B = self._B
            zero = torch.zeros((B, self.msg_dim), device=self._dev, dtype=self._dt)
            m = torch.stack([torch.cat([up[c] if c >= 0 else zero for c in tr.children[i]], dim=-1) for i in level])
            out = fn(level, m)
            for k, i in enumerate(level):
                up[i] = out[k]
        return up

    def _top_down(self, fn):
        """fn(nodes, msg_in [n, B, msg_dim]) -> msg_down [n, B, msg_dim * max_children]; root level first."""
        tr = self._tree
        down = [None] * tr.L
        for level in tr.levels:
            zero = torch.zeros((self._B, self.msg_dim * self.max_children), device=self._dev, dtype=self._dt)
            ms = []
            for i in level:
                pm = down[tr.parents[i]] if tr.parents[i] >= 0 else zero
                ms.append(pm[:, tr.slot[i] * self.msg_dim:(tr.slot[i] + 1) * self.msg_dim])
            out = fn(level, torch.stack(ms))
            for k, i in enumerate(level):
                down[i] = out[k]
        return down


class ActorGraphPolicy(_GraphModule):
    """Drop-in for reference ModularActor.ActorGraphPolicy (constructor of ModularActor.py:102-115)."""
    _lists = ("sNet", "actor")

    def __init__(self, state_dim, action_dim, msg_dim, batch_size, max_action, max_children, disable_fold, td, bu,
                 args=None, device=None):
        super().__init__()
        self._init_common(state_dim, action_dim, msg_dim, batch_size, max_children, disable_fold, td, bu)
        self.max_action = max_action
        if bu:
            self.sNet = nn.ModuleList([ActorUp(state_dim, msg_dim, max_children)])
        self.actor = nn.ModuleList([ActorDownAction(msg_dim if bu else state_dim, action_dim, msg_dim, max_action, max_children)])
        if device is not None:
            self.to(device)

    def clear_buffer(self):
        self.action = None

    def forward(self, state, mode="train"):
        x = self._split(state, self.state_dim)
        self._B, self._dev, self._dt = state.shape[0], state.device, state.dtype
        act = [None] * self.num_limbs
        if self.bu:
            up = self._bottom_up(lambda level, m: self.sNet[0](x[level], m))

        def down_fn(level, m):
            # both ways: a limb's own input is the message it sent up (ModularActor.py:294-297)
            a, msg = self.actor[0](torch.stack([up[i] for i in level]) if self.bu else x[level], m)
            for k, i in enumerate(level):
                act[i] = a[k]
            return msg
        self._top_down(down_fn)
        self.action = torch.stack(act, dim=1).reshape(state.shape[0], -1)       # [B, L, A] -> [B, L * A]
        return self.action


class CriticGraphPolicy(_GraphModule):
    """Drop-in for reference ModularCritic.CriticGraphPolicy: twin Q values, per-limb outputs summed over the limbs -> [B, 1]
    each (ModularCritic.py:286-290)."""
    _lists = ("sNet", "critic")

    def __init__(self, state_dim, action_dim, msg_dim, batch_size, max_children, disable_fold, td, bu, args=None,
                 device=None):
        super().__init__()
        self._init_common(state_dim, action_dim, msg_dim, batch_size, max_children, disable_fold, td, bu)
        if bu:
            self.sNet = nn.ModuleList([CriticUp(state_dim, action_dim, msg_dim, max_children)])
        self.critic = nn.ModuleList([CriticDownAction(msg_dim if bu else state_dim, action_dim, msg_dim, max_children)])
        if device is not None:
            self.to(device)

    def clear_buffer(self):
        self.x1 = self.x2 = None

    def _run(self, state, action, twin):
        x, u = self._split(state, self.state_dim), self._split(action, self.action_dim)
        self._B, self._dev, self._dt = state.shape[0], state.device, state.dtype
        L = self.num_limbs
        q1, q2 = [None] * L, [None] * L
        if self.bu:
            up = self._bottom_up(lambda level, m: self.sNet[0](x[level], u[level], m))

        def down_fn(level, m):
            xs = torch.stack([up[i] for i in level]) if self.bu else x[level]
            a, b, msg = self.critic[0](xs, u[level], m, twin=twin)
            for k, i in enumerate(level):
                q1[i] = a[k]
                q2[i] = b[k] if twin else None
            return msg
        self._top_down(down_fn)
        self.x1 = torch.stack(q1, dim=-1).sum(dim=-1).reshape(state.shape[0], -1)
        self.x2 = torch.stack(q2, dim=-1).sum(dim=-1).reshape(state.shape[0], -1) if twin else None
        return self.x1, self.x2

    def forward(self, state, action):
        return self._run(state, action, True)

    def Q1(self, state, action):
        return self._run(state, action, False)[0]
