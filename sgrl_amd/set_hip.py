"""Python front-end of the HIP SET-actor forward (C ABI: include/sgrl_set.h).

`HipSetActor` packs the parameters of an `SEPolicy` (set_policy.py, reference-compatible state_dict) into the flat
device buffer the kernels expect, re-packing automatically when the parameters change (optimizer steps bump the
tensors' version counters), describes the batch structure (morphologies x env counts) to the engine and runs
`actions = max_action * tanh(actor(obs))` for all environments in one call.  No CPU fallback: a missing extension
or device raises `SgrlError`.
"""
import ctypes

import numpy as np
import torch

from . import _lib

NGLOBAL, NLAYER, LAYERS = 25, 30, 3
NW = NGLOBAL + LAYERS * NLAYER


def _bind(L):
    if getattr(L, "_set_bound", False):
        return
    vp = ctypes.c_void_p
    L.sgrl_set_create.argtypes = [ctypes.POINTER(vp)]
    L.sgrl_set_destroy.argtypes = [vp]
    L.sgrl_set_destroy.restype = None
    L.sgrl_set_weights.argtypes = [vp, vp, vp, ctypes.c_int]
    L.sgrl_set_graph.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp]
    L.sgrl_set_forward.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_float, vp]
    L.sgrl_set_forward_q.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp]
    L.sgrl_set_time_forward.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, vp,
                                        ctypes.POINTER(ctypes.c_float)]
    L.sgrl_set_num_nodes.argtypes = [vp]
    L.sgrl_set_workspace_bytes.argtypes = [vp]
    L.sgrl_set_workspace_bytes.restype = ctypes.c_int64
    L.sgrl_set_peek.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int64]
    L.sgrl_set_last_error.restype = ctypes.c_char_p
    L._set_bound = True


def _check(L, rc, what):
    if rc != 0:
        raise _lib.SgrlError("%s failed (%d): %s" % (what, rc, L.sgrl_set_last_error().decode()))


def fold_gram_weight(w):
    """[out, 1024] weight acting on vec(G) of a symmetric 32x32 G -> [out, 544] acting on its packed lower triangle."""
    out_f = w.shape[0]
    w3 = w.reshape(out_f, 32, 32)
    sym = w3 + w3.transpose(1, 2)
    idx_a, idx_b = torch.tril_indices(32, 32)          # row-major lower triangle: k = a(a+1)/2 + b
    f = sym[:, idx_a, idx_b].clone()
    diag = idx_a == idx_b
    f[:, diag] = w3[:, idx_a[diag], idx_b[diag]]
    return torch.cat([f, f.new_zeros(out_f, 544 - f.shape[1])], dim=1).contiguous()


def pack_tensors(sd, prefix="actor.", critic=False):
    """[(tensor float32, ...)] in slot order (include/sgrl_set.h) from a state_dict-like mapping of torch tensors.
    critic=True packs a critic TransformerModel (scalar head: decoder_ng in the DECG / L1M_B slots, see sgrl_set.h)."""
    g = lambda k: sd[prefix + k].detach().float()
    out = [None] * NW
    out[0:3] = [g("pos_encoder.embeddings.%d.weight" % i) for i in range(3)]
    out[3], out[4] = g("transformer_encoder.rel_encoder.weight"), g("transformer_encoder.rel_encoder.bias")
    out[5], out[6] = g("transformer_encoder.norm.weight"), g("transformer_encoder.norm.bias")
    out[7], out[8], out[9] = g("g_encoder.weight"), g("encoder.weight"), g("encoder.bias")
    out[10] = g("gg_proj.weight")
    out[11], out[12], out[13], out[14] = fold_gram_weight(g("linear1_g.weight")), g("linear1_g.bias"), g("linear2_g.weight"), g("linear2_g.bias")
    w = g("linear1_ng.weight")
    out[15] = torch.cat([w, w.new_zeros(w.shape[0], 160 - w.shape[1])], dim=1)
    out[16], out[17], out[18] = g("linear1_ng.bias"), g("linear2_ng.weight"), g("linear2_ng.bias")
    if critic:
        z = w.new_zeros(1)
        out[19], out[21] = g("decoder_ng.weight").reshape(-1), g("decoder_ng.bias").reshape(-1)
        out[20], out[22], out[23], out[24] = z, z, z, z
    else:
        out[19] = g("decoder_g.weight").reshape(-1)
        out[20], out[21], out[22], out[23] = g("linear1_m.weight"), g("linear1_m.bias"), g("linear2_m.weight"), g("linear2_m.bias")
        out[24] = g("g_proj.weight")
    scaling = float(128) ** -0.5   # (2 * head_dim)^-0.5, reference subequivariant_attentions.py:88
    for l in range(LAYERS):
        p = "transformer_encoder.layers.%d." % l
        a = p + "self_attn."
        b = NGLOBAL + l * NLAYER
        vg = g(a + "vg_proj.weight")
        out[b:b + NLAYER] = [
            g(a + "g_proj.weight"), fold_gram_weight(g(a + "linear_g1.weight")), g(a + "linear_g1.bias"), g(a + "linear_g2.weight"),
            g(a + "linear_g2.bias"),
            torch.cat([g(a + "q_proj.weight") * scaling, g(a + "k_proj.weight"), g(a + "v_proj.weight")], 0),
            torch.cat([g(a + "q_proj.bias") * scaling, g(a + "k_proj.bias"), g(a + "v_proj.bias")], 0),
            torch.cat([vg, vg.new_zeros(256 - vg.shape[0], vg.shape[1])], 0),
            g(a + "ng_out.weight"), g(a + "ng_out.bias"), g(a + "g_out.weight"),
            g(p + "g_proj2.weight"), g(p + "g_proj3.weight"), fold_gram_weight(g(p + "linear_g1.weight")), g(p + "linear_g1.bias"),
            g(p + "linear_g2.weight"), g(p + "linear_g2.bias"), g(p + "linear3.weight"), g(p + "linear3.bias"),
            g(p + "linear4.weight"), g(p + "linear4.bias"), g(p + "linear5.weight"), g(p + "linear1.weight"),
            g(p + "linear1.bias"), g(p + "linear2.weight"), g(p + "linear2.bias"), g(p + "norm1.weight"),
            g(p + "norm1.bias"), g(p + "norm2.weight"), g(p + "norm2.bias")]
    return out


class HipSetActor(object):
    """HIP forward of one SET network: the actor of an `SEPolicy` (default) or, with `net=` / `critic=True`, one critic
    `TransformerModel` of an `SECritic` (see `HipSetCritic`)."""

    def __init__(self, policy, device=None, net=None, critic=False):
        if not torch.cuda.is_available():
            raise _lib.SgrlError("HipSetActor needs an MI355X (no CPU fallback)")
        self.L = _lib.lib()
        _bind(self.L)
        self.policy = policy
        self.net = net if net is not None else policy.actor
        self.critic = bool(critic)
        self.device = torch.device(device) if device is not None else next(self.net.parameters()).device
        if self.device.type != "cuda":
            raise _lib.SgrlError("the SEPolicy must live on the GPU for the HIP path")
        h = ctypes.c_void_p()
        _check(self.L, self.L.sgrl_set_create(ctypes.byref(h)), "sgrl_set_create")
        self.h = h
        self._wbuf = None
        self._wver = None
        self._cfg_key = None
        self.n_env = 0
        self.act_ld = 0
        self._single_cache = {}

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.sgrl_set_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------------
    def sync_weights(self, force=False):
        params = list(self.net.parameters())
        ver = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        if not force and ver == self._wver:
            return
        sd = {"actor." + k: v for k, v in self.net.state_dict().items()}
        tens = pack_tensors(sd, critic=self.critic)
        offs = np.zeros(NW, dtype=np.int64)
        pos = 0
        for i, t in enumerate(tens):
            offs[i] = pos
            pos += (t.numel() + 63) // 64 * 64    # keep every tensor 256-byte aligned
        buf = torch.zeros(pos, dtype=torch.float32, device=self.device)
        for i, t in enumerate(tens):
            buf[offs[i]:offs[i] + t.numel()] = t.reshape(-1).to(self.device)
        torch.cuda.synchronize(self.device)
        self._wbuf = buf
        _check(self.L, self.L.sgrl_set_weights(self.h, ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(offs.ctypes.data),
                                               NW), "sgrl_set_weights")
        self._wver = ver

    # ---- batch structure ------------------------------------------------------------------------------
    def configure(self, graphs, counts):
        """graphs: per-morphology dicts with 'traversals' (3 index vectors) and 'relation' [L,L,3]; counts: envs each."""
        key = (tuple(id(g) for g in graphs), tuple(int(c) for c in counts))
        if key == self._cfg_key:
            return
        Ls, trav, rel = [], [], []
        for g in graphs:
            t = [np.asarray(v.cpu() if torch.is_tensor(v) else v, dtype=np.int32) for v in g["traversals"]]
            Ls.append(len(t[0]))
            trav.append(np.concatenate(t))
            r = g["relation"]
            rel.append(np.asarray(r.detach().cpu() if torch.is_tensor(r) else r, dtype=np.float32).reshape(-1))
        Ls = np.asarray(Ls, dtype=np.int32)
        cnt = np.asarray(counts, dtype=np.int32)
        trav = np.ascontiguousarray(np.concatenate(trav), dtype=np.int32)
        rel = np.ascontiguousarray(np.concatenate(rel), dtype=np.float32)
        vp = lambda a: ctypes.c_void_p(a.ctypes.data)
        _check(self.L, self.L.sgrl_set_graph(self.h, len(Ls), vp(Ls), vp(cnt), vp(trav), vp(rel)), "sgrl_set_graph")
        self._cfg_key = key
        self.n_env = int(cnt.sum())
        self.max_limbs = int(Ls.max())
        self.num_nodes = self.L.sgrl_set_num_nodes(self.h)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def forward_batch(self, obs, out=None, act_ld=None):
        """obs: float32 CUDA [n_env, obs_ld] (rows zero padded beyond 41*L) -> actions float32 [n_env, act_ld]."""
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.dim() == 2 and obs.stride(1) == 1
        assert obs.shape[0] == self.n_env
        self.sync_weights()
        act_ld = act_ld or 3 * self.max_limbs
        if out is None:
            out = torch.empty((self.n_env, act_ld), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.shape == (self.n_env, act_ld)
        _check(self.L, self.L.sgrl_set_forward(self.h, ctypes.c_void_p(obs.data_ptr()), int(obs.stride(0)),
                                               ctypes.c_void_p(out.data_ptr()), int(act_ld),
                                               ctypes.c_float(float(self.policy.max_action)), self._stream()),
               "sgrl_set_forward")
        return out

    def forward_q(self, obs, action, out=None, q_ld=None):
        """critic network: obs [n_env, obs_ld], action [n_env, act_ld] (3 slots per limb) -> per-limb Q [n_env, q_ld]."""
        assert self.critic, "forward_q needs a handle created with critic=True"
        for t in (obs, action):
            assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == self.n_env
        self.sync_weights()
        q_ld = q_ld or self.max_limbs
        if out is None:
            out = torch.empty((self.n_env, q_ld), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.shape == (self.n_env, q_ld)
        _check(self.L, self.L.sgrl_set_forward_q(self.h, ctypes.c_void_p(obs.data_ptr()), int(obs.stride(0)),
                                                 ctypes.c_void_p(action.data_ptr()), int(action.stride(0)),
                                                 ctypes.c_void_p(out.data_ptr()), int(q_ld), self._stream()),
               "sgrl_set_forward_q")
        return out

    def time_forward(self, obs, out, reps):
        self.sync_weights()
        ms = ctypes.c_float(0)
        _check(self.L, self.L.sgrl_set_time_forward(self.h, ctypes.c_void_p(obs.data_ptr()), int(obs.stride(0)),
                                                    ctypes.c_void_p(out.data_ptr()), int(out.stride(0)),
                                                    ctypes.c_float(float(self.policy.max_action)), int(reps),
                                                    self._stream(), ctypes.byref(ms)), "sgrl_set_time_forward")
        return float(ms.value)

    def forward_single(self, state, graph):
        """SEPolicy.forward(state [B, 41*L]) for the current morphology (reference agent.py:197)."""
        B = state.shape[0]
        self.configure([graph], [B])
        return self.forward_batch(state.contiguous().float(), act_ld=3 * len(graph["parents"]))

    def peek(self, which, per_node):
        out = np.zeros((self.num_nodes, per_node), dtype=np.float32)
        _check(self.L, self.L.sgrl_set_peek(self.h, which, ctypes.c_void_p(out.ctypes.data), out.size), "sgrl_set_peek")
        return out


class HipSetCritic(object):
    """Twin critics of an `SECritic` on the HIP path (inference only: the TD3 target values, reference agent.py:136-148):
    two handles, one per TransformerModel, sharing the batch structure."""

    def __init__(self, critic_module, device=None):
        self.q1 = HipSetActor(critic_module, device=device, net=critic_module.critic1, critic=True)
        self.q2 = HipSetActor(critic_module, device=device, net=critic_module.critic2, critic=True)

    def configure(self, graphs, counts):
        self.q1.configure(graphs, counts)
        self.q2.configure(graphs, counts)

    def forward_batch(self, obs, action, q_ld=None, which=(1, 2)):
        out = []
        if 1 in which:
            out.append(self.q1.forward_q(obs, action, q_ld=q_ld))
        if 2 in which:
            out.append(self.q2.forward_q(obs, action, q_ld=q_ld))
        return tuple(out)

    def forward_single(self, state, action, graph, which=(1, 2)):
        """SECritic.forward(state [B, 41 L], action [B, 3 L]) for one morphology -> per-limb Q [B, L] each."""
        B, L = state.shape[0], len(graph["parents"])
        self.configure([graph], [B])
        return self.forward_batch(state.contiguous().float(), action.contiguous().float(), q_ld=L, which=which)
