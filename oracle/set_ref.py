"""oracle/set_ref.py -- TEST INFRASTRUCTURE: NumPy restatement of the SET actor forward.

Follows, function by function:
  SEPolicy.forward                      reference src/SEActor.py:334-347
  TransformerModel.forward              reference src/SEActor.py:237-287
  RepeatTransformerEncoder.forward      reference src/SEActor.py:138-167
  MyTransformerEncoderLayer.forward     reference src/SEActor.py:82-125
  multi_head_attention_forward          reference src/subequivariant_attentions.py:4-154
  ConcatPositionalEmbedding.forward     reference src/SEActor.py:29-31

Pinned by tests/test_oracle_set.py against tests/golden/set_forward.npz and set_probes_walker7.npz, which were
produced by executing the reference SEPolicy (tools/capture_golden.py).  Works in the dtype of the inputs
(float64 for the tight pin, float32 to mimic the reference's arithmetic).  Node-major layout [B, L, ...]
instead of the reference's [L, B, ...]; only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np

D = 128
H = 2
HD2 = 128  # 2 * head_dim


def _lin(x, sd, name, bias=True):
    w = sd[name + ".weight"]
    y = x @ w.T
    if bias and (name + ".bias") in sd:
        y = y + sd[name + ".bias"]
    return y


def _layer_norm(x, sd, name, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * sd[name + ".weight"] + sd[name + ".bias"]


def _invariants(x, gdir, sd, proj, lin1, lin2):
    """x [B,L,3,C] -> (features [B,L,out], F_norm [B,L,1]).  SEActor.py:93-101 / subequivariant_attentions.py:90-97."""
    z = np.concatenate([_lin(x, sd, proj, bias=False), gdir], axis=-1)           # [B,L,3,32]
    gram = np.einsum("blsa,blsc->blac", z, z)                                      # [B,L,32,32]
    fn = np.sqrt((gram ** 2).sum((-2, -1)))[..., None] + 1.0
    flat = gram.reshape(gram.shape[0], gram.shape[1], 1024)
    h = _lin(np.maximum(_lin(flat, sd, lin1), 0), sd, lin2)
    return h, fn


def _attention(g, ng, gdir, sd, p, bias):
    B, L = ng.shape[:2]
    inv, fn = _invariants(g, gdir, sd, p + "g_proj", p + "linear_g1", p + "linear_g2")
    c = np.concatenate([inv, ng], -1)
    scaling = float(HD2) ** -0.5
    q = _lin(c, sd, p + "q_proj") / fn * scaling
    k = _lin(c, sd, p + "k_proj") / fn
    v = _lin(c, sd, p + "v_proj") / fn
    vg = _lin(g, sd, p + "vg_proj", bias=False).reshape(B, L, 3, H, HD2 - 2)
    vg = np.concatenate([vg, np.repeat(gdir[:, :, :, None, :], H, axis=3)], -1)     # [B,L,3,H,128]
    q = q.reshape(B, L, H, HD2)
    k = k.reshape(B, L, H, HD2)
    v = v.reshape(B, L, H, HD2)
    s = np.einsum("bihd,bjhd->bhij", q, k)
    if bias is not None:
        s = s + bias[None]
    s = s - s.max(-1, keepdims=True)
    w = np.exp(s)
    w = w / w.sum(-1, keepdims=True)
    o = np.einsum("bhij,bjhd->bihd", w, v).reshape(B, L, H * HD2)
    og = np.einsum("bhij,bjshd->bishd", w, vg).reshape(B, L, 3, H * HD2)
    return _lin(og, sd, p + "g_out", bias=False), _lin(o, sd, p + "ng_out")


def _layer(g, ng, gdir, sd, p, bias):
    g1, ng1 = _attention(g, ng, gdir, sd, p + "self_attn.", bias)
    g = g + g1
    ng = _layer_norm(ng + ng1, sd, p + "norm1")
    inv, fn = _invariants(g1, gdir, sd, p + "g_proj2", p + "linear_g1", p + "linear_g2")
    c = np.concatenate([inv, ng], -1)
    mat = (_lin(np.maximum(_lin(c, sd, p + "linear3"), 0), sd, p + "linear4") / fn).reshape(*ng.shape[:2], 32, 32)
    z3 = np.concatenate([_lin(g1, sd, p + "g_proj3", bias=False), gdir], -1)
    g = g + _lin(np.einsum("blsa,blac->blsc", z3, mat), sd, p + "linear5", bias=False)
    ng = _layer_norm(ng + _lin(np.maximum(_lin(c, sd, p + "linear1"), 0), sd, p + "linear2") / fn, sd, p + "norm2")
    return g, ng


def set_actor_forward(sd, obs, traversals, relation, max_action=1.0, n_layers=3, probes=None):
    """sd: {name: ndarray} with the 'actor.' prefix stripped; obs [B, 41*L]; traversals 3 x int[L];
    relation [L, L, 3].  Returns actions [B, 3*L]."""
    dt = obs.dtype
    sd = {k: np.asarray(v, dtype=dt) for k, v in sd.items()}
    relation = np.asarray(relation, dtype=dt)
    B = obs.shape[0]
    L = len(traversals[0])
    x = obs.reshape(B, L, 41)
    g0 = np.swapaxes(x[..., :24].reshape(B, L, 8, 3), -1, -2)                      # [B,L,3,8]
    n0 = x[..., 24:]
    gdir = g0[..., 1:3]
    g = _lin(g0, sd, "g_encoder", bias=False) * np.sqrt(dt.type(D))
    ng = _lin(n0, sd, "encoder") * np.sqrt(dt.type(D))
    pos = np.concatenate([sd["pos_encoder.embeddings.%d.weight" % i][np.asarray(t)] for i, t in enumerate(traversals)], 1)
    ng = ng + pos[None]
    rel = _lin(relation, sd, "transformer_encoder.rel_encoder")                    # [L,L,H]
    bias0 = np.transpose(rel, (2, 0, 1))                                           # [H,i,j]
    for li in range(n_layers):
        g, ng = _layer(g, ng, gdir, sd, "transformer_encoder.layers.%d." % li, bias0 if li == 0 else None)
        if probes is not None:
            probes["layer%d/g" % li] = g.copy()
            probes["layer%d/ng" % li] = ng.copy()
    ng = _layer_norm(ng, sd, "transformer_encoder.norm")
    out_ng = np.concatenate([n0, ng], -1)
    out_g = np.concatenate([g0, g], -1)
    inv, fn = _invariants(out_g, gdir, sd, "gg_proj", "linear1_g", "linear2_g")
    hng = _lin(np.maximum(_lin(out_ng, sd, "linear1_ng"), 0), sd, "linear2_ng")
    c = np.concatenate([inv, hng], -1)
    mat = (_lin(np.maximum(_lin(c, sd, "linear1_m"), 0), sd, "linear2_m") / fn).reshape(B, L, 32, 32)
    zh = np.concatenate([_lin(out_g, sd, "g_proj", bias=False), gdir], -1)
    vec = _lin(np.einsum("blsa,blac->blsc", zh, mat), sd, "decoder_g", bias=False)[..., 0]   # [B,L,3]
    axes = g0[..., 5:8]                                                            # [B,L,3(spatial),3(axis)]
    act = np.einsum("blsk,bls->blk", axes, vec)
    return (max_action * np.tanh(act)).reshape(B, 3 * L)


def formula_state_dict(keys_shapes, dtype=np.float64):
    """State dict (prefix 'actor.' stripped) filled with oracle.formula values."""
    from .formula import formula_values
    out = {}
    for k, shp in keys_shapes.items():
        out[k[len("actor."):] if k.startswith("actor.") else k] = formula_values(k, tuple(shp)).astype(dtype)
    return out
