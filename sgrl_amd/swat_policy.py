"""SWAT (structure-aware transformer) actor / critic behind the reference's module surfaces.

`StructurePolicy` / `CriticStructurePolicy` keep the constructor signatures, `forward`, `Q1`, `change_morphology` and the
`state_dict()` keys of reference src/StructureActor.py:176-273 / src/StructureCritic.py:8-125: a plain transformer over the
limbs (embedding size 128, 2 heads, 3 layers) with the three traversal-index position embeddings added once and the
relation bias (PPR, symmetric Laplacian, distance -> one additive bias per head) on the first layer only.  Plain
differentiable PyTorch -- this baseline has no HIP fast path (the SET model is the one the north star names); outputs are
pinned to fixtures produced by executing the reference's own modules (tests/golden/swat_forward.npz,
tools/capture_golden_swat.py).
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .set_policy import ConcatPositionalEmbedding


class _SelfAttention(nn.Module):
    """Parameters of torch.nn.MultiheadAttention (in_proj_weight / in_proj_bias / out_proj), forward of the reference's
    attentions.multi_head_attention_forward: q scaled by head_dim^-0.5, additive float mask [B * H, L, L]."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, x, bias=None):
        """x [B, L, E]; bias [H, L, L] or None."""
        B, L, E = x.shape
        H = self.num_heads
        hd = E // H
        q, k, v = F.linear(x, self.in_proj_weight, self.in_proj_bias).chunk(3, dim=-1)
        q = (q * float(hd) ** -0.5).view(B, L, H, hd)
        k, v = k.view(B, L, H, hd), v.view(B, L, H, hd)
        s = torch.einsum("bihd,bjhd->bhij", q, k)
        if bias is not None:
            s = s + bias.unsqueeze(0)
        w = F.softmax(s, dim=-1)
        o = torch.einsum("bhij,bjhd->bihd", w, v).reshape(B, L, E)
        return self.out_proj(o)


class _EncoderLayer(nn.Module):
    """reference MyTransformerEncoderLayer (StructureActor.py:48-66): post-norm, ReLU feed-forward, dropout = identity at
    the reference's dropout_rate 0."""

    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.self_attn = _SelfAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x, bias=None):
        x = self.norm1(x + self.self_attn(x, bias))
        return self.norm2(x + self.linear2(F.relu(self.linear1(x))))


class _Encoder(nn.Module):
    """reference RepeatTransformerEncoder (StructureActor.py:68-107)."""

    def __init__(self, layer, num_layers, nhead, norm=None, d_rel=3):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.norm = norm
        self.nhead = nhead
        self.rel_encoder = nn.Linear(d_rel, nhead)

    def forward(self, x, pos, rel):
        x = x + pos.unsqueeze(0)
        bias = self.rel_encoder(rel).permute(2, 0, 1)          # [H, i, j]
        for i, layer in enumerate(self.layers):
            x = layer(x, bias if i == 0 else None)
        return self.norm(x) if self.norm is not None else x


class TransformerModel(nn.Module):
    """reference StructureActor.TransformerModel (StructureActor.py:110-168)."""

    def __init__(self, feature_size, output_size, ninp, nhead, nhid, nlayers, dropout=0.0, condition_decoder=False,
                 transformer_norm=False, num_positions=0, rel_size=1):
        super().__init__()
        self.model_type = "Structure"
        self.pos_encoder = ConcatPositionalEmbedding(ninp, num_positions=num_positions)
        self.transformer_encoder = _Encoder(_EncoderLayer(ninp, nhead, nhid), nlayers, nhead,
                                            norm=nn.LayerNorm(ninp) if transformer_norm else None, d_rel=rel_size)
        self.encoder = nn.Linear(feature_size, ninp)
        self.ninp = ninp
        self.condition_decoder = bool(condition_decoder)
        self.decoder = nn.Linear(ninp + feature_size if condition_decoder else ninp, output_size)
        with torch.no_grad():
            self.encoder.weight.uniform_(-0.1, 0.1)
            self.decoder.bias.zero_()
            self.decoder.weight.uniform_(-0.1, 0.1)

    def forward(self, x, graph):
        """x [B, L, feature] (node-major) -> [B, L, output_size]."""
        h = self.encoder(x) * math.sqrt(self.ninp)
        h = self.transformer_encoder(h, self.pos_encoder(graph["traversals"]), graph["relation"])
        if self.condition_decoder:
            h = torch.cat([h, x], dim=2)
        return self.decoder(h)


def _model(feature, out, args):
    return TransformerModel(feature, out, args.attention_embedding_size, args.attention_heads, args.attention_hidden_size,
                            args.attention_layers, args.dropout_rate, condition_decoder=args.condition_decoder_on_features,
                            transformer_norm=args.transformer_norm, num_positions=len(args.traversal_types),
                            rel_size=args.rel_size)


class StructurePolicy(nn.Module):
    """Drop-in for reference StructureActor.StructurePolicy (constructor of StructureActor.py:179-191)."""

    def __init__(self, state_dim, action_dim, msg_dim, batch_size, max_action, max_children, disable_fold, td, bu,
                 args=None, device=None):
        super().__init__()
        self.num_limbs = 1
        self.max_action = max_action
        self.msg_dim, self.batch_size, self.max_children, self.disable_fold = msg_dim, batch_size, max_children, disable_fold
        self.state_dim, self.action_dim = state_dim, action_dim
        self.actor = _model(state_dim, action_dim, args)
        if device is not None:
            self.actor.to(device)
        self.graph = None

    def clear_buffer(self):
        self.action = None
        self.input_state = None

    def forward(self, state, mode="train"):
        self.clear_buffer()
        B = state.shape[0]
        x = state.reshape(B, self.num_limbs, -1)
        self.action = (self.max_action * torch.tanh(self.actor(x, self.graph))).reshape(B, -1)
        return self.action

    def change_morphology(self, graph):
        self.graph = graph
        self.parents = graph["parents"]
        self.num_limbs = len(self.parents)


class CriticStructurePolicy(nn.Module):
    """Drop-in for reference StructureCritic.CriticStructurePolicy (per-limb twin Q values [B, L])."""

    def __init__(self, state_dim, action_dim, msg_dim, batch_size, max_children, disable_fold, td, bu, args=None,
                 device=None):
        super().__init__()
        self.num_limbs = 1
        self.msg_dim, self.batch_size, self.max_children, self.disable_fold = msg_dim, batch_size, max_children, disable_fold
        self.state_dim, self.action_dim = state_dim, action_dim
        self.critic1 = _model(state_dim + action_dim, 1, args)
        self.critic2 = _model(state_dim + action_dim, 1, args)
        if device is not None:
            self.to(device)
        self.graph = None

    def _input(self, state, action):
        B = state.shape[0]
        assert state.shape[1] == self.state_dim * self.num_limbs, \
            "state.shape[1] expects {} but got {}".format(self.state_dim * self.num_limbs, state.shape[1])
        return torch.cat([state.reshape(B, self.num_limbs, -1), action.reshape(B, self.num_limbs, -1)], dim=2)

    def forward(self, state, action):
        x = self._input(state, action)
        B = x.shape[0]
        return self.critic1(x, self.graph).reshape(B, -1), self.critic2(x, self.graph).reshape(B, -1)

    def Q1(self, state, action):
        x = self._input(state, action)
        return self.critic1(x, self.graph).reshape(x.shape[0], -1)

    def clear_buffer(self):
        pass

    def change_morphology(self, graph):
        self.graph = graph
        self.parents = graph["parents"]
        self.num_limbs = len(self.parents)
