"""TEST INFRASTRUCTURE (oracle/): closed-form parameter values for the SET actor.

Golden SET-forward vectors are produced by the *reference* SEPolicy (src/SEActor.py:290-356)
whose parameters were overwritten with these formulas, so that only (inputs, outputs) have to
be committed under tests/golden/ -- not an 18.9 MB weight blob.  The same formulas are applied
to the build's SEPolicy / HIP weight pack in the tests.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import anything under oracle/.
"""
import zlib

import numpy as np


# per-parameter gains chosen so that every branch of the network contributes O(0.1..1) to the output
# (with a plain 1/sqrt(fan_in) scale the equivariant stream dwarfs the rest and actions are ~1e-4,
# which would make an absolute-tolerance parity test blind to most of the graph)
_GAINS = (
    ("g_encoder.weight", 0.04),
    ("decoder_g.weight", 6.0),
    ("linear5.weight", 4.0),
    ("g_out.weight", 0.5),
    ("vg_proj.weight", 0.5),
    ("ng_out.weight", 6.0),
    ("linear_g1.weight", 0.5),
    ("linear1_g.weight", 0.5),
)


def _gain(name):
    for suffix, g in _GAINS:
        if name.endswith(suffix):
            return g
    return 1.0


def formula_values(name, shape):
    """Deterministic float64 array for parameter `name` (state_dict key without prefix games)."""
    numel = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(numel, dtype=np.float64)
    h = float(zlib.crc32(name.encode("utf-8")) % 9973)
    v = 0.5 * np.sin(i * 0.7853 + h * 0.37) + 0.5 * np.sin(i * 0.0137 + h)
    if len(shape) == 2:
        if "embeddings" in name:
            out = 0.3 * v
        else:
            fan_in = shape[1]
            out = v * (1.2 / np.sqrt(fan_in)) * _gain(name)
    else:
        if "norm" in name and name.endswith("weight"):
            out = 1.0 + 0.1 * v
        else:
            out = 0.05 * v
    return out.reshape(shape)


def apply_formula_(module, dtype=None):
    """Overwrite every parameter of a torch module in place (module = SEPolicy-like)."""
    import torch
    with torch.no_grad():
        for name, p in module.state_dict().items():
            vals = formula_values(name, tuple(p.shape))
            p.copy_(torch.from_numpy(vals).to(p.dtype if dtype is None else dtype))
    return module


def synth_obs(num_limbs, batch, seed):
    """Synthetic but plausible per-limb observations (layout of <env>.py:116-140), float64 [B, 41*L]."""
    rng = np.random.RandomState(seed)
    L = num_limbs
    obs = np.zeros((batch, L, 41))
    obs[:, :, 0:3] = rng.uniform(-0.6, 0.6, size=(batch, L, 3))
    obs[:, 0, 0:3] = 0.0
    obs[:, :, 5] = -9.81
    ang = rng.uniform(-np.pi, np.pi, size=(batch, 1))
    obs[:, :, 6] = np.cos(ang)
    obs[:, :, 7] = np.sin(ang)
    obs[:, :, 9:12] = np.clip(rng.normal(0, 2.0, size=(batch, L, 3)), -10, 10)
    obs[:, :, 12:15] = rng.normal(0, 3.0, size=(batch, L, 3))
    for k in range(3):
        ax = rng.normal(size=(batch, L, 3))
        ax /= np.linalg.norm(ax, axis=-1, keepdims=True)
        obs[:, :, 15 + 3 * k:18 + 3 * k] = ax
    obs[:, 0, 15:24] = 0.0
    obs[:, :, 24:27] = rng.uniform(-1.0, 1.0, size=(batch, L, 3))
    obs[:, 0, 24:27] = 0.0
    for k in range(3):
        obs[:, :, 27 + 3 * k] = rng.uniform(-0.1, 1.1, size=(batch, L))
        lo = rng.uniform(0.05, 0.5, size=(batch, L))
        obs[:, :, 28 + 3 * k] = lo
        obs[:, :, 29 + 3 * k] = lo + rng.uniform(0.01, 0.45, size=(batch, L))
    obs[:, 0, 27:36] = 0.5
    types = rng.randint(0, 5, size=(batch, L))
    for t in range(4):
        obs[:, :, 36 + t] = (types == t + 1)
    obs[:, 0, 36:40] = [1, 0, 0, 0]
    obs[:, :, 40] = rng.uniform(0.0, 1.6, size=(batch, L))
    return obs.reshape(batch, L * 41)


def scripted_batch(L, B, seed):
    """A replay batch made from seeds only (tools/capture_golden_update.py feeds it to the reference's Agent.update; the tests
    regenerate it instead of storing a 256-row batch): observations synth_obs(seed) / synth_obs(seed + 1), uniform actions,
    N(1, 0.5) rewards, 30 % done flags.  float32, the dtype the replay buffer hands out."""
    rng = np.random.RandomState(seed)
    return dict(obs=synth_obs(L, B, seed).astype(np.float32), next_obs=synth_obs(L, B, seed + 1).astype(np.float32),
                action=rng.uniform(-1, 1, size=(B, 3 * L)).astype(np.float32),
                reward=rng.normal(1.0, 0.5, size=(B, 1)).astype(np.float32),
                done=(rng.uniform(size=(B, 1)) < 0.3).astype(np.float32))
