"""sgrl_amd: MI355X-native batched rollout engine for the SGRL hot path.

Path (BASELINE.json north_star): VecEnv.step -> rigid-body step -> per-limb obs scatter -> SET actor forward.
Only the pieces of that path live here; see DESIGN.md.
"""
__version__ = "0.1.0"
