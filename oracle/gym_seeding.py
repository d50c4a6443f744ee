"""Test infrastructure (oracle): the reference's reset stream, restated.

The reference seeds every worker's environment with `e.seed(seed)` (reference src/utils.py:19), which for a gym 0.17.2
MujocoEnv is `self.np_random, seed = gym.utils.seeding.np_random(seed)`; `reset_model` (src/environments/<name>.py:150-164)
then draws from that MT19937 stream.  gym is an un-vendored third-party dependency (`gym==0.17.2`, reference
requirements.txt:4) absent from /root/reference and from this image, so `np_random` / `hash_seed` / `create_seed` are
restated here from the published 0.17.2 algorithm [3P-knowledge]:

    seed    -> create_seed: seed mod 2^64
            -> hash_seed:   first 8 bytes of sha512(str(seed)), read as little-endian 32-bit words (zero padded by one word)
            -> int list:    the 32-bit limbs of that integer, least significant first
            -> numpy.random.RandomState().seed(list)          (MT19937 init_by_array)

PARITY UNPINNED against real gym (nothing here can run it).  What IS pinned: the ORDER, distribution and size of the draws
`reset_model` takes from the stream -- `reset_draws` below reproduces, call for call, the draws recorded while executing the
reference's own `reset_model` of all 19 distinct env files on a seeded RandomState (tests/golden/env_arith.npz
`<env>/reset/draws`; tests/test_gym_seeding.py) -- and the mapping draws -> (qpos, qvel, target) (tests/test_oracle_env_arith.py).
The engine itself deliberately uses a counter RNG: identical per-worker MT19937 streams would make the
thousands of replicas of a morphology identical.  This module exists so that a parity run can START the engine / the C
oracle from exactly the state the reference's worker would have after its first `reset()` (`first_reset_state`).
"""
import hashlib
import struct

import numpy as np


def create_seed(a, max_bytes=8):
    """gym.utils.seeding.create_seed for an int seed."""
    if not (isinstance(a, (int, np.integer)) and a >= 0):
        raise ValueError("Seed must be a non-negative integer")
    return int(a) % 2 ** (8 * max_bytes)


def _bigint_from_bytes(b):
    sizeof_int = 4
    padding = sizeof_int - len(b) % sizeof_int          # a full zero word when the length already is a multiple of four
    b += b"\0" * padding
    words = struct.unpack("{}I".format(len(b) // sizeof_int), b)
    return sum(2 ** (sizeof_int * 8 * i) * v for i, v in enumerate(words))


def hash_seed(seed, max_bytes=8):
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode("utf8")).digest()[:max_bytes])


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise ValueError("Seed must be non-negative")
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def np_random(seed):
    """gym.utils.seeding.np_random(seed) -> (RandomState, seed)."""
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed


def reset_draws(rng, envname, nq, nv):
    """The draws one `reset_model` call takes from `rng`, in the reference's order (src/environments/<name>.py:150-164;
    cheetah files: U(-.1,.1) position noise and GAUSSIAN velocity noise, 3d_cheetah_14_full.py:157-159; `_v2_` tasks draw
    the target distance too).  Layout: [yaw angle | nq position-noise values | nv velocity-noise values | target angle | (target
    distance)] -- the layout tests/golden/env_arith.npz `<env>/reset/draws` uses."""
    cheetah = "cheetah" in envname
    d = [rng.uniform(low=-np.pi, high=np.pi)]
    d += list(rng.uniform(low=-0.1, high=0.1, size=nq) if cheetah else rng.uniform(low=-0.005, high=0.005, size=nq))
    d += list(rng.randn(nv) if cheetah else rng.uniform(low=-0.005, high=0.005, size=nv))
    d.append(rng.uniform(low=-np.pi, high=np.pi))
    if "_v2_" in envname:
        d.append(rng.uniform(10, 20))
    return np.array(d, dtype=np.float64)


def state_from_draws(draws, envname, qpos0, nq, nv):
    """draws -> (qpos, qvel, target): the mapping pinned by tests/test_oracle_env_arith.py (reference <name>.py:150-164:
    yaw-only root quaternion from half the first angle, additive noise, cheetah velocity noise scaled by 0.1)."""
    cheetah = "cheetah" in envname
    q = np.array(qpos0, dtype=np.float64).copy()
    half = draws[0] / 2
    q[3], q[6] = np.cos(half), np.sin(half)
    q[4] = q[5] = 0.0
    q = q + draws[1:1 + nq]
    v = draws[1 + nq:1 + nq + nv] * (0.1 if cheetah else 1.0)
    r = draws[1 + nq + nv]
    ln = draws[2 + nq + nv] if "_v2_" in envname else 10000.0
    return q, v, np.array([np.cos(r), np.sin(r)]) * ln


def first_reset_state(seed, envname, qpos0, nq, nv):
    """State of a reference worker seeded with `seed` after its first reset()."""
    rng, _ = np_random(seed)
    return state_from_draws(reset_draws(rng, envname, nq, nv), envname, qpos0, nq, nv)
