"""The lazily materialised `infos` of BatchedModularVecEnv.step_wait (sgrl_amd/vec_env.py StepInfos): behaves like the tuple of dicts
the reference's SubprocVecEnv returns (reference src/subproc_vec_env.py:59-63; the trainer reads infos[i]['dist'] in the video path
only, reference src/common/trainer.py:216) without building n dicts per step."""
import numpy as np
import pytest

from sgrl_amd.vec_env import StepInfos


def test_step_infos_is_a_sequence_of_dicts_made_when_read():
    dist = np.array([0.5, 1.5, 2.5, 3.5], dtype=np.float32)
    trunc = np.array([0, 1, 0, 0], dtype=np.uint8)
    infos = StepInfos(dist, trunc, {2: {"constraint_rows_dropped": 3}})
    assert len(infos) == 4
    assert infos[0] == {"dist": 0.5} and isinstance(infos[0]["dist"], float)
    assert infos[1] == {"dist": 1.5, "TimeLimit.truncated": True}
    assert infos[2] == {"dist": 2.5, "constraint_rows_dropped": 3}
    assert infos[-1] == {"dist": 3.5}
    assert infos[1:3] == ({"dist": 1.5, "TimeLimit.truncated": True}, {"dist": 2.5, "constraint_rows_dropped": 3})
    assert list(infos) == [infos[i] for i in range(4)]
    assert infos == tuple(infos) and infos == list(infos) and not (infos == [{"dist": 0.0}] * 4)
    with pytest.raises(IndexError):
        infos[4]
    # the reference's consumers: zip(*results)-style unpacking and per-environment lookups
    a, b, c, d = infos
    assert d["dist"] == 3.5
    # the arrays behind it are the step's own (a later step hands out new ones): mutating a read dict changes nothing
    infos[0]["dist"] = 99.0
    assert infos[0]["dist"] == 0.5
