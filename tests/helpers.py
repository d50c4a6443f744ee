"""Shared test helpers (models, oracle access)."""
import functools
import os

import numpy as np

from sgrl_amd import mjcf, model_pack
from sgrl_amd.env_spec import env_spec_for


@functools.lru_cache(maxsize=None)
def packed(env_name, max_rows=model_pack.DEFAULT_MAX_ROWS, pgs_iters=model_pack.DEFAULT_PGS_ITERS,
           pgs_tol=model_pack.DEFAULT_PGS_TOL, solver=model_pack.DEFAULT_SOLVER):
    """(model, ib, fb) for an environment name (v2 names share the v1 morphology)."""
    xml_name = env_name.replace("_v2_", "_")
    m = mjcf.load_asset(xml_name)
    ib, fb = model_pack.pack_model(m, spec=env_spec_for(env_name), max_rows=max_rows, pgs_iters=pgs_iters,
                                   pgs_tol=pgs_tol, solver=solver)
    return m, ib, fb


def oracle_model(env_name, **kw):
    from oracle import physics_ref
    m, ib, fb = packed(env_name, **kw)
    return m, physics_ref.OracleModel(ib, fb)


WALKERS = ["3d_walker_2_right_leg_left_knee", "3d_walker_3_left_leg_right_foot", "3d_walker_3_left_knee_right_knee",
           "3d_walker_4_right_knee_left_foot", "3d_walker_5_foot", "3d_walker_5_left_knee", "3d_walker_6_right_foot",
           "3d_walker_7_full"]
HOPPERS = ["3d_hopper_3_shin", "3d_hopper_4_lower_shin", "3d_hopper_5_full"]
