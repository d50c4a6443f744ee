"""Pack a compiled `mjcf.Model` + `EnvSpec` into the (int32, float64) blobs of include/sgrl_model.h."""
import numpy as np

from . import mjcf
from .env_spec import env_spec_for

# must match the enums in include/sgrl_model.h
NHDR = 24
NFHDR = 16
(H_MAGIC, H_NBODY, H_NJNT, H_NQ, H_NV, H_NU, H_NGEOM, H_NPAIR, H_INTEGRATOR, H_FRAME_SKIP, H_DONE_RULE,
 H_TARGET_V2, H_RESET_VEL_NORMAL, H_NHEIGHT_BODIES, H_HEIGHT_BODY0, H_HEIGHT_BODY1, H_MAX_ROWS,
 H_PGS_ITERS, H_SOLVER) = range(19)
(F_TIMESTEP, F_GRAV_X, F_GRAV_Y, F_GRAV_Z, F_HEIGHT_LO, F_HEIGHT_HI, F_ANG_LIMIT, F_ALIVE_BONUS,
 F_HEADING_WEIGHT, F_CTRL_COST, F_RESET_POS_NOISE, F_RESET_VEL_NOISE, F_PGS_TOL, F_TOTAL_MASS) = range(14)

DEFAULT_MAX_ROWS = 48
DEFAULT_PGS_ITERS = 300
DEFAULT_PGS_TOL = 1e-10
DEFAULT_SOLVER = 1


MAXDEPTH = 8


def _derived_int_tables(model):
    """body_depth, body_path, body_subend, body_dofmask, dof_act (see include/sgrl_model.h)."""
    nb, nv = model.nbody, model.nv
    depth = np.zeros(nb, dtype=np.int32)
    path = np.full((nb, MAXDEPTH), -1, dtype=np.int32)
    subend = np.zeros(nb, dtype=np.int32)
    mask = np.zeros((nb, 2), dtype=np.int32)
    for b in range(1, nb):
        chain = []
        a = b
        while a > 0:
            chain.append(a)
            a = int(model.body_parent[a])
        chain.reverse()
        if len(chain) > MAXDEPTH:
            raise ValueError("kinematic chain deeper than %d" % MAXDEPTH)
        depth[b] = len(chain)
        path[b, :len(chain)] = chain
        bits = 0
        for c in chain:
            for d in range(int(model.body_dofadr[c]), int(model.body_dofadr[c]) + int(model.body_dofnum[c])):
                bits |= 1 << d
        lo, hi = bits & 0xFFFFFFFF, (bits >> 32) & 0xFFFFFFFF
        mask[b] = np.array([lo, hi], dtype=np.uint32).view(np.int32)
    # pre-order numbering => contiguous subtrees
    for b in range(nb - 1, 0, -1):
        subend[b] = max(subend[b], b + 1)
        p = int(model.body_parent[b])
        if p > 0:
            subend[p] = max(subend[p], subend[b])
    for b in range(1, nb):
        for c in range(b + 1, nb):
            inside = c < subend[b]
            a = c
            anc = False
            while a > 0:
                if a == b:
                    anc = True
                a = int(model.body_parent[a])
            if inside != anc:
                raise ValueError("bodies are not in pre-order")
    if nv > 64:
        raise ValueError("nv > 64 unsupported")
    dof_act = np.full(nv, -1, dtype=np.int32)
    for u in range(model.nu):
        dof_act[int(model.act_dof[u])] = u
    return [depth, path.ravel(), subend, mask.ravel(), dof_act]


def worst_case_rows(model):
    """Upper bound on simultaneously active constraint rows: one per limited joint + every contact of every pair."""
    rows = int(np.sum(np.asarray(model.jnt_limited) != 0))
    for k in range(model.npair):
        g2 = int(model.pair_g2[k])
        ncon = 2 if (int(model.geom_type[int(model.pair_g1[k])]) == mjcf.GEOM_PLANE and int(model.geom_type[g2]) == mjcf.GEOM_CAPSULE) else 1
        dim = int(model.pair_condim[k])
        rows += ncon * (1 if dim == 1 else 2 * (dim - 1))
    return rows


def pack_model(model, spec=None, env_name=None, max_rows=DEFAULT_MAX_ROWS, pgs_iters=DEFAULT_PGS_ITERS,
               pgs_tol=DEFAULT_PGS_TOL, solver=DEFAULT_SOLVER):
    """Return (ib int32[...], fb float64[...])."""
    if spec is None:
        spec = env_spec_for(env_name or model.name)
    hdr = np.zeros(NHDR, dtype=np.int32)
    hdr[H_MAGIC] = mjcf.MAGIC
    hdr[H_NBODY], hdr[H_NJNT], hdr[H_NQ], hdr[H_NV] = model.nbody, model.njnt, model.nq, model.nv
    hdr[H_NU], hdr[H_NGEOM], hdr[H_NPAIR] = model.nu, model.ngeom, model.npair
    hdr[H_INTEGRATOR] = model.integrator
    hdr[H_FRAME_SKIP] = spec.frame_skip
    hdr[H_DONE_RULE] = spec.done_rule
    hdr[H_TARGET_V2] = 1 if spec.target_v2 else 0
    hdr[H_RESET_VEL_NORMAL] = 1 if spec.reset_vel_normal else 0
    hb = [model.body_names.index(n) for n in spec.height_bodies if n in model.body_names]
    hdr[H_NHEIGHT_BODIES] = len(hb)
    for i, b in enumerate(hb[:2]):
        hdr[H_HEIGHT_BODY0 + i] = b
    hdr[H_MAX_ROWS] = min(int(max_rows), max(1, worst_case_rows(model)))   # small morphologies need less LDS
    hdr[H_PGS_ITERS] = pgs_iters
    hdr[H_SOLVER] = solver
    ints = [hdr]
    for k in mjcf.Model.INT_FIELDS:
        ints.append(np.asarray(getattr(model, k), dtype=np.int32).ravel())
    ints.extend(_derived_int_tables(model))
    fh = np.zeros(NFHDR, dtype=np.float64)
    fh[F_TIMESTEP] = model.timestep
    fh[F_GRAV_X:F_GRAV_Z + 1] = model.gravity
    fh[F_HEIGHT_LO], fh[F_HEIGHT_HI], fh[F_ANG_LIMIT] = spec.height_lo, spec.height_hi, spec.ang_limit
    fh[F_ALIVE_BONUS], fh[F_HEADING_WEIGHT], fh[F_CTRL_COST] = spec.alive_bonus, spec.heading_weight, spec.ctrl_cost
    fh[F_RESET_POS_NOISE], fh[F_RESET_VEL_NOISE] = spec.reset_pos_noise, spec.reset_vel_noise
    fh[F_PGS_TOL] = pgs_tol
    fh[F_TOTAL_MASS] = float(np.sum(model.body_mass[1:]))
    fls = [fh]
    for k in mjcf.Model.F64_FIELDS:
        fls.append(np.asarray(getattr(model, k), dtype=np.float64).ravel())
    return np.ascontiguousarray(np.concatenate(ints)), np.ascontiguousarray(np.concatenate(fls))
