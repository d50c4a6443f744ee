/* sgrl_model.h -- packed morphology + task description shared by the HIP engine (libsgrl_hip.so)
 * and the CPU oracle (oracle/libsgrl_oracle.so).  Plain C, no dependencies.
 *
 * A morphology is two flat blobs produced by sgrl_amd.model_pack.pack_model():
 *   int32  ib[]:  header (SGRL_NHDR ints) followed by the integer tables, in the order of the
 *                 SGRL_INT_FIELDS list below;
 *   double fb[]:  SGRL_NFHDR scalars followed by the float tables in SGRL_F64_FIELDS order.
 * This replaces what the reference obtains from MuJoCo's compiled `mjModel` through mujoco-py
 * (reference src/environments/ModularEnv.py:12 `MujocoEnv.__init__(self, xml, 4)`) plus the
 * per-file task constants of reference src/environments/<name>.py:15-44,150-164.
 */
#ifndef SGRL_MODEL_H
#define SGRL_MODEL_H

#include <stdint.h>

#ifdef __HIPCC__
#define SGRL_HD __host__ __device__ inline
#else
#define SGRL_HD static inline
#endif

#define SGRL_MAGIC 0x5347524C
#define SGRL_MAXDEPTH 8

/* ---- integer header -------------------------------------------------------------------------- */
enum {
  SGRL_H_MAGIC = 0,
  SGRL_H_NBODY,      /* bodies incl. world (index 0); limbs L = nbody-1 */
  SGRL_H_NJNT,
  SGRL_H_NQ,
  SGRL_H_NV,
  SGRL_H_NU,
  SGRL_H_NGEOM,
  SGRL_H_NPAIR,
  SGRL_H_INTEGRATOR, /* 0 Euler (implicit joint damping), 1 RK4 */
  SGRL_H_FRAME_SKIP, /* 4 */
  SGRL_H_DONE_RULE,  /* 0 walker/humanoid, 1 hopper, 2 cheetah */
  SGRL_H_TARGET_V2,  /* 0: radius 10000 about origin; 1: radius U(10,20) about current position */
  SGRL_H_RESET_VEL_NORMAL, /* 0 uniform, 1 normal (cheetah) */
  SGRL_H_NHEIGHT_BODIES,   /* cheetah: bodies whose z also bounds the height (0..2) */
  SGRL_H_HEIGHT_BODY0,
  SGRL_H_HEIGHT_BODY1,
  SGRL_H_MAX_ROWS,   /* cap on constraint rows per evaluation */
  SGRL_H_PGS_ITERS,  /* maximum number of projected Gauss-Seidel sweeps per evaluation */
  SGRL_H_SOLVER,     /* 0: projected Gauss-Seidel only; 1: block-pivot direct solve (<= 32 rows) with PGS fallback */
  SGRL_NHDR = 24
};

/* ---- float header ---------------------------------------------------------------------------- */
enum {
  SGRL_F_TIMESTEP = 0,
  SGRL_F_GRAV_X, SGRL_F_GRAV_Y, SGRL_F_GRAV_Z,
  SGRL_F_HEIGHT_LO, SGRL_F_HEIGHT_HI, SGRL_F_ANG_LIMIT,
  SGRL_F_ALIVE_BONUS, SGRL_F_HEADING_WEIGHT, SGRL_F_CTRL_COST,
  SGRL_F_RESET_POS_NOISE, SGRL_F_RESET_VEL_NOISE,
  SGRL_F_PGS_TOL,    /* stop when max_i |(A_ii+R_i) df_i| < tol * (1 + max_i |b_i|) over one sweep */
  SGRL_F_TOTAL_MASS, /* sum of body masses (derived) */
  SGRL_NFHDR = 16
};

enum { SGRL_GEOM_PLANE = 0, SGRL_GEOM_SPHERE = 2, SGRL_GEOM_CAPSULE = 3 };
enum { SGRL_JNT_FREE = 0, SGRL_JNT_HINGE = 3 };

/* Address-space qualifiers of the four kinds of pointers in the view.  Empty for host code and the oracle.  The HIP
 * engine defines them before including this header: header copies and float tables in constant memory (uniform
 * reads become scalar loads, the rest plain global loads that do not touch the LDS queue), int tables in LDS. */
#ifndef SGRL_CONST_AS
#define SGRL_CONST_AS
#endif
#ifndef SGRL_ITAB_AS
#define SGRL_ITAB_AS
#endif
#ifndef SGRL_FTAB_AS
#define SGRL_FTAB_AS
#endif
typedef const SGRL_CONST_AS int32_t* sgrl_hdr_t;
typedef const SGRL_CONST_AS double* sgrl_fhdr_t;
typedef const SGRL_ITAB_AS int32_t* sgrl_itab_t;
typedef const SGRL_FTAB_AS double* sgrl_ftab_t;

typedef struct SgrlModelView {
  sgrl_hdr_t hdr;
  sgrl_fhdr_t fhdr;
  int nbody, njnt, nq, nv, nu, ngeom, npair;
  /* int tables */
  sgrl_itab_t body_parent, body_jntadr, body_jntnum, body_dofadr, body_dofnum, body_limbtype;
  sgrl_itab_t jnt_type, jnt_body, jnt_qposadr, jnt_dofadr, jnt_limited;
  sgrl_itab_t dof_body, dof_jnt, dof_parent;
  sgrl_itab_t geom_type, geom_body;
  sgrl_itab_t pair_g1, pair_g2, pair_condim;
  sgrl_itab_t act_dof, act_slot;
  /* derived by the packer (not part of the compiled asset): */
  sgrl_itab_t body_depth;   /* [nbody] number of bodies on the path torso..b (torso = 1) */
  sgrl_itab_t body_path;    /* [nbody*8] path[0] = 1 (torso) ... path[depth-1] = b, padded with -1 */
  sgrl_itab_t body_subend;  /* [nbody] bodies are in pre-order: subtree(b) = [b, subend[b]) */
  sgrl_itab_t body_dofmask; /* [nbody*2] 64-bit mask (lo, hi) of the dofs that move body b */
  sgrl_itab_t dof_act;      /* [nv] actuator driving this dof or -1 */
  /* float tables */
  sgrl_ftab_t qpos0;
  sgrl_ftab_t body_pos, body_quat, body_ipos, body_inertia, body_mass, body_invweight0;
  sgrl_ftab_t jnt_pos, jnt_axis, jnt_range, jnt_stiffness, jnt_solref, jnt_solimp, jnt_margin;
  sgrl_ftab_t dof_armature, dof_damping, dof_invweight0;
  sgrl_ftab_t geom_pos, geom_quat, geom_size;
  sgrl_ftab_t pair_mu, pair_margin, pair_solref, pair_solimp;
  sgrl_ftab_t act_gear, act_ctrlrange;
  int n_int, n_f64; /* total blob lengths */
} SgrlModelView;

/* Set up table pointers into the blobs (ib, fb).  The table sizes and the two headers the view exposes are read from
 * `hdr_src` / `fhdr_src`, which may be different copies of the same blobs: the HIP engine passes the copies in
 * constant memory there (scalar loads -> sizes, offsets and every header constant stay in scalar registers) while ib
 * points at the LDS copy of the int tables.  Returns 0, or -1 on a bad magic. */
/* `dims`: the SGRL_NHDR header VALUES the table sizes are taken from (the HIP engine passes them in registers: copies of the
 * header made wave-uniform, or compile-time constants in a kernel instance built for one dimension set); hdr_src / fhdr_src:
 * the header copies the view exposes for later reads of task constants. */
SGRL_HD int sgrl_model_view_dims(const int32_t* dims, sgrl_hdr_t hdr_src, sgrl_fhdr_t fhdr_src, sgrl_itab_t ib, sgrl_ftab_t fb,
                                 SgrlModelView* v) {
  if (dims[SGRL_H_MAGIC] != SGRL_MAGIC) return -1;
  v->hdr = hdr_src;
  v->fhdr = fhdr_src;
  const int nb = dims[SGRL_H_NBODY], nj = dims[SGRL_H_NJNT], nq = dims[SGRL_H_NQ], nv = dims[SGRL_H_NV];
  const int nu = dims[SGRL_H_NU], ng = dims[SGRL_H_NGEOM], np = dims[SGRL_H_NPAIR];
  v->nbody = nb; v->njnt = nj; v->nq = nq; v->nv = nv; v->nu = nu; v->ngeom = ng; v->npair = np;
  sgrl_itab_t p = ib + SGRL_NHDR;
  v->body_parent = p; p += nb;
  v->body_jntadr = p; p += nb;
  v->body_jntnum = p; p += nb;
  v->body_dofadr = p; p += nb;
  v->body_dofnum = p; p += nb;
  v->body_limbtype = p; p += nb;
  v->jnt_type = p; p += nj;
  v->jnt_body = p; p += nj;
  v->jnt_qposadr = p; p += nj;
  v->jnt_dofadr = p; p += nj;
  v->jnt_limited = p; p += nj;
  v->dof_body = p; p += nv;
  v->dof_jnt = p; p += nv;
  v->dof_parent = p; p += nv;
  v->geom_type = p; p += ng;
  v->geom_body = p; p += ng;
  v->pair_g1 = p; p += np;
  v->pair_g2 = p; p += np;
  v->pair_condim = p; p += np;
  v->act_dof = p; p += nu;
  v->act_slot = p; p += nu;
  v->body_depth = p; p += nb;
  v->body_path = p; p += 8 * nb;
  v->body_subend = p; p += nb;
  v->body_dofmask = p; p += 2 * nb;
  v->dof_act = p; p += nv;
  v->n_int = (int)(p - ib);
  sgrl_ftab_t f = fb + SGRL_NFHDR;
  v->qpos0 = f; f += nq;
  v->body_pos = f; f += 3 * nb;
  v->body_quat = f; f += 4 * nb;
  v->body_ipos = f; f += 3 * nb;
  v->body_inertia = f; f += 6 * nb;
  v->body_mass = f; f += nb;
  v->body_invweight0 = f; f += 2 * nb;
  v->jnt_pos = f; f += 3 * nj;
  v->jnt_axis = f; f += 3 * nj;
  v->jnt_range = f; f += 2 * nj;
  v->jnt_stiffness = f; f += nj;
  v->jnt_solref = f; f += 2 * nj;
  v->jnt_solimp = f; f += 5 * nj;
  v->jnt_margin = f; f += nj;
  v->dof_armature = f; f += nv;
  v->dof_damping = f; f += nv;
  v->dof_invweight0 = f; f += nv;
  v->geom_pos = f; f += 3 * ng;
  v->geom_quat = f; f += 4 * ng;
  v->geom_size = f; f += 3 * ng;
  v->pair_mu = f; f += np;
  v->pair_margin = f; f += np;
  v->pair_solref = f; f += 2 * np;
  v->pair_solimp = f; f += 5 * np;
  v->act_gear = f; f += nu;
  v->act_ctrlrange = f; f += 2 * nu;
  v->n_f64 = (int)(f - fb);
  return 0;
}

SGRL_HD int sgrl_model_view_from(sgrl_hdr_t hdr_src, sgrl_fhdr_t fhdr_src, sgrl_itab_t ib, sgrl_ftab_t fb,
                                 SgrlModelView* v) {
  int32_t dims[SGRL_NHDR];
  for (int k = 0; k < SGRL_NHDR; k++) dims[k] = hdr_src[k];
  return sgrl_model_view_dims(dims, hdr_src, fhdr_src, ib, fb, v);
}

/* Blob lengths implied by a header (the same sums sgrl_model_view_from walks; tests/test_abi.py checks them against
 * the packer for every shipped morphology). */
SGRL_HD void sgrl_model_blob_sizes(const int32_t* hdr, int* n_int, int* n_f64) {
  const int nb = hdr[SGRL_H_NBODY], nj = hdr[SGRL_H_NJNT], nq = hdr[SGRL_H_NQ], nv = hdr[SGRL_H_NV];
  const int nu = hdr[SGRL_H_NU], ng = hdr[SGRL_H_NGEOM], np = hdr[SGRL_H_NPAIR];
  *n_int = SGRL_NHDR + 18 * nb + 5 * nj + 4 * nv + 2 * ng + 3 * np + 2 * nu;
  *n_f64 = SGRL_NFHDR + nq + 19 * nb + 17 * nj + 3 * nv + 10 * ng + 9 * np + 3 * nu;
}

SGRL_HD int sgrl_model_view(const int32_t* ib, const double* fb, SgrlModelView* v) {
  return sgrl_model_view_from((sgrl_hdr_t)ib, (sgrl_fhdr_t)fb, (sgrl_itab_t)ib, (sgrl_ftab_t)fb, v);
}

/* Per-environment persistent state (one per env; SoA in the HIP engine, AoS in the oracle):
 *   qpos[nq], qvel[nv]                 generalized state (quaternion normalised by kinematics, as mj 2.1.0 does)
 *   torso_xy_stale[2]                  torso xpos[:2] left by the last forward pass (reference <env>.py:22 reads it
 *                                      BEFORE do_simulation, i.e. it lags qpos by one RK stage)
 *   target[2], step_count, episode     task state
 */

#endif /* SGRL_MODEL_H */
