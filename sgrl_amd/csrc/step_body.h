// step_body.h -- the batched rollout engine's per-environment step, written against a tiny
// wavefront-execution interface `W` (see wave_hip.h for the gfx950 implementation).
//
// One environment is advanced by ONE 64-lane wavefront; all per-environment working data lives in
// an LDS slab `S` (doubles) + `I` (ints).  The computation replaces, per environment,
//   ModularEnv.step / do_simulation(a, 4) / _get_obs / reset_model
//   (reference src/environments/3d_walker_7_full.py:15-164 and family variants) and the worker-side
//   auto-reset of reference src/subproc_vec_env.py:12-15,
// i.e. 4 x mj_step (RK4 or semi-implicit Euler) of MuJoCo-style articulated rigid-body dynamics:
// kinematics -> CRBA mass matrix -> Cholesky -> RNE bias -> collision -> soft-constraint rows ->
// dual LCP (block principal pivoting, projected Gauss-Seidel as fallback) -> integrate, then the
// 41-float-per-limb observation scatter, reward, termination and counter-RNG reset.
//
// Parallel decomposition (lane = body | dof | contact pair | constraint row, depending on phase):
//   joint rotations          : lane per hinge (sincos once per joint), constants staged in dead LDS arrays
//   kinematics / velocities  : lane b walks its own root->b chain (no barriers inside)
//   composite inertias       : lane b sums its pre-order-contiguous subtree
//   mass matrix              : lane i fills row i (packed lower triangle) along its dof-ancestor chain
//   Cholesky                 : wave policy: registers, with the explicit inverse of the factor (nv <= 24);
//                              else right-looking root-free elimination in LDS, one phase per pivot
//   half-solves Y = L^-1 J'  : with L^-1: triangular matrix product (no recurrence); else lane r owns
//                              constraint row r (+ one extra row for the smooth force)
//   dual LCP                 : free set as a 64-bit ballot mask; A_FF formed from Y; factor + solve per round
//   PGS (fallback)           : lane d owns v[d] = (Y' f)[d]; row residual = wave reduction
//   observation              : lane per output float, coalesced store
// Constraint rows live in LDS for the common case and in a per-environment HBM slab when an evaluation
// has more rows than the LDS arrays hold (struct Rows): the stage is inlined once per variant.
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/sgrl_model.h"

#ifdef __HIPCC__
#define SGRL_DEV __device__ __forceinline__
#else
#define SGRL_DEV inline
#endif

namespace sgrl {

constexpr double kMinVal = 1e-15;
constexpr double kMinImp = 0.0001;
constexpr double kMaxImp = 0.9999;
constexpr double kPi = 3.14159265358979323846;
constexpr int kNAMax = 32;                        // most rows whose factor the LDS scratch (dead zone) is asked to hold
constexpr int kBppMaxIter = 40;                   // block-pivot rounds before giving up (-> Gauss-Seidel)
constexpr int kSlabRows = 256;                    // most constraint rows an evaluation may have (HBM slab path; the row cap of a morphology)
constexpr int kPivotRows = 64;                    // ... of which the exact block-pivot solve takes up to this many (free set = 64-bit mask);
                                                  // evaluations with more rows run Gauss-Seidel over the rows in the slab (pgs_big)
constexpr int kPrevRows = 48;                     // warm-start memory in LDS: the first rows of the previous evaluation (the rest: HBM slab)
constexpr int kSlabLdy = 47;                      // largest Y row stride served: nv <= 46
// per-env HBM slab (Engine::rows_hbm): factor of a <= 64-row free set | Y | five row arrays | four int row arrays | warm-start
// memory of the rows beyond kPrevRows (forces, then keys) (+ ldy (ldy + 1) / 2 behind the slab: the Euler integrator's copy of
// the mass matrix, see Layout::mfull_hbm)
SGRL_HD int slab_prev_offset(int maxrows, int ldy) { return kPivotRows * (kPivotRows + 1) / 2 + (maxrows + 1) * ldy + 5 * maxrows + 2 * maxrows; }
SGRL_HD int slab_rows_doubles(int maxrows, int ldy) { return slab_prev_offset(maxrows, ldy) + maxrows + maxrows / 2 + 1 + 8; }
SGRL_HD int slab_doubles(int maxrows, int ldy) { return slab_rows_doubles(maxrows, ldy) + ldy * (ldy + 1) / 2; }
constexpr int kScratchDoublesMax = 2080 + 257 * 47 + 7 * 256 + 256 + 129 + 8 + 47 * 48 / 2;     // slab_doubles(kSlabRows, kSlabLdy): what a test harness may allocate

// ------------------------------------------------------------------------------------------------
// LDS layout (offsets in doubles for S, in ints for I)
struct Layout {
  int nb, nj, nq, nv, nu, np, ncon, maxrows, lrows, ld, ldy;
  int prevcap;     // warm-start rows remembered in LDS (the rest: the HBM slab): min(maxrows, kPrevRows); paired layouts: no more than lrows
  int fstride;     // doubles per contact frame slot: 9 (normal, tangent 1, [tangent 2: recomputed, slot unused]); paired layouts and layouts without an LDS copy of the int tables: 6
  // S
  int qpos, qvel, q0, v0, xv, fq, dvacc, daacc, ctrl, act;
  int xpos, xquat, xmat, xipos, xanchor, xaxis;
  int cinert, crb, cdof, cfrc;
  int dead, dead_len, na_max;
  int L, dinv, qfs, qacc, vpgs;
  int con_pos, con_frame, con_dist;
  int Y, eR, earef, eb, ef, eidg, prev_f;
  int misc;  // 16 scalars
  int Mfull; // Euler only: copy of M (lower triangle incl. diag)
  int mfull_hbm;   // 1: the copy waits in the environment's HBM scratch between CRBA and the end of the substep and is
                   // factored inside the (then idle) constraint-row block, Mfull = Y: no LDS of its own (cheetah_14: 8.3 KB)
  int model_f; // LDS copy of the float model blob (n_f64 doubles)
  int s_total;
  // I
  int con_valid, row_kind, row_src, row_sub, prev_key, flist, ecnt, icnt;
  int model_i; // LDS copy of the int model blob (n_int ints)
  int i_total;
  // TWO environments of one morphology per workgroup (wave_half.h): environment B's slab lies pair_stride doubles behind A's
  // (S_B = S_A + pair_stride, I_B = I_A + 2 pair_stride) and ONE copy of the int model tables behind both (model_i is relative
  // to I_A).  0: one environment per workgroup.
  int pair_stride;
};

SGRL_HD int full_na(int len) {      // rows whose packed factor fits `len` doubles (at most kNAMax)
  int na = 0;
  while (na < kNAMax && (na + 1) * (na + 2) / 2 <= len) na++;
  return na;
}
// lrows_cut: rows taken off the natural size of the LDS row arrays (make_layout picks it)
SGRL_HD void make_layout_rows(const int32_t* hdr, Layout* o, int n_int, int n_f64, int lrows_cut, bool pair = false) {
  const int nb = hdr[SGRL_H_NBODY], nj = hdr[SGRL_H_NJNT], nq = hdr[SGRL_H_NQ], nv = hdr[SGRL_H_NV];
  const int nu = hdr[SGRL_H_NU], np = hdr[SGRL_H_NPAIR];
  o->nb = nb; o->nj = nj; o->nq = nq; o->nv = nv; o->nu = nu; o->np = np;
  o->ncon = 2 * np;
  o->maxrows = hdr[SGRL_H_MAX_ROWS];
  o->ld = nv | 1;
  o->ldy = nv | 1;
  int p = 0;
  // the Runge-Kutta stage state (q0, v0, xv, dvacc, daacc) exists only where the integrator is RK4: the Euler step never touches it
  const bool rk4_state = hdr[SGRL_H_INTEGRATOR] != 0;
  o->qpos = p; p += nq; o->qvel = p; p += nv;
  o->q0 = rk4_state ? p : o->qpos; p += rk4_state ? nq : 0; o->v0 = rk4_state ? p : o->qvel; p += rk4_state ? nv : 0;
  // PAIRED layouts (two slabs must fit one workgroup's LDS at eight workgroups per CU: wave_half.h) are put on a diet that the
  // one-environment layouts do not need (theirs stay byte for byte what rounds 3-4 tuned): xv is qvel (mj_step writes both with the
  // same value and nothing changes qvel in between), the composite inertias live in the constraint-row block (dead until the rows
  // are built), contact frames keep two vectors (the third is their cross product), the reciprocal pivots of the unused in-place
  // factorisation have no array, the warm-start memory in LDS ends with the LDS rows.
  const bool xv_own = rk4_state && !pair;
  o->xv = xv_own ? p : o->qvel; p += xv_own ? nv : 0; o->fq = p; p += nv;
  o->dvacc = rk4_state ? p : o->qvel; p += rk4_state ? nv : 0; o->daacc = rk4_state ? p : o->qvel; p += rk4_state ? nv : 0;
  o->ctrl = p; p += nu + 1;
  o->xpos = p; p += 3 * nb; o->xaxis = p; p += 3 * nj; o->cdof = p; p += 6 * nv;
  // "dead zone": everything below is no longer needed once the constraint rows of an evaluation are built, so the
  // LCP solver reuses the span as scratch for its Cholesky factor
  o->dead = p;
  o->xquat = p; p += 4 * nb; o->xmat = p; p += 9 * nb; o->xipos = p; p += 3 * nb; o->xanchor = p; p += 3 * nj;
  o->fstride = (pair || n_int == 0) ? 6 : 9;     // (n_int == 0: a kernel that reads the int tables from global memory -- the dieted slabs)
  o->cinert = p; p += 10 * nb; o->crb = p; p += pair ? 0 : 10 * nb; o->cfrc = p; p += 6 * nb;
  o->con_pos = p; p += 3 * o->ncon; o->con_frame = p; p += o->fstride * o->ncon; o->con_dist = p; p += o->ncon;
  o->dead_len = p - o->dead;
  // rows the factor scratch can serve: from the span a one-environment layout has (a paired layout keeps the same LDS row count and
  // pads its shorter span up to the factor's size below)
  const int full_len = o->dead_len + (pair ? 10 * nb + 3 * o->ncon : 0);
  {
    int na = 0;
    while (na < kNAMax && (na + 1) * (na + 2) / 2 <= full_len) na++;
    o->na_max = na;
  }
  if (pair) {
    const int want = (o->maxrows < o->na_max ? o->maxrows : o->na_max) - lrows_cut;      // = lrows below
    const int need = want * (want + 1) / 2;
    if (o->dead_len < need) { p += need - o->dead_len; o->dead_len = need; }
    int na = 0;
    while (na < kNAMax && (na + 1) * (na + 2) / 2 <= o->dead_len) na++;
    o->na_max = na < want ? want : na;          // (na >= want by construction; the scratch serves free sets up to na rows)
  }
  const bool dinv_own = !(pair && rk4_state && nv <= 15);      // paired RK4 sets factor on registers with the explicit inverse (HalfWaveT::chol_inv_packed: n <= 15, the same bound): dinv is never touched
  o->L = p; p += nv * (nv + 1) / 2; o->dinv = dinv_own ? p : o->L; p += dinv_own ? nv : 0;
  o->qfs = p; p += nv; o->qacc = p; p += nv; o->vpgs = p; p += nv;
  // The constraint-row arrays in LDS hold `lrows` rows: as many as the factor scratch (dead zone) can serve.  The rare
  // evaluations with more rows (up to maxrows <= 64) run their whole constraint stage out of the environment's HBM slab
  // instead (Engine::Rows) -- sizing the slab for the common case buys one or two more workgroups per CU.
  o->lrows = pair ? (o->maxrows < full_na(full_len) ? o->maxrows : full_na(full_len)) - lrows_cut
                  : (o->maxrows < o->na_max ? o->maxrows : o->na_max) - lrows_cut;
  o->Y = p; p += (o->lrows + 1) * o->ldy;
  if (pair) o->crb = o->Y;      // 10 nb doubles of a block of (lrows + 1) ldy: written and read inside crba_and_factor only
  o->eR = p; p += o->lrows; o->earef = p; p += o->lrows; o->eb = p; p += o->lrows;
  o->ef = p; p += o->lrows;
  // (the block-pivot solve parks up to lrows values in prev_f: at least lrows entries)
  const int prevfull = o->maxrows < kPrevRows ? o->maxrows : kPrevRows;
  const int prevcap = (pair && o->lrows < prevfull) ? o->lrows : prevfull;
  o->prevcap = prevcap;
  o->eidg = p; p += o->lrows; o->prev_f = p; p += prevcap;
  o->misc = p; p += 16;
  o->Mfull = p;
  o->mfull_hbm = 0;
  if (hdr[SGRL_H_INTEGRATOR] == 0) {
    if ((o->lrows + 1) * o->ldy >= nv * (nv + 1) / 2) { o->Mfull = o->Y; o->mfull_hbm = 1; }
    else p += nv * (nv + 1) / 2;
  }
  o->model_f = p; p += n_f64;
  o->s_total = p;
  int q = 0;
  o->con_valid = q; q += o->ncon;
  o->row_kind = q; q += o->lrows; o->row_src = q; q += o->lrows; o->row_sub = q; q += o->lrows;
  o->prev_key = q; q += prevcap;
  o->flist = q; q += o->lrows;
  o->ecnt = q; q += 2 * nj + o->ncon;
  o->icnt = q; q += 8;
  o->pair_stride = 0;
  if (pair) {      // [S_A | I_A] [S_B | I_B] [model ints]: the shared tables start where environment B's slab ends
    q = (q + 1) & ~1;
    o->pair_stride = o->s_total + q / 2;
    q += 2 * o->pair_stride;
  }
  o->model_i = q; q += n_int;
  o->i_total = q;
}

SGRL_HD int layout_bytes(const Layout* o) { return o->s_total * 8 + ((o->i_total + 1) & ~1) * 4; }

// LDS is allocated in 1280-byte granules, 160 KB per CU, and the register budget admits 8 workgroups per CU.
SGRL_HD int workgroups_per_cu(int lds_bytes) {
  const int per = (160 * 1024) / (((lds_bytes + 1279) / 1280) * 1280);
  return per > 8 ? 8 : per;
}

constexpr int kMaxRowCut = 16;
// The layout with the LDS row arrays at their natural size (what the factor scratch can serve) -- or up to eight rows
// shorter (never below 20) when that is what it takes to fit one more workgroup per CU: walker_7 22 392 B (32 rows, 7 per
// CU) -> 20 344 B (24 rows, 8 per CU), measured 10 % faster on the walker mix although more evaluations (25..32 rows) then
// take the HBM slab path.  Evaluations with more rows than the arrays hold use the slab either way: nothing is dropped.
SGRL_HD void make_layout(const int32_t* hdr, Layout* o, int n_int = 0, int n_f64 = 0, bool pair = false) {
  make_layout_rows(hdr, o, n_int, n_f64, 0, pair);
  const int base = workgroups_per_cu(layout_bytes(o));
  if (pair && base < 8) {
    // a slab pair is only used at eight workgroups per CU (the engine declines it otherwise): the smallest cut that gets there
    // (never below 16 rows: a foot flat on the floor is 8 rows, two are 16)
    for (int cut = 1; cut <= kMaxRowCut && o->lrows - cut >= 16; cut++) {
      Layout t;
      make_layout_rows(hdr, &t, n_int, n_f64, cut, pair);
      if (workgroups_per_cu(layout_bytes(&t)) >= 8) { *o = t; return; }
    }
    return;
  }
  if (n_int == 0) {
    // Dieted layouts (a kernel that reads the int tables from global memory: engine_kernel.h SGRL_ITAB_GLOBAL): the row cut that gives
    // the MOST residents, down to 19 rows -- humanoid_9 then fits seven times into a CU's LDS (23 008 B) where its default layout
    // already runs at 21 rows for six.  Two plain loops (the highest count, then the first cut that reaches it) so that a
    // fixed-dimension kernel still folds the whole layout into constants; the default layouts below keep the loop rounds 3-4 tuned
    // (any other shape of it cost the walker kernel its constant layout: 71 spilled registers, twice the store traffic).
    const int natural = o->lrows;
    int best = base;
    for (int cut = 1; cut <= kMaxRowCut && natural - cut >= 19; cut++) {
      Layout t;
      make_layout_rows(hdr, &t, n_int, n_f64, cut, pair);
      const int w = workgroups_per_cu(layout_bytes(&t));
      if (w > best) best = w;
    }
    if (best == base) return;
    for (int cut = 1; cut <= kMaxRowCut && natural - cut >= 19; cut++) {
      Layout t;
      make_layout_rows(hdr, &t, n_int, n_f64, cut, pair);
      if (workgroups_per_cu(layout_bytes(&t)) == best) { *o = t; return; }
    }
    return;
  }
  for (int cut = 1; cut <= kMaxRowCut && o->lrows - cut >= 20; cut++) {
    Layout t;
    make_layout_rows(hdr, &t, n_int, n_f64, cut, pair);
    if (workgroups_per_cu(layout_bytes(&t)) > base) { *o = t; return; }
  }
}

// misc slots
enum { MS_COM = 0, /* 3 */ MS_REWARD = 4, MS_DIST = 5, MS_PREQUAT = 6 /* 4 */, MS_PREPOS = 10 /* 2 */ };
// icnt slots
enum { IC_NROW = 0, IC_NROW_WANTED = 1, IC_OVERFLOW = 2, IC_DONE = 3, IC_TRUNC = 4, IC_PREV_N = 5, IC_SWEEPS = 6, IC_ROWSUM = 7 };
enum { ROW_LIMIT_LO = 0, ROW_LIMIT_HI = 1, ROW_CON1 = 2, ROW_PYR = 3 };

// ------------------------------------------------------------------------------------------------
// small vector helpers (register arrays with static indexing)
SGRL_DEV void cross3(double* r, const double* a, const double* b) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
SGRL_DEV double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
SGRL_DEV double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
SGRL_DEV void quat_mul(double* r, const double* a, const double* b) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
SGRL_DEV double inv_sqrt(double x) {
#ifdef __HIPCC__
  return rsqrt(x);        // one transcendental + refinement instead of sqrt followed by a division (dependent chain)
#else
  return 1.0 / sqrt(x);
#endif
}
SGRL_DEV void quat_normalize(double* q) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 < kMinVal * kMinVal) { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; return; }
  const double s = inv_sqrt(n2);
  q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
SGRL_DEV void quat2mat(double* m, const double* q) {
  const double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q[1] * q[2] - q[0] * q[3]); m[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
  m[3] = 2 * (q[1] * q[2] + q[0] * q[3]); m[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
  m[6] = 2 * (q[1] * q[3] - q[0] * q[2]); m[7] = 2 * (q[2] * q[3] + q[0] * q[1]);
}
SGRL_DEV void mat_vec(double* r, const double* m, const double* v) {
  const double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  const double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  const double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
SGRL_DEV void axisangle2quat(double* q, const double* axis, double angle) {
  if (angle == 0.0) { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; return; }
  double s, c;
  sincos(0.5 * angle, &s, &c);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
// spatial inertia (Ixx Iyy Izz Ixy Ixz Iyz | hx hy hz | m) times motion [w; v] -> force [tau; F]
SGRL_DEV void inert_mul(double* f, const double* I, const double* mv) {
  const double* w = mv; const double* v = mv + 3; const double* h = I + 6;
  double hv[3], hw[3];
  cross3(hv, h, v); cross3(hw, h, w);
  f[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2] + hv[0];
  f[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2] + hv[1];
  f[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2] + hv[2];
  f[3] = I[9] * v[0] - hw[0]; f[4] = I[9] * v[1] - hw[1]; f[5] = I[9] * v[2] - hw[2];
}
SGRL_DEV void cross_motion(double* r, const double* vel, const double* m) {
  double a[3], b[3], c[3];
  cross3(a, vel, m); cross3(b, vel, m + 3); cross3(c, vel + 3, m);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
SGRL_DEV void cross_force(double* r, const double* vel, const double* f) {
  double a[3], b[3], c[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
SGRL_DEV void ld3(double* r, const double* p) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; }
SGRL_DEV void ld6(double* r, const double* p) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3]; r[4] = p[4]; r[5] = p[5]; }
SGRL_DEV int popcount64(uint64_t x) {
#ifdef __HIPCC__
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}
SGRL_DEV int clz64(uint64_t x) {
#ifdef __HIPCC__
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}
SGRL_DEV int tri(int i) { return i * (i + 1) / 2; }   // packed lower triangle: (i, j) at tri(i) + j
// p = a(a+1)/2 + b with b <= a  ->  (a, b)
SGRL_DEV void tri_decode(int p, int* a_out, int* b_out) {
  // float sqrt is within one of the exact root for p < 2^20; two branch-free corrections make it exact
  int a = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
  a += ((a + 1) * (a + 2) / 2 <= p) ? 1 : 0;
  a -= (a * (a + 1) / 2 > p) ? 1 : 0;
  *a_out = a; *b_out = p - a * (a + 1) / 2;
}
SGRL_DEV bool dof_in_mask(sgrl_itab_t mask2, int d) {
  const uint32_t w = (uint32_t)(d < 32 ? mask2[0] : mask2[1]);
  return (w >> (d & 31)) & 1u;
}

// counter RNG shared bit-for-bit with the CPU oracle (Philox4x32-10)
SGRL_DEV double rng_uniform01(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t stream, uint32_t idx) {
  uint32_t c0 = idx >> 2, c1 = episode, c2 = stream, c3 = (uint32_t)(seed >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = env_id;
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const uint32_t lane = idx & 3u;
  const uint32_t x = lane == 0 ? c0 : (lane == 1 ? c1 : (lane == 2 ? c2 : c3));
  return ((double)x + 0.5) * (1.0 / 4294967296.0);
}

#ifdef SGRL_PHASE_PROF
#define SGRL_TICK(id) w.tick(id)
#else
#define SGRL_TICK(id)
#endif

// ------------------------------------------------------------------------------------------------
template <class W>
struct Engine {
  W& w;
  SgrlModelView m;             // own copy: the int-table pointers are re-fenced every evaluation (fence_view)
  const Layout& o;
  double* S;
  int32_t* I;
  double* big_scratch = nullptr;   // optional per-env HBM slab: 2 * 64*65/2 doubles
  bool linv = false;               // this evaluation's S[o.L] holds L^-1 (explicit inverse of the factor), not L

  SGRL_DEV Engine(W& w_, const SgrlModelView& m_, const Layout& o_, double* S_, int32_t* I_)
      : w(w_), m(m_), o(o_), S(S_), I(I_) {}

  // The int tables live in LDS: an access needs their (uniform) base in a VGPR.  Left alone, the compiler materialises
  // all ~26 bases as VGPRs once and keeps them live across the whole step (then spills them); making the bases opaque
  // at the top of every evaluation confines those copies to where they are used.
  SGRL_DEV void fence_view() {
    if (W::kFixedDims) return;     // a fixed-dimension kernel instance: the table addresses are compile-time constants
    m.body_parent = w.fenced(m.body_parent); m.body_jntadr = w.fenced(m.body_jntadr);
    m.body_jntnum = w.fenced(m.body_jntnum); m.body_dofadr = w.fenced(m.body_dofadr);
    m.body_dofnum = w.fenced(m.body_dofnum); m.body_limbtype = w.fenced(m.body_limbtype);
    m.jnt_type = w.fenced(m.jnt_type); m.jnt_body = w.fenced(m.jnt_body);
    m.jnt_qposadr = w.fenced(m.jnt_qposadr); m.jnt_dofadr = w.fenced(m.jnt_dofadr);
    m.jnt_limited = w.fenced(m.jnt_limited); m.dof_body = w.fenced(m.dof_body); m.dof_jnt = w.fenced(m.dof_jnt);
    m.dof_parent = w.fenced(m.dof_parent); m.geom_type = w.fenced(m.geom_type);
    m.geom_body = w.fenced(m.geom_body); m.pair_g1 = w.fenced(m.pair_g1); m.pair_g2 = w.fenced(m.pair_g2);
    m.pair_condim = w.fenced(m.pair_condim); m.act_dof = w.fenced(m.act_dof); m.act_slot = w.fenced(m.act_slot);
    m.body_depth = w.fenced(m.body_depth); m.body_path = w.fenced(m.body_path);
    m.body_subend = w.fenced(m.body_subend); m.body_dofmask = w.fenced(m.body_dofmask);
    m.dof_act = w.fenced(m.dof_act);
  }

  // ---- position stage -------------------------------------------------------------------------
  // rotate v by the unit quaternion q (through its matrix, as the oracle does)
  SGRL_DEV static void quat_rot(double* r, const double* q, const double* v) {
    double mtx[9];
    quat2mat(mtx, q);
    mat_vec(r, mtx, v);
  }

  // Body poses.  A body's pose is the product of the LOCAL transforms along its chain, and what made the chain walk slow was
  // not its arithmetic but the look-ups inside it (path -> joints of the body -> per-joint constants: four dependent LDS round
  // trips per joint, three joints per level).  So the local transform of every body -- body offset, then its hinges: rotation
  // about the anchor -- is composed FIRST, one lane per body (phase 2), and the walk (phase 3) is one quaternion product and one
  // rotated offset per level with every operand address known up front.  Joint anchors / axes in the world frame follow from
  // the parent's pose, one lane per hinge (phase 4).  Same products as oracle/physics.c kinematics() in another association
  // (and one normalisation per body at the end of its own chain instead of one per level): agreement to rounding.
  // Scratch: the constraint rows' Y block, dead until build_rows -- per joint 17 doubles (rotation 4 | axis 3 | position 3 |
  // local quaternion before the joint 4 | local anchor 3), per body 7 (local offset 3 | local quaternion 4).
  SGRL_DEV void kinematics() {
    // root quaternion normalised in place first (mj_kinematics of 2.1.0 [3P-knowledge]; oracle/physics.c kinematics())
    w.lanes(1, [&](int) {
      double q[4] = {S[o.qpos + 3], S[o.qpos + 4], S[o.qpos + 5], S[o.qpos + 6]};
      quat_normalize(q);
      S[o.qpos + 3] = q[0]; S[o.qpos + 4] = q[1]; S[o.qpos + 5] = q[2]; S[o.qpos + 6] = q[3];
    });
    const int TJ = o.Y, TB = o.Y + 17 * o.nj;
    // phase 1: one lane per hinge -- its rotation (sincos once per joint) and its constants from the float tables (L2)
    w.lanes(o.nj, [&](int j) {
      if (m.jnt_type[j] == SGRL_JNT_FREE) return;
      double ja[3], ql[4];
      ld3(ja, m.jnt_axis + 3 * j);
      const int qa = m.jnt_qposadr[j];
      axisangle2quat(ql, ja, S[o.qpos + qa] - m.qpos0[qa]);
      double* t = S + TJ + 17 * j;
      for (int k = 0; k < 4; k++) t[k] = ql[k];
      for (int k = 0; k < 3; k++) { t[4 + k] = ja[k]; t[7 + k] = m.jnt_pos[3 * j + k]; }
    });
    // phase 2: one lane per body below the torso -- local transform (parent frame -> body frame after its joints)
    w.lanes(o.nb, [&](int c) {
      if (c < 2) return;
      double p[3], q[4];
      ld3(p, m.body_pos + 3 * c);
      for (int k = 0; k < 4; k++) q[k] = m.body_quat[4 * c + k];
      const int j0 = m.body_jntadr[c], jn = m.body_jntnum[c];
      for (int j = j0; j < j0 + jn; j++) {
        double* t = S + TJ + 17 * j;
        double ql[4], jp[3], anchor[3], v[3], qn[4];
        for (int k = 0; k < 4; k++) ql[k] = t[k];
        ld3(jp, t + 7);
        quat_rot(v, q, jp);
        for (int k = 0; k < 3; k++) anchor[k] = p[k] + v[k];
        for (int k = 0; k < 4; k++) t[10 + k] = q[k];              // local orientation BEFORE this joint: what turns its axis
        for (int k = 0; k < 3; k++) t[14 + k] = anchor[k];
        quat_mul(qn, q, ql);
        for (int k = 0; k < 4; k++) q[k] = qn[k];
        quat_rot(v, q, jp);                                        // exactly zero for a joint at the body origin
        for (int k = 0; k < 3; k++) p[k] = anchor[k] - v[k];
      }
      double* u = S + TB + 7 * c;
      for (int k = 0; k < 3; k++) u[k] = p[k];
      for (int k = 0; k < 4; k++) u[3 + k] = q[k];
    });
    // phase 3: one lane per body -- its own chain of local transforms
    w.lanes(o.nb, [&](int b) {
      if (b == 0) {
        for (int k = 0; k < 3; k++) S[o.xpos + k] = 0;
        S[o.xquat] = 1; S[o.xquat + 1] = 0; S[o.xquat + 2] = 0; S[o.xquat + 3] = 0;
        for (int k = 0; k < 9; k++) S[o.xmat + k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
      }
      double pos[3], quat[4], mat[9];
      const int depth = m.body_depth[b];
      // level 0 = torso with the free joint
      for (int k = 0; k < 3; k++) pos[k] = S[o.qpos + k];
      for (int k = 0; k < 4; k++) quat[k] = S[o.qpos + 3 + k];
      quat_normalize(quat);
      for (int lvl = 1; lvl < depth; lvl++) {
        const int c = m.body_path[8 * b + lvl];
        const double* u = S + TB + 7 * c;
        double pl[3], qloc[4], t[3], qn[4];
        ld3(pl, u);
        for (int k = 0; k < 4; k++) qloc[k] = u[3 + k];
        quat_rot(t, quat, pl);
        for (int k = 0; k < 3; k++) pos[k] += t[k];
        quat_mul(qn, quat, qloc);
        for (int k = 0; k < 4; k++) quat[k] = qn[k];
      }
      quat_normalize(quat);
      quat2mat(mat, quat);
      for (int k = 0; k < 3; k++) S[o.xpos + 3 * b + k] = pos[k];
      for (int k = 0; k < 4; k++) S[o.xquat + 4 * b + k] = quat[k];
      for (int k = 0; k < 9; k++) S[o.xmat + 9 * b + k] = mat[k];
      double ip[3], t[3];
      ld3(ip, m.body_ipos + 3 * b);
      mat_vec(t, mat, ip);
      for (int k = 0; k < 3; k++) S[o.xipos + 3 * b + k] = pos[k] + t[k];
    });
    // phase 4: one lane per joint -- anchor and axis in the world frame from the PARENT body's pose
    w.lanes(o.nj, [&](int j) {
      if (m.jnt_type[j] == SGRL_JNT_FREE) {
        const int b = m.jnt_body[j];
        for (int k = 0; k < 3; k++) { S[o.xanchor + 3 * j + k] = S[o.xpos + 3 * b + k]; S[o.xaxis + 3 * j + k] = (k == 2) ? 1.0 : 0.0; }
        return;
      }
      const int par = m.body_parent[m.jnt_body[j]];
      const double* t = S + TJ + 17 * j;
      double pp[3], qp[4], qb[4], ql[4], al[3], ja[3], v[3], ax[3];
      ld3(pp, S + o.xpos + 3 * par);
      for (int k = 0; k < 4; k++) { qp[k] = S[o.xquat + 4 * par + k]; ql[k] = t[10 + k]; }
      ld3(al, t + 14); ld3(ja, t + 4);
      quat_rot(v, qp, al);
      quat_mul(qb, qp, ql);
      quat_rot(ax, qb, ja);
      for (int k = 0; k < 3; k++) { S[o.xanchor + 3 * j + k] = pp[k] + v[k]; S[o.xaxis + 3 * j + k] = ax[k]; }
    });
  }

  SGRL_DEV void com_pos() {
    const double mt = m.fhdr[SGRL_F_TOTAL_MASS];
    // the three wave sums come back in every lane: they are used straight from registers by the phase below (one phase
    // instead of seven); lane 0 also parks them in LDS for the later stages (constraint rows, body velocities)
    double comr[3];
    for (int k = 0; k < 3; k++)
      comr[k] = w.sum(o.nb, [&](int b) { return b == 0 ? 0.0 : m.body_mass[b] * S[o.xipos + 3 * b + k]; }) / mt;
    w.lanes(o.nb > o.nv ? o.nb : o.nv, [&](int i) {
      const double com[3] = {comr[0], comr[1], comr[2]};
      if (i == 0) for (int k = 0; k < 3; k++) S[o.misc + MS_COM + k] = com[k];
      if (i >= 1 && i < o.nb) {
        const int b = i;
        double R[9], Wm[9], RI[9];
        for (int k = 0; k < 9; k++) R[k] = S[o.xmat + 9 * b + k];
        sgrl_ftab_t ib = m.body_inertia + 6 * b;
        const double Im[9] = {ib[0], ib[3], ib[4], ib[3], ib[1], ib[5], ib[4], ib[5], ib[2]};
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) RI[3 * r + c] = R[3 * r] * Im[c] + R[3 * r + 1] * Im[3 + c] + R[3 * r + 2] * Im[6 + c];
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) Wm[3 * r + c] = RI[3 * r] * R[3 * c] + RI[3 * r + 1] * R[3 * c + 1] + RI[3 * r + 2] * R[3 * c + 2];
        double d[3];
        const double mass = m.body_mass[b];
        for (int k = 0; k < 3; k++) d[k] = S[o.xipos + 3 * b + k] - com[k];
        const double dd = dot3(d, d);
        double* ci = S + o.cinert + 10 * b;
        ci[0] = Wm[0] + mass * (dd - d[0] * d[0]); ci[1] = Wm[4] + mass * (dd - d[1] * d[1]);
        ci[2] = Wm[8] + mass * (dd - d[2] * d[2]);
        ci[3] = Wm[1] - mass * d[0] * d[1]; ci[4] = Wm[2] - mass * d[0] * d[2]; ci[5] = Wm[5] - mass * d[1] * d[2];
        ci[6] = mass * d[0]; ci[7] = mass * d[1]; ci[8] = mass * d[2]; ci[9] = mass;
      }
      if (i < o.nv) {
        const int d = i, j = m.dof_jnt[d], b = m.dof_body[d];
        double* c = S + o.cdof + 6 * d;
        if (m.jnt_type[j] == SGRL_JNT_FREE) {
          const int k = d - m.jnt_dofadr[j];
          if (k < 3) {
            c[0] = 0; c[1] = 0; c[2] = 0; c[3] = (k == 0); c[4] = (k == 1); c[5] = (k == 2);
          } else {
            const int kk = k - 3;
            double ax[3] = {S[o.xmat + 9 * b + kk], S[o.xmat + 9 * b + 3 + kk], S[o.xmat + 9 * b + 6 + kk]};
            double off[3], cr[3];
            for (int t = 0; t < 3; t++) off[t] = com[t] - S[o.xpos + 3 * b + t];
            cross3(cr, ax, off);
            c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2]; c[3] = cr[0]; c[4] = cr[1]; c[5] = cr[2];
          }
        } else {
          double ax[3], off[3], cr[3];
          ld3(ax, S + o.xaxis + 3 * j);
          for (int t = 0; t < 3; t++) off[t] = com[t] - S[o.xanchor + 3 * j + t];
          cross3(cr, ax, off);
          c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2]; c[3] = cr[0]; c[4] = cr[1]; c[5] = cr[2];
        }
      }
    });
  }

  SGRL_DEV void crba_and_factor() {
    const int nv = o.nv;
    // composite inertias: subtree(b) = [b, subend[b]) in pre-order; and zero the lower triangle of M
    w.lanes(o.nb > nv ? o.nb : nv, [&](int i) {
      if (i >= 1 && i < o.nb) {
        double acc[10];
        for (int k = 0; k < 10; k++) acc[k] = 0;
        const int e = m.body_subend[i];
        for (int c = i; c < e; c++)
          for (int k = 0; k < 10; k++) acc[k] += S[o.cinert + 10 * c + k];
        for (int k = 0; k < 10; k++) S[o.crb + 10 * i + k] = acc[k];
      }
      if (i < nv) for (int k = 0; k <= i; k++) S[o.L + tri(i) + k] = 0;
    });
    w.lanes(nv, [&](int i) {
      double buf[6], ci[10], cd[6];
      const int b = m.dof_body[i];
      for (int k = 0; k < 10; k++) ci[k] = S[o.crb + 10 * b + k];
      ld6(cd, S + o.cdof + 6 * i);
      inert_mul(buf, ci, cd);
      S[o.L + tri(i) + i] = dot6(cd, buf) + m.dof_armature[i];
      for (int j = m.dof_parent[i]; j >= 0; j = m.dof_parent[j]) {
        double cj[6];
        ld6(cj, S + o.cdof + 6 * j);
        S[o.L + tri(i) + j] = dot6(cj, buf);
      }
    });
    if (W::hdr_const(m, SGRL_H_INTEGRATOR) == 0) {  // Euler needs M again for (M + h D)
      if (o.mfull_hbm) {
        double* const mh = big_scratch + slab_rows_doubles(o.maxrows, o.ldy);
        const int nt = nv * (nv + 1) / 2;
        w.lanes(64, [&](int l) { for (int k = l; k < nt; k += 64) mh[k] = S[o.L + k]; });
      } else {
        w.lanes(nv, [&](int i) { for (int k = 0; k <= i; k++) S[o.Mfull + tri(i) + k] = S[o.L + tri(i) + k]; });
      }
    }
    SGRL_TICK(2);
    // small systems: Cholesky and the explicit inverse of the factor in one register sweep (wave policy); the
    // triangular solves below then become recurrence-free dot products.  Otherwise factor in place.
    linv = w.chol_inv_packed(nv, S + o.L, kMinVal);
    if (!linv) cholesky(o.L);
    SGRL_TICK(8);
  }

  // in-place lower Cholesky of the matrix at S[base] (packed lower triangle); diag reciprocals -> dinv
  SGRL_DEV void cholesky(int base) {
    const int nv = o.nv;
    // right-looking root-free elimination: after pivot j every trailing entry (i, k), j < k <= i, takes ONE fused
    // update  M_ik -= M_ij M_kj / M_jj  -- constant depth per pivot (one barrier), all lanes busy on the triangle
    for (int j = 0; j < nv - 1; j++) {
      double pj = S[base + tri(j) + j];
      if (pj < kMinVal) pj = kMinVal;
      const double wj = 1.0 / pj;
      const int sdim = nv - j - 1;
      w.lanes(sdim * (sdim + 1) / 2, [&](int p) {
        int a, b;
        tri_decode(p, &a, &b);
        const int i = j + 1 + a, k = j + 1 + b;
        S[base + tri(i) + k] -= S[base + tri(i) + j] * S[base + tri(k) + j] * wj;
      });
    }
    // L[i][j] = C[i][j] / sqrt(C[j][j]),  dinv[j] = 1 / sqrt(C[j][j])
    w.lanes(nv, [&](int j) {
      double pj = S[base + tri(j) + j];
      if (pj < kMinVal) pj = kMinVal;
      S[o.dinv + j] = sqrt(1.0 / pj);
    });
    w.lanes(nv * (nv - 1) / 2, [&](int p) {
      int a, b;
      tri_decode(p, &a, &b);          // strict lower triangle: i = a + 1 > j = b
      S[base + tri(a + 1) + b] *= S[o.dinv + b];
    });
  }

  // x (at S[xo], nv values) <- L^-T x  /  L^-1 x
  SGRL_DEV void solve_upper_inplace(int base, int xo) { w.trsv_upper(o.nv, S + base, S + o.dinv, S + xo); }
  SGRL_DEV void solve_lower_inplace(int base, int xo) { w.trsv_lower(o.nv, S + base, S + o.dinv, S + xo); }

  // ---- collision ------------------------------------------------------------------------------
  SGRL_DEV void geom_pose(int g, double* pos, double* mat) {
    const int b = m.geom_body[g];
    double t[3], gp[3], gq[4], gm[9], xm[9];
    ld3(gp, m.geom_pos + 3 * g);
    for (int k = 0; k < 9; k++) xm[k] = S[o.xmat + 9 * b + k];
    mat_vec(t, xm, gp);
    for (int k = 0; k < 3; k++) pos[k] = S[o.xpos + 3 * b + k] + t[k];
    for (int k = 0; k < 4; k++) gq[k] = m.geom_quat[4 * g + k];
    quat2mat(gm, gq);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) mat[3 * r + c] = xm[3 * r] * gm[c] + xm[3 * r + 1] * gm[3 + c] + xm[3 * r + 2] * gm[6 + c];
  }
  SGRL_DEV void make_frame(double* fr) {
    const double n2 = sqrt(dot3(fr + 3, fr + 3));
    if (n2 < 0.5) {
      fr[3] = 0; fr[4] = 0; fr[5] = 0;
      if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1; else fr[5] = 1;
    }
    const double d = dot3(fr, fr + 3);
    for (int k = 0; k < 3; k++) fr[3 + k] -= d * fr[k];
    const double n = sqrt(dot3(fr + 3, fr + 3));
    if (n < kMinVal) { fr[3] = 1; fr[4] = 0; fr[5] = 0; }
    else for (int k = 0; k < 3; k++) fr[3 + k] /= n;
    cross3(fr + 6, fr, fr + 3);
  }
  SGRL_DEV void put_contact(int slot, bool valid, double dist, const double* pos, const double* normal, const double* tangent) {
    I[o.con_valid + slot] = valid ? 1 : 0;
    if (!valid) return;
    double fr[9];
    for (int k = 0; k < 3; k++) { fr[k] = normal[k]; fr[3 + k] = tangent ? tangent[k] : 0.0; }
    make_frame(fr);
    S[o.con_dist + slot] = dist;
    for (int k = 0; k < 3; k++) S[o.con_pos + 3 * slot + k] = pos[k];
    for (int k = 0; k < 6; k++) S[o.con_frame + o.fstride * slot + k] = fr[k];      // normal, tangent 1 (tangent 2 = normal x tangent 1: build_rows)
  }
  SGRL_DEV void plane_sphere(int slot, double margin, const double* ppos, const double* n, const double* c, double r, const double* tangent) {
    const double d[3] = {c[0] - ppos[0], c[1] - ppos[1], c[2] - ppos[2]};
    const double dist = dot3(d, n) - r;
    double pos[3];
    for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * (r + 0.5 * dist);
    put_contact(slot, dist < margin, dist, pos, n, tangent);
  }
  SGRL_DEV void collide() {
    w.lanes(o.np, [&](int p) {
      const int g1 = m.pair_g1[p], g2 = m.pair_g2[p];
      double p1[3], m1[9], p2[3], m2[9];
      geom_pose(g1, p1, m1); geom_pose(g2, p2, m2);
      const double margin = m.pair_margin[p];
      const int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
      if (t1 == SGRL_GEOM_PLANE) {
        const double n[3] = {m1[2], m1[5], m1[8]};
        if (t2 == SGRL_GEOM_SPHERE) {
          plane_sphere(2 * p, margin, p1, n, p2, m.geom_size[3 * g2], nullptr);
          I[o.con_valid + 2 * p + 1] = 0;
        } else {
          const double ax[3] = {m2[2], m2[5], m2[8]};
          const double h = m.geom_size[3 * g2 + 1], r = m.geom_size[3 * g2];
          double ca[3], cb[3];
          for (int k = 0; k < 3; k++) { ca[k] = p2[k] + ax[k] * h; cb[k] = p2[k] - ax[k] * h; }
          plane_sphere(2 * p, margin, p1, n, ca, r, ax);
          plane_sphere(2 * p + 1, margin, p1, n, cb, r, ax);
        }
      } else {
        // capsule - capsule (oracle/physics.c collide()): closest points of the two segments; parallel axes: the end
        // spheres of each capsule against the other's axis, at most two contacts
        const double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
        const double h1 = m.geom_size[3 * g1 + 1], h2 = m.geom_size[3 * g2 + 1];
        const double r1 = m.geom_size[3 * g1], r2 = m.geom_size[3 * g2];
        const double d[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        const double bb = dot3(a1, a2), c = dot3(a1, d), f = dot3(a2, d);
        const double den = 1.0 - bb * bb;
        const bool general = fabs(den) >= kMinVal;
        int made = 0;
        I[o.con_valid + 2 * p] = 0;
        I[o.con_valid + 2 * p + 1] = 0;
        for (int q = 0; q < (general ? 1 : 4) && made < 2; q++) {
          double s, t;
          if (general) {
            s = (bb * f - c) / den;
            if (s > h1) s = h1;
            if (s < -h1) s = -h1;
            t = bb * s + f;
            if (t > h2) { t = h2; s = bb * t - c; if (s > h1) s = h1; if (s < -h1) s = -h1; }
            else if (t < -h2) { t = -h2; s = bb * t - c; if (s > h1) s = h1; if (s < -h1) s = -h1; }
          } else if (q < 2) {
            s = q ? -h1 : h1; t = bb * s + f;
            if (t > h2) t = h2;
            if (t < -h2) t = -h2;
          } else {
            t = (q & 1) ? -h2 : h2; s = bb * t - c;
            if (s > h1) s = h1;
            if (s < -h1) s = -h1;
          }
          double c1[3], nn[3];
          for (int k = 0; k < 3; k++) { c1[k] = p1[k] + s * a1[k]; nn[k] = (p2[k] + t * a2[k]) - c1[k]; }
          const double len = sqrt(dot3(nn, nn));
          const double dist = len - r1 - r2;
          if (dist >= margin) continue;
          if (len < kMinVal) { nn[0] = 1; nn[1] = 0; nn[2] = 0; } else for (int k = 0; k < 3; k++) nn[k] /= len;
          double pos[3];
          for (int k = 0; k < 3; k++) pos[k] = c1[k] + nn[k] * (r1 + 0.5 * dist);
          put_contact(2 * p + made, true, dist, pos, nn, nullptr);
          made++;
        }
      }
    });
  }

  // ---- velocity stage: cvel / cacc along each body's own chain, RNE bias, smooth force ------------
  SGRL_DEV void bias_and_smooth_force() {
    w.lanes(o.nb, [&](int b) {
      if (b == 0) { for (int k = 0; k < 6; k++) S[o.cfrc + k] = 0; return; }
      double cv[6] = {0, 0, 0, 0, 0, 0};
      double ca[6] = {0, 0, 0, -m.fhdr[SGRL_F_GRAV_X], -m.fhdr[SGRL_F_GRAV_Y], -m.fhdr[SGRL_F_GRAV_Z]};
      const int depth = m.body_depth[b];
      for (int lvl = 0; lvl < depth; lvl++) {
        const int c = m.body_path[8 * b + lvl];
        const int j0 = m.body_jntadr[c], jn = m.body_jntnum[c];
        for (int j = j0; j < j0 + jn; j++) {
          const int d0 = m.jnt_dofadr[j];
          if (m.jnt_type[j] == SGRL_JNT_FREE) {
            for (int d = 0; d < 3; d++) {
              const double qv = S[o.qvel + d0 + d];
              for (int k = 0; k < 6; k++) cv[k] += S[o.cdof + 6 * (d0 + d) + k] * qv;
            }
            double cd[3][6], dd[6];
            for (int d = 0; d < 3; d++) ld6(cd[d], S + o.cdof + 6 * (d0 + 3 + d));
            for (int d = 0; d < 3; d++) {
              cross_motion(dd, cv, cd[d]);
              const double qv = S[o.qvel + d0 + 3 + d];
              for (int k = 0; k < 6; k++) ca[k] += dd[k] * qv;
            }
            for (int d = 0; d < 3; d++) {
              const double qv = S[o.qvel + d0 + 3 + d];
              for (int k = 0; k < 6; k++) cv[k] += cd[d][k] * qv;
            }
          } else {
            double cd[6], dd[6];
            ld6(cd, S + o.cdof + 6 * d0);
            cross_motion(dd, cv, cd);
            const double qv = S[o.qvel + d0];
            for (int k = 0; k < 6; k++) { ca[k] += dd[k] * qv; cv[k] += cd[k] * qv; }
          }
        }
      }
      double ci[10], f1[6], iv[6], f2[6];
      for (int k = 0; k < 10; k++) ci[k] = S[o.cinert + 10 * b + k];
      inert_mul(f1, ci, ca);
      inert_mul(iv, ci, cv);
      cross_force(f2, cv, iv);
      for (int k = 0; k < 6; k++) S[o.cfrc + 6 * b + k] = f1[k] + f2[k];
    });
    w.lanes(o.nv, [&](int d) {
      const int b = m.dof_body[d], e = m.body_subend[b];
      double f[6] = {0, 0, 0, 0, 0, 0}, cd[6];
      for (int c = b; c < e; c++)
        for (int k = 0; k < 6; k++) f[k] += S[o.cfrc + 6 * c + k];
      ld6(cd, S + o.cdof + 6 * d);
      const double bias = dot6(cd, f);
      const int j = m.dof_jnt[d];
      double passive = -m.dof_damping[d] * S[o.qvel + d];
      if (m.jnt_type[j] == SGRL_JNT_HINGE) {
        const int qa = m.jnt_qposadr[j];
        passive -= m.jnt_stiffness[j] * (S[o.qpos + qa] - m.qpos0[qa]);
      }
      double q = passive - bias;
      const int u = m.dof_act[d];
      if (u >= 0) {
        double c = S[o.ctrl + u];
        const double lo = m.act_ctrlrange[2 * u], hi = m.act_ctrlrange[2 * u + 1];
        if (c < lo) c = lo;
        if (c > hi) c = hi;
        q += m.act_gear[u] * c;
      }
      S[o.qfs + d] = q;
    });
  }

  // ---- constraint rows ------------------------------------------------------------------------
  SGRL_DEV double impedance(const double* solimp, double x_raw) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    if (dmin < kMinImp) dmin = kMinImp;
    if (dmin > kMaxImp) dmin = kMaxImp;
    if (dmax < kMinImp) dmax = kMinImp;
    if (dmax > kMaxImp) dmax = kMaxImp;
    if (width < 0) width = 0;
    if (mid < kMinImp) mid = kMinImp;
    if (mid > kMaxImp) mid = kMaxImp;
    if (power < 1) power = 1;
    if (dmin == dmax || width <= kMinVal) return 0.5 * (dmin + dmax);
    const double x = fabs(x_raw) / width;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    double y;
    if (power == 1) y = x;
    else if (power == 2) y = (x <= mid) ? (x * x) / mid : 1 - ((1 - x) * (1 - x)) / (1 - mid);   // the MJCF default
    else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
    else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
  }
  SGRL_DEV void kb(const double* solref, const double* solimp, double* K, double* B) {
    double dmax = solimp[1];
    if (dmax < kMinImp) dmax = kMinImp;
    if (dmax > kMaxImp) dmax = kMaxImp;
    double tc = solref[0];
    const double dr = solref[1], h2 = 2 * m.fhdr[SGRL_F_TIMESTEP];
    if (tc < h2) tc = h2;
    *K = 1.0 / (dmax * dmax * tc * tc * dr * dr);
    *B = 2.0 / (dmax * tc);
  }

  // row table (serial, lane 0): limits in joint order, then contacts in slot order; same cap rule as the oracle
  // The arrays of one evaluation's constraint rows: in the LDS slab (common case) or in the environment's HBM scratch
  // slab (more rows than the LDS arrays hold).  Every constraint-stage function is inlined once per variant, so the
  // compiler sees LDS or global pointers, never generic ones.
  struct Rows {
    double *Y, *eR, *earef, *eb, *ef, *eidg, *C;
    int32_t *kind, *src, *sub, *flist;
  };
  SGRL_DEV Rows rows_lds() const {
    return Rows{S + o.Y, S + o.eR, S + o.earef, S + o.eb, S + o.ef, S + o.eidg, S + o.dead,
                I + o.row_kind, I + o.row_src, I + o.row_sub, I + o.flist};
  }
  SGRL_DEV Rows rows_hbm() const {   // slab: C[64*65/2] | Y[maxrows + 1][ldy] | 5 x maxrows doubles | 4 x maxrows ints  (slab_doubles)
    double* p = big_scratch;
    const int mr = o.maxrows;
    Rows r;
    r.C = p; p += kPivotRows * (kPivotRows + 1) / 2;
    r.Y = p; p += (mr + 1) * o.ldy;
    r.eR = p; p += mr; r.earef = p; p += mr; r.eb = p; p += mr; r.ef = p; p += mr;
    r.eidg = p; p += mr;
    int32_t* q = reinterpret_cast<int32_t*>(p);
    r.kind = q; r.src = q + mr; r.sub = q + 2 * mr; r.flist = q + 3 * mr;
    return r;
  }

  // number of rows this evaluation wants (before the cap): lane-parallel count per item + a wave sum
  SGRL_DEV int count_rows() {
    // lane-parallel: item t = (joint, side) for t < 2*nj, contact slot otherwise; rows wanted per item, then every
    // lane takes the prefix sum of the items before it.  Falls back to the serial walk only when the cap would bite.
    const int nitem = 2 * o.nj + o.ncon;
    w.lanes(nitem, [&](int t) {
      int c = 0;
      if (t < 2 * o.nj) {
        const int j = t >> 1, side = (t & 1) ? 1 : -1;
        if (m.jnt_limited[j]) {
          const double q = S[o.qpos + m.jnt_qposadr[j]];
          const double dist = side * (m.jnt_range[2 * j + (side + 1) / 2] - q);
          if (dist < m.jnt_margin[j]) c = 1;
        }
      } else {
        const int sl = t - 2 * o.nj;
        if (I[o.con_valid + sl]) { const int dim = m.pair_condim[sl >> 1]; c = (dim == 1) ? 1 : 2 * (dim - 1); }
      }
      I[o.ecnt + t] = c;
    });
    return (int)w.sum(nitem, [&](int t) { return (double)I[o.ecnt + t]; });     // small integers: exact in f64
  }

  // row table (kind, source, sub-index) into the arrays of R; sets IC_NROW / IC_NROW_WANTED and counts overflows
  SGRL_DEV void fill_rows(const Rows& R, int wanted_total, int cap) {
    const int nitem = 2 * o.nj + o.ncon;
    if (wanted_total <= cap) {
      w.lanes(nitem, [&](int t) {
        int pre = 0;
        for (int u = 0; u < t; u++) pre += I[o.ecnt + u];
        const int c = I[o.ecnt + t];
        if (t == 0) { I[o.icnt + IC_NROW_WANTED] = wanted_total; I[o.icnt + IC_NROW] = wanted_total; }
        for (int k = 0; k < c; k++) {
          if (t < 2 * o.nj) { R.kind[pre] = (t & 1) ? ROW_LIMIT_HI : ROW_LIMIT_LO; R.src[pre] = t >> 1; R.sub[pre] = 0; }
          else { R.kind[pre + k] = (c == 1) ? ROW_CON1 : ROW_PYR; R.src[pre + k] = t - 2 * o.nj; R.sub[pre + k] = k; }
        }
      });
      return;
    }
    // the cap bites (never observed at the shipped caps): serial walk that drops what does not fit
    w.lanes(1, [&](int) {
      int nrow = 0, wanted = 0;
      const int maxrows = cap;
      for (int j = 0; j < o.nj; j++) {
        if (!m.jnt_limited[j]) continue;
        const double q = S[o.qpos + m.jnt_qposadr[j]];
        for (int side = -1; side <= 1; side += 2) {
          const double dist = side * (m.jnt_range[2 * j + (side + 1) / 2] - q);
          if (dist >= m.jnt_margin[j]) continue;
          wanted++;
          if (nrow >= maxrows) continue;
          R.kind[nrow] = side < 0 ? ROW_LIMIT_LO : ROW_LIMIT_HI;
          R.src[nrow] = j; R.sub[nrow] = 0;
          nrow++;
        }
      }
      for (int s = 0; s < o.ncon; s++) {
        if (!I[o.con_valid + s]) continue;
        const int p = s >> 1, dim = m.pair_condim[p];
        const int nr = (dim == 1) ? 1 : 2 * (dim - 1);
        wanted += nr;
        if (nrow + nr > maxrows) continue;
        for (int k = 0; k < nr; k++) {
          R.kind[nrow] = dim == 1 ? ROW_CON1 : ROW_PYR;
          R.src[nrow] = s; R.sub[nrow] = k;
          nrow++;
        }
      }
      I[o.icnt + IC_NROW] = nrow; I[o.icnt + IC_NROW_WANTED] = wanted;
      if (wanted > nrow) I[o.icnt + IC_OVERFLOW] += 1;
    });
  }

  template <bool BIG>
  SGRL_DEV void build_rows_and_halfsolve(const Rows& R) {
    const int nv = o.nv, ldy = o.ldy;
    const int nrow = I[o.icnt + IC_NROW];
    w.lanes(nrow + 1, [&](int r) {
      double* Yr = R.Y + r * ldy;
      if (r == nrow) {  // extra right-hand side: the smooth force
        for (int d = 0; d < nv; d++) Yr[d] = S[o.qfs + d];
        return;
      }
      const int kind = R.kind[r], src = R.src[r], sub = R.sub[r];
      double Rreg, aref;
      if (kind == ROW_LIMIT_LO || kind == ROW_LIMIT_HI) {
        const int j = src, side = kind == ROW_LIMIT_LO ? -1 : 1, dof = m.jnt_dofadr[j];
        for (int d = 0; d < nv; d++) Yr[d] = 0;
        Yr[dof] = -side;
        const double q = S[o.qpos + m.jnt_qposadr[j]];
        const double margin = m.jnt_margin[j];
        const double dist = side * (m.jnt_range[2 * j + (side + 1) / 2] - q);
        double si[5], sr[2], K, B;
        for (int k = 0; k < 5; k++) si[k] = m.jnt_solimp[5 * j + k];
        sr[0] = m.jnt_solref[2 * j]; sr[1] = m.jnt_solref[2 * j + 1];
        const double imp = impedance(si, dist - margin);
        kb(sr, si, &K, &B);
        Rreg = (1 - imp) / imp * m.dof_invweight0[dof];
        if (Rreg < kMinVal) Rreg = kMinVal;
        aref = -B * (-side * S[o.qvel + dof]) - K * imp * (dist - margin);
      } else {
        const int s = src, p = s >> 1;
        const int b1 = m.geom_body[m.pair_g1[p]], b2 = m.geom_body[m.pair_g2[p]];
        const double margin = m.pair_margin[p], dist = S[o.con_dist + s], mu = m.pair_mu[p];
        double off[3], dir[3];
        const double* fr = S + o.con_frame + o.fstride * s;     // normal, tangent 1; tangent 2 is their cross product (make_frame)
        for (int k = 0; k < 3; k++) off[k] = S[o.con_pos + 3 * s + k] - S[o.misc + MS_COM + k];
        if (kind == ROW_CON1) {
          for (int k = 0; k < 3; k++) dir[k] = fr[k];
        } else {
          const double sg = (sub & 1) ? -mu : mu;
          double tg[3];
          if (sub >> 1) cross3(tg, fr, fr + 3);                    // tangent 2 (what make_frame left in fr[6..8])
          else ld3(tg, fr + 3);
          for (int k = 0; k < 3; k++) dir[k] = fr[k] + sg * tg[k];
        }
        sgrl_itab_t mk1 = m.body_dofmask + 2 * b1;
        sgrl_itab_t mk2 = m.body_dofmask + 2 * b2;
        double vel = 0;
        for (int d = 0; d < nv; d++) {
          double sgn = 0;
          if (b2 > 0 && dof_in_mask(mk2, d)) sgn += 1.0;
          if (b1 > 0 && dof_in_mask(mk1, d)) sgn -= 1.0;
          double val = 0;
          if (sgn != 0) {
            double cd[6], v[3];
            ld6(cd, S + o.cdof + 6 * d);
            cross3(v, cd, off);
            v[0] += cd[3]; v[1] += cd[4]; v[2] += cd[5];
            val = sgn * dot3(dir, v);
          }
          Yr[d] = val;
          vel += val * S[o.qvel + d];
        }
        double si[5], sr[2], K, B;
        for (int k = 0; k < 5; k++) si[k] = m.pair_solimp[5 * p + k];
        sr[0] = m.pair_solref[2 * p]; sr[1] = m.pair_solref[2 * p + 1];
        const double imp = impedance(si, dist - margin);
        kb(sr, si, &K, &B);
        const double tran = m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2];
        if (kind == ROW_CON1) {
          Rreg = (1 - imp) / imp * tran;
          if (Rreg < kMinVal) Rreg = kMinVal;
        } else {
          double R0 = (1 - imp) / imp * (tran + mu * mu * tran);
          if (R0 < kMinVal) R0 = kMinVal;
          Rreg = 2 * mu * mu * R0;
          if (Rreg < kMinVal) Rreg = kMinVal;
        }
        aref = -B * vel - K * imp * (dist - margin);
      }
      R.eR[r] = Rreg; R.earef[r] = aref;
      // warm start from the previous evaluation of this env-step: same constraint (kind, source, edge) -> same force.  The
      // memory of the first kPrevRows rows is in LDS, that of the rest (contact-rich states only) in the environment's HBM slab
      const int key = (kind << 16) | (src << 3) | sub;
      double f0 = 0;
      const int pn = I[o.icnt + IC_PREV_N];
      const int pcap = o.prevcap, pl = pn < pcap ? pn : pcap;
      for (int k = 0; k < pl; k++) if (I[o.prev_key + k] == key) f0 = S[o.prev_f + k];
      if (pn > pcap) {
        const double* hf = big_scratch + slab_prev_offset(o.maxrows, o.ldy);
        const int32_t* hk = reinterpret_cast<const int32_t*>(hf + o.maxrows);
        for (int k = pcap; k < pn; k++) if (hk[k] == key) f0 = hf[k];
      }
      R.ef[r] = f0;
    });
    SGRL_TICK(6);
    // half-solve Y <- L^-1 Y for all right-hand sides (rows + the smooth force), one lane per right-hand side
    if (linv) {
      // Y_r = L^-1 J_r as a triangular matrix product: one lane per right-hand side, entries from the last to the first
      // so that the row is overwritten in place (Y_r[d] needs J_r[0..d] only)
      if (BIG || !w.trmm_rows(nrow + 1, nv, S + o.L, R.Y, ldy))      // register version where the policy has one (LDS rows)
      w.lanes(nrow + 1, [&](int r) {
        double* Yr = R.Y + r * ldy;
        for (int d = nv - 1; d >= 0; d--) {
          const double* Ld = S + o.L + tri(d);
          double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
          int c = 0;
          for (; c + 4 <= d + 1; c += 4) {
            const double a0 = Ld[c], a1 = Ld[c + 1], a2 = Ld[c + 2], a3 = Ld[c + 3];
            const double b0 = Yr[c], b1 = Yr[c + 1], b2 = Yr[c + 2], b3 = Yr[c + 3];
            s0 += a0 * b0; s1 += a1 * b1; s2 += a2 * b2; s3 += a3 * b3;
          }
          for (; c <= d; c++) s0 += Ld[c] * Yr[c];
          Yr[d] = (s0 + s1) + (s2 + s3);
        }
      });
    } else {
      w.trsm_lower_rows(nrow + 1, nv, S + o.L, S + o.dinv, R.Y, ldy);
    }
    SGRL_TICK(9);
    w.lanes(nrow > nv ? nrow : nv, [&](int r) {
      if (r < nrow) {
        const double* Yr = R.Y + r * ldy;
        const double* ys = R.Y + nrow * ldy;      // the half-solved smooth force: the extra right-hand side
        double q0 = 0, q1 = 0, t0 = 0, t1 = 0;    // two chains per dot product
        int d = 0;
        for (; d + 2 <= nv; d += 2) {
          const double y0 = Yr[d], y1 = Yr[d + 1];
          q0 += y0 * y0; q1 += y1 * y1; t0 += y0 * ys[d]; t1 += y1 * ys[d + 1];
        }
        for (; d < nv; d++) { q0 += Yr[d] * Yr[d]; t0 += Yr[d] * ys[d]; }
        const double Rr = R.eR[r];
        R.eidg[r] = 1.0 / ((q0 + q1) + Rr);
        R.eb[r] = (t0 + t1) - R.earef[r];
      }
      if (r < nv) S[o.vpgs + r] = 0;
    });
  }

  // Exact solve of the dual LCP  0 <= f  _|_  A f + b >= 0  (A = Y Y' + R, SPD) by block principal pivoting (Judice &
  // Pires): guess the free set F, solve A_FF x = -b_F by a lane-parallel root-free Cholesky, exchange every index that
  // violates x_F >= 0 or (A x + b)_G >= 0 (a single index once the number of violations has stopped shrinking), repeat.
  // A is never stored: the principal block A_FF is formed straight from the half-solved rows Y into the factor's
  // scratch C (packed lower triangle), and the complement test uses  (A x)_G = Y_G (Y_F' x).  The free set is
  // warm-started from the previous evaluation.  On success S[vpgs] = Y' f.  The result is the same unique optimum
  // projected Gauss-Seidel converges to; PGS remains the fallback.
  template <bool SMALL, bool CLDS>
  SGRL_DEV bool lcp_block_pivot(const Rows& R, int n, double thresh, int* iters_out) {
    double* const wvp = R.earef;          // scratch: reciprocal pivots / intermediate vector
    // scratch: right-hand side -> solution (compact, free-set order; up to 64 entries).  LDS rows: the warm-start array (n <=
    // lrows <= its size); slab rows: the LDS Y block, idle while the rows live in the slab
    double* const xwp = SMALL ? S + o.prev_f : S + o.Y;
    const int nv = o.nv, ldy = o.ldy;
    uint64_t F = w.ballot(n, [&](int i) { return R.ef[i] > 0.0; });
    int patience = 3, best = n + 1;
    for (int iter = 0; iter < kBppMaxIter; iter++) {
      w.fence_lane();            // nothing lane-dependent is carried across pivoting rounds (registers)
      const int nf = popcount64(F);
      // factor scratch: the LDS dead zone (LDS rows; slab-path evaluations whose WHOLE row set is small enough for it: the
      // elimination then runs out of LDS, not global memory) or the slab's own.  The choice is a template parameter, never a
      // run-time select between an LDS and a global pointer: that select makes the accesses FLAT, and in the fixed-dimension
      // kernels (LDS address a compile-time constant) such a flat access faulted (memory aperture violation on a humanoid_7
      // lying on the floor, round 3: tools/diag/contact_stress.py); with a static address space there are no flat accesses.
      double* const C = CLDS ? S + o.dead : R.C;
      w.lanes(n, [&](int i) {
        if ((F >> i) & 1ull) {
          const int pos = popcount64(F & ((1ull << i) - 1ull));
          R.flist[pos] = i;
          xwp[pos] = -R.eb[i];
        }
      });
      // A_FF = Y_F Y_F' + diag(R_F), compact packed lower triangle: on the FP64 matrix cores where the wave policy has them
      // and the free set is large enough to pay, else one lane per entry
      if (!w.aff_rows(nf, nv, R.Y, ldy, R.flist, R.eR, C))
      w.lanes(nf * (nf + 1) / 2, [&](int p) {
        int i, j;
        tri_decode(p, &i, &j);
        const int fi = R.flist[i], fj = R.flist[j];
        const double* yi = R.Y + fi * ldy;
        const double* yj = R.Y + fj * ldy;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;      // four independent chains: a dot product over nv is latency, not work
        int d = 0;
        for (; d + 4 <= nv; d += 4) {
          const double a0 = yi[d], a1 = yi[d + 1], a2 = yi[d + 2], a3 = yi[d + 3];
          const double b0 = yj[d], b1 = yj[d + 1], b2 = yj[d + 2], b3 = yj[d + 3];
          s0 += a0 * b0; s1 += a1 * b1; s2 += a2 * b2; s3 += a3 * b3;
        }
        for (; d < nv; d++) s0 += yi[d] * yj[d];
        double a = (s0 + s1) + (s2 + s3);
        if (i == j) a += R.eR[fi];
        C[p] = a;
      });
      SGRL_TICK(12);
      // factor A_FF and solve.  Small free sets in LDS: Cholesky + explicit inverse of the factor on registers (wave
      // policy), then x = T'(T rhs) as two recurrence-free products; otherwise the lane-parallel root-free elimination
      if (SMALL && w.chol_inv_packed(nf, C, kMinVal)) {
        w.lanes(nf, [&](int i) {
          double z = 0;
          for (int c = 0; c <= i; c++) z += C[i * (i + 1) / 2 + c] * xwp[c];
          wvp[i] = z;
        });
        w.lanes(nf, [&](int k) {
          double x = 0;
          for (int i = k; i < nf; i++) x += C[i * (i + 1) / 2 + k] * wvp[i];
          xwp[k] = x;
        });
      } else {
        // root-free right-looking Cholesky of A_FF: one fused update per trailing entry per pivot;  w[k] = 1 / C[k][k]
        for (int j = 0; j < nf - 1; j++) {
          double pj = C[j * (j + 1) / 2 + j];
          if (pj < kMinVal) pj = kMinVal;
          const double wj = 1.0 / pj;
          const int sdim = nf - j - 1;
          w.lanes(sdim * (sdim + 1) / 2, [&](int p) {
            int a, b;
            tri_decode(p, &a, &b);
            const int i = j + 1 + a, k = j + 1 + b;
            C[i * (i + 1) / 2 + k] -= C[i * (i + 1) / 2 + j] * C[k * (k + 1) / 2 + j] * wj;
          });
        }
        w.lanes(nf, [&](int j) {
          double pj = C[j * (j + 1) / 2 + j];
          if (pj < kMinVal) pj = kMinVal;
          wvp[j] = 1.0 / pj;
        });
        // forward substitution (column sweeps), diagonal scaling, backward substitution
        for (int j = 0; j < nf; j++) {
          const double zj = xwp[j] * wvp[j];
          w.lanes_from(j + 1, nf, [&](int i) { xwp[i] -= C[i * (i + 1) / 2 + j] * zj; });
        }
        w.lanes(nf, [&](int i) { xwp[i] *= wvp[i]; });
        for (int j = nf - 1; j > 0; j--) {
          const double xj = xwp[j];
          w.lanes(j, [&](int k) { xwp[k] -= C[j * (j + 1) / 2 + k] * wvp[k] * xj; });
        }
      }
      SGRL_TICK(13);
      // u = Y_F' x  (= Y' f for the candidate f);  violations: x_i < 0 on F, (Y_i u + b_i) < -thresh on the complement
      w.lanes(nv, [&](int d) {
        double u0 = 0, u1 = 0;
        int k = 0;
        for (; k + 2 <= nf; k += 2) { u0 += R.Y[R.flist[k] * ldy + d] * xwp[k]; u1 += R.Y[R.flist[k + 1] * ldy + d] * xwp[k + 1]; }
        if (k < nf) u0 += R.Y[R.flist[k] * ldy + d] * xwp[k];
        S[o.vpgs + d] = u0 + u1;
      });
      const uint64_t V = w.ballot(n, [&](int i) {
        if ((F >> i) & 1ull) return xwp[popcount64(F & ((1ull << i) - 1ull))] < -thresh * R.eidg[i];
        const double* yi = R.Y + i * ldy;
        double y0 = R.eb[i], y1 = 0, y2 = 0, y3 = 0;
        int d = 0;
        for (; d + 4 <= nv; d += 4) {
          y0 += yi[d] * S[o.vpgs + d]; y1 += yi[d + 1] * S[o.vpgs + d + 1];
          y2 += yi[d + 2] * S[o.vpgs + d + 2]; y3 += yi[d + 3] * S[o.vpgs + d + 3];
        }
        for (; d < nv; d++) y0 += yi[d] * S[o.vpgs + d];
        return (y0 + y1) + (y2 + y3) < -thresh;
      });
      SGRL_TICK(14);
      if (V == 0) {
        w.lanes(n, [&](int i) {
          R.ef[i] = ((F >> i) & 1ull) ? xwp[popcount64(F & ((1ull << i) - 1ull))] : 0.0;
        });
        *iters_out = iter + 1;
        return true;
      }
      const int nviol = popcount64(V);
      if (nviol < best) { best = nviol; patience = 3; F ^= V; }
      else if (patience > 0) { patience--; F ^= V; }
      else { F ^= (1ull << (63 - clz64(V))); }   // backup rule: only the highest violating index
    }
    // not converged (never observed): leave a feasible point for the Gauss-Seidel fallback
    w.lanes(n, [&](int i) { if (R.ef[i] < 0.0) R.ef[i] = 0.0; });
    *iters_out = kBppMaxIter;
    return false;
  }

  template <bool BIG>
  SGRL_DEV void pgs_and_finish(const Rows& R) {
    const int nv = o.nv, ldy = o.ldy;
    const int nrow = I[o.icnt + IC_NROW];
    int sweeps = 0, diag_code = 0;
    if (nrow > 0) {
      const double bmax = w.maxabs(nrow, [&](int r) { return R.eb[r]; });
      const double thresh = m.fhdr[SGRL_F_PGS_TOL] * (1.0 + bmax);
      bool solved = false;
      if (W::hdr_const(m, SGRL_H_SOLVER) == 1) {
        if (!BIG) {
          solved = lcp_block_pivot<true, true>(R, nrow, thresh, &sweeps);       // rows and factor scratch in LDS
          if (!solved) diag_code |= 1 << 8;                                     // diagnostics: block pivoting gave up
        } else if (nrow <= kPivotRows) {
          // rare (a few envs per 8192-env launch): more rows than the LDS arrays hold -> the same exact solve with
          // rows and factor in this environment's HBM scratch slab
          solved = nrow <= o.na_max ? lcp_block_pivot<false, true>(R, nrow, thresh, &sweeps)
                                    : lcp_block_pivot<false, false>(R, nrow, thresh, &sweeps);
          diag_code |= (1 << 16) | (solved ? 0 : 1 << 8);
        }
      }
      if (!solved) {
        // matrix-free projected Gauss-Seidel, the policy keeps v = Y'f one entry per lane (also the SOLVER = 0 path)
        diag_code |= 1;
        if (BIG && nrow > kPivotRows)     // more rows than the register version keeps: rows streamed from the slab
          sweeps = w.pgs_big(nrow, nv, R.Y, ldy, R.eb, R.eR, R.eidg, R.ef, S + o.vpgs, m.hdr[SGRL_H_PGS_ITERS], thresh);
        else
          sweeps = w.pgs(nrow, nv, R.Y, ldy, R.eb, R.eR, R.eidg, R.ef, S + o.vpgs, m.hdr[SGRL_H_PGS_ITERS], thresh);
      }
    }
    SGRL_TICK(7);
    // remember the solution for the next evaluation's warm start
    const int prevcap = o.prevcap;       // LDS keeps the first rows, the HBM slab the others
    const int nkeep = (nrow <= prevcap || big_scratch != nullptr) ? nrow : prevcap;
    w.lanes(nkeep > 0 ? nkeep : 1, [&](int r) {
      if (r < nkeep) {
        const int key = (R.kind[r] << 16) | (R.src[r] << 3) | R.sub[r];
        if (r < prevcap) {
          I[o.prev_key + r] = key;
          S[o.prev_f + r] = R.ef[r];
        } else {
          double* hf = big_scratch + slab_prev_offset(o.maxrows, o.ldy);
          hf[r] = R.ef[r];
          reinterpret_cast<int32_t*>(hf + o.maxrows)[r] = key;
        }
      }
      if (r == 0) { I[o.icnt + IC_PREV_N] = nkeep; I[o.icnt + IC_SWEEPS] += sweeps; I[o.icnt + IC_ROWSUM] += diag_code; }
    });
    // qacc = L^-T (ys + Y' f)
    if (linv) {
      // qacc = L^-T w as a matrix-vector product with the explicit inverse (w parked in vpgs: no in-place hazard)
      w.lanes(nv, [&](int d) { S[o.vpgs + d] = R.Y[nrow * ldy + d] + (nrow > 0 ? S[o.vpgs + d] : 0.0); });
      w.lanes(nv, [&](int k) {
        double s0 = 0, s1 = 0;
        int i = k;
        for (; i + 1 < nv; i += 2) { s0 += S[o.L + tri(i) + k] * S[o.vpgs + i]; s1 += S[o.L + tri(i + 1) + k] * S[o.vpgs + i + 1]; }
        if (i < nv) s0 += S[o.L + tri(i) + k] * S[o.vpgs + i];
        S[o.qacc + k] = s0 + s1;
      });
    } else {
      w.lanes(nv, [&](int d) { S[o.qacc + d] = R.Y[nrow * ldy + d] + (nrow > 0 ? S[o.vpgs + d] : 0.0); });
      solve_upper_inplace(o.L, o.qacc);
    }
  }

  SGRL_DEV void forward() {
    w.fence_lane();
    fence_view();
    SGRL_TICK(-1);
    kinematics();               SGRL_TICK(0);
    com_pos();                  SGRL_TICK(1);
    crba_and_factor();          SGRL_TICK(2);
    collide();                  SGRL_TICK(3);
    bias_and_smooth_force();    SGRL_TICK(4);
    const int wanted = count_rows();
#ifdef SGRL_FORCE_SLAB_ROWS      // diagnostic build only: every evaluation keeps its constraint rows in the HBM slab (prices Y out of LDS)
    if (big_scratch != nullptr) {
#else
    if (wanted > o.lrows && big_scratch != nullptr) {
#endif
      const Rows R = rows_hbm();
      fill_rows(R, wanted, o.maxrows);           SGRL_TICK(5);
      build_rows_and_halfsolve<true>(R);         SGRL_TICK(10);
      pgs_and_finish<true>(R);                   SGRL_TICK(11);
    } else {
      const Rows R = rows_lds();
      fill_rows(R, wanted, o.lrows);             SGRL_TICK(5);
      build_rows_and_halfsolve<false>(R);        SGRL_TICK(10);
      pgs_and_finish<false>(R);                  SGRL_TICK(11);
    }
  }

  // ---- integration ----------------------------------------------------------------------------
  // S[dst..] <- S[src..] (+) h * S[vel..]
  SGRL_DEV void integrate_pos(int dst, int src, int vel, double h) {
    w.lanes(o.nj, [&](int j) {
      const int qa = m.jnt_qposadr[j], d = m.jnt_dofadr[j];
      if (m.jnt_type[j] == SGRL_JNT_FREE) {
        for (int k = 0; k < 3; k++) S[dst + qa + k] = S[src + qa + k] + h * S[vel + d + k];
        double ax[3] = {S[vel + d + 3], S[vel + d + 4], S[vel + d + 5]};
        const double n = sqrt(dot3(ax, ax));
        double ang;
        if (n < kMinVal) { ax[0] = 1; ax[1] = 0; ax[2] = 0; ang = 0; }
        else { ax[0] /= n; ax[1] /= n; ax[2] /= n; ang = h * n; }
        double qr[4], qn[4], q[4] = {S[src + qa + 3], S[src + qa + 4], S[src + qa + 5], S[src + qa + 6]};
        axisangle2quat(qr, ax, ang);
        quat_normalize(q);
        quat_mul(qn, q, qr);
        for (int k = 0; k < 4; k++) S[dst + qa + 3 + k] = qn[k];
      } else {
        S[dst + qa] = S[src + qa] + h * S[vel + d];
      }
    });
  }

  SGRL_DEV void mj_step() {
    const int nv = o.nv, nq = o.nq;
    const double h = m.fhdr[SGRL_F_TIMESTEP];
    const bool rk4 = W::hdr_const(m, SGRL_H_INTEGRATOR) == 1;
    {
      // RK4 (four stages) or Euler (one); the kinematics left in LDS afterwards are those of the last stage (what
      // _get_obs reads).  One loop with a single inlined copy of forward() keeps the kernel's code near the I-cache.
#pragma unroll 1
      for (int st = 0; st < (rk4 ? 4 : 1); st++) {
        const double a = (st == 3) ? 1.0 : 0.5;
        const double bw = (st == 3) ? (1.0 / 6) : (1.0 / 3);
        if (st > 0) {
          integrate_pos(o.qpos, o.q0, o.xv, a * h);
          w.lanes(nv, [&](int d) {
            const double v = S[o.v0 + d] + a * h * S[o.fq + d];
            S[o.qvel + d] = v; S[o.xv + d] = v;
            S[o.dvacc + d] += bw * v;
          });
        }
        forward();
        if (!rk4) break;
        if (st == 0) {
          w.lanes(nq > nv ? nq : nv, [&](int i) {
            if (i < nq) S[o.q0 + i] = S[o.qpos + i];
            if (i < nv) {
              S[o.v0 + i] = S[o.qvel + i]; S[o.xv + i] = S[o.qvel + i]; S[o.fq + i] = S[o.qacc + i];
              S[o.dvacc + i] = (1.0 / 6) * S[o.qvel + i]; S[o.daacc + i] = (1.0 / 6) * S[o.qacc + i];
            }
          });
        } else {
          w.lanes(nv, [&](int d) { S[o.fq + d] = S[o.qacc + d]; S[o.daacc + d] += bw * S[o.qacc + d]; });
        }
      }
    }
    if (rk4) {
      integrate_pos(o.qpos, o.q0, o.dvacc, h);
      w.lanes(nv, [&](int d) { S[o.qvel + d] = S[o.v0 + d] + h * S[o.daacc + d]; });
    } else {
      // semi-implicit Euler with implicit joint damping: (M + h D) a = M qacc
      if (o.mfull_hbm) {       // the copy of M comes back from HBM into the constraint-row block, idle once qacc is known
        const double* const mh = big_scratch + slab_rows_doubles(o.maxrows, o.ldy);
        const int nt = nv * (nv + 1) / 2;
        w.lanes(64, [&](int l) { for (int k = l; k < nt; k += 64) S[o.Mfull + k] = mh[k]; });
      }
      w.lanes(nv, [&](int i) {
        double s = 0;
        for (int j = 0; j < nv; j++) s += (j <= i ? S[o.Mfull + tri(i) + j] : S[o.Mfull + tri(j) + i]) * S[o.qacc + j];
        S[o.fq + i] = s;
      });
      w.lanes(nv, [&](int i) { S[o.Mfull + tri(i) + i] += h * m.dof_damping[i]; });
      cholesky(o.Mfull);
      solve_lower_inplace(o.Mfull, o.fq);
      solve_upper_inplace(o.Mfull, o.fq);
      w.lanes(nv, [&](int d) { S[o.qvel + d] += h * S[o.fq + d]; });
      integrate_pos(o.qpos, o.qpos, o.qvel, h);
    }
  }

  // ---- observation / reward / done --------------------------------------------------------------
  // body-origin velocities from the kinematics in LDS and S[qvel]: xvelr -> cfrc[6b..], xvelp -> cfrc[6b+3..]
  SGRL_DEV void body_velocities() {
    w.lanes(o.nb, [&](int b) {
      double cv[6] = {0, 0, 0, 0, 0, 0};
      if (b > 0) {
        for (int d = m.body_dofadr[b] + m.body_dofnum[b] - 1; d >= 0; d = m.dof_parent[d]) {
          const double qv = S[o.qvel + d];
          for (int k = 0; k < 6; k++) cv[k] += S[o.cdof + 6 * d + k] * qv;
        }
        double off[3], t[3];
        for (int k = 0; k < 3; k++) off[k] = S[o.xpos + 3 * b + k] - S[o.misc + MS_COM + k];
        cross3(t, cv, off);
        cv[3] += t[0]; cv[4] += t[1]; cv[5] += t[2];
      }
      for (int k = 0; k < 6; k++) S[o.cfrc + 6 * b + k] = cv[k];
    });
  }

  SGRL_DEV double obs_element(int b, int k) const {
    // b = body id (1..nb-1), k = 0..40; reference <env>.py:116-140
    const double R2D = 180.0 / kPi;
    if (k < 3) return S[o.xpos + 3 * b + k] - S[o.xpos + 3 + k];
    if (k < 5) return 0.0;
    if (k == 5) return -9.81;
    if (k < 8) {
      const double dx = S[o.misc + 12] - S[o.xpos + 3], dy = S[o.misc + 13] - S[o.xpos + 4];
      const double dn = sqrt(dx * dx + dy * dy);
      return (k == 6 ? dx : dy) / dn;
    }
    if (k == 8) return 0.0;
    if (k < 12) { const double v = S[o.cfrc + 6 * b + 3 + (k - 9)]; return v < -10 ? -10 : (v > 10 ? 10 : v); }
    if (k < 15) return S[o.cfrc + 6 * b + (k - 12)];
    if (b == 1) {
      if (k < 27) return 0.0;
      if (k < 36) return 0.5;
    } else {
      const int j0 = m.body_jntadr[b];
      if (k < 24) return S[o.xaxis + 3 * (j0 + (k - 15) / 3) + (k - 15) % 3];
      if (k < 27) return S[o.qpos + m.jnt_qposadr[j0 + (k - 24)]];
      if (k < 36) {
        const int jj = (k - 27) / 3, which = (k - 27) % 3, j = j0 + jj;
        const double lo = m.jnt_range[2 * j] * R2D, hi = m.jnt_range[2 * j + 1] * R2D;
        if (which == 0) return (S[o.qpos + m.jnt_qposadr[j]] * R2D - lo) / (hi - lo);
        if (which == 1) return (180.0 + lo) / 360.0;
        return (180.0 + hi) / 360.0;
      }
    }
    if (k < 40) return (m.body_limbtype[b] == (k - 36) + 1) ? 1.0 : 0.0;
    return S[o.xpos + 3 * b + 2];
  }

  // obs from the kinematics/velocities in LDS; target must be in misc[12..13]
  SGRL_DEV void write_obs(float* obs32, double* obs64, int obs_max_len) {
    const int n = 41 * (o.nb - 1);
    w.lanes(obs_max_len, [&](int i) {
      double v = 0.0;
      if (i < n) v = obs_element(1 + i / 41, i % 41);
      if (obs32) obs32[i] = (float)v;
      if (obs64) obs64[i] = v;
    });
  }

  // reward + done (serial scalar arithmetic on lane 0)
  SGRL_DEV void reward_done() {
    w.lanes(1, [&](int) {
      const double* q = S + o.misc + MS_PREQUAT;
      const double qw = q[0], x = q[1], y = q[2], z = q[3];
      const double r00 = 1 - 2 * y * y - 2 * z * z, r10 = 2 * x * y + 2 * z * qw;
      const double r20 = 2 * x * z - 2 * y * qw, r21 = 2 * y * z + 2 * x * qw, r22 = 1 - 2 * x * x - 2 * y * y;
      const double heading = atan2(r10, r00);
      const double pitch = atan2(-r20, sqrt(r21 * r21 + r22 * r22));
      const double roll = atan2(r21, r22);
      const double tx = S[o.misc + 12], ty = S[o.misc + 13];
      const double pbx = S[o.misc + MS_PREPOS], pby = S[o.misc + MS_PREPOS + 1];
      const double dist_before = sqrt((tx - pbx) * (tx - pbx) + (ty - pby) * (ty - pby));
      const double pax = S[o.xpos + 3], pay = S[o.xpos + 4];
      const double dist_after = sqrt((tx - pax) * (tx - pax) + (ty - pay) * (ty - pay));
      const double dt = m.fhdr[SGRL_F_TIMESTEP] * W::hdr_const(m, SGRL_H_FRAME_SKIP);
      double height = S[o.qpos + 2];
      double r = (dist_before - dist_after) / dt;
      if (m.fhdr[SGRL_F_HEADING_WEIGHT] != 0.0) r += ((pax - pbx) * cos(heading) + (pay - pby) * sin(heading)) / dt;
      if (m.fhdr[SGRL_F_ALIVE_BONUS] != 0.0) r += m.fhdr[SGRL_F_ALIVE_BONUS];
      double sq = 0;
      for (int u = 0; u < o.nu; u++) sq += S[o.ctrl + u] * S[o.ctrl + u];
      r -= m.fhdr[SGRL_F_CTRL_COST] * sq;
      S[o.misc + MS_REWARD] = r; S[o.misc + MS_DIST] = dist_after;
      const double lo = m.fhdr[SGRL_F_HEIGHT_LO], hi = m.fhdr[SGRL_F_HEIGHT_HI], al = m.fhdr[SGRL_F_ANG_LIMIT];
      const int rule = m.hdr[SGRL_H_DONE_RULE];
      bool ok;
      if (rule == 0) {
        ok = height > lo && height < hi && fabs(pitch) < al && fabs(roll) < al;
      } else if (rule == 1) {
        const double* qq = S + o.qpos + 3;
        const double ang = 2 * atan2(sqrt(qq[1] * qq[1] + qq[2] * qq[2]), sqrt(qq[0] * qq[0] + qq[3] * qq[3]));
        bool fin = true, small = true;
        for (int i = 0; i < o.nq; i++) if (!isfinite(S[o.qpos + i])) fin = false;
        for (int i = 0; i < o.nv; i++) if (!isfinite(S[o.qvel + i])) fin = false;
        for (int i = 3; i < o.nq; i++) if (!(fabs(S[o.qpos + i]) < 100)) small = false;
        for (int i = 0; i < o.nv; i++) if (!(fabs(S[o.qvel + i]) < 100)) small = false;
        ok = fin && small && height > lo && fabs(ang) < al;
      } else {
        for (int i = 0; i < m.hdr[SGRL_H_NHEIGHT_BODIES]; i++) {
          const double zb = S[o.xpos + 3 * m.hdr[SGRL_H_HEIGHT_BODY0 + i] + 2];
          if (zb < height) height = zb;
        }
        double s2 = 0;
        for (int i = 0; i < o.nv; i++) s2 += S[o.qvel + i] * S[o.qvel + i];
        ok = height > lo && fabs(pitch) < al && fabs(roll) < al && s2 > 1;
      }
      I[o.icnt + IC_DONE] = ok ? 0 : 1;
    });
  }

  // reset_model with the counter RNG (draw order of reference <env>.py:150-164); leaves fresh kinematics in LDS
  SGRL_DEV void reset_state(uint64_t seed, uint32_t env_id, uint32_t episode) {
    const int nq = o.nq, nv = o.nv;
    const double pn = m.fhdr[SGRL_F_RESET_POS_NOISE], vn = m.fhdr[SGRL_F_RESET_VEL_NOISE];
    const bool normal = m.hdr[SGRL_H_RESET_VEL_NORMAL] != 0;
    w.lanes(nq > nv ? nq : nv, [&](int i) {
      if (i < nq) {
        double q = m.qpos0[i];
        if (i == 3 || i == 6) {
          const double rad = (-kPi + 2 * kPi * rng_uniform01(seed, env_id, episode, 0, 0)) / 2;
          q = (i == 3) ? cos(rad) : sin(rad);
        }
        S[o.qpos + i] = q + (-pn + 2 * pn * rng_uniform01(seed, env_id, episode, 0, 1 + i));
      }
      if (i < nv) {
        if (normal) {
          const double u1 = rng_uniform01(seed, env_id, episode, 0, 1 + nq + 2 * i);
          const double u2 = rng_uniform01(seed, env_id, episode, 0, 2 + nq + 2 * i);
          S[o.qvel + i] = vn * sqrt(-2.0 * log(u1)) * cos(2 * kPi * u2);
        } else {
          S[o.qvel + i] = -vn + 2 * vn * rng_uniform01(seed, env_id, episode, 0, 1 + nq + i);
        }
      }
      if (i == 0) {
        const uint32_t base = 1 + nq + (normal ? 2 * nv : nv);
        const double r = -kPi + 2 * kPi * rng_uniform01(seed, env_id, episode, 0, base);
        double len = 10000.0;
        if (m.hdr[SGRL_H_TARGET_V2]) len = 10.0 + 10.0 * rng_uniform01(seed, env_id, episode, 0, base + 1);
        S[o.misc + 12] = cos(r) * len; S[o.misc + 13] = sin(r) * len;
      }
    });
  }

  // kinematics + velocities at the current S[qpos], S[qvel]  (gym set_state -> sim.forward)
  SGRL_DEV void refresh_kinematics() {
    kinematics();
    com_pos();
    body_velocities();
  }
};

// ------------------------------------------------------------------------------------------------
// Per-environment persistent record in HBM (array of records; one wave reads/writes one record contiguously):
//   double rec[stride]: qpos[nq] | qvel[nv] | torso_xy_stale[2] | target[2]
//   int32  cnt[4]:      step_count, episode, overflow_total, reserved
struct StepIO {
  double* rec;           // this env's record
  int32_t* cnt;          // this env's counters
  const float* action;   // [action_max_len] policy-ordered, first 3 = torso dummies
  float* obs32;          // [obs_max_len] or null
  double* obs64;         // [obs_max_len] or null
  float* reward; uint8_t* done; float* dist; uint8_t* truncated;  // scalars for this env (nullable)
  double* reward64;      // nullable
  int obs_max_len;
  double* scratch;       // per-env HBM slab (slab_doubles(max_rows, ldy) doubles).  Never null in the engine and the emulator: Euler
                         // layouts keep their mass-matrix copy there (Layout::mfull_hbm) and the warm start its rows beyond kPrevRows
  uint64_t seed; uint32_t env_id; int max_episode_steps; int auto_reset;
};

template <class W>
SGRL_DEV void load_state(Engine<W>& e, const StepIO& io) {
  const Layout& o = e.o;
  e.w.lanes(o.nq + o.nv + 4, [&](int i) {
    const double v = io.rec[i];
    if (i < o.nq) e.S[o.qpos + i] = v;
    else if (i < o.nq + o.nv) e.S[o.qvel + (i - o.nq)] = v;
    else if (i < o.nq + o.nv + 2) e.S[o.misc + MS_PREPOS + (i - o.nq - o.nv)] = v;
    else e.S[o.misc + 12 + (i - o.nq - o.nv - 2)] = v;
  });
}
template <class W>
SGRL_DEV void store_state(Engine<W>& e, const StepIO& io, bool fresh_xy) {
  const Layout& o = e.o;
  e.w.lanes(o.nq + o.nv + 4, [&](int i) {
    double v;
    if (i < o.nq) v = e.S[o.qpos + i];
    else if (i < o.nq + o.nv) v = e.S[o.qvel + (i - o.nq)];
    else if (i < o.nq + o.nv + 2) v = e.S[o.xpos + 3 + (i - o.nq - o.nv)];
    else v = e.S[o.misc + 12 + (i - o.nq - o.nv - 2)];
    io.rec[i] = v;
  });
  (void)fresh_xy;
}

// VecEnv.reset() for one env
template <class W>
SGRL_DEV void env_reset(W& w, const SgrlModelView& m, const Layout& o, double* S, int32_t* I, const StepIO& io, bool bump_episode) {
  Engine<W> e(w, m, o, S, I);
  int32_t episode = io.cnt[1];
  if (bump_episode) episode += 1;
  e.reset_state(io.seed, io.env_id, (uint32_t)episode);
  e.refresh_kinematics();
  e.write_obs(io.obs32, io.obs64, io.obs_max_len);
  store_state(e, io, true);
  w.lanes(1, [&](int) { io.cnt[0] = 0; io.cnt[1] = episode; });
}

// make the record consistent after an external set_state (qpos/qvel/target already in the record)
template <class W>
SGRL_DEV void env_refresh(W& w, const SgrlModelView& m, const Layout& o, double* S, int32_t* I, const StepIO& io) {
  Engine<W> e(w, m, o, S, I);
  load_state(e, io);
  e.refresh_kinematics();
  e.write_obs(io.obs32, io.obs64, io.obs_max_len);
  store_state(e, io, true);
}

// VecEnv.step() for one env (reference subproc_vec_env.py:12-15 + <env>.py:15-44)
template <class W>
SGRL_DEV void env_step(W& w, const SgrlModelView& m, const Layout& o, double* S, int32_t* I, const StepIO& io) {
  Engine<W> e(w, m, o, S, I);
  e.big_scratch = io.scratch;
  load_state(e, io);
  w.lanes(o.nu > 8 ? o.nu : 8, [&](int u) {
    if (u < o.nu) { const int s = m.act_slot[u]; S[o.ctrl + u] = s >= 0 ? (double)io.action[s] : 0.0; }
    if (u < 4) S[o.misc + MS_PREQUAT + u] = S[o.qpos + 3 + u];
    if (u == 4) { I[o.icnt + IC_OVERFLOW] = 0; I[o.icnt + IC_PREV_N] = 0; I[o.icnt + IC_SWEEPS] = 0; I[o.icnt + IC_ROWSUM] = 0; }
  });
  const int fs = W::hdr_const(m, SGRL_H_FRAME_SKIP);
#pragma unroll 1
  for (int s = 0; s < fs; s++) e.mj_step();
  e.body_velocities();
  e.reward_done();
  e.write_obs(io.obs32, io.obs64, io.obs_max_len);
  // bookkeeping: target resampling (<env>.py:41-43), time limit (gym TimeLimit), outputs
  w.lanes(1, [&](int) {
    const double dist = S[o.misc + MS_DIST];
    const double tx = S[o.misc + 12], ty = S[o.misc + 13];
    const uint32_t ep = (uint32_t)io.cnt[1], sc = (uint32_t)io.cnt[0];
    if (dist < 1.0 && sqrt(tx * tx + ty * ty) > 1.0) {
      const double r = -kPi + 2 * kPi * rng_uniform01(io.seed, io.env_id, ep, 1, 2 * sc);
      if (m.hdr[SGRL_H_TARGET_V2]) {
        const double len = 10.0 + 10.0 * rng_uniform01(io.seed, io.env_id, ep, 1, 2 * sc + 1);
        S[o.misc + 12] = S[o.xpos + 3] + cos(r) * len; S[o.misc + 13] = S[o.xpos + 4] + sin(r) * len;
      } else { S[o.misc + 12] = cos(r) * 10000.0; S[o.misc + 13] = sin(r) * 10000.0; }
    }
    int done = I[o.icnt + IC_DONE], trunc = 0;
    const int steps = io.cnt[0] + 1;
    if (io.max_episode_steps > 0 && steps >= io.max_episode_steps) { trunc = !done; done = 1; }
    I[o.icnt + IC_DONE] = done; I[o.icnt + IC_TRUNC] = trunc;
    io.cnt[0] = steps;
    io.cnt[2] += I[o.icnt + IC_OVERFLOW];
    io.cnt[3] = I[o.icnt + IC_ROWSUM];   // diagnostics: (#block-pivot failures << 8) | #evaluations on the matrix-free PGS path
    if (io.reward) *io.reward = (float)S[o.misc + MS_REWARD];
    if (io.reward64) *io.reward64 = S[o.misc + MS_REWARD];
    if (io.done) *io.done = (uint8_t)done;
    if (io.dist) *io.dist = (float)dist;
    if (io.truncated) *io.truncated = (uint8_t)trunc;
  });
  store_state(e, io, false);
  if (I[o.icnt + IC_DONE] && io.auto_reset) env_reset(w, m, o, S, I, io, true);
}

}  // namespace sgrl
