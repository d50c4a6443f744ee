/* oracle/physics.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C FP64 restatement of the CPU path the reference runs per environment step:
 *   ModularEnv.step          reference src/environments/3d_walker_7_full.py:15-44  (and family variants,
 *                            3d_hopper_3_shin.py:29-42, 3d_humanoid_9_full.py:35, 3d_cheetah_14_full.py:29-37)
 *   do_simulation(a, 4)      reference src/environments/3d_walker_7_full.py:24  -> gym MujocoEnv -> mj_step x4
 *   _get_obs                 reference src/environments/3d_walker_7_full.py:46-148
 *   reset_model              reference src/environments/3d_walker_7_full.py:150-164
 *
 * PARITY UNPINNED for the physics: `mj_step` lives in MuJoCo 2.1.0 (mujoco-py==2.1.2.14, gym==0.17.2;
 * reference requirements.txt:3-5), an un-vendored C library that is absent from /root/reference and from this
 * image.  The functions below restate its published computation pipeline (kinematics,
 * CRBA, RNE, plane/capsule collision, soft-constraint rows with solref/solimp, RK4 / semi-implicit Euler)
 * from documentation and recollection; they are pinned only by the physical known-answer tests in
 * tests/test_oracle_physics.py.  The env arithmetic around the simulator (reward / done / 41-float limb
 * observation / reset draws) IS pinned: tests/test_oracle_env_arith.py checks sgrl_oracle_env_epilogue()
 * against tests/golden/env_arith.npz, which was produced by executing the reference's own env files.
 *
 * The constraint solver is projected Gauss-Seidel on the dual (north_star asks for PGS; MuJoCo's default is
 * Newton on the primal -- both converge to the unique optimum of the same strictly convex problem).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sgrl_model.h"

#define NB 16
#define NJ 44
#define NVM 48
#define NQM 49
#define NG 24
#define MAXCON 96
#define MAXROW 320
#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define PI 3.14159265358979323846

typedef struct {
  double xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3];
  double xanchor[NJ][3], xaxis[NJ][3];
  double com[3];
  double cinert[NB][10], crb[NB][10];
  double cdof[NVM][6], cdof_dot[NVM][6];
  double cvel[NB][6], cacc[NB][6], cfrc[NB][6];
  double M[NVM][NVM], L[NVM][NVM];
  double qfrc_smooth[NVM], qacc_smooth[NVM], qacc[NVM];
  /* contacts */
  int ncon;
  double con_pos[MAXCON][3], con_frame[MAXCON][9], con_dist[MAXCON];
  int con_pair[MAXCON], con_slot[MAXCON];
  /* constraint rows */
  int nrow, nrow_wanted;
  double J[MAXROW][NVM], Y[MAXROW][NVM];
  double efc_R[MAXROW], efc_aref[MAXROW], efc_b[MAXROW], efc_f[MAXROW];
  int row_key[MAXROW];
  /* warm start across the evaluations of one call (same constraint -> previous force) */
  int prev_n, prev_key[MAXROW];
  double prev_f[MAXROW];
  double pgs_last_change;
  int pgs_iters_used;
} Work;

/* ------------------------------------------------------------------------------------------------ */
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
static void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = 1.0 / n;
  q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
static void quat2mat(double* m, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q[1] * q[2] - q[0] * q[3]); m[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
  m[3] = 2 * (q[1] * q[2] + q[0] * q[3]); m[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
  m[6] = 2 * (q[1] * q[3] - q[0] * q[2]); m[7] = 2 * (q[2] * q[3] + q[0] * q[1]);
}
static void mat_vec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  if (angle == 0.0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* spatial inertia (10: Ixx Iyy Izz Ixy Ixz Iyz, hx hy hz, m) times motion vector [w; v] -> force [tau; F] */
static void inert_mul(double* f, const double* I, const double* mv) {
  const double* w = mv; const double* v = mv + 3; const double* h = I + 6;
  double hv[3], hw[3];
  cross3(hv, h, v); cross3(hw, h, w);
  f[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2] + hv[0];
  f[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2] + hv[1];
  f[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2] + hv[2];
  f[3] = I[9] * v[0] - hw[0]; f[4] = I[9] * v[1] - hw[1]; f[5] = I[9] * v[2] - hw[2];
}
static void cross_motion(double* r, const double* vel, const double* m) {
  double a[3], b[3], c[3];
  cross3(a, vel, m); cross3(b, vel, m + 3); cross3(c, vel + 3, m);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double* r, const double* vel, const double* f) {
  double a[3], b[3], c[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

/* ------------------------------------------------------------------------------------------------ */
/* position stage: kinematics, COM frame quantities, CRBA, Cholesky.         */
static void kinematics(const SgrlModelView* m, double* qpos, Work* w) {
  w->xpos[0][0] = w->xpos[0][1] = w->xpos[0][2] = 0;
  w->xquat[0][0] = 1; w->xquat[0][1] = w->xquat[0][2] = w->xquat[0][3] = 0;
  quat2mat(w->xmat[0], w->xquat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b], j0 = m->body_jntadr[b], jn = m->body_jntnum[b];
    double pos[3], quat[4];
    if (jn == 1 && m->jnt_type[j0] == SGRL_JNT_FREE) {
      int qa = m->jnt_qposadr[j0];
      quat_normalize(qpos + qa + 3); /* in place, as mj_kinematics of 2.1.0 does [3P-knowledge] */
      for (int k = 0; k < 3; k++) pos[k] = qpos[qa + k];
      for (int k = 0; k < 4; k++) quat[k] = qpos[qa + 3 + k];
      for (int k = 0; k < 3; k++) { w->xanchor[j0][k] = pos[k]; w->xaxis[j0][k] = (k == 2); }
    } else {
      double t[3];
      mat_vec(t, w->xmat[p], m->body_pos + 3 * b);
      for (int k = 0; k < 3; k++) pos[k] = w->xpos[p][k] + t[k];
      quat_mul(quat, w->xquat[p], m->body_quat + 4 * b);
      for (int j = j0; j < j0 + jn; j++) {
        double r[9], ql[4], qn[4], v[3];
        quat2mat(r, quat);
        mat_vec(t, r, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) w->xanchor[j][k] = pos[k] + t[k];
        mat_vec(w->xaxis[j], r, m->jnt_axis + 3 * j);
        int qa = m->jnt_qposadr[j];
        axisangle2quat(ql, m->jnt_axis + 3 * j, qpos[qa] - m->qpos0[qa]);
        quat_mul(qn, quat, ql);
        for (int k = 0; k < 4; k++) quat[k] = qn[k];
        quat2mat(r, quat);
        mat_vec(v, r, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) pos[k] = w->xanchor[j][k] - v[k];
      }
    }
    quat_normalize(quat);
    for (int k = 0; k < 3; k++) w->xpos[b][k] = pos[k];
    for (int k = 0; k < 4; k++) w->xquat[b][k] = quat[k];
    quat2mat(w->xmat[b], quat);
    double t[3];
    mat_vec(t, w->xmat[b], m->body_ipos + 3 * b);
    for (int k = 0; k < 3; k++) w->xipos[b][k] = pos[k] + t[k];
  }
}

static void com_pos(const SgrlModelView* m, Work* w) {
  double mt = 0, c[3] = {0, 0, 0};
  for (int b = 1; b < m->nbody; b++) {
    mt += m->body_mass[b];
    for (int k = 0; k < 3; k++) c[k] += m->body_mass[b] * w->xipos[b][k];
  }
  for (int k = 0; k < 3; k++) w->com[k] = c[k] / mt;
  for (int b = 1; b < m->nbody; b++) {
    const double* R = w->xmat[b]; const double* ib = m->body_inertia + 6 * b;
    double I[9] = {ib[0], ib[3], ib[4], ib[3], ib[1], ib[5], ib[4], ib[5], ib[2]};
    double RI[9], W[9];
    for (int r = 0; r < 3; r++) for (int c2 = 0; c2 < 3; c2++) {
      double s = 0; for (int k = 0; k < 3; k++) s += R[3 * r + k] * I[3 * k + c2];
      RI[3 * r + c2] = s;
    }
    for (int r = 0; r < 3; r++) for (int c2 = 0; c2 < 3; c2++) {
      double s = 0; for (int k = 0; k < 3; k++) s += RI[3 * r + k] * R[3 * c2 + k];
      W[3 * r + c2] = s;
    }
    double d[3], mass = m->body_mass[b];
    for (int k = 0; k < 3; k++) d[k] = w->xipos[b][k] - w->com[k];
    double dd = dot3(d, d);
    double* ci = w->cinert[b];
    ci[0] = W[0] + mass * (dd - d[0] * d[0]); ci[1] = W[4] + mass * (dd - d[1] * d[1]);
    ci[2] = W[8] + mass * (dd - d[2] * d[2]);
    ci[3] = W[1] - mass * d[0] * d[1]; ci[4] = W[2] - mass * d[0] * d[2]; ci[5] = W[5] - mass * d[1] * d[2];
    ci[6] = mass * d[0]; ci[7] = mass * d[1]; ci[8] = mass * d[2]; ci[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    int d0 = m->jnt_dofadr[j], b = m->jnt_body[j];
    if (m->jnt_type[j] == SGRL_JNT_FREE) {
      double off[3];
      for (int k = 0; k < 3; k++) off[k] = w->com[k] - w->xpos[b][k];
      for (int k = 0; k < 3; k++) {
        double* c = w->cdof[d0 + k];
        c[0] = c[1] = c[2] = 0; c[3] = (k == 0); c[4] = (k == 1); c[5] = (k == 2);
        double ax[3] = {w->xmat[b][k], w->xmat[b][3 + k], w->xmat[b][6 + k]};
        double* cr = w->cdof[d0 + 3 + k];
        cr[0] = ax[0]; cr[1] = ax[1]; cr[2] = ax[2];
        cross3(cr + 3, ax, off);
      }
    } else {
      double off[3];
      for (int k = 0; k < 3; k++) off[k] = w->com[k] - w->xanchor[j][k];
      double* c = w->cdof[d0];
      for (int k = 0; k < 3; k++) c[k] = w->xaxis[j][k];
      cross3(c + 3, w->xaxis[j], off);
    }
  }
}

static void crba(const SgrlModelView* m, Work* w) {
  int nv = m->nv;
  for (int b = 1; b < m->nbody; b++) memcpy(w->crb[b], w->cinert[b], sizeof(double) * 10);
  for (int b = m->nbody - 1; b > 1; b--) {
    int p = m->body_parent[b];
    if (p > 0) for (int k = 0; k < 10; k++) w->crb[p][k] += w->crb[b][k];
  }
  for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) w->M[i][j] = 0;
  for (int i = 0; i < nv; i++) {
    double buf[6];
    inert_mul(buf, w->crb[m->dof_body[i]], w->cdof[i]);
    w->M[i][i] = dot6(w->cdof[i], buf) + m->dof_armature[i];
    for (int j = m->dof_parent[i]; j >= 0; j = m->dof_parent[j]) {
      double v = dot6(w->cdof[j], buf);
      w->M[i][j] = v; w->M[j][i] = v;
    }
  }
}

static int cholesky(int n, double A[NVM][NVM], double L[NVM][NVM]) {
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) L[i][j] = 0;
  for (int j = 0; j < n; j++) {
    double s = A[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (s < MINVAL) s = MINVAL;
    double d = sqrt(s);
    L[j][j] = d;
    double inv = 1.0 / d;
    for (int i = j + 1; i < n; i++) {
      double t = A[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t * inv;
    }
  }
  return 0;
}
static void solve_lower(int n, double L[NVM][NVM], double* x) { /* L y = x */
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * x[k];
    x[i] = s / L[i][i];
  }
}
static void solve_upper(int n, double L[NVM][NVM], double* x) { /* L^T y = x */
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
}

/* velocity stage: cvel / cdof_dot (mj_comVel) and the RNE bias force (mj_rne, flg_acc = 0)          */
static void com_vel(const SgrlModelView* m, const double* qvel, Work* w) {
  for (int k = 0; k < 6; k++) w->cvel[0][k] = 0;
  for (int b = 1; b < m->nbody; b++) {
    double cv[6];
    memcpy(cv, w->cvel[m->body_parent[b]], sizeof(cv));
    int j0 = m->body_jntadr[b], jn = m->body_jntnum[b];
    for (int j = j0; j < j0 + jn; j++) {
      int d0 = m->jnt_dofadr[j];
      if (m->jnt_type[j] == SGRL_JNT_FREE) {
        for (int d = 0; d < 3; d++) {
          for (int k = 0; k < 6; k++) { w->cdof_dot[d0 + d][k] = 0; cv[k] += w->cdof[d0 + d][k] * qvel[d0 + d]; }
        }
        for (int d = 3; d < 6; d++) cross_motion(w->cdof_dot[d0 + d], cv, w->cdof[d0 + d]);
        for (int d = 3; d < 6; d++) for (int k = 0; k < 6; k++) cv[k] += w->cdof[d0 + d][k] * qvel[d0 + d];
      } else {
        cross_motion(w->cdof_dot[d0], cv, w->cdof[d0]);
        for (int k = 0; k < 6; k++) cv[k] += w->cdof[d0][k] * qvel[d0];
      }
    }
    memcpy(w->cvel[b], cv, sizeof(cv));
  }
}

static void rne_bias(const SgrlModelView* m, const double* qvel, Work* w, double* bias) {
  const double* g = m->fhdr + SGRL_F_GRAV_X;
  w->cacc[0][0] = w->cacc[0][1] = w->cacc[0][2] = 0;
  w->cacc[0][3] = -g[0]; w->cacc[0][4] = -g[1]; w->cacc[0][5] = -g[2];
  for (int k = 0; k < 6; k++) w->cfrc[0][k] = 0;
  for (int b = 1; b < m->nbody; b++) {
    double ca[6];
    memcpy(ca, w->cacc[m->body_parent[b]], sizeof(ca));
    int d0 = m->body_dofadr[b], dn = m->body_dofnum[b];
    for (int d = d0; d < d0 + dn; d++) for (int k = 0; k < 6; k++) ca[k] += w->cdof_dot[d][k] * qvel[d];
    memcpy(w->cacc[b], ca, sizeof(ca));
    double f1[6], iv[6], f2[6];
    inert_mul(f1, w->cinert[b], ca);
    inert_mul(iv, w->cinert[b], w->cvel[b]);
    cross_force(f2, w->cvel[b], iv);
    for (int k = 0; k < 6; k++) w->cfrc[b][k] = f1[k] + f2[k];
  }
  for (int b = m->nbody - 1; b > 1; b--) {
    int p = m->body_parent[b];
    if (p > 0) for (int k = 0; k < 6; k++) w->cfrc[p][k] += w->cfrc[b][k];
  }
  for (int d = 0; d < m->nv; d++) bias[d] = dot6(w->cdof[d], w->cfrc[m->dof_body[d]]);
}

/* ------------------------------------------------------------------------------------------------ */
/* collision: plane-sphere, plane-capsule, capsule-capsule                             */
static void geom_pose(const SgrlModelView* m, const Work* w, int g, double* pos, double* mat) {
  int b = m->geom_body[g];
  double t[3], gm[9];
  mat_vec(t, w->xmat[b], m->geom_pos + 3 * g);
  for (int k = 0; k < 3; k++) pos[k] = w->xpos[b][k] + t[k];
  quat2mat(gm, m->geom_quat + 4 * g);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
    double s = 0; for (int k = 0; k < 3; k++) s += w->xmat[b][3 * r + k] * gm[3 * k + c];
    mat[3 * r + c] = s;
  }
}
static void make_frame(double* fr) {
  /* fr[0:3] unit normal; fr[3:6] preferred tangent or zero */
  double n2 = sqrt(dot3(fr + 3, fr + 3));
  if (n2 < 0.5) {
    fr[3] = fr[4] = fr[5] = 0;
    if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1; else fr[5] = 1;
  }
  double d = dot3(fr, fr + 3);
  for (int k = 0; k < 3; k++) fr[3 + k] -= d * fr[k];
  double n = sqrt(dot3(fr + 3, fr + 3));
  if (n < MINVAL) { fr[3] = 1; fr[4] = 0; fr[5] = 0; }
  else for (int k = 0; k < 3; k++) fr[3 + k] /= n;
  cross3(fr + 6, fr, fr + 3);
}
static void add_contact(Work* w, int pair, int slot, double dist, const double* pos, const double* normal, const double* tangent) {
  if (w->ncon >= MAXCON) return;
  int c = w->ncon++;
  w->con_pair[c] = pair; w->con_slot[c] = slot; w->con_dist[c] = dist;
  for (int k = 0; k < 3; k++) { w->con_pos[c][k] = pos[k]; w->con_frame[c][k] = normal[k]; w->con_frame[c][3 + k] = tangent ? tangent[k] : 0; }
  make_frame(w->con_frame[c]);
}
static void plane_sphere(Work* w, int pair, int slot, double margin, const double* ppos, const double* n, const double* c, double r, const double* tangent) {
  double d[3] = {c[0] - ppos[0], c[1] - ppos[1], c[2] - ppos[2]};
  double dist = dot3(d, n) - r;
  if (dist >= margin) return;
  double pos[3];
  for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * (r + 0.5 * dist);
  add_contact(w, pair, slot, dist, pos, n, tangent);
}
static void collide(const SgrlModelView* m, Work* w) {
  w->ncon = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_g1[p], g2 = m->pair_g2[p];
    double p1[3], m1[9], p2[3], m2[9];
    geom_pose(m, w, g1, p1, m1); geom_pose(m, w, g2, p2, m2);
    double margin = m->pair_margin[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    if (t1 == SGRL_GEOM_PLANE) {
      double n[3] = {m1[2], m1[5], m1[8]};
      if (t2 == SGRL_GEOM_SPHERE) {
        plane_sphere(w, p, 2 * p, margin, p1, n, p2, m->geom_size[3 * g2], 0);
      } else {
        double ax[3] = {m2[2], m2[5], m2[8]}, h = m->geom_size[3 * g2 + 1], r = m->geom_size[3 * g2];
        double ca[3], cb[3];
        for (int k = 0; k < 3; k++) { ca[k] = p2[k] + ax[k] * h; cb[k] = p2[k] - ax[k] * h; }
        plane_sphere(w, p, 2 * p, margin, p1, n, ca, r, ax); /* frames aligned with the capsule axis [3P-knowledge] */
        plane_sphere(w, p, 2 * p + 1, margin, p1, n, cb, r, ax);
      }
    } else { /* capsule - capsule: restates mjc_CapsuleCapsule of MuJoCo 2.1 [3P-knowledge] -- one contact between the
              * closest points of the two segments; for PARALLEL axes the closest "point" is a whole stretch, and up to
              * two contacts are made by testing the end spheres of each capsule against the other's axis */
      double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
      double h1 = m->geom_size[3 * g1 + 1], h2 = m->geom_size[3 * g2 + 1];
      double r1 = m->geom_size[3 * g1], r2 = m->geom_size[3 * g2];
      /* points p1 + s*a1 (|s|<=h1), p2 + t*a2 (|t|<=h2) */
      double d[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
      double b = dot3(a1, a2), c = dot3(a1, d), f = dot3(a2, d);
      double den = 1.0 - b * b;
      double ss[4], tt[4];
      int ncand = 0, made = 0;
      if (fabs(den) >= MINVAL) {
        double s = (b * f - c) / den, t;
        if (s > h1) s = h1; if (s < -h1) s = -h1;
        t = b * s + f;
        if (t > h2) { t = h2; s = b * t - c; if (s > h1) s = h1; if (s < -h1) s = -h1; }
        else if (t < -h2) { t = -h2; s = b * t - c; if (s > h1) s = h1; if (s < -h1) s = -h1; }
        ss[0] = s; tt[0] = t; ncand = 1;
      } else {
        for (int e = 0; e < 2; e++) {            /* ends of capsule 1 against axis 2 */
          double s = e ? -h1 : h1, t = b * s + f;
          if (t > h2) t = h2; if (t < -h2) t = -h2;
          ss[ncand] = s; tt[ncand] = t; ncand++;
        }
        for (int e = 0; e < 2; e++) {            /* ends of capsule 2 against axis 1 */
          double t = e ? -h2 : h2, s = b * t - c;
          if (s > h1) s = h1; if (s < -h1) s = -h1;
          ss[ncand] = s; tt[ncand] = t; ncand++;
        }
      }
      for (int q = 0; q < ncand && made < 2; q++) {
        double c1[3], c2[3], nn[3];
        for (int k = 0; k < 3; k++) { c1[k] = p1[k] + ss[q] * a1[k]; c2[k] = p2[k] + tt[q] * a2[k]; nn[k] = c2[k] - c1[k]; }
        double len = sqrt(dot3(nn, nn));
        double dist = len - r1 - r2;
        if (dist >= margin) continue;
        if (len < MINVAL) { nn[0] = 1; nn[1] = 0; nn[2] = 0; } else for (int k = 0; k < 3; k++) nn[k] /= len;
        double pos[3];
        for (int k = 0; k < 3; k++) pos[k] = c1[k] + nn[k] * (r1 + 0.5 * dist);
        add_contact(w, p, 2 * p + made, dist, pos, nn, 0);
        made++;
      }
    }
  }
}

/* point Jacobian difference (body b2 minus body b1) at world point p, projected on direction dir */
static void jac_dir(const SgrlModelView* m, const Work* w, int b1, int b2, const double* p, const double* dir, double sign_scale, double* row, int accumulate) {
  int nv = m->nv;
  if (!accumulate) for (int d = 0; d < nv; d++) row[d] = 0;
  double off[3] = {p[0] - w->com[0], p[1] - w->com[1], p[2] - w->com[2]};
  for (int pass = 0; pass < 2; pass++) {
    int b = pass ? b1 : b2;
    double sg = pass ? -sign_scale : sign_scale;
    if (b <= 0) continue;
    int d = m->body_dofadr[b] + m->body_dofnum[b] - 1;
    /* bodies always carry dofs in these models; walk the dof ancestor chain */
    for (; d >= 0; d = m->dof_parent[d]) {
      double v[3];
      cross3(v, w->cdof[d], off);
      v[0] += w->cdof[d][3]; v[1] += w->cdof[d][4]; v[2] += w->cdof[d][5];
      row[d] += sg * dot3(dir, v);
    }
  }
}

static double impedance(const double* solimp, double x_raw) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin < MINIMP) dmin = MINIMP; if (dmin > MAXIMP) dmin = MAXIMP;
  if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
  if (width < 0) width = 0;
  if (mid < MINIMP) mid = MINIMP; if (mid > MAXIMP) mid = MAXIMP;
  if (power < 1) power = 1;
  if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
  double x = fabs(x_raw) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}
static void kb(const SgrlModelView* m, const double* solref, const double* solimp, double* K, double* B) {
  double dmax = solimp[1];
  if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
  double tc = solref[0], dr = solref[1], h2 = 2 * m->fhdr[SGRL_F_TIMESTEP];
  if (tc < h2) tc = h2;
  *K = 1.0 / (dmax * dmax * tc * tc * dr * dr);
  *B = 2.0 / (dmax * tc);
}

static void make_constraints(const SgrlModelView* m, const double* qpos, const double* qvel, Work* w) {
  int nv = m->nv, maxrows = m->hdr[SGRL_H_MAX_ROWS];
  if (maxrows > MAXROW) maxrows = MAXROW;
  w->nrow = 0; w->nrow_wanted = 0;
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j]) continue;
    double q = qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - q);
      double margin = m->jnt_margin[j];
      if (dist >= margin) continue;
      w->nrow_wanted++;
      if (w->nrow >= maxrows) continue;
      int r = w->nrow++;
      for (int d = 0; d < nv; d++) w->J[r][d] = 0;
      int dof = m->jnt_dofadr[j];
      w->J[r][dof] = -side;
      double imp = impedance(m->jnt_solimp + 5 * j, dist - margin), K, B;
      kb(m, m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, &K, &B);
      double R = (1 - imp) / imp * m->dof_invweight0[dof];
      if (R < MINVAL) R = MINVAL;
      w->efc_R[r] = R;
      w->efc_aref[r] = -B * (-side * qvel[dof]) - K * imp * (dist - margin);
      w->row_key[r] = ((side < 0 ? 0 : 1) << 16) | (j << 3);
    }
  }
  /* contacts */
  for (int c = 0; c < w->ncon; c++) {
    int p = w->con_pair[c];
    int b1 = m->geom_body[m->pair_g1[p]], b2 = m->geom_body[m->pair_g2[p]];
    double margin = m->pair_margin[p], dist = w->con_dist[c], mu = m->pair_mu[p];
    int dim = m->pair_condim[p];
    int nr = (dim == 1) ? 1 : 2 * (dim - 1);
    w->nrow_wanted += nr;
    if (w->nrow + nr > maxrows) continue;
    double imp = impedance(m->pair_solimp + 5 * p, dist - margin), K, B;
    kb(m, m->pair_solref + 2 * p, m->pair_solimp + 5 * p, &K, &B);
    double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    const double* fr = w->con_frame[c];
    int r0 = w->nrow;
    if (dim == 1) {
      jac_dir(m, w, b1, b2, w->con_pos[c], fr, 1.0, w->J[r0], 0);
      double R = (1 - imp) / imp * tran;
      if (R < MINVAL) R = MINVAL;
      w->efc_R[r0] = R;
    } else {
      for (int k = 0; k < dim - 1; k++) {
        for (int s = 0; s < 2; s++) {
          int r = r0 + 2 * k + s;
          jac_dir(m, w, b1, b2, w->con_pos[c], fr, 1.0, w->J[r], 0);
          jac_dir(m, w, b1, b2, w->con_pos[c], fr + 3 * (k + 1), s ? -mu : mu, w->J[r], 1);
        }
      }
      double R0 = (1 - imp) / imp * (tran + mu * mu * tran);
      if (R0 < MINVAL) R0 = MINVAL;
      double Rpy = 2 * mu * mu * R0;
      if (Rpy < MINVAL) Rpy = MINVAL;
      for (int r = r0; r < r0 + nr; r++) w->efc_R[r] = Rpy;
    }
    for (int r = r0; r < r0 + nr; r++) {
      w->row_key[r] = ((dim == 1 ? 2 : 3) << 16) | (w->con_slot[c] << 3) | (r - r0);
      double vel = 0;
      for (int d = 0; d < nv; d++) vel += w->J[r][d] * qvel[d];
      w->efc_aref[r] = -B * vel - K * imp * (dist - margin);
    }
    w->nrow += nr;
  }
}

/* Exact dual LCP solve by block principal pivoting (Judice & Pires) on A = Y Y' + R: guess the free set F from the
 * warm start, solve A_FF x = -b_F, exchange all indices violating x_F >= 0 / (A x + b)_G >= -thresh (one index after
 * the violation count stopped shrinking three times), repeat.  Returns 1 when a complementary solution was found. */
static int lcp_block_pivot(int n, int nv, Work* w, const double* diag, double thresh) {
  double A[64][64];   /* automatic storage: the oracle is re-entrant (tests step several oracle envs from a thread pool) */
  double C[64][64];
  double x[64], rhs[64];
  int F[64], list[64];
  for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) {
    double s = 0;
    for (int d = 0; d < nv; d++) s += w->Y[i][d] * w->Y[j][d];
    if (i == j) s += w->efc_R[i];
    A[i][j] = s; A[j][i] = s;
  }
  for (int i = 0; i < n; i++) F[i] = w->efc_f[i] > 0.0;
  int patience = 3, best = n + 1;
  for (int iter = 0; iter < 40; iter++) {
    int nf = 0;
    for (int i = 0; i < n; i++) if (F[i]) list[nf++] = i;
    for (int j = 0; j < nf; j++) {            /* Cholesky A_FF = C C' */
      double s = A[list[j]][list[j]];
      for (int k = 0; k < j; k++) s -= C[j][k] * C[j][k];
      if (s < MINVAL) s = MINVAL;
      C[j][j] = sqrt(s);
      for (int i = j + 1; i < nf; i++) {
        double t = A[list[i]][list[j]];
        for (int k = 0; k < j; k++) t -= C[i][k] * C[j][k];
        C[i][j] = t / C[j][j];
      }
    }
    for (int i = 0; i < nf; i++) { double s = -w->efc_b[list[i]]; for (int k = 0; k < i; k++) s -= C[i][k] * rhs[k]; rhs[i] = s / C[i][i]; }
    for (int i = nf - 1; i >= 0; i--) { double s = rhs[i]; for (int k = i + 1; k < nf; k++) s -= C[k][i] * x[k]; x[i] = s / C[i][i]; }
    int viol[64], nviol = 0, top = -1, pos = 0;
    for (int i = 0; i < n; i++) {
      int bad;
      if (F[i]) { bad = x[pos] < -thresh / diag[i]; pos++; }
      else {
        double y = w->efc_b[i];
        for (int k = 0; k < nf; k++) y += A[i][list[k]] * x[k];
        bad = y < -thresh;
      }
      viol[i] = bad;
      if (bad) { nviol++; top = i; }
    }
    if (nviol == 0) {
      pos = 0;
      for (int i = 0; i < n; i++) w->efc_f[i] = F[i] ? x[pos++] : 0.0;
      w->pgs_iters_used = iter + 1;
      return 1;
    }
    if (nviol < best) { best = nviol; patience = 3; for (int i = 0; i < n; i++) F[i] ^= viol[i]; }
    else if (patience > 0) { patience--; for (int i = 0; i < n; i++) F[i] ^= viol[i]; }
    else F[top] ^= 1;
  }
  for (int i = 0; i < n; i++) if (w->efc_f[i] < 0) w->efc_f[i] = 0;
  return 0;
}

/* dual PGS: min 1/2 f'(A+R)f + f'b, f >= 0, A = J M^-1 J' = Y Y' with Y = (L^-1 J')'               */
static void solve_constraints(const SgrlModelView* m, Work* w) {
  int nv = m->nv, n = w->nrow, iters = m->hdr[SGRL_H_PGS_ITERS];
  for (int d = 0; d < nv; d++) w->qacc[d] = w->qacc_smooth[d];
  w->pgs_last_change = 0;
  w->pgs_iters_used = 0;
  if (n == 0) { w->prev_n = 0; return; }
  for (int r = 0; r < n; r++) {
    double s = 0;
    for (int d = 0; d < nv; d++) { s += w->J[r][d] * w->qacc_smooth[d]; w->Y[r][d] = w->J[r][d]; }
    w->efc_b[r] = s - w->efc_aref[r];
    solve_lower(nv, w->L, w->Y[r]);
    double f0 = 0;
    for (int k = 0; k < w->prev_n; k++) if (w->prev_key[k] == w->row_key[r]) f0 = w->prev_f[k];
    w->efc_f[r] = f0;
  }
  /* v = Y' f maintained incrementally */
  double v[NVM];
  for (int d = 0; d < nv; d++) { double s = 0; for (int r = 0; r < n; r++) s += w->Y[r][d] * w->efc_f[r]; v[d] = s; }
  double diag[MAXROW], idiag[MAXROW];
  for (int r = 0; r < n; r++) {
    double s = 0;
    for (int d = 0; d < nv; d++) s += w->Y[r][d] * w->Y[r][d];
    diag[r] = s + w->efc_R[r];
    idiag[r] = 1.0 / diag[r];
  }
  double bmax = 0;
  for (int r = 0; r < n; r++) if (fabs(w->efc_b[r]) > bmax) bmax = fabs(w->efc_b[r]);
  const double thresh = m->fhdr[SGRL_F_PGS_TOL] * (1.0 + bmax);
  int solved = 0;
  {
    /* same dispatch rule as the engine: exact block-pivot solve up to 64 rows, Gauss-Seidel beyond */
    if (m->hdr[SGRL_H_SOLVER] == 1 && n <= 64) solved = lcp_block_pivot(n, nv, w, diag, thresh);
  }
  if (solved) {
    for (int d = 0; d < nv; d++) { double s = 0; for (int r = 0; r < n; r++) s += w->Y[r][d] * w->efc_f[r]; v[d] = s; }
    iters = 0;
  }
  for (int it = 0; it < iters; it++) {
    double change = 0;
    for (int r = 0; r < n; r++) {
      double res = w->efc_b[r] + w->efc_R[r] * w->efc_f[r];
      for (int d = 0; d < nv; d++) res += w->Y[r][d] * v[d];
      double fn = w->efc_f[r] - res * idiag[r];
      if (fn < 0) fn = 0;
      double df = fn - w->efc_f[r];
      if (df != 0) {
        for (int d = 0; d < nv; d++) v[d] += w->Y[r][d] * df;
        w->efc_f[r] = fn;
        if (fabs(df) * diag[r] > change) change = fabs(df) * diag[r];
      }
    }
    w->pgs_last_change = change;
    w->pgs_iters_used = it + 1;
    if (change < thresh) break;
  }
  /* the warm-start memory keeps every row (the engine: the first kPrevRows in LDS, the rest in its HBM slab) */
  w->prev_n = n;
  for (int r = 0; r < w->prev_n; r++) { w->prev_key[r] = w->row_key[r]; w->prev_f[r] = w->efc_f[r]; }
  solve_upper(nv, w->L, v); /* M^-1 J' f = L^-T (Y' f) */
  for (int d = 0; d < nv; d++) w->qacc[d] += v[d];
}

/* full forward dynamics at (qpos, qvel, ctrl): fills w->qacc                                        */
static void forward(const SgrlModelView* m, double* qpos, const double* qvel, const double* ctrl, Work* w) {
  int nv = m->nv;
  kinematics(m, qpos, w);
  com_pos(m, w);
  crba(m, w);
  cholesky(nv, w->M, w->L);
  collide(m, w);
  com_vel(m, qvel, w);
  double bias[NVM];
  rne_bias(m, qvel, w, bias);
  for (int d = 0; d < nv; d++) {
    int j = m->dof_jnt[d];
    double passive = -m->dof_damping[d] * qvel[d];
    if (m->jnt_type[j] == SGRL_JNT_HINGE) {
      int qa = m->jnt_qposadr[j];
      passive -= m->jnt_stiffness[j] * (qpos[qa] - m->qpos0[qa]);
    }
    w->qfrc_smooth[d] = passive - bias[d];
  }
  for (int u = 0; u < m->nu; u++) {
    double c = ctrl[u], lo = m->act_ctrlrange[2 * u], hi = m->act_ctrlrange[2 * u + 1];
    if (c < lo) c = lo; if (c > hi) c = hi;
    w->qfrc_smooth[m->act_dof[u]] += m->act_gear[u] * c;
  }
  for (int d = 0; d < nv; d++) w->qacc_smooth[d] = w->qfrc_smooth[d];
  solve_lower(nv, w->L, w->qacc_smooth);
  solve_upper(nv, w->L, w->qacc_smooth);
  make_constraints(m, qpos, qvel, w);
  solve_constraints(m, w);
}

/* qpos <- qpos (+) h * vel  (mj_integratePos)                                                      */
static void integrate_pos(const SgrlModelView* m, double* qpos, const double* vel, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], d = m->jnt_dofadr[j];
    if (m->jnt_type[j] == SGRL_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * vel[d + k];
      double ax[3] = {vel[d + 3], vel[d + 4], vel[d + 5]};
      double n = sqrt(dot3(ax, ax)), ang;
      if (n < MINVAL) { ax[0] = 1; ax[1] = 0; ax[2] = 0; ang = 0; }
      else { ax[0] /= n; ax[1] /= n; ax[2] /= n; ang = h * n; }
      double qr[4], qn[4];
      axisangle2quat(qr, ax, ang);
      quat_normalize(qpos + qa + 3);
      quat_mul(qn, qpos + qa + 3, qr);
      for (int k = 0; k < 4; k++) qpos[qa + 3 + k] = qn[k];
    } else {
      qpos[qa] += h * vel[d];
    }
  }
}

static void mj_step_once(const SgrlModelView* m, double* qpos, double* qvel, const double* ctrl, Work* w) {
  int nq = m->nq, nv = m->nv;
  double h = m->fhdr[SGRL_F_TIMESTEP];
  forward(m, qpos, qvel, ctrl, w);
  if (m->hdr[SGRL_H_INTEGRATOR] == 1) {
    /* classic RK4 on (qpos, qvel); stage states keep the kinematics of the LAST stage in w */
    static const double A[3] = {0.5, 0.5, 1.0};
    static const double Bw[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
    double q0[NQM], v0[NVM], Xv[4][NVM], F[4][NVM], q[NQM], v[NVM];
    memcpy(q0, qpos, sizeof(double) * nq); memcpy(v0, qvel, sizeof(double) * nv);
    memcpy(Xv[0], qvel, sizeof(double) * nv); memcpy(F[0], w->qacc, sizeof(double) * nv);
    for (int i = 1; i < 4; i++) {
      memcpy(q, q0, sizeof(double) * nq);
      integrate_pos(m, q, Xv[i - 1], A[i - 1] * h);
      for (int d = 0; d < nv; d++) v[d] = v0[d] + A[i - 1] * h * F[i - 1][d];
      memcpy(Xv[i], v, sizeof(double) * nv);
      forward(m, q, v, ctrl, w);
      memcpy(F[i], w->qacc, sizeof(double) * nv);
    }
    double dv[NVM], da[NVM];
    for (int d = 0; d < nv; d++) {
      dv[d] = Bw[0] * Xv[0][d] + Bw[1] * Xv[1][d] + Bw[2] * Xv[2][d] + Bw[3] * Xv[3][d];
      da[d] = Bw[0] * F[0][d] + Bw[1] * F[1][d] + Bw[2] * F[2][d] + Bw[3] * F[3][d];
    }
    memcpy(qpos, q0, sizeof(double) * nq);
    integrate_pos(m, qpos, dv, h);
    for (int d = 0; d < nv; d++) qvel[d] = v0[d] + h * da[d];
  } else {
    /* semi-implicit Euler with implicit joint damping: (M + h D) a = M qacc */
    double rhs[NVM];
    int any = 0;
    for (int d = 0; d < nv; d++) if (m->dof_damping[d] > 0) any = 1;
    if (any) {
      for (int i = 0; i < nv; i++) { double s = 0; for (int j = 0; j < nv; j++) s += w->M[i][j] * w->qacc[j]; rhs[i] = s; }
      double MH[NVM][NVM], LH[NVM][NVM];   /* automatic: re-entrant */
      for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) MH[i][j] = w->M[i][j] + (i == j ? h * m->dof_damping[i] : 0);
      cholesky(nv, MH, LH);
      solve_lower(nv, LH, rhs); solve_upper(nv, LH, rhs);
    } else {
      memcpy(rhs, w->qacc, sizeof(double) * nv);
    }
    for (int d = 0; d < nv; d++) qvel[d] += h * rhs[d];
    integrate_pos(m, qpos, qvel, h);
  }
}

/* body-origin velocities from the (stale) kinematics in w and the given qvel: mj_jacBody * qvel     */
static void body_velocities(const SgrlModelView* m, const Work* w, const double* qvel, double* xvelp, double* xvelr) {
  for (int b = 1; b < m->nbody; b++) {
    double cv[6] = {0, 0, 0, 0, 0, 0};
    for (int d = m->body_dofadr[b] + m->body_dofnum[b] - 1; d >= 0; d = m->dof_parent[d])
      for (int k = 0; k < 6; k++) cv[k] += w->cdof[d][k] * qvel[d];
    double off[3] = {w->xpos[b][0] - w->com[0], w->xpos[b][1] - w->com[1], w->xpos[b][2] - w->com[2]};
    double t[3];
    cross3(t, cv, off);
    for (int k = 0; k < 3; k++) { xvelr[3 * b + k] = cv[k]; xvelp[3 * b + k] = cv[3 + k] + t[k]; }
  }
  for (int k = 0; k < 3; k++) xvelp[k] = xvelr[k] = 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* env arithmetic: reward / done / observation (pinned by tests/golden/env_arith.npz)               */
typedef struct {
  /* inputs taken before do_simulation */
  double quat_before[4];     /* qpos[3:7]            <env>.py:16 */
  double pos_before[2];      /* torso xpos[:2]       <env>.py:22 */
} PreStep;

static void quat2mat_ref(double r[3][3], const double* q) { /* reference utils.py:82-104 form */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  r[0][0] = 1 - 2 * y * y - 2 * z * z; r[0][1] = 2 * x * y - 2 * z * w; r[0][2] = 2 * x * z + 2 * y * w;
  r[1][0] = 2 * x * y + 2 * z * w; r[1][1] = 1 - 2 * x * x - 2 * z * z; r[1][2] = 2 * y * z - 2 * x * w;
  r[2][0] = 2 * x * z - 2 * y * w; r[2][1] = 2 * y * z + 2 * x * w; r[2][2] = 1 - 2 * x * x - 2 * y * y;
}

/* obs for all limbs, 41 doubles each.  xpos/xvelp/xvelr are [nbody][3], xaxis [njnt][3]. <env>.py:46-148 */
static void get_obs(const SgrlModelView* m, const double* xpos, const double* xvelp, const double* xvelr,
                    const double* xaxis, const double* qpos, const double* target, double* obs) {
  const double* tp = xpos + 3;
  double dir[2] = {target[0] - tp[0], target[1] - tp[1]};
  double dn = sqrt(dir[0] * dir[0] + dir[1] * dir[1]);
  dir[0] /= dn; dir[1] /= dn;
  const double R2D = 180.0 / PI;
  for (int b = 1; b < m->nbody; b++) {
    double* o = obs + 41 * (b - 1);
    for (int k = 0; k < 41; k++) o[k] = 0;
    for (int k = 0; k < 3; k++) o[k] = xpos[3 * b + k] - tp[k];
    o[5] = -9.81;
    o[6] = dir[0]; o[7] = dir[1];
    for (int k = 0; k < 3; k++) {
      double v = xvelp[3 * b + k];
      o[9 + k] = v < -10 ? -10 : (v > 10 ? 10 : v);
      o[12 + k] = xvelr[3 * b + k];
    }
    if (b == 1) {
      for (int k = 0; k < 3; k++) { o[27 + 3 * k] = 0.5; o[28 + 3 * k] = 0.5; o[29 + 3 * k] = 0.5; }
    } else {
      int j0 = m->body_jntadr[b];
      for (int k = 0; k < 3; k++) {
        int j = j0 + k;
        for (int c = 0; c < 3; c++) o[15 + 3 * k + c] = xaxis[3 * j + c];
        double a0 = qpos[m->jnt_qposadr[j]];
        double lo = m->jnt_range[2 * j] * R2D, hi = m->jnt_range[2 * j + 1] * R2D;
        o[24 + k] = a0;
        o[27 + 3 * k] = (a0 * R2D - lo) / (hi - lo);
        o[28 + 3 * k] = (180.0 + lo) / 360.0;
        o[29 + 3 * k] = (180.0 + hi) / 360.0;
      }
    }
    int lt = m->body_limbtype[b];
    if (lt >= 1 && lt <= 4) o[36 + lt - 1] = 1.0;
    o[40] = xpos[3 * b + 2];
  }
}

/* reward / done.  a = actuator-ordered action (nu).  Returns done flag.  <env>.py:15-44 */
static int reward_done(const SgrlModelView* m, const PreStep* pre, const double* a, const double* xpos_after,
                       const double* qpos_after, const double* qvel_after, const double* target,
                       double* reward, double* dist_out) {
  double rm[3][3];
  quat2mat_ref(rm, pre->quat_before);
  double heading = atan2(rm[1][0], rm[0][0]);
  double pitch = atan2(-rm[2][0], sqrt(rm[2][1] * rm[2][1] + rm[2][2] * rm[2][2]));
  double roll = atan2(rm[2][1], rm[2][2]);
  double hd[2] = {cos(heading), sin(heading)};
  double db[2] = {target[0] - pre->pos_before[0], target[1] - pre->pos_before[1]};
  double dist_before = sqrt(db[0] * db[0] + db[1] * db[1]);
  const double* pa = xpos_after + 3;
  double da[2] = {target[0] - pa[0], target[1] - pa[1]};
  double dist_after = sqrt(da[0] * da[0] + da[1] * da[1]);
  double dt = m->fhdr[SGRL_F_TIMESTEP] * m->hdr[SGRL_H_FRAME_SKIP];
  double height = qpos_after[2];
  double r = (dist_before - dist_after) / dt;
  if (m->fhdr[SGRL_F_HEADING_WEIGHT] != 0.0)
    r += ((pa[0] - pre->pos_before[0]) * hd[0] + (pa[1] - pre->pos_before[1]) * hd[1]) / dt;
  if (m->fhdr[SGRL_F_ALIVE_BONUS] != 0.0) r += m->fhdr[SGRL_F_ALIVE_BONUS];
  double sq = 0;
  for (int u = 0; u < m->nu; u++) sq += a[u] * a[u];
  r -= m->fhdr[SGRL_F_CTRL_COST] * sq;
  *reward = r; *dist_out = dist_after;
  double lo = m->fhdr[SGRL_F_HEIGHT_LO], hi = m->fhdr[SGRL_F_HEIGHT_HI], al = m->fhdr[SGRL_F_ANG_LIMIT];
  int rule = m->hdr[SGRL_H_DONE_RULE], ok;
  if (rule == 0) {
    ok = height > lo && height < hi && fabs(pitch) < al && fabs(roll) < al;
  } else if (rule == 1) {
    const double* q = qpos_after + 3;
    double ang = 2 * atan2(sqrt(q[1] * q[1] + q[2] * q[2]), sqrt(q[0] * q[0] + q[3] * q[3]));
    int fin = 1, small = 1;
    for (int i = 0; i < m->nq; i++) if (!isfinite(qpos_after[i])) fin = 0;
    for (int i = 0; i < m->nv; i++) if (!isfinite(qvel_after[i])) fin = 0;
    for (int i = 3; i < m->nq; i++) if (!(fabs(qpos_after[i]) < 100)) small = 0;
    for (int i = 0; i < m->nv; i++) if (!(fabs(qvel_after[i]) < 100)) small = 0;
    ok = fin && small && height > lo && fabs(ang) < al;
  } else {
    for (int i = 0; i < m->hdr[SGRL_H_NHEIGHT_BODIES]; i++) {
      double z = xpos_after[3 * m->hdr[SGRL_H_HEIGHT_BODY0 + i] + 2];
      if (z < height) height = z;
    }
    double s2 = 0;
    for (int i = 0; i < m->nv; i++) s2 += qvel_after[i] * qvel_after[i];
    ok = height > lo && fabs(pitch) < al && fabs(roll) < al && s2 > 1;
  }
  return !ok;
}

/* counter-based RNG shared with the HIP engine (Philox4x32-10)                                      */
static void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static double rng_uniform01(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t stream, uint32_t idx) {
  uint32_t c[4] = {idx >> 2, episode, stream, (uint32_t)(seed >> 32)};
  philox4x32(c, (uint32_t)seed, env_id);
  return ((double)c[idx & 3] + 0.5) * (1.0 / 4294967296.0);
}

/* ------------------------------------------------------------------------------------------------ */
/* exported API                                                                                     */
typedef struct {
  double qpos[NQM], qvel[NVM];
  double torso_xy_stale[2], target[2];
  int32_t step_count, episode;
} OracleEnv;

int sgrl_oracle_sizeof_env(void) { return (int)sizeof(OracleEnv); }

/* forward dynamics only; qpos quaternion is normalised in place */
int sgrl_oracle_forward(const int32_t* ib, const double* fb, double* qpos, const double* qvel, const double* ctrl,
                        double* qacc, double* M_out, double* diag /* [8] */) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  Work* w = (Work*)calloc(1, sizeof(Work));
  forward(&m, qpos, qvel, ctrl, w);
  for (int d = 0; d < m.nv; d++) qacc[d] = w->qacc[d];
  if (M_out) for (int i = 0; i < m.nv; i++) for (int j = 0; j < m.nv; j++) M_out[i * m.nv + j] = w->M[i][j];
  if (diag) { diag[0] = w->ncon; diag[1] = w->nrow; diag[2] = w->nrow_wanted; diag[3] = w->pgs_last_change;
              diag[4] = w->com[0]; diag[5] = w->com[1]; diag[6] = w->com[2]; diag[7] = w->pgs_iters_used; }
  free(w);
  return 0;
}

/* n mj_steps from (qpos, qvel) with constant ctrl; kin (optional) receives the stale kinematics:
 * xpos[nbody*3], xaxis[njnt*3], xvelp[nbody*3], xvelr[nbody*3] */
int sgrl_oracle_mj_step(const int32_t* ib, const double* fb, double* qpos, double* qvel, const double* ctrl,
                        int nsteps, double* kin) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  Work* w = (Work*)calloc(1, sizeof(Work));
  int overflow = 0;
  for (int s = 0; s < nsteps; s++) { mj_step_once(&m, qpos, qvel, ctrl, w); if (w->nrow_wanted > w->nrow) overflow++; }
  if (kin) {
    double* xpos = kin; double* xaxis = xpos + 3 * m.nbody; double* xvelp = xaxis + 3 * m.njnt; double* xvelr = xvelp + 3 * m.nbody;
    for (int b = 0; b < m.nbody; b++) for (int k = 0; k < 3; k++) xpos[3 * b + k] = w->xpos[b][k];
    for (int j = 0; j < m.njnt; j++) for (int k = 0; k < 3; k++) xaxis[3 * j + k] = w->xaxis[j][k];
    body_velocities(&m, w, qvel, xvelp, xvelr);
  }
  free(w);
  return overflow;
}

/* env arithmetic on injected kinematic snapshots (golden-vector entry point).
 * Returns done; writes obs[41*L], reward, dist. */
int sgrl_oracle_env_epilogue(const int32_t* ib, const double* fb, const double* quat_before, const double* pos_before,
                             const double* env_action, const double* xpos_after, const double* xvelp_after,
                             const double* xvelr_after, const double* xaxis_after, const double* qpos_after,
                             const double* qvel_after, const double* target, double* obs, double* reward,
                             double* dist) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  PreStep pre;
  for (int k = 0; k < 4; k++) pre.quat_before[k] = quat_before[k];
  pre.pos_before[0] = pos_before[0]; pre.pos_before[1] = pos_before[1];
  int done = reward_done(&m, &pre, env_action, xpos_after, qpos_after, qvel_after, target, reward, dist);
  get_obs(&m, xpos_after, xvelp_after, xvelr_after, xaxis_after, qpos_after, target, obs);
  return done;
}

static void fresh_obs(const SgrlModelView* m, OracleEnv* e, Work* w, double* obs) {
  kinematics(m, e->qpos, w);
  com_pos(m, w);
  double xpos[NB * 3], xaxis[NJ * 3], xvelp[NB * 3], xvelr[NB * 3];
  for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 3; k++) xpos[3 * b + k] = w->xpos[b][k];
  for (int j = 0; j < m->njnt; j++) for (int k = 0; k < 3; k++) xaxis[3 * j + k] = w->xaxis[j][k];
  body_velocities(m, w, e->qvel, xvelp, xvelr);
  e->torso_xy_stale[0] = w->xpos[1][0]; e->torso_xy_stale[1] = w->xpos[1][1];
  get_obs(m, xpos, xvelp, xvelr, xaxis, e->qpos, e->target, obs);
}

/* reset_model with the engine's counter RNG (draw order of <env>.py:150-164). */
static void reset_env(const SgrlModelView* m, OracleEnv* e, uint64_t seed, uint32_t env_id, Work* w, double* obs) {
  int nq = m->nq, nv = m->nv;
  uint32_t ep = (uint32_t)e->episode, i = 0;
  double pn = m->fhdr[SGRL_F_RESET_POS_NOISE], vn = m->fhdr[SGRL_F_RESET_VEL_NOISE];
  for (int k = 0; k < nq; k++) e->qpos[k] = m->qpos0[k];
  double rad = (-PI + 2 * PI * rng_uniform01(seed, env_id, ep, 0, i++)) / 2;
  e->qpos[3] = cos(rad); e->qpos[6] = sin(rad);
  for (int k = 0; k < nq; k++) e->qpos[k] += -pn + 2 * pn * rng_uniform01(seed, env_id, ep, 0, i++);
  if (m->hdr[SGRL_H_RESET_VEL_NORMAL]) {
    for (int k = 0; k < nv; k++) {
      double u1 = rng_uniform01(seed, env_id, ep, 0, i++), u2 = rng_uniform01(seed, env_id, ep, 0, i++);
      e->qvel[k] = vn * sqrt(-2.0 * log(u1)) * cos(2 * PI * u2);
    }
  } else {
    for (int k = 0; k < nv; k++) e->qvel[k] = -vn + 2 * vn * rng_uniform01(seed, env_id, ep, 0, i++);
  }
  double r = -PI + 2 * PI * rng_uniform01(seed, env_id, ep, 0, i++);
  double len = 10000.0;
  if (m->hdr[SGRL_H_TARGET_V2]) len = 10.0 + 10.0 * rng_uniform01(seed, env_id, ep, 0, i++);
  e->target[0] = cos(r) * len; e->target[1] = sin(r) * len;
  e->step_count = 0;
  fresh_obs(m, e, w, obs);
}

int sgrl_oracle_env_reset(const int32_t* ib, const double* fb, OracleEnv* e, uint64_t seed, uint32_t env_id, double* obs) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  Work* w = (Work*)calloc(1, sizeof(Work));
  reset_env(&m, e, seed, env_id, w, obs);
  free(w);
  return 0;
}

/* make the env state consistent after an external set of qpos/qvel (gym set_state -> sim.forward) */
int sgrl_oracle_env_refresh(const int32_t* ib, const double* fb, OracleEnv* e, double* obs) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  Work* w = (Work*)calloc(1, sizeof(Work));
  fresh_obs(&m, e, w, obs);
  free(w);
  return 0;
}

/* One VecEnv step for one env: action = policy-ordered slots [3L] (first 3 = torso dummies).
 * obs[41L] receives the post-step observation, or the reset observation when done (auto-reset,
 * reference src/subproc_vec_env.py:12-15).  Returns done (0/1), <0 on error.  info[0]=dist, info[1]=row overflow,
 * info[2]=truncated-by-time-limit. */
int sgrl_oracle_env_step(const int32_t* ib, const double* fb, OracleEnv* e, const double* action, uint64_t seed,
                         uint32_t env_id, int max_episode_steps, int auto_reset, double* obs, double* reward,
                         double* info) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  Work* w = (Work*)calloc(1, sizeof(Work));
  PreStep pre;
  for (int k = 0; k < 4; k++) pre.quat_before[k] = e->qpos[3 + k];
  pre.pos_before[0] = e->torso_xy_stale[0]; pre.pos_before[1] = e->torso_xy_stale[1];
  double a[NVM];
  for (int u = 0; u < m.nu; u++) a[u] = m.act_slot[u] >= 0 ? action[m.act_slot[u]] : 0.0;
  int overflow = 0;
  for (int s = 0; s < m.hdr[SGRL_H_FRAME_SKIP]; s++) {
    mj_step_once(&m, e->qpos, e->qvel, a, w);
    if (w->nrow_wanted > w->nrow) overflow++;
  }
  double xpos[NB * 3], xaxis[NJ * 3], xvelp[NB * 3], xvelr[NB * 3];
  for (int b = 0; b < m.nbody; b++) for (int k = 0; k < 3; k++) xpos[3 * b + k] = w->xpos[b][k];
  for (int j = 0; j < m.njnt; j++) for (int k = 0; k < 3; k++) xaxis[3 * j + k] = w->xaxis[j][k];
  body_velocities(&m, w, e->qvel, xvelp, xvelr);
  double dist;
  int done = reward_done(&m, &pre, a, xpos, e->qpos, e->qvel, e->target, reward, &dist);
  get_obs(&m, xpos, xvelp, xvelr, xaxis, e->qpos, e->target, obs);
  e->torso_xy_stale[0] = xpos[3]; e->torso_xy_stale[1] = xpos[4];
  /* target resampling, <env>.py:41-43 */
  double tn = sqrt(e->target[0] * e->target[0] + e->target[1] * e->target[1]);
  if (dist < 1.0 && tn > 1.0) {
    uint32_t ep = (uint32_t)e->episode, sc = (uint32_t)e->step_count;
    double r = -PI + 2 * PI * rng_uniform01(seed, env_id, ep, 1, 2 * sc);
    if (m.hdr[SGRL_H_TARGET_V2]) {
      double len = 10.0 + 10.0 * rng_uniform01(seed, env_id, ep, 1, 2 * sc + 1);
      e->target[0] = xpos[3] + cos(r) * len; e->target[1] = xpos[4] + sin(r) * len;
    } else { e->target[0] = cos(r) * 10000.0; e->target[1] = sin(r) * 10000.0; }
  }
  e->step_count++;
  int truncated = 0;
  if (max_episode_steps > 0 && e->step_count >= max_episode_steps) { truncated = !done; done = 1; }
  info[0] = dist; info[1] = overflow; info[2] = truncated;
  if (done && auto_reset) { e->episode++; reset_env(&m, e, seed, env_id, w, obs); }
  free(w);
  return done;
}

/* raw RNG access for bit-exactness tests against the HIP engine */
double sgrl_oracle_rng_uniform01(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t stream, uint32_t idx) {
  return rng_uniform01(seed, env_id, episode, stream, idx);
}

/* total energy (kinetic + gravitational potential) for conservation KATs */
int sgrl_oracle_energy(const int32_t* ib, const double* fb, double* qpos, const double* qvel, double* out) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  Work* w = (Work*)calloc(1, sizeof(Work));
  kinematics(&m, qpos, w); com_pos(&m, w); crba(&m, w);
  double ke = 0;
  for (int i = 0; i < m.nv; i++) for (int j = 0; j < m.nv; j++) ke += 0.5 * qvel[i] * w->M[i][j] * qvel[j];
  double pe = 0;
  const double* g = m.fhdr + SGRL_F_GRAV_X;
  for (int b = 1; b < m.nbody; b++) pe -= m.body_mass[b] * dot3(g, w->xipos[b]);
  out[0] = ke; out[1] = pe;
  free(w);
  return 0;
}
