// wave_hip.h -- gfx950 implementation of the wavefront-execution interface used by step_body.h.
// One workgroup = one 64-lane wavefront = one environment.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sgrl_model.h"

namespace sgrl {

template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
// 64-lane sum with a wave-uniform result: xor butterflies inside each 16-lane row on the DPP network
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then four v_readlane row totals.
__device__ __forceinline__ double wave_sum(double x) {
  x += dpp_move<0xB1>(x);
  x += dpp_move<0x4E>(x);
  x += dpp_move<0x141>(x);
  x += dpp_move<0x140>(x);
  return (read_lane(x, 0) + read_lane(x, 16)) + (read_lane(x, 32) + read_lane(x, 48));
}
__device__ __forceinline__ double wave_max(double x) {
  x = fmax(x, dpp_move<0xB1>(x));
  x = fmax(x, dpp_move<0x4E>(x));
  x = fmax(x, dpp_move<0x141>(x));
  x = fmax(x, dpp_move<0x140>(x));
  return fmax(fmax(read_lane(x, 0), read_lane(x, 16)), fmax(read_lane(x, 32), read_lane(x, 48)));
}

// Dimension policy of a kernel instance.  DimsAny: everything comes from the morphology's header (one kernel for all
// morphologies).  DimsFixed<...>: the counts the LDS layout, the model view and every loop bound derive from are compile-time
// constants -- layout offsets become immediates of the LDS instructions, table bases constants, the scalar registers that held
// ~120 offsets / pointers (and spilled: 483 spills in the generic kernel) are free; measured on 8192 x 3d_walker_7_full:
// 4.52-4.78 -> 4.07-4.29 ms per launch (profiles/r3_variant_probe_1.log).  The host only launches an instance on environments
// whose header matches it field for field (engine.hip spec_for).
struct DimsAny {
  static constexpr bool kFixed = false;
  static constexpr bool kPair = false;
  __device__ static __forceinline__ void apply(int32_t*) {}
  template <class M> __device__ static __forceinline__ int hdr_const(const M& m, int idx) { return m.hdr[idx]; }
};
struct DimsAny;
template <int NB, int NJ, int NQ, int NV, int NU, int NG, int NP, int INTEG, int FSKIP, int MAXROWS, int SOLVER, int PAIR = 0>
struct DimsFixed {
  static constexpr bool kFixed = true;
  static constexpr int kNv = NV;
  static constexpr bool kPair = PAIR != 0;     // the family kernel also holds the two-environments-per-wavefront instance of this set
  // a half wave factors on registers up to 15 dofs (wave_half.h chol_inv_packed) and the pair layout aliases dinv onto L on that promise
  static_assert(PAIR == 0 || NV <= 15, "two-environments-per-wavefront instances are built for at most 15 dofs");
  __device__ static __forceinline__ void apply(int32_t* hdr) {
    hdr[SGRL_H_NBODY] = NB; hdr[SGRL_H_NJNT] = NJ; hdr[SGRL_H_NQ] = NQ; hdr[SGRL_H_NV] = NV; hdr[SGRL_H_NU] = NU;
    hdr[SGRL_H_NGEOM] = NG; hdr[SGRL_H_NPAIR] = NP; hdr[SGRL_H_INTEGRATOR] = INTEG; hdr[SGRL_H_FRAME_SKIP] = FSKIP;
    hdr[SGRL_H_MAX_ROWS] = MAXROWS; hdr[SGRL_H_SOLVER] = SOLVER;
  }
  // header fields the engine branches on: constants of the instance
  template <class M> __device__ static __forceinline__ int hdr_const(const M& m, int idx) {
    return idx == SGRL_H_INTEGRATOR ? INTEG : (idx == SGRL_H_SOLVER ? SOLVER : (idx == SGRL_H_FRAME_SKIP ? FSKIP : m.hdr[idx]));
  }
};

typedef double sgrl_v4d __attribute__((ext_vector_type(4)));
// FP64 matrix cores (v_mfma_f64_16x16x4_f64; tools/micro/mfma_f64_lab.hip, profiles/r3_mfma_f64_lab.log: 145 cycles issue
// interval, 180 dependent -- 7 FMA/clk/SIMD, under HALF the vector unit's 16, so they pay only where the vector form leaves
// most lanes idle or waits on LDS: the Gram product A_FF = Y_F Y_F' (11.3 k -> 3.5 k cycles at nf = 24 at the engine's
// residency) and the triangular product Y <- Y L^-T (7.6 k -> 4.8 k at n = 24)).  Operand maps: A[row = lane & 15][k = lane >> 4],
// B[k = lane >> 4][col = lane & 15], D register i = row (lane >> 4) + 4 i, column lane & 15.
#ifndef SGRL_AFF_MFMA_MIN
#define SGRL_AFF_MFMA_MIN 8       // free sets from this size on form A_FF on the matrix cores
#endif
#ifndef SGRL_TRMM_MFMA_MIN
#define SGRL_TRMM_MFMA_MIN 9      // dof counts from this on run the half-solve product on the matrix cores
#endif

// NVCAP: the largest dof count whose register solver instances are compiled in (24 = all; a kernel variant built for the
// small morphologies only carries the instances it can meet: fewer registers)
template <int NVCAP, class D = DimsAny>
struct HipWaveT {
  static constexpr bool kFixedDims = D::kFixed;     // the kernel instance knows the morphology's dimensions at compile time
  template <class M> __device__ static __forceinline__ int hdr_const(const M& m, int idx) { return D::hdr_const(m, idx); }
  int lane;
#ifdef SGRL_PHASE_PROF
  // diagnostic build only (tools/phase_prof.py): s_memtime deltas per phase of Engine::forward, lane 0 accumulates
  long long t_last = 0;
  unsigned long long* prof = nullptr;   // [16] per workgroup
  __device__ __forceinline__ void tick(int id) {
    const long long t = __builtin_readcyclecounter();
    if (id >= 0 && prof && lane == 0) prof[id] += (unsigned long long)(t - t_last);
    t_last = __builtin_readcyclecounter();
  }
#endif
  __device__ __forceinline__ HipWaveT() : lane(threadIdx.x & 63) {}
  // Called at the top of every dynamics evaluation: makes the lane id opaque again so that lane-dependent address
  // arithmetic is not hoisted out of the stage / frame-skip loops (it would stay live across them and spill).
  __device__ __forceinline__ void fence_lane() { asm volatile("" : "+v"(lane)); }
  // same for a wave-uniform value held in scalar registers (e.g. the LDS base of a table)
  template <class T> __device__ __forceinline__ T fenced(T v) {   // 32-bit values on the device (LDS pointers, ints)
    int t = __builtin_amdgcn_readfirstlane((int)(unsigned)(__UINTPTR_TYPE__)v);
    asm volatile("" : "+s"(t));
    return (T)(__UINTPTR_TYPE__)(unsigned)t;
  }
  template <class F> __device__ __forceinline__ void lanes(int n, F f) {
    for (int i = lane; i < n; i += 64) f(i);
    __syncthreads();
  }
  template <class F> __device__ __forceinline__ void lanes_from(int lo, int hi, F f) {
    for (int i = lo + lane; i < hi; i += 64) f(i);
    __syncthreads();
  }
  template <class F> __device__ __forceinline__ double sum(int n, F f) {
    double p = 0.0;
    for (int i = lane; i < n; i += 64) p += f(i);
    return wave_sum(p);
  }
  // Projected Gauss-Seidel on the dual, n <= 64 rows, nv <= 64 dofs, all state in registers:
  //   lane d keeps v[d] = (Y'f)[d]; lane r keeps (b, R, diag, 1/diag, f) of row r, broadcast with v_readlane;
  //   the row residual is one DPP wave reduction; Y rows stream from LDS one row ahead of their use.
  __device__ __forceinline__ int pgs(int n_in, int nv_in, const double* Y, int ldy, const double* b, const double* R,
                                      const double* idg, double* f, double* v, int iters_in,
                                      double thresh) {
    const int n = __builtin_amdgcn_readfirstlane(n_in), nv = __builtin_amdgcn_readfirstlane(nv_in);
    const int iters = __builtin_amdgcn_readfirstlane(iters_in);
    const bool rowl = lane < n, dofl = lane < nv;
    double rb = rowl ? b[lane] : 0.0, rR = rowl ? R[lane] : 0.0;
    double ridg = rowl ? idg[lane] : 0.0, rf = rowl ? f[lane] : 0.0;
    const double rdg = rowl ? 1.0 / ridg : 0.0;       // the row's diagonal A_ii + R_i (only its reciprocal is stored)
    double vv = 0.0;
    for (int r = 0; r < n; r++) {
      const double fr = read_lane(rf, r);
      if (dofl) vv += Y[r * ldy + lane] * fr;
    }
    int it = 0;
    for (; it < iters; it++) {
      double change = 0.0;
      double ynext = dofl ? Y[lane] : 0.0;
      for (int r = 0; r < n; r++) {
        const double y = ynext;
        if (r + 1 < n) ynext = dofl ? Y[(r + 1) * ldy + lane] : 0.0;
        const double dot = wave_sum(y * vv);
        const double fr = read_lane(rf, r);
        const double res = read_lane(rb, r) + read_lane(rR, r) * fr + dot;
        double fn = fr - res * read_lane(ridg, r);
        fn = fn < 0.0 ? 0.0 : fn;
        const double df = fn - fr;
        vv += y * df;
        if (lane == r) rf = fn;
        change = fmax(change, fabs(df) * read_lane(rdg, r));
      }
      if (change < thresh) { it++; break; }
    }
    if (rowl) f[lane] = rf;
    if (dofl) v[lane] = vv;
    __syncthreads();
    return it;
  }
  // The same iteration for ANY number of rows (evaluations with more than 64: the block-pivot solve does not take them, the
  // register version above does not hold them): row scalars and Y rows stream from the environment's HBM slab, the operands
  // of row r + 1 are requested before the arithmetic of row r; lane d keeps v[d].  Slow (a few hundred cycles per row and
  // sweep) and rare: contact-rich states of the many-geom morphologies (a cheetah lying on the floor).
  __device__ __forceinline__ int pgs_big(int n_in, int nv_in, const double* Y, int ldy, const double* b, const double* R,
                                          const double* idg, double* f, double* v, int iters_in, double thresh) {
    const int n = __builtin_amdgcn_readfirstlane(n_in), nv = __builtin_amdgcn_readfirstlane(nv_in);
    const int iters = __builtin_amdgcn_readfirstlane(iters_in);
    const bool dofl = lane < nv;
    const int dl = dofl ? lane : 0;
    double vv = 0.0;
    for (int r = 0; r < n; r++) {
      const double fr = f[r];
      const double y = Y[r * ldy + dl];
      if (dofl) vv += y * fr;
    }
    int it = 0;
    for (; it < iters; it++) {
      double change = 0.0;
      double yn = Y[dl], bn = b[0], Rn = R[0], in_ = idg[0], fn_ = f[0];
      for (int r = 0; r < n; r++) {
        const double y = dofl ? yn : 0.0, br = bn, Rr = Rn, ir = in_, fr = fn_;
        if (r + 1 < n) { yn = Y[(r + 1) * ldy + dl]; bn = b[r + 1]; Rn = R[r + 1]; in_ = idg[r + 1]; fn_ = f[r + 1]; }
        const double dot = wave_sum(y * vv);
        const double res = br + Rr * fr + dot;
        double fnew = fr - res * ir;
        fnew = fnew < 0.0 ? 0.0 : fnew;
        const double df = fnew - fr;
        vv += y * df;
        if (lane == 0) f[r] = fnew;
        change = fmax(change, fabs(df) / ir);
      }
      __syncthreads();                  // this sweep's stores of f before the next sweep's loads
      if (change < thresh) { it++; break; }
    }
    if (dofl) v[lane] = vv;
    __syncthreads();
    return it;
  }
  // ---- triangular solves on a packed lower triangle in LDS (entry (i, j) at i(i+1)/2 + j), n <= 64 ---------------------
  // The serial recurrences run on registers: lane = matrix row (or right-hand side), pivots / solution entries are
  // broadcast with v_readlane, so the dependent chain per step is a few ALU ops instead of an LDS round trip.
  //
  // Cholesky AND explicit inverse of the factor in one register-resident sweep: lane i owns row i of M (-> L) and row
  // i of the identity (-> L^-1).  At pivot j the finished row j of L^-1 is broadcast entry by entry and every later
  // lane folds it in with the same multiplier l_ij it uses for the trailing update -- the inverse costs no extra
  // dependent steps, only issue slots.  On return P holds L^-1 (packed lower triangle INCLUDING its diagonal): the
  // triangular solves of the caller become plain dot products with no recurrence.
  template <int NMAX>
  __device__ __forceinline__ void chol_inv_reg(int n, double* P, double minval) {
    const bool act = lane < n;
    const int li = act ? lane : 0;
    double* rowp = P + li * (li + 1) / 2;
    // ONE register array for both matrices: row entry k of M lives in a[k + 1] until pivot k has consumed it, entry c of
    // the inverse row is born in a[c] at pivot c -- the slot the M entry c - 1 has just vacated.  NMAX + 1 doubles.
    double a[NMAX + 1];
#pragma unroll
    for (int k = 0; k < NMAX; k++) a[k + 1] = rowp[k < li ? k : li];      // k > i re-reads the diagonal: unused filler
    double mydj = 0.0;
#pragma unroll
    for (int j = 0; j < NMAX; j++) {
      if (j < n) {
        const double c = a[j + 1];
        double pj = read_lane(c, j);
        pj = pj < minval ? minval : pj;
        const double dj = rsqrt(pj);
        const double l = c * dj;
        mydj = lane == j ? dj : mydj;
        const double lm = lane > j ? l * dj : 0.0;       // multiplier for the (unscaled) row j of the inverse
        a[j] = lane == j ? 1.0 : 0.0;                    // column j of the identity enters now
#pragma unroll
        for (int c2 = 0; c2 <= j; c2++) a[c2] -= lm * read_lane(a[c2], j);
#pragma unroll
        for (int k = j + 1; k < NMAX; k++) a[k + 1] -= l * read_lane(l, k);
      }
    }
    if (act) {
#pragma unroll
      for (int k = NMAX - 1; k >= 0; k--) rowp[k < li ? k : li] = a[k] * mydj;   // k > i lands on the diagonal slot first, k = i fixes it
    }
    __syncthreads();
  }
  // true when a register version covers n (the caller falls back to its LDS factorisation otherwise)
  __device__ __forceinline__ bool chol_inv_packed(int n_in, double* P, double minval) {
    const int n = __builtin_amdgcn_readfirstlane(n_in);
    if (NVCAP >= 9 && n <= 9) chol_inv_reg<9>(n, P, minval);            // one instance per dof count of the shipped nv <= 24 morphologies
    else if (NVCAP >= 12 && n <= 12) chol_inv_reg<12>(n, P, minval);
    else if (NVCAP >= 15 && n <= 15) chol_inv_reg<15>(n, P, minval);
    else if (NVCAP >= 18 && n <= 18) chol_inv_reg<18>(n, P, minval);
    else if (NVCAP >= 21 && n <= 21) chol_inv_reg<21>(n, P, minval);
    else if (NVCAP >= 24 && n <= 24) chol_inv_reg<24>(n, P, minval);
    else return false;
    return true;
  }
  // Y_r <- T Y_r for the rows r < nrhs, T = packed lower-triangular matrix (the explicit L^-1).  Two lanes share one
  // right-hand side: lane h of the pair keeps the entries c = h (mod 2) of the row in registers and accumulates its half
  // of every output; the halves meet over one DPP exchange.  Every entry of T is an LDS read shared by all pairs, all
  // outputs accumulate independently (no recurrence: pure issue throughput), 32 right-hand sides per pass.
  template <int NMAX>
  __device__ __forceinline__ void trmm_rows_reg(int nrhs, int n, const double* T, double* Y, int ldy) {
    constexpr int NH = (NMAX + 1) / 2;
    const int h = lane & 1;
    for (int r0 = 0; r0 < nrhs; r0 += 32) {
      const int r = r0 + (lane >> 1);
      const bool act = r < nrhs;
      double* y = Y + (act ? r : nrhs - 1) * ldy;
      double jr[NH];
#pragma unroll
      for (int u = 0; u < NH; u++) {
        const int c = 2 * u + h;
        const double v = y[c < n ? c : n - 1];
        jr[u] = c < n ? v : 0.0;
      }
      // two outputs (d, d - 1) per straight-line block: four independent accumulation chains; blocks whose rows lie
      // beyond n are skipped by a uniform branch
#pragma unroll
      for (int d = (NMAX - 1) | 1; d >= 1; d -= 2) {
        if (d - 1 < n) {
          const double* Td = T + d * (d + 1) / 2 + h;        // this lane's columns of row d: c = 2u + h
          const double* Te = T + (d - 1) * d / 2 + h;
          double s0 = 0.0, s1 = 0.0, e0 = 0.0, e1 = 0.0;
#pragma unroll
          for (int u = 0; u < NH; u++) {
            // row d (odd) has columns 0..d: both lanes own u <= (d - 1) / 2.  Row d - 1 (even) ends at column d - 1:
            // its last pair u = (d - 1) / 2 exists for h = 0 only.
            if (2 * u + 1 <= d) {
              const double td = Td[2 * u], te = Te[2 * u];
              const double tem = (2 * u + 1 <= d - 1) ? te : (h == 0 ? te : 0.0);
              if (u & 1) { s1 += td * jr[u]; e1 += tem * jr[u]; } else { s0 += td * jr[u]; e0 += tem * jr[u]; }
            }
          }
          double sd = s0 + s1, se = e0 + e1;
          sd += dpp_move<0xB1>(sd);                          // quad_perm [1,0,3,2]: the partner lane's half
          se += dpp_move<0xB1>(se);
          if (act) { if (h == 0) y[d - 1] = se; else if (d < n) y[d] = sd; }
        }
      }
    }
    __syncthreads();
  }
  // The same product on the matrix cores: Y[r][d] <- sum_{c <= d} T[d][c] Y[r][c], 16 right-hand sides x 16 outputs per tile,
  // only the k-steps of the triangle.  n <= 24 (six k-steps kept in registers), in place: a tile's operands are in registers
  // before its first store, tiles of different right-hand sides do not overlap.
  __device__ __forceinline__ void trmm_rows_mfma(int nrhs, int n, const double* T, double* Y, int ldy) {
    const int lo = lane & 15, hi = lane >> 4;
    const int ksteps = (n + 3) >> 2;
    for (int r0 = 0; r0 < nrhs; r0 += 16) {
      const int r = r0 + lo;
      const bool rv = r < nrhs;
      const double* yr = Y + (rv ? r : 0) * ldy;
      double b[6];
#pragma unroll
      for (int s = 0; s < 6; s++) {
        const int c = 4 * s + hi;
        const double v = yr[c < n ? c : 0];
        b[s] = (rv && c < n) ? v : 0.0;
      }
#pragma unroll
      for (int dt = 0; dt < 2; dt++) {
        if (16 * dt < n) {                                   // uniform
          const int d = 16 * dt + lo;
          const bool dv = d < n;
          const double* Td = T + (dv ? d * (d + 1) / 2 : 0);
          sgrl_v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
          for (int s = 0; s < 6; s++) {
            if (s < ksteps && s < 4 * dt + 4) {              // uniform: k-steps of the triangle only
              const int c = 4 * s + hi;
              const double t = Td[(dv && c <= d) ? c : 0];
              const double a = (dv && c <= d) ? t : 0.0;
              if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s], acc1, 0, 0, 0);
              else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s], acc0, 0, 0, 0);
            }
          }
          if (rv) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const int dd = 16 * dt + hi + 4 * i;
              if (dd < n) Y[r * ldy + dd] = acc0[i] + acc1[i];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // A_FF = Y_F Y_F' + diag(R_F) into the packed lower triangle C (free-set order: row i of the block = constraint row fl[i]),
  // 16 x 16 tiles of the lower triangle, k over the dofs.  false: free set too small to pay (the caller's lane-per-entry form).
  __device__ __forceinline__ bool aff_rows(int nf_in, int nv_in, const double* Y, int ldy, const int32_t* fl, const double* eR, double* C) {
    const int nf = __builtin_amdgcn_readfirstlane(nf_in), nv = __builtin_amdgcn_readfirstlane(nv_in);
    if (nf < SGRL_AFF_MFMA_MIN) return false;
    const int lo = lane & 15, hi = lane >> 4;
    const int nt = (nf + 15) >> 4, ksteps = (nv + 3) >> 2;
    for (int ti = 0; ti < nt; ti++) {
      const int ri = 16 * ti + lo;
      const double* yi = Y + fl[ri < nf ? ri : 0] * ldy;
      for (int tj = 0; tj <= ti; tj++) {
        const int rj = 16 * tj + lo;
        const double* yj = Y + fl[rj < nf ? rj : 0] * ldy;
        sgrl_v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int s = 0; s < ksteps; s += 2) {
          const int d0 = 4 * s + hi, d1 = d0 + 4;
          const double a0 = yi[d0 < nv ? d0 : 0], b0 = yj[d0 < nv ? d0 : 0];
          const double a1 = yi[d1 < nv ? d1 : 0], b1 = yj[d1 < nv ? d1 : 0];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((d0 < nv && ri < nf) ? a0 : 0.0, (d0 < nv && rj < nf) ? b0 : 0.0, acc0, 0, 0, 0);
          if (s + 1 < ksteps)
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((d1 < nv && ri < nf) ? a1 : 0.0, (d1 < nv && rj < nf) ? b1 : 0.0, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int r = 16 * ti + hi + 4 * i, c = 16 * tj + lo;
          if (r < nf && c <= r) {
            double v = acc0[i] + acc1[i];
            if (r == c) v += eR[fl[r]];
            C[r * (r + 1) / 2 + c] = v;
          }
        }
      }
    }
    __syncthreads();
    return true;
  }
  __device__ __forceinline__ bool trmm_rows(int nrhs_in, int n_in, const double* T, double* Y, int ldy) {
#ifdef SGRL_NO_TRMM
    return false;
#endif
    const int n = __builtin_amdgcn_readfirstlane(n_in), nrhs = __builtin_amdgcn_readfirstlane(nrhs_in);
    if (n >= SGRL_TRMM_MFMA_MIN && n <= 24) { trmm_rows_mfma(nrhs, n, T, Y, ldy); return true; }
    if (NVCAP >= 9 && n <= 9) trmm_rows_reg<9>(nrhs, n, T, Y, ldy);
    else if (NVCAP >= 12 && n <= 12) trmm_rows_reg<12>(nrhs, n, T, Y, ldy);
    else if (NVCAP >= 15 && n <= 15) trmm_rows_reg<15>(nrhs, n, T, Y, ldy);
    else if (NVCAP >= 18 && n <= 18) trmm_rows_reg<18>(nrhs, n, T, Y, ldy);
    else if (NVCAP >= 21 && n <= 21) trmm_rows_reg<21>(nrhs, n, T, Y, ldy);
    else if (NVCAP >= 24 && n <= 24) trmm_rows_reg<24>(nrhs, n, T, Y, ldy);
    else return false;
    return true;
  }
  // x <- L^-1 x (lane i keeps x_i; column j of L streams from LDS one step ahead)
  __device__ __forceinline__ double trsv_lower_reg(int n, const double* P, double dv, double x) {
    const double* row = P + lane * (lane + 1) / 2;
    double lnext = (lane > 0 && lane < n) ? row[0] : 0.0;
    for (int j = 0; j < n; j++) {
      const double lj = lnext;
      if (j + 1 < n) lnext = (lane > j + 1 && lane < n) ? row[j + 1] : 0.0;
      const double xj = read_lane(x, j) * read_lane(dv, j);
      x = lane == j ? xj : (lane > j ? x - lj * xj : x);
    }
    return x;
  }
  // x <- L^-T x (lane k keeps x_k; row i of L streams from LDS one step ahead)
  __device__ __forceinline__ double trsv_upper_reg(int n, const double* P, double dv, double x) {
    double lnext = (n > 0 && lane < n - 1) ? P[(n - 1) * n / 2 + lane] : 0.0;
    for (int i = n - 1; i >= 0; i--) {
      const double li = lnext;
      if (i > 0) lnext = lane < i - 1 ? P[(i - 1) * i / 2 + lane] : 0.0;
      const double xi = read_lane(x, i) * read_lane(dv, i);
      x = lane == i ? xi : (lane < i ? x - li * xi : x);
    }
    return x;
  }
  __device__ __forceinline__ void trsv_lower(int n_in, const double* P, const double* dinv, double* xs) {
    const int n = __builtin_amdgcn_readfirstlane(n_in);
    const double dv = lane < n ? dinv[lane] : 0.0;
    const double x = trsv_lower_reg(n, P, dv, lane < n ? xs[lane] : 0.0);
    if (lane < n) xs[lane] = x;
    __syncthreads();
  }
  __device__ __forceinline__ void trsv_upper(int n_in, const double* P, const double* dinv, double* xs) {
    const int n = __builtin_amdgcn_readfirstlane(n_in);
    const double dv = lane < n ? dinv[lane] : 0.0;
    const double x = trsv_upper_reg(n, P, dv, lane < n ? xs[lane] : 0.0);
    if (lane < n) xs[lane] = x;
    __syncthreads();
  }
  // Y_r <- L^-1 Y_r for the rows r < nrhs (row stride ldy), left-looking so that each entry is written once.  SPLIT
  // lanes share one right-hand side: lane q of the group takes the terms j = q (mod SPLIT) of every dot product and the
  // partial sums are folded on the DPP network, so 64 / SPLIT right-hand sides advance per pass.  The newest entry
  // y_{i-1} is taken from a register (every lane of the group has it) instead of waiting for its LDS round trip.
  template <int SPLIT>
  __device__ __forceinline__ void trsm_rows_split(int nrhs, int n, const double* P, const double* dinv, double* Y, int ldy) {
    constexpr int PER = 64 / SPLIT;
    const int q = lane & (SPLIT - 1);
    for (int r0 = 0; r0 < nrhs; r0 += PER) {
      const int r = r0 + lane / SPLIT;
      const bool act = r < nrhs;
      double* y = Y + (act ? r : nrhs - 1) * ldy;
      double yprev = 0.0;
      for (int i = 0; i < n; i++) {
        const double* Li = P + i * (i + 1) / 2;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = q;
        for (; j + 3 * SPLIT < i - 1; j += 4 * SPLIT) {
          const double a0 = Li[j], a1 = Li[j + SPLIT], a2 = Li[j + 2 * SPLIT], a3 = Li[j + 3 * SPLIT];
          const double b0 = y[j], b1 = y[j + SPLIT], b2 = y[j + 2 * SPLIT], b3 = y[j + 3 * SPLIT];
          s0 -= a0 * b0; s1 -= a1 * b1; s2 -= a2 * b2; s3 -= a3 * b3;
        }
        for (; j < i - 1; j += SPLIT) s0 -= Li[j] * y[j];
        double sum = (s0 + s1) + (s2 + s3);
        if (SPLIT >= 2) sum += dpp_move<0xB1>(sum);      // quad_perm [1,0,3,2]
        if (SPLIT >= 4) sum += dpp_move<0x4E>(sum);      // quad_perm [2,3,0,1]
        double t = y[i] + sum;
        if (i > 0) t -= Li[i - 1] * yprev;
        yprev = t * dinv[i];
        if (act && q == 0) y[i] = yprev;
      }
    }
    __syncthreads();
  }
  __device__ __forceinline__ void trsm_lower_rows(int nrhs_in, int n_in, const double* P, const double* dinv, double* Y,
                                                  int ldy) {
    const int n = __builtin_amdgcn_readfirstlane(n_in), nrhs = __builtin_amdgcn_readfirstlane(nrhs_in);
    if (nrhs <= 16) trsm_rows_split<4>(nrhs, n, P, dinv, Y, ldy);
    else if (nrhs <= 32) trsm_rows_split<2>(nrhs, n, P, dinv, Y, ldy);
    else trsm_rows_split<1>(nrhs, n, P, dinv, Y, ldy);
  }
  template <class F> __device__ __forceinline__ uint64_t ballot(int n, F f) {
    const bool p = (lane < n) ? (bool)f(lane) : false;
    return (uint64_t)__ballot(p);
  }
  template <class F> __device__ __forceinline__ double maxabs(int n, F f) {
    double p = 0.0;
    for (int i = lane; i < n; i += 64) p = fmax(p, fabs(f(i)));
    return wave_max(p);
  }
};
using HipWave = HipWaveT<24>;

// gfx950 primitives of the half-wave interface (wave_half.h): lanes 0..31 = environment A, 32..63 = environment B.  A half is two
// 16-lane DPP rows; everything here stays inside the caller's half, so it also holds when only one half is active (the two
// environments' data-dependent branches diverge at half granularity).
template <class D>
struct HipHalfPrim {
  static constexpr bool kFixedDims = D::kFixed;
  template <class M> __device__ static __forceinline__ int hdr_const(const M& m, int idx) { return D::hdr_const(m, idx); }
  int lane;      // logical lane inside the half
  int half;
  __device__ __forceinline__ HipHalfPrim() : lane(threadIdx.x & 31), half((threadIdx.x >> 5) & 1) {}
#ifdef SGRL_PHASE_PROF
  __device__ __forceinline__ void tick(int) {}      // the phase profile is taken on the one-environment instances
#endif
  __device__ __forceinline__ void fence_lane() { asm volatile("" : "+v"(lane)); }
  template <class T> __device__ __forceinline__ T fenced(T v) {
    int t = __builtin_amdgcn_readfirstlane((int)(unsigned)(__UINTPTR_TYPE__)v);
    asm volatile("" : "+s"(t));
    return (T)(__UINTPTR_TYPE__)(unsigned)t;
  }
  __device__ static __forceinline__ void sync() { __syncthreads(); }
  // row_newbcast:j -- lane j of every 16-lane row to the whole row.  The callers keep their operands in logical lanes 0..15 = the
  // FIRST row of the half and read the result there (the second row computes on its own lanes' values, never stored).
  __device__ static __forceinline__ double bcast16(double x, int j) {
    switch (j) {       // j is a constant after unrolling: one DPP move per half of the double
      case 0: return dpp_move<0x150>(x);   case 1: return dpp_move<0x151>(x);   case 2: return dpp_move<0x152>(x);
      case 3: return dpp_move<0x153>(x);   case 4: return dpp_move<0x154>(x);   case 5: return dpp_move<0x155>(x);
      case 6: return dpp_move<0x156>(x);   case 7: return dpp_move<0x157>(x);   case 8: return dpp_move<0x158>(x);
      case 9: return dpp_move<0x159>(x);   case 10: return dpp_move<0x15A>(x);  case 11: return dpp_move<0x15B>(x);
      case 12: return dpp_move<0x15C>(x);  case 13: return dpp_move<0x15D>(x);  case 14: return dpp_move<0x15E>(x);
      default: return dpp_move<0x15F>(x);
    }
  }
  __device__ static __forceinline__ double xor1(double x) { return dpp_move<0xB1>(x); }      // quad_perm [1,0,3,2]
  // v_permlane16_swap (gfx950): odd rows of the first operand <-> even rows of the second; with both = x the pair comes back as
  // (row0 row0 row2 row2), (row1 row1 row3 row3): every lane holds both row values of its half
  __device__ static __forceinline__ void rows_of_half(double x, double* even, double* odd) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    *even = __hiloint2double((int)h[0], (int)l[0]);
    *odd = __hiloint2double((int)h[1], (int)l[1]);
  }
  __device__ static __forceinline__ double half_sum(double x) {
    x += dpp_move<0xB1>(x);
    x += dpp_move<0x4E>(x);
    x += dpp_move<0x141>(x);
    x += dpp_move<0x140>(x);
    double e, o;
    rows_of_half(x, &e, &o);
    return e + o;
  }
  __device__ static __forceinline__ double half_max(double x) {
    x = fmax(x, dpp_move<0xB1>(x));
    x = fmax(x, dpp_move<0x4E>(x));
    x = fmax(x, dpp_move<0x141>(x));
    x = fmax(x, dpp_move<0x140>(x));
    double e, o;
    rows_of_half(x, &e, &o);
    return fmax(e, o);
  }
  __device__ __forceinline__ uint32_t half_ballot(bool p) const {
    const uint64_t b = (uint64_t)__ballot(p);
    return half ? (uint32_t)(b >> 32) : (uint32_t)b;
  }
};

}  // namespace sgrl
