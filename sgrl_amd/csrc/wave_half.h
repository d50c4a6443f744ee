// wave_half.h -- TWO environments per 64-lane wavefront: the wavefront-execution interface of step_body.h for a HALF wave.
//
// The light morphologies (nv <= 12: walker_2 / walker_3, hopper_3) keep 9..12 of a wavefront's 64 lanes busy, and the step kernel is
// bound by the latency of its ~35 dependent phases per dynamics evaluation, not by issue slots: two
// environments of the SAME morphology therefore share one wavefront -- lanes 0..31 advance environment A, lanes 32..63 environment
// B, each on its own LDS slab (step_body.h Layout::pair_stride) -- and both ride the one instruction stream.  step_body.h needs no
// change for this: it is written in SIMT form against `W`; with per-lane slab pointers every value it loads (row counts, free sets,
// pivot counts) is a per-lane value that is uniform within a half, and data-dependent control flow simply diverges BETWEEN the
// halves (the hardware serialises the two sides under the EXEC mask).  Cross-lane traffic never leaves a half.
//
// The algorithms below are written against a primitive set `P` so that the SAME source runs on the gfx950 (wave_hip.h HipHalfPrim:
// DPP row broadcasts, v_permlane16_swap, ballot halves) and in the CPU-only build container (tests/emu/emu_pair.cpp: 32 fibers per
// half that meet at every cross-lane operation):
//   int  lane                      logical lane 0..31 inside the half
//   void sync()                    phase boundary (LDS writes of the phase visible to the next one)
//   double bcast16(double x, int j)   value of logical lane j (< 16) of the caller's half, valid in logical lanes 0..15
//   double xor1(double x)          value of lane ^ 1
//   double half_sum(double) / half_max(double)   reduction over the 32 lanes of the half, result in every lane
//   uint32_t half_ballot(bool)     bit l = predicate of logical lane l of the half
//   fence_lane / fenced / kFixedDims / hdr_const   as in HipWaveT
#pragma once
#include <stdint.h>

#include "step_body.h"

namespace sgrl {

template <int NVCAP, class P>
struct HalfWaveT : P {
  static_assert(NVCAP <= 16, "the register solvers of a half wave keep one matrix row per lane of ONE 16-lane DPP row");
  using P::lane;
  static constexpr int kLanes = 32;

  template <class F> SGRL_DEV void lanes(int n, F f) {
    for (int i = lane; i < n; i += kLanes) f(i);
    P::sync();
  }
  template <class F> SGRL_DEV void lanes_from(int lo, int hi, F f) {
    for (int i = lo + lane; i < hi; i += kLanes) f(i);
    P::sync();
  }
  template <class F> SGRL_DEV double sum(int n, F f) {
    double p = 0.0;
    for (int i = lane; i < n; i += kLanes) p += f(i);
    return P::half_sum(p);
  }
  template <class F> SGRL_DEV double maxabs(int n, F f) {
    double p = 0.0;
    for (int i = lane; i < n; i += kLanes) p = fmax(p, fabs(f(i)));
    return P::half_max(p);
  }
  // bit i = f(i), i < n <= 64: two passes of 32 lanes
  template <class F> SGRL_DEV uint64_t ballot(int n, F f) {
    uint64_t m = P::half_ballot(lane < n ? (bool)f(lane) : false);
    if (n > kLanes) m |= (uint64_t)P::half_ballot(lane + kLanes < n ? (bool)f(lane + kLanes) : false) << 32;
    return m;
  }

  // ---- Cholesky + explicit inverse of the factor on registers (HipWaveT::chol_inv_reg, same arithmetic in the same order): lane i of
  // the half owns row i; pivots and finished inverse rows are broadcast inside the half's first 16-lane row
  template <int NMAX>
  SGRL_DEV void chol_inv_reg(int n, double* Pm, double minval) {
    const bool act = lane < n;
    const int li = act ? lane : 0;
    double* rowp = Pm + li * (li + 1) / 2;
    double a[NMAX + 1];
#pragma unroll
    for (int k = 0; k < NMAX; k++) a[k + 1] = rowp[k < li ? k : li];
    double mydj = 0.0;
#pragma unroll
    for (int j = 0; j < NMAX; j++) {
      if (j < n) {
        const double c = a[j + 1];
        double pj = P::bcast16(c, j);
        pj = pj < minval ? minval : pj;
        const double dj = inv_sqrt(pj);
        const double l = c * dj;
        mydj = lane == j ? dj : mydj;
        const double lm = lane > j ? l * dj : 0.0;
        a[j] = lane == j ? 1.0 : 0.0;
#pragma unroll
        for (int c2 = 0; c2 <= j; c2++) a[c2] -= lm * P::bcast16(a[c2], j);
#pragma unroll
        for (int k = j + 1; k < NMAX; k++) a[k + 1] -= l * P::bcast16(l, k);
      }
    }
    if (act) {
#pragma unroll
      for (int k = NMAX - 1; k >= 0; k--) rowp[k < li ? k : li] = a[k] * mydj;
    }
    P::sync();
  }
  // The instance is picked PER HALF (two free sets of different size classes run one after the other): the result of one
  // environment never depends on what its wave mate is doing.
  SGRL_DEV bool chol_inv_packed(int n, double* Pm, double minval) {
    if (NVCAP >= 9 && n <= 9) chol_inv_reg<9>(n, Pm, minval);
    else if (NVCAP >= 12 && n <= 12) chol_inv_reg<12>(n, Pm, minval);
    else if (NVCAP >= 15 && n <= 15) chol_inv_reg<15>(n, Pm, minval);
    else return false;
    return true;
  }

  // Y_r <- T Y_r for the rows r < nrhs (HipWaveT::trmm_rows_reg): two lanes share one right-hand side, 16 right-hand sides of each
  // environment per pass
  template <int NMAX>
  SGRL_DEV void trmm_rows_reg(int nrhs, int n, const double* T, double* Y, int ldy) {
    constexpr int NH = (NMAX + 1) / 2;
    const int h = lane & 1;
    for (int r0 = 0; r0 < nrhs; r0 += kLanes / 2) {
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
#pragma unroll
      for (int d = (NMAX - 1) | 1; d >= 1; d -= 2) {
        if (d - 1 < n) {
          const double* Td = T + d * (d + 1) / 2 + h;
          const double* Te = T + (d - 1) * d / 2 + h;
          double s0 = 0.0, s1 = 0.0, e0 = 0.0, e1 = 0.0;
#pragma unroll
          for (int u = 0; u < NH; u++) {
            if (2 * u + 1 <= d) {
              const double td = Td[2 * u], te = Te[2 * u];      // d == n: row d lies behind the triangle (read, never stored)
              const double tem = (2 * u + 1 <= d - 1) ? te : (h == 0 ? te : 0.0);
              if (u & 1) { s1 += td * jr[u]; e1 += tem * jr[u]; } else { s0 += td * jr[u]; e0 += tem * jr[u]; }
            }
          }
          double sd = s0 + s1, se = e0 + e1;
          sd += P::xor1(sd);
          se += P::xor1(se);
          if (act) { if (h == 0) y[d - 1] = se; else if (d < n) y[d] = sd; }
        }
      }
    }
    P::sync();
  }
  SGRL_DEV bool trmm_rows(int nrhs, int n, const double* T, double* Y, int ldy) {
#ifdef SGRL_NO_TRMM
    return false;
#endif
    if (NVCAP >= 9 && n <= 9) trmm_rows_reg<9>(nrhs, n, T, Y, ldy);
    else if (NVCAP >= 12 && n <= 12) trmm_rows_reg<12>(nrhs, n, T, Y, ldy);
    else if (NVCAP >= 15 && n <= 15) trmm_rows_reg<15>(nrhs, n, T, Y, ldy);
    else return false;
    return true;
  }
  // the FP64 matrix cores take a whole wavefront: a half keeps the lane-per-entry form of the caller
  SGRL_DEV bool aff_rows(int, int, const double*, int, const int32_t*, const double*, double*) { return false; }

  // Projected Gauss-Seidel on the dual for any number of rows (the fallback of the block-pivot solve; HipWaveT::pgs_big): lane d of
  // the half keeps v[d] = (Y'f)[d] (nv <= 32), the scalars of row r are read by every lane from the row arrays (LDS or the HBM slab)
  SGRL_DEV int pgs(int n, int nv, const double* Y, int ldy, const double* b, const double* R, const double* idg, double* f, double* v,
                   int iters, double thresh) {
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
      for (int r = 0; r < n; r++) {
        const double yl = Y[r * ldy + dl];
        const double y = dofl ? yl : 0.0, br = b[r], Rr = R[r], ir = idg[r], fr = f[r];
        const double dot = P::half_sum(y * vv);
        const double res = br + Rr * fr + dot;
        double fnew = fr - res * ir;
        fnew = fnew < 0.0 ? 0.0 : fnew;
        const double df = fnew - fr;
        vv += y * df;
        if (lane == 0) f[r] = fnew;
        change = fmax(change, fabs(df) / ir);
      }
      P::sync();                       // this sweep's stores of f before the next sweep's loads
      if (change < thresh) { it++; break; }
    }
    if (dofl) v[lane] = vv;
    P::sync();
    return it;
  }
  SGRL_DEV int pgs_big(int n, int nv, const double* Y, int ldy, const double* b, const double* R, const double* idg, double* f,
                       double* v, int iters, double thresh) { return pgs(n, nv, Y, ldy, b, R, idg, f, v, iters, thresh); }

  // substitutions with the factor itself (dof counts beyond NVCAP only: never taken by the instances that are built, kept correct)
  SGRL_DEV void trsv_lower(int n, const double* Pm, const double* dinv, double* x) {
    lanes(1, [&](int) {
      for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int j = 0; j < i; j++) s -= Pm[i * (i + 1) / 2 + j] * x[j];
        x[i] = s * dinv[i];
      }
    });
  }
  SGRL_DEV void trsv_upper(int n, const double* Pm, const double* dinv, double* x) {
    lanes(1, [&](int) {
      for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= Pm[k * (k + 1) / 2 + i] * x[k];
        x[i] = s * dinv[i];
      }
    });
  }
  SGRL_DEV void trsm_lower_rows(int nrhs, int n, const double* Pm, const double* dinv, double* Y, int ldy) {
    lanes(nrhs, [&](int r) {
      double* y = Y + r * ldy;
      for (int i = 0; i < n; i++) {
        double s = y[i];
        for (int j = 0; j < i; j++) s -= Pm[i * (i + 1) / 2 + j] * y[j];
        y[i] = s * dinv[i];
      }
    });
  }
};

}  // namespace sgrl
