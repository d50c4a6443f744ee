// tests/emu/emu_step.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// Instantiates sgrl_amd/csrc/step_body.h with a *serial lane emulator* so that the exact engine source can be
// unit-tested on the CPU-only build container against the oracle before it is run on a GPU.  `lanes(n, f)` runs the
// lane bodies one after another -- in ascending or descending order (set_reverse) -- which also exposes any
// intra-phase cross-lane dependency (the two orders must agree bit for bit).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sgrl_amd/csrc/step_body.h"

namespace {
bool g_reverse = false;
bool g_linv = true;   // exercise the explicit-inverse path (what HipWave takes for nv <= 24); 0 = always the L path

struct EmuWave {
  static constexpr bool kFixedDims = false;
  template <class M> static int hdr_const(const M& m, int idx) { return m.hdr[idx]; }
  void fence_lane() {}
  template <class T> T fenced(T v) { return v; }
  template <class F> void lanes(int n, F f) {
    if (g_reverse) for (int i = n - 1; i >= 0; i--) f(i);
    else for (int i = 0; i < n; i++) f(i);
  }
  template <class F> void lanes_from(int lo, int hi, F f) {
    if (g_reverse) for (int i = hi - 1; i >= lo; i--) f(i);
    else for (int i = lo; i < hi; i++) f(i);
  }
  // same association as the device reduction is NOT required (rounding-level differences are expected)
  template <class F> double sum(int n, F f) { double s = 0; for (int i = 0; i < n; i++) s += f(i); return s; }
  // projected Gauss-Seidel sweeps (reference semantics of HipWave::pgs): f warm-started, v = Y'f maintained
  int pgs(int n, int nv, const double* Y, int ldy, const double* b, const double* R,
           const double* idg, double* f, double* v, int iters, double thresh) {
    for (int d = 0; d < nv; d++) { double s = 0; for (int r = 0; r < n; r++) s += Y[r * ldy + d] * f[r]; v[d] = s; }
    for (int it = 0; it < iters; it++) {
      double change = 0;
      for (int r = 0; r < n; r++) {
        double dot = 0;
        for (int d = 0; d < nv; d++) dot += Y[r * ldy + d] * v[d];
        const double res = b[r] + R[r] * f[r] + dot;
        double fn = f[r] - res * idg[r];
        if (fn < 0) fn = 0;
        const double df = fn - f[r];
        if (df != 0) {
          for (int d = 0; d < nv; d++) v[d] += Y[r * ldy + d] * df;
          f[r] = fn;
          const double c = std::fabs(df) / idg[r];
          if (c > change) change = c;
        }
      }
      if (change < thresh) return it + 1;
    }
    return iters;
  }
  int pgs_big(int n, int nv, const double* Y, int ldy, const double* b, const double* R, const double* idg, double* f, double* v,
              int iters, double thresh) { return pgs(n, nv, Y, ldy, b, R, idg, f, v, iters, thresh); }
  // triangular solves on a packed lower triangle (reference semantics of the HipWave register versions)
  bool trmm_rows(int, int, const double*, double*, int) { return false; }   // the emulator takes the generic lanes() form
  bool aff_rows(int, int, const double*, int, const int32_t*, const double*, double*) { return false; }
  // reference semantics of HipWave::chol_inv_packed: P <- L^-1 (packed, with diagonal)
  bool chol_inv_packed(int n, double* P, double minval) {
    if (!g_linv || n > 24) return false;
    std::vector<double> dinv(n);
    chol_ref(n, P, dinv.data(), minval);
    std::vector<double> X((size_t)n * n, 0.0);
    for (int c = 0; c < n; c++) {
      for (int i = c; i < n; i++) {
        double s = (i == c) ? 1.0 : 0.0;
        for (int j = c; j < i; j++) s -= P[i * (i + 1) / 2 + j] * X[(size_t)j * n + c];
        X[(size_t)i * n + c] = s * dinv[i];
      }
    }
    for (int i = 0; i < n; i++) for (int c = 0; c <= i; c++) P[i * (i + 1) / 2 + c] = X[(size_t)i * n + c];
    return true;
  }
  static void chol_ref(int n, double* P, double* dinv, double minval) {
    for (int j = 0; j < n; j++) {
      double pj = P[j * (j + 1) / 2 + j];
      if (pj < minval) pj = minval;
      const double dj = 1.0 / std::sqrt(pj);
      dinv[j] = dj;
      for (int i = j + 1; i < n; i++) P[i * (i + 1) / 2 + j] *= dj;
      for (int i = j + 1; i < n; i++)
        for (int k = j + 1; k <= i; k++) P[i * (i + 1) / 2 + k] -= P[i * (i + 1) / 2 + j] * P[k * (k + 1) / 2 + j];
    }
  }   // the emulator always takes the generic path
  void trsv_lower(int n, const double* P, const double* dinv, double* x) {
    for (int i = 0; i < n; i++) {
      double s = x[i];
      for (int j = 0; j < i; j++) s -= P[i * (i + 1) / 2 + j] * x[j];
      x[i] = s * dinv[i];
    }
  }
  void trsv_upper(int n, const double* P, const double* dinv, double* x) {
    for (int i = n - 1; i >= 0; i--) {
      double s = x[i];
      for (int k = i + 1; k < n; k++) s -= P[k * (k + 1) / 2 + i] * x[k];
      x[i] = s * dinv[i];
    }
  }
  void trsm_lower_rows(int nrhs, int n, const double* P, const double* dinv, double* Y, int ldy) {
    for (int r = 0; r < nrhs; r++) trsv_lower(n, P, dinv, Y + r * ldy);
  }
  template <class F> uint64_t ballot(int n, F f) { uint64_t m = 0; for (int i = 0; i < n; i++) if (f(i)) m |= (1ull << i); return m; }
  template <class F> double maxabs(int n, F f) { double s = 0; for (int i = 0; i < n; i++) { double v = std::fabs(f(i)); if (v > s) s = v; } return s; }
};
}  // namespace

extern "C" {

void sgrl_emu_set_reverse(int r) { g_reverse = r != 0; }
void sgrl_emu_set_linv(int v) { g_linv = v != 0; }

int sgrl_emu_layout_doubles(const int32_t* ib) { sgrl::Layout o; sgrl::make_layout(ib, &o); return o.s_total; }
int sgrl_emu_layout_bytes(const int32_t* ib) { sgrl::Layout o; sgrl::make_layout(ib, &o); return sgrl::layout_bytes(&o); }

// forward dynamics at (qpos, qvel, ctrl) -> qacc; also returns nrow / ncon
int sgrl_emu_forward(const int32_t* ib, const double* fb, double* qpos, const double* qvel, const double* ctrl,
                     double* qacc, double* diag) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  sgrl::Layout o; sgrl::make_layout(ib, &o);
  std::vector<double> S(o.s_total, NAN);
  std::vector<int32_t> I(o.i_total + 2, -12345);
  EmuWave w;
  sgrl::Engine<EmuWave> e(w, m, o, S.data(), I.data());
  static std::vector<double> fscratch(sgrl::kScratchDoublesMax);
  e.big_scratch = fscratch.data();
  for (int i = 0; i < m.nq; i++) S[o.qpos + i] = qpos[i];
  for (int i = 0; i < m.nv; i++) S[o.qvel + i] = qvel[i];
  for (int i = 0; i < m.nu; i++) S[o.ctrl + i] = ctrl[i];
  I[o.icnt + sgrl::IC_OVERFLOW] = 0;
  I[o.icnt + sgrl::IC_PREV_N] = 0;
  e.forward();
  for (int i = 0; i < m.nv; i++) qacc[i] = S[o.qacc + i];
  for (int i = 0; i < m.nq; i++) qpos[i] = S[o.qpos + i];
  if (diag) {
    int ncon = 0;
    for (int s = 0; s < o.ncon; s++) ncon += I[o.con_valid + s];
    diag[0] = ncon; diag[1] = I[o.icnt + sgrl::IC_NROW]; diag[2] = I[o.icnt + sgrl::IC_NROW_WANTED];
  }
  return 0;
}

// generic env call: op 0 = reset, 1 = step, 2 = refresh.  rec/cnt are the persistent record of one env.
int sgrl_emu_env(int op, const int32_t* ib, const double* fb, double* rec, int32_t* cnt, const float* action,
                 float* obs32, double* obs64, int obs_max_len, uint64_t seed, uint32_t env_id, int max_episode_steps,
                 int auto_reset, double* reward64, uint8_t* done, float* dist, uint8_t* truncated) {
  SgrlModelView m;
  if (sgrl_model_view(ib, fb, &m)) return -1;
  sgrl::Layout o; sgrl::make_layout(ib, &o);
  std::vector<double> S(o.s_total, NAN);
  std::vector<int32_t> I(o.i_total + 2, -12345);
  EmuWave w;
  sgrl::StepIO io;
  io.rec = rec; io.cnt = cnt; io.action = action; io.obs32 = obs32; io.obs64 = obs64; io.reward = nullptr;
  io.done = done; io.dist = dist; io.truncated = truncated; io.reward64 = reward64; io.obs_max_len = obs_max_len;
  static std::vector<double> scratch(sgrl::kScratchDoublesMax);
  io.scratch = scratch.data();
  io.seed = seed; io.env_id = env_id; io.max_episode_steps = max_episode_steps; io.auto_reset = auto_reset;
  if (op == 0) sgrl::env_reset(w, m, o, S.data(), I.data(), io, false);
  else if (op == 1) sgrl::env_step(w, m, o, S.data(), I.data(), io);
  else sgrl::env_refresh(w, m, o, S.data(), I.data(), io);
  return 0;
}
}
