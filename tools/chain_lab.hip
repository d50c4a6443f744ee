// tools/chain_lab.hip -- diagnostic only: the fused back-to-back products of chain_f16.h against the single products they
// replace (gemm_f32.h), on the shapes of one SET forward: time of each form, the largest difference between their outputs, and
// (mode r) both forms against float64 with operands at 1e-20 .. 1e8 -- the range check of the row-scaled two-piece products.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/chain_lab.exe tools/chain_lab.hip && tools/chain_lab.exe [nodes] [r]
//   ... -DSGRL_CHAIN_PROF -o tools/chain_lab_prof.exe: also prints the per-phase cycle shares of the site kernel (s_memtime stamps)
#include "../sgrl_amd/csrc/chain_f16.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace sgrl_gemm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class F>
static float time_us(F&& launch, int reps = 20) {
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  for (int w = 0; w < 3; w++) launch();
  hipEventRecord(t0, 0);
  for (int r = 0; r < reps; r++) launch();
  hipEventRecord(t1, 0);
  hipEventSynchronize(t1);
  float ms = 0;
  hipEventElapsedTime(&ms, t0, t1);
  if (hipGetLastError() != hipSuccess) printf("  launch error!\n");
  return ms * 1e3f / reps;
}

constexpr auto kSplitPlain = k_gemm3<0, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
constexpr auto kSplitRelu = k_gemm3<EPI_RELU, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
constexpr auto kSplitRowdiv = k_gemm3<EPI_ROWDIV, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
constexpr auto kSplitLn = k_gemm3<EPI_ROWDIV | EPI_LN, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
constexpr auto kGram = k_gemm3<EPI_RELU, 4, 2, 1, 2, 16, 2, false, false, false, 0, true, 2, true, 2>;
constexpr auto kProj = k_gemm3<EPI_ZSPLIT, 4, 1, 1, 2, 16, 2, false, false, false, 0, false, 2, false, 2>;
constexpr int kLds128 = TileCfg3<4, 2, 1, 2, 16, 2>::kLdsBytes;
constexpr int kLdsProj = TileCfg3<4, 1, 1, 2, 16, 2>::kLdsBytes;

template <class K> static void raise(K k, int lds) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }

static std::vector<float> fetch(const float* d, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  return h;
}
static void report(const char* what, const std::vector<float>& a, const std::vector<float>& b, int ld, int cols) {
  double worst = 0, mag = 0;
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); i++) {
    if ((int)(i % ld) >= cols) continue;
    const double d = fabs((double)a[i] - b[i]);
    if (!(d == d)) bad++;
    if (d > worst) worst = d;
    if (fabs(a[i]) > mag) mag = fabs(a[i]);
  }
  printf("      %-28s max |fused - single| %.3e (largest value %.3e, NaN %zu)\n", what, worst, mag, bad);
}

static unsigned g_seed = 12345;
static float rnd() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffff) / 65536.0f - 0.5f; }
struct Host { std::vector<float> h; float* d; };
static Host dev(size_t n, float scale, float offset = 0.f) {
  Host r; r.h.resize(n);
  for (auto& v : r.h) v = rnd() * scale + offset;
  CK(hipMalloc(&r.d, n * 4)); CK(hipMemcpy(r.d, r.h.data(), n * 4, hipMemcpyHostToDevice));
  return r;
}
// weights as k_encode_rows leaves them: words of the rows scaled by powers of two + the inverse scales
struct Wt { unsigned* w; float* sc; };
static EncMat* g_lab_mat;      // the descriptor of the last words() call (device)
static Wt words(const float* w, int rows, int K) {
  Wt r;
  CK(hipMalloc(&r.w, (size_t)rows * K * 4)); CK(hipMalloc(&r.sc, (size_t)rows * 4));
  EncMat m{0, rows, K, 0, 1};
  EncMat* dm; CK(hipMalloc(&dm, sizeof(EncMat))); CK(hipMemcpy(dm, &m, sizeof(EncMat), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_encode_rows, dim3((rows + 3) / 4), dim3(256), 0, 0, w, r.w, r.sc, dm, 1, rows);
  g_lab_mat = dm;
  return r;
}
static GemmArgs gemm(const float* A, int lda, const Wt& W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K, int flags,
                     const float* rowdiv = nullptr, float* C2 = nullptr, int ldc2 = 0) {
  GemmArgs g{A, lda, reinterpret_cast<const float*>(W.w), ldw, bias, C, ldc, M, N, K, flags, rowdiv, C2, ldc2};
  g.wscale = W.sc;
  return g;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 35840;
  const int N3 = 3 * N;
  const char mode = argc > 2 ? argv[2][0] : ' ';
  raise(kSplitPlain, kLds128); raise(kSplitRelu, kLds128); raise(kSplitRowdiv, kLds128); raise(kSplitLn, kLds128); raise(kGram, kLds128); raise(kProj, kLdsProj);
  raise(k_chain<0, 256, EPI_ROWDIV | EPI_LN, 0>, kChainLds); raise(k_chain<0, 256, 0, 0>, kChainLds); raise(k_chain<1, 256, 0, 0>, kChainLds);
  raise(k_chain<1, 256, 0, 1>, kChainLds); raise(k_chain<1, 256, 0, 2>, kChainLds); raise(k_chain<0, 128, 0, 0>, kChainLds);
  raise(k_chain<1, 128, 0, 2>, kChainLds);
  const int tiles = (N + 127) / 128, blocks = (N + kChainRows - 1) / kChainRows;

  Host cat = dev((size_t)N * 256, 2.0f);               // [inv | ng]
  Host fn = dev((size_t)N, 1.0f, 2.0f);
  Host W1 = dev(256 * 256, 0.2f), b1 = dev(256, 1.0f), W2 = dev(128 * 256, 0.2f), b2 = dev(128, 1.0f);
  Host lnw = dev(128, 1.0f, 1.0f), lnb = dev(128, 1.0f);
  Wt W1w = words(W1.d, 256, 256), W2w = words(W2.d, 128, 256);
  float *h256, *out_a, *out_b, *ng_a, *ng_b;
  CK(hipMalloc(&h256, (size_t)N * 256 * 4)); CK(hipMalloc(&out_a, (size_t)N * 256 * 4)); CK(hipMalloc(&out_b, (size_t)N * 256 * 4));
  CK(hipMalloc(&ng_a, (size_t)N * 256 * 4)); CK(hipMalloc(&ng_b, (size_t)N * 256 * 4));
  CK(hipMemset(out_a, 0, (size_t)N * 256 * 4)); CK(hipMemset(out_b, 0, (size_t)N * 256 * 4));
  printf("chain lab: %d nodes (%d row tiles of 128, %d workgroups of %d rows)\n", N, tiles, blocks, kChainRows);

  if (mode == 'r') {
    // ---- range: single product and fused pair against float64, activations x sa, weights x sw -----------------------------------
    const int M = 4096;
    for (float sa : {1.f, 1e8f, 1e-20f, 3e4f})
      for (float sw : {1.f, 1e4f, 1e-6f}) {
        std::vector<float> hA((size_t)M * 256), hW1(256 * 256), hW2(128 * 256);
        for (auto& v : hA) v = rnd() * 2.0f * sa * std::pow(10.0f, 3.0f * rnd());      // three decades inside a row
        for (int r = 0; r < M; r += 7) for (int k = 0; k < 16; k++) hA[(size_t)r * 256 + k] = 0.f;   // rows whose first k-tile is zero: the estimate fails
        for (int r = 3; r < M; r += 11) for (int k = 0; k < 16; k++) hA[(size_t)r * 256 + k] *= 1e-4f; // ... or is far too small
        for (auto& v : hW1) v = rnd() * 0.2f * sw;
        for (auto& v : hW2) v = rnd() * 0.2f * sw;
        float *dA, *dW1, *dW2, *dC, *dH;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW1, hW1.size() * 4)); CK(hipMalloc(&dW2, hW2.size() * 4));
        CK(hipMalloc(&dC, (size_t)M * 256 * 4)); CK(hipMalloc(&dH, (size_t)M * 256 * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW1, hW1.data(), hW1.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW2, hW2.data(), hW2.size() * 4, hipMemcpyHostToDevice));
        Wt w1 = words(dW1, 256, 256), w2 = words(dW2, 128, 256);
        GemmArgs g1 = gemm(dA, 256, w1, 256, b1.d, dH, 256, M, 256, 256, EPI_RELU);
        hipLaunchKernelGGL(kSplitRelu, dim3((M / 128) * 2), dim3(512), kLds128, 0, g1);
        ChainArgs c{};
        c.A = dA; c.lda = 256; c.W1 = w1.w; c.ldw1 = 256; c.b1 = b1.d; c.W2 = w2.w; c.ldw2 = 256; c.b2 = b2.d; c.C = dC; c.ldc = 256; c.M = M; c.K1 = 256;
        c.ws1 = w1.sc; c.ws2 = w2.sc;
        hipLaunchKernelGGL((k_chain<0, 256, 0, 0>), dim3(M / kChainRows), dim3(512), kChainLds, 0, c);
        CK(hipDeviceSynchronize());
        std::vector<float> gH = fetch(dH, (size_t)M * 256), gC = fetch(dC, (size_t)M * 256);
        double w1e = 0, w2e = 0;
        size_t nan = 0;
        unsigned q = 777;
        for (int smp = 0; smp < 3000; smp++) {
          q = q * 1664525u + 1013904223u;
          const int m = smp < 600 ? (smp % 2 ? 7 * (smp % 500) : 3 + 11 * (smp % 300)) % M : (q >> 4) % M;
          q = q * 1664525u + 1013904223u; const int n = (q >> 4) % 128;
          std::vector<double> H(256);
          for (int h = 0; h < 256; h++) {
            double r = b1.h[h], mag = fabs((double)b1.h[h]);
            for (int k = 0; k < 256; k++) { const double p = (double)hA[(size_t)m * 256 + k] * hW1[(size_t)h * 256 + k]; r += p; mag += fabs(p); }
            H[h] = r > 0 ? r : 0;
            if (h == n || h == n + 128) {
              const double e = fabs((double)gH[(size_t)m * 256 + h] - H[h]) / mag;
              if (!(e == e)) nan++; else if (e > w1e) w1e = e;
            }
          }
          double r = b2.h[n], mag = fabs((double)b2.h[n]);
          for (int h = 0; h < 256; h++) { const double p = H[h] * hW2[(size_t)n * 256 + h]; r += p; mag += fabs(p); }
          const double e = fabs((double)gC[(size_t)m * 256 + n] - r) / mag;
          if (!(e == e)) nan++; else if (e > w2e) w2e = e;
        }
        printf("  activations x %-7.0e weights x %-6.0e: single product err/sum|a w| %.2e | fused pair %.2e | NaN %zu\n", sa, sw, w1e, w2e, nan);
        hipFree(dA); hipFree(dW1); hipFree(dW2); hipFree(dC); hipFree(dH);
      }
    return 0;
  }

  // ---- (1) linear1 -> ReLU -> linear2 -> / fn -> residual + LayerNorm (in place on ng = cat[:, 128:]) ----------------------
  {
    GemmArgs g1 = gemm(cat.d, 256, W1w, 256, b1.d, h256, 256, N, 256, 256, EPI_RELU);
    GemmArgs g2 = gemm(h256, 256, W2w, 256, b2.d, nullptr, 0, N, 128, 256, EPI_ROWDIV | EPI_LN, fn.d);
    g2.ln_io = ng_a + 128; g2.ln_ld = 256; g2.ln_w = lnw.d; g2.ln_b = lnb.d;
    ChainArgs c{};
    c.A = cat.d; c.lda = 256; c.W1 = W1w.w; c.ldw1 = 256; c.b1 = b1.d; c.W2 = W2w.w; c.ldw2 = 256; c.b2 = b2.d; c.M = N; c.K1 = 256;
    c.rowdiv = fn.d; c.ln_io = ng_b + 128; c.ln_ld = 256; c.ln_w = lnw.d; c.ln_b = lnb.d; c.ws1 = W1w.sc; c.ws2 = W2w.sc;
    CK(hipMemcpy(ng_a, cat.d, (size_t)N * 256 * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(ng_b, cat.d, (size_t)N * 256 * 4, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, g1);
    hipLaunchKernelGGL(kSplitLn, dim3(tiles), dim3(512), kLds128, 0, g2);
    hipLaunchKernelGGL((k_chain<0, 256, EPI_ROWDIV | EPI_LN, 0>), dim3(blocks), dim3(512), kChainLds, 0, c);
    CK(hipDeviceSynchronize());
    report("l1 -> l2 + LayerNorm", fetch(ng_a, (size_t)N * 256), fetch(ng_b, (size_t)N * 256), 256, 256);
    const float ta = time_us([&] { hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, g1); });
    const float tb = time_us([&] { hipLaunchKernelGGL(kSplitLn, dim3(tiles), dim3(512), kLds128, 0, g2); });
    const float tab = time_us([&] { hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, g1); hipLaunchKernelGGL(kSplitLn, dim3(tiles), dim3(512), kLds128, 0, g2); });
    const float tf = time_us([&] { hipLaunchKernelGGL((k_chain<0, 256, EPI_ROWDIV | EPI_LN, 0>), dim3(blocks), dim3(512), kChainLds, 0, c); });
    printf("  l1 -> l2 + LN   : single products %.1f + %.1f us (back to back %.1f) | fused %.1f us\n", ta, tb, tab, tf);
    GemmArgs g2p = gemm(h256, 256, W2w, 256, b2.d, out_a, 256, N, 128, 256, 0);
    ChainArgs cp = c; cp.C = out_b; cp.ldc = 256;
    hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, g1);
    hipLaunchKernelGGL(kSplitPlain, dim3(tiles), dim3(512), kLds128, 0, g2p);
    hipLaunchKernelGGL((k_chain<0, 256, 0, 0>), dim3(blocks), dim3(512), kChainLds, 0, cp);
    CK(hipDeviceSynchronize());
    report("l1 -> l2 (plain store)", fetch(out_a, (size_t)N * 256), fetch(out_b, (size_t)N * 256), 256, 128);
    const float tp = time_us([&] { hipLaunchKernelGGL(kSplitPlain, dim3(tiles), dim3(512), kLds128, 0, g2p); });
    const float tfp = time_us([&] { hipLaunchKernelGGL((k_chain<0, 256, 0, 0>), dim3(blocks), dim3(512), kChainLds, 0, cp); });
    printf("  l1 -> l2 plain  : single products %.1f + %.1f us | fused %.1f us\n", ta, tp, tfp);
  }
  // ---- (2) projection -> Gram operand -> linear_g1 -> ReLU -> linear_g2 ----------------------------------------------------
  {
    Host g = dev((size_t)N3 * 128, 2.0f);
    Host Wp = dev(64 * 128, 0.3f);
    CK(hipMemset(Wp.d + 30 * 128, 0, 2 * 128 * 4)); CK(hipMemset(Wp.d + 62 * 128, 0, 2 * 128 * 4));
    Wt Wpw = words(Wp.d, 64, 128);
    Host Wg = dev(256 * 576, 0.05f), bg = dev(256, 1.0f);
    Wt Wgw = words(Wg.d, 256, 576);
    Host zc0 = dev((size_t)N3 * 32, 1.0f);
    float *zc_a = zc0.d, *zc_b, *z2_a, *z2_b, *fn_a, *fn_b;
    CK(hipMalloc(&zc_b, (size_t)N3 * 32 * 4)); CK(hipMalloc(&z2_a, (size_t)N3 * 32 * 4)); CK(hipMalloc(&z2_b, (size_t)N3 * 32 * 4));
    CK(hipMalloc(&fn_a, (size_t)N * 4)); CK(hipMalloc(&fn_b, (size_t)N * 4));
    CK(hipMemcpy(zc_b, zc_a, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(z2_a, zc_a, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(z2_b, zc_a, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice));
    for (int nz = 1; nz <= 2; nz++) {
      GemmArgs p = gemm(g.d, 128, Wpw, 128, nullptr, zc_a, 32, N3, nz == 2 ? 64 : 32, 128, EPI_ZSPLIT, nullptr, nz == 2 ? z2_a : nullptr, 32);
      GemmArgs gg = gemm(zc_a, 96, Wgw, 576, bg.d, h256, 256, N, 256, 576, EPI_RELU);
      gg.rowdiv_out = fn_a;
      GemmArgs g2p = gemm(h256, 256, W2w, 256, b2.d, out_a, 256, N, 128, 256, 0);
      ChainArgs c{};
      c.A = zc_b; c.W1 = Wgw.w; c.ldw1 = 576; c.b1 = bg.d; c.W2 = W2w.w; c.ldw2 = 256; c.b2 = b2.d; c.C = out_b; c.ldc = 256; c.M = N; c.K1 = 576;
      c.fn_out = fn_b; c.X = g.d; c.ldx = 128; c.Kp = 128; c.Wp = Wpw.w; c.zc = zc_b; c.z2 = nz == 2 ? z2_b : nullptr;
      c.ws1 = Wgw.sc; c.ws2 = W2w.sc; c.wsp = Wpw.sc;
      const int pgrid = (N3 + 127) / 128;
      auto single = [&] {
        hipLaunchKernelGGL(kProj, dim3(pgrid), dim3(256), kLdsProj, 0, p);
        hipLaunchKernelGGL(kGram, dim3(tiles * 2), dim3(512), kLds128, 0, gg);
        hipLaunchKernelGGL(kSplitPlain, dim3(tiles), dim3(512), kLds128, 0, g2p);
      };
      auto fused = [&] {
        if (nz == 1) hipLaunchKernelGGL((k_chain<1, 256, 0, 1>), dim3(blocks), dim3(512), kChainLds, 0, c);
        else hipLaunchKernelGGL((k_chain<1, 256, 0, 2>), dim3(blocks), dim3(512), kChainLds, 0, c);
      };
      single(); fused();
      CK(hipDeviceSynchronize());
      printf("  site with %d projection(s):\n", nz);
      report("zc", fetch(zc_a, (size_t)N3 * 32), fetch(zc_b, (size_t)N3 * 32), 32, 32);
      if (nz == 2) report("z2", fetch(z2_a, (size_t)N3 * 32), fetch(z2_b, (size_t)N3 * 32), 32, 32);
      report("fn", fetch(fn_a, N), fetch(fn_b, N), 1, 1);
      report("proj -> lg1 -> lg2", fetch(out_a, (size_t)N * 256), fetch(out_b, (size_t)N * 256), 256, 128);
      const float t1 = time_us([&] { hipLaunchKernelGGL(kProj, dim3(pgrid), dim3(256), kLdsProj, 0, p); });
      const float t2 = time_us([&] { hipLaunchKernelGGL(kGram, dim3(tiles * 2), dim3(512), kLds128, 0, gg); });
      const float t3 = time_us([&] { hipLaunchKernelGGL(kSplitPlain, dim3(tiles), dim3(512), kLds128, 0, g2p); });
      const float ts = time_us(single), tf = time_us(fused);
      if (nz == 2) {
        // round 6: the same site kernel with the prologue's X arriving as row-scaled words (k_chain DBG = 8, lab only)
        Wt gw = words(g.d, N3, 128);
        ChainArgs cw = c; cw.X = reinterpret_cast<const float*>(gw.w); cw.xsc = gw.sc;
        float *zc_w, *z2_w, *out_w;
        CK(hipMalloc(&zc_w, (size_t)N3 * 32 * 4)); CK(hipMalloc(&z2_w, (size_t)N3 * 32 * 4)); CK(hipMalloc(&out_w, (size_t)N * 256 * 4));
        CK(hipMemset(out_w, 0, (size_t)N * 256 * 4));
        CK(hipMemcpy(zc_w, zc0.d, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(z2_w, zc0.d, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice));   // (columns 30, 31 are never written)
        cw.zc = zc_w; cw.z2 = z2_w; cw.A = zc_w; cw.C = out_w;
        raise(k_chain<1, 256, 0, 2, 8>, kChainLds);
        auto fusedw = [&] { hipLaunchKernelGGL((k_chain<1, 256, 0, 2, 8>), dim3(blocks), dim3(512), kChainLds, 0, cw); };
        fusedw();
        CK(hipDeviceSynchronize());
        report("zc, X as words vs X as f32", fetch(zc_b, (size_t)N3 * 32), fetch(zc_w, (size_t)N3 * 32), 32, 32);
        report("site output, X as words", fetch(out_b, (size_t)N * 256), fetch(out_w, (size_t)N * 256), 256, 128);
        std::vector<float> a7, b7;
        for (int r = 0; r < 7; r++) { a7.push_back(time_us(fused, 40)); b7.push_back(time_us(fusedw, 40)); }
        std::sort(a7.begin(), a7.end()); std::sort(b7.begin(), b7.end());
        printf("    site kernel (two projections), alternating 7 x 40 launches: median %.1f us, best %.1f | X as words: median %.1f us, best %.1f\n", a7[3], a7[0], b7[3], b7[0]);
      }
      ChainArgs c0 = c; c0.A = zc_a;
      const float tf0 = time_us([&] { hipLaunchKernelGGL((k_chain<1, 256, 0, 0>), dim3(blocks), dim3(512), kChainLds, 0, c0); });
      printf("    single products proj %.1f + lg1 %.1f + lg2 %.1f us (back to back %.1f) | fused site %.1f us | fused lg1 -> lg2 without the projection %.1f us\n",
             t1, t2, t3, ts, tf, tf0);
#ifdef SGRL_CHAIN_PROF
      {   // where a workgroup of the site kernel spends its cycles (s_memtime at the phase boundaries, wave 0 of every workgroup)
        unsigned long long z8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pr[8];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_prof), z8, sizeof(z8)));
        fused();
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_chain_prof), sizeof(pr)));
        const char* nm[6] = {"prologue (projections)", "phase-1 set-up (Gram norm, stage 0)", "phase-1 k-loop (36 k-tiles)", "hand-off (bias, ReLU, split)", "phase 2 (16 k-steps)", "epilogue (stores)"};
        const double wg = (double)pr[7], tot = (double)pr[6];
        printf("    site kernel, %d workgroups, mean %.0f cycles (100 MHz s_memtime ticks x clock ratio; shares matter) per workgroup:\n", (int)wg, tot / wg);
        for (int i = 0; i < 6; i++) printf("      %-40s %8.0f  %5.1f %%\n", nm[i], pr[i] / wg, 100.0 * pr[i] / tot);
      }
#endif
    }
    {
      // the head's site: K = 144 projections, hidden width 128; linear1_ng -> linear2_ng: K 160, hidden 128
      Host og = dev((size_t)N3 * 144, 2.0f), Wp2 = dev(64 * 144, 0.3f), Wh = dev(128 * 576, 0.05f), W2h = dev(128 * 128, 0.2f);
      Wt Wp2w = words(Wp2.d, 64, 144), Whw = words(Wh.d, 128, 576), W2hw = words(W2h.d, 128, 128);
      GemmArgs p = gemm(og.d, 144, Wp2w, 144, nullptr, zc_a, 32, N3, 64, 144, EPI_ZSPLIT, nullptr, z2_a, 32);
      GemmArgs gg = gemm(zc_a, 96, Whw, 576, bg.d, h256, 128, N, 128, 576, EPI_RELU);
      gg.rowdiv_out = fn_a;
      GemmArgs g2p = gemm(h256, 128, W2hw, 128, b2.d, out_a, 256, N, 128, 128, 0);
      ChainArgs c{};
      c.A = zc_b; c.W1 = Whw.w; c.ldw1 = 576; c.b1 = bg.d; c.W2 = W2hw.w; c.ldw2 = 128; c.b2 = b2.d; c.C = out_b; c.ldc = 256; c.M = N; c.K1 = 576;
      c.fn_out = fn_b; c.X = og.d; c.ldx = 144; c.Kp = 144; c.Wp = Wp2w.w; c.zc = zc_b; c.z2 = z2_b; c.ws1 = Whw.sc; c.ws2 = W2hw.sc; c.wsp = Wp2w.sc;
      auto single = [&] {
        hipLaunchKernelGGL(kProj, dim3((N3 + 127) / 128), dim3(256), kLdsProj, 0, p);
        hipLaunchKernelGGL(kGram, dim3(tiles), dim3(512), kLds128, 0, gg);
        hipLaunchKernelGGL(kSplitPlain, dim3(tiles), dim3(512), kLds128, 0, g2p);
      };
      auto fused = [&] { hipLaunchKernelGGL((k_chain<1, 128, 0, 2>), dim3(blocks), dim3(512), kChainLds, 0, c); };
      single(); fused();
      CK(hipDeviceSynchronize());
      printf("  head site (K 144 projections, hidden 128):\n");
      report("zc", fetch(zc_a, (size_t)N3 * 32), fetch(zc_b, (size_t)N3 * 32), 32, 32);
      report("z2", fetch(z2_a, (size_t)N3 * 32), fetch(z2_b, (size_t)N3 * 32), 32, 32);
      report("proj -> l1g -> l2g", fetch(out_a, (size_t)N * 256), fetch(out_b, (size_t)N * 256), 256, 128);
      printf("    single products back to back %.1f us | fused %.1f us\n", time_us(single), time_us(fused));
      Host ong = dev((size_t)N * 160, 2.0f), Wn = dev(128 * 160, 0.2f);
      Wt Wnw = words(Wn.d, 128, 160);
      GemmArgs n1 = gemm(ong.d, 160, Wnw, 160, b1.d, h256, 128, N, 128, 160, EPI_RELU);
      ChainArgs cn{};
      cn.A = ong.d; cn.lda = 160; cn.W1 = Wnw.w; cn.ldw1 = 160; cn.b1 = b1.d; cn.W2 = W2hw.w; cn.ldw2 = 128; cn.b2 = b2.d; cn.C = out_b; cn.ldc = 256; cn.M = N; cn.K1 = 160;
      cn.ws1 = Wnw.sc; cn.ws2 = W2hw.sc;
      auto single_n = [&] {
        hipLaunchKernelGGL(kSplitRelu, dim3(tiles), dim3(512), kLds128, 0, n1);
        hipLaunchKernelGGL(kSplitPlain, dim3(tiles), dim3(512), kLds128, 0, g2p);
      };
      auto fused_n = [&] { hipLaunchKernelGGL((k_chain<0, 128, 0, 0>), dim3(blocks), dim3(512), kChainLds, 0, cn); };
      single_n(); fused_n();
      CK(hipDeviceSynchronize());
      report("l1ng -> l2ng", fetch(out_a, (size_t)N * 256), fetch(out_b, (size_t)N * 256), 256, 128);
      printf("    l1ng -> l2ng: single products back to back %.1f us | fused %.1f us\n", time_us(single_n), time_us(fused_n));
    }
  }
  // ---- (2b) linear3 -> ReLU -> linear4 -> contraction: fused vs the two launches; what the parts of the fused phase cost -----------
  {
    constexpr auto kEquivG = k_gemm3<EPI_ROWDIV | EPI_EQUIV, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
    constexpr int kLdsEq = TileCfg3<4, 2, 1, 2, 16, 3>::kLdsBytes > 128 * 100 * 4 ? TileCfg3<4, 2, 1, 2, 16, 3>::kLdsBytes : 128 * 100 * 4;   // + the epilogue's z rows
    raise(kEquivG, kLdsEq);
    Host W4 = dev(1024 * 256, 0.2f), b4 = dev(1024, 1.0f), zq = dev((size_t)N * 96, 2.0f);
    Wt W4w = words(W4.d, 1024, 256);
    float *t_a, *t_b;
    CK(hipMalloc(&t_a, (size_t)N * 96 * 4)); CK(hipMalloc(&t_b, (size_t)N * 96 * 4));
    GemmArgs g1 = gemm(cat.d, 256, W1w, 256, b1.d, h256, 256, N, 256, 256, EPI_RELU);
    GemmArgs g4 = gemm(h256, 256, W4w, 256, b4.d, nullptr, 0, N, 1024, 256, EPI_ROWDIV | EPI_EQUIV, fn.d);
    g4.zq = zq.d; g4.tout = t_a;
    ChainArgs c{};
    c.A = cat.d; c.lda = 256; c.W1 = W1w.w; c.ldw1 = 256; c.b1 = b1.d; c.W2 = W4w.w; c.ldw2 = 256; c.b2 = b4.d; c.M = N; c.K1 = 256;
    c.rowdiv = fn.d; c.zq = zq.d; c.tout = t_b; c.ws1 = W1w.sc; c.ws2 = W4w.sc;
    auto single = [&] {
      hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, g1);
      hipLaunchKernelGGL(kEquivG, dim3(tiles * 8), dim3(512), kLdsEq, 0, g4);
    };
#define EQ(D) k_chain<0, 256, EPI_ROWDIV | EPI_EQUIV, 0, D>
    raise(EQ(0), kChainEqLds); raise(EQ(1), kChainEqLds); raise(EQ(2), kChainEqLds); raise(EQ(4), kChainEqLds); raise(EQ(7), kChainEqLds);
    single();
    hipLaunchKernelGGL((EQ(0)), dim3(blocks), dim3(512), kChainEqLds, 0, c);
    CK(hipDeviceSynchronize());
    report("l3 -> l4 -> contraction", fetch(t_a, (size_t)N * 96), fetch(t_b, (size_t)N * 96), 96, 96);
    printf("  l3 -> l4 -> contraction: single products back to back %.1f us | fused %.1f us | without sub-slice hand-over %.1f | without the contraction %.1f | without W2 staging %.1f | none of the three %.1f\n",
           time_us(single), time_us([&] { hipLaunchKernelGGL((EQ(0)), dim3(blocks), dim3(512), kChainEqLds, 0, c); }),
           time_us([&] { hipLaunchKernelGGL((EQ(1)), dim3(blocks), dim3(512), kChainEqLds, 0, c); }),
           time_us([&] { hipLaunchKernelGGL((EQ(2)), dim3(blocks), dim3(512), kChainEqLds, 0, c); }),
           time_us([&] { hipLaunchKernelGGL((EQ(4)), dim3(blocks), dim3(512), kChainEqLds, 0, c); }),
           time_us([&] { hipLaunchKernelGGL((EQ(7)), dim3(blocks), dim3(512), kChainEqLds, 0, c); }));
  }
  // ---- (3) the wide single products (reference timings for the forward's budget) -------------------------------------------------
  // ... and (round 6) the same products with the ACTIVATION operand arriving as row-scaled words too, as if its
  // producer had emitted the split form: no estimate, no vote, no split in the consumer's k-loop (k_gemm3 WORDS = 3, lab only).
  // The words are made here by k_encode_rows (exact row maxima); the f32 form scales by its sampled estimate: same value to rounding.
  {
    constexpr auto kWordsRowdiv = k_gemm3<EPI_ROWDIV, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 3>;
    constexpr auto kWordsRelu = k_gemm3<EPI_RELU, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 3>;
    raise(kWordsRowdiv, kLds128); raise(kWordsRelu, kLds128);
    Host Wq = dev(1024 * 256, 0.2f), bq = dev(1024, 1.0f);
    Wt Wqw = words(Wq.d, 1024, 256);
    Wt catw = words(cat.d, N, 256);
    float *big, *big2; CK(hipMalloc(&big, (size_t)N * 1024 * 4)); CK(hipMalloc(&big2, (size_t)N * 1024 * 4));
    for (int Nx : {1024, 768, 256}) {
      GemmArgs q = gemm(cat.d, 256, Wqw, 256, bq.d, big, Nx, N, Nx, 256, EPI_ROWDIV, fn.d);
      GemmArgs q2 = q; q2.flags = EPI_RELU;
      GemmArgs qw = q; qw.A = reinterpret_cast<const float*>(catw.w); qw.ascale = catw.sc; qw.C = big2;
      GemmArgs qw2 = qw; qw2.flags = EPI_RELU;
      const dim3 grid(tiles * (Nx / 128));
      hipLaunchKernelGGL(kSplitRowdiv, grid, dim3(512), kLds128, 0, q);
      hipLaunchKernelGGL(kWordsRowdiv, grid, dim3(512), kLds128, 0, qw);
      CK(hipDeviceSynchronize());
      report("A as words vs A as f32", fetch(big, (size_t)N * Nx), fetch(big2, (size_t)N * Nx), Nx, Nx);
      // alternating, seven rounds of 40 launches each: median and best of each form (the first window after a switch of kernels runs
      // slow on this part -- single windows of 20 launches spread over 15 %)
      std::vector<float> tf[2], tw[2];
      for (int r = 0; r < 7; r++) {
        tf[0].push_back(time_us([&] { hipLaunchKernelGGL(kSplitRowdiv, grid, dim3(512), kLds128, 0, q); }, 40));
        tw[0].push_back(time_us([&] { hipLaunchKernelGGL(kWordsRowdiv, grid, dim3(512), kLds128, 0, qw); }, 40));
        tf[1].push_back(time_us([&] { hipLaunchKernelGGL(kSplitRelu, grid, dim3(512), kLds128, 0, q2); }, 40));
        tw[1].push_back(time_us([&] { hipLaunchKernelGGL(kWordsRelu, grid, dim3(512), kLds128, 0, qw2); }, 40));
      }
      auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      auto best = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[0]; };
      printf("  product M %d N %4d K 256: row division median %.1f us, best %.1f (A as words %.1f, %.1f) | ReLU %.1f, %.1f (A as words %.1f, %.1f)\n",
             N, Nx, med(tf[0]), best(tf[0]), med(tw[0]), best(tw[0]), med(tf[1]), best(tf[1]), med(tw[1]), best(tw[1]));
    }
    const float te = time_us([&] { hipLaunchKernelGGL(k_encode_rows, dim3((N + 3) / 4), dim3(256), 0, 0, cat.d, catw.w, catw.sc, g_lab_mat, 1, N); });
    printf("  (a separate pass that turns a [%d, 256] f32 operand into words + row scales: %.1f us)\n", N, te);
  }
  return 0;
}
