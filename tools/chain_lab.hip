// tools/chain_lab.hip -- diagnostic only: the fused back-to-back products of chain_f16.h against the single products they
// replace (gemm_f32.h), on the shapes of one SET forward: time of each form and the largest difference between their outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/chain_lab.exe tools/chain_lab.hip && tools/chain_lab.exe [nodes]
#include "../sgrl_amd/csrc/chain_f16.h"

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

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 35840;
  const int N3 = 3 * N;
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  auto dev = [&](size_t n, float scale, float offset = 0.f) {
    std::vector<float> h(n);
    for (auto& v : h) v = rnd() * scale + offset;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
  };
  auto words = [&](const float* w, size_t n) {
    unsigned* d; CK(hipMalloc(&d, n * 4));
    hipLaunchKernelGGL(k_encode_words, dim3(256), dim3(256), 0, 0, w, d, (long long)n, (unsigned*)nullptr);
    return d;
  };
  raise(kSplitPlain, kLds128); raise(kSplitRelu, kLds128); raise(kSplitLn, kLds128); raise(kGram, kLds128); raise(kProj, kLdsProj);
  raise(k_chain<0, 256, EPI_ROWDIV | EPI_LN, 0>, kChainLds); raise(k_chain<0, 256, 0, 0>, kChainLds); raise(k_chain<1, 256, 0, 0>, kChainLds);
  raise(k_chain<1, 256, 0, 1>, kChainLds); raise(k_chain<1, 256, 0, 2>, kChainLds); raise(k_chain<0, 128, 0, 0>, kChainLds);
  raise(k_chain<1, 128, 0, 2>, kChainLds);
  unsigned* ev; CK(hipMalloc(&ev, 4)); CK(hipMemset(ev, 0, 4));
  const int tiles = (N + 127) / 128, blocks = (N + kChainRows - 1) / kChainRows;

  float* cat = dev((size_t)N * 256, 2.0f);               // [inv | ng]
  float* fn = dev((size_t)N, 1.0f, 2.0f);
  float* W1 = dev(256 * 256, 0.2f); float* b1 = dev(256, 1.0f);
  float* W2 = dev(128 * 256, 0.2f); float* b2 = dev(128, 1.0f);
  float* lnw = dev(128, 1.0f, 1.0f); float* lnb = dev(128, 1.0f);
  unsigned* W1w = words(W1, 256 * 256); unsigned* W2w = words(W2, 128 * 256);
  float *h256, *out_a, *out_b, *ng_a, *ng_b;
  CK(hipMalloc(&h256, (size_t)N * 256 * 4)); CK(hipMalloc(&out_a, (size_t)N * 256 * 4)); CK(hipMalloc(&out_b, (size_t)N * 256 * 4));
  CK(hipMalloc(&ng_a, (size_t)N * 256 * 4)); CK(hipMalloc(&ng_b, (size_t)N * 256 * 4));
  CK(hipMemset(out_a, 0, (size_t)N * 256 * 4)); CK(hipMemset(out_b, 0, (size_t)N * 256 * 4));

  printf("chain lab: %d nodes (%d row tiles of 128, %d workgroups of %d rows)\n", N, tiles, blocks, kChainRows);
  // ---- (1) linear1 -> ReLU -> linear2 -> / fn -> residual + LayerNorm (in place on ng = cat[:, 128:]) ----------------------
  {
    GemmArgs g1{cat, 256, reinterpret_cast<const float*>(W1w), 256, b1, h256, 256, N, 256, 256, EPI_RELU, nullptr, nullptr, 0};
    g1.range_events = ev;
    GemmArgs g2{h256, 256, reinterpret_cast<const float*>(W2w), 256, b2, nullptr, 0, N, 128, 256, EPI_ROWDIV | EPI_LN, fn, nullptr, 0};
    g2.ln_io = ng_a + 128; g2.ln_ld = 256; g2.ln_w = lnw; g2.ln_b = lnb; g2.range_events = ev;
    ChainArgs c{};
    c.A = cat; c.lda = 256; c.W1 = W1w; c.ldw1 = 256; c.b1 = b1; c.W2 = W2w; c.ldw2 = 256; c.b2 = b2; c.M = N; c.K1 = 256;
    c.rowdiv = fn; c.ln_io = ng_b + 128; c.ln_ld = 256; c.ln_w = lnw; c.ln_b = lnb; c.range_events = ev;
    CK(hipMemcpy(ng_a, cat, (size_t)N * 256 * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(ng_b, cat, (size_t)N * 256 * 4, hipMemcpyDeviceToDevice));
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
    // plain second epilogue (the same pair as linear3-shaped products would use it)
    GemmArgs g2p{h256, 256, reinterpret_cast<const float*>(W2w), 256, b2, out_a, 256, N, 128, 256, 0, nullptr, nullptr, 0};
    g2p.range_events = ev;
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
    float* g = dev((size_t)N3 * 128, 2.0f);
    float* Wp = dev(64 * 128, 0.3f);
    CK(hipMemset(Wp + 30 * 128, 0, 2 * 128 * 4)); CK(hipMemset(Wp + 62 * 128, 0, 2 * 128 * 4));
    unsigned* Wpw = words(Wp, 64 * 128);
    float* Wg = dev(256 * 576, 0.05f); float* bg = dev(256, 1.0f);
    unsigned* Wgw = words(Wg, 256 * 576);
    float* zc_a = dev((size_t)N3 * 32, 1.0f);
    float *zc_b, *z2_a, *z2_b, *fn_a, *fn_b;
    CK(hipMalloc(&zc_b, (size_t)N3 * 32 * 4)); CK(hipMalloc(&z2_a, (size_t)N3 * 32 * 4)); CK(hipMalloc(&z2_b, (size_t)N3 * 32 * 4));
    CK(hipMalloc(&fn_a, (size_t)N * 4)); CK(hipMalloc(&fn_b, (size_t)N * 4));
    CK(hipMemcpy(zc_b, zc_a, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(z2_a, zc_a, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(z2_b, zc_a, (size_t)N3 * 32 * 4, hipMemcpyDeviceToDevice));
    for (int nz = 1; nz <= 2; nz++) {
      GemmArgs p{g, 128, reinterpret_cast<const float*>(Wpw), 128, nullptr, zc_a, 32, N3, nz == 2 ? 64 : 32, 128, EPI_ZSPLIT, nullptr, nz == 2 ? z2_a : nullptr, 32};
      p.range_events = ev;
      GemmArgs gg{zc_a, 96, reinterpret_cast<const float*>(Wgw), 576, bg, h256, 256, N, 256, 576, EPI_RELU, nullptr, nullptr, 0};
      gg.rowdiv_out = fn_a; gg.range_events = ev;
      GemmArgs g2p{h256, 256, reinterpret_cast<const float*>(W2w), 256, b2, out_a, 256, N, 128, 256, 0, nullptr, nullptr, 0};
      g2p.range_events = ev;
      ChainArgs c{};
      c.A = zc_b; c.W1 = Wgw; c.ldw1 = 576; c.b1 = bg; c.W2 = W2w; c.ldw2 = 256; c.b2 = b2; c.C = out_b; c.ldc = 256; c.M = N; c.K1 = 576;
      c.fn_out = fn_b; c.X = g; c.ldx = 128; c.Kp = 128; c.Wp = Wpw; c.zc = zc_b; c.z2 = nz == 2 ? z2_b : nullptr; c.range_events = ev;
      const int pgrid = ((N3 + 127) / 128) * (nz == 2 ? 1 : 1);
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
      ChainArgs c0 = c; c0.A = zc_a;
      const float tf0 = time_us([&] { hipLaunchKernelGGL((k_chain<1, 256, 0, 0>), dim3(blocks), dim3(512), kChainLds, 0, c0); });
      printf("    single products proj %.1f + lg1 %.1f + lg2 %.1f us (back to back %.1f) | fused site %.1f us | fused lg1 -> lg2 without the projection %.1f us\n",
             t1, t2, t3, ts, tf, tf0);
    }
    // the head's site: K = 144 projections, hidden width 128
    {
      float* og = dev((size_t)N3 * 144, 2.0f);
      float* Wp2 = dev(64 * 144, 0.3f);
      unsigned* Wp2w = words(Wp2, 64 * 144);
      float* Wh = dev(128 * 576, 0.05f); unsigned* Whw = words(Wh, 128 * 576);
      float* W2h = dev(128 * 128, 0.2f); unsigned* W2hw = words(W2h, 128 * 128);
      GemmArgs p{og, 144, reinterpret_cast<const float*>(Wp2w), 144, nullptr, zc_a, 32, N3, 64, 144, EPI_ZSPLIT, nullptr, z2_a, 32};
      p.range_events = ev;
      GemmArgs gg{zc_a, 96, reinterpret_cast<const float*>(Whw), 576, bg, h256, 128, N, 128, 576, EPI_RELU, nullptr, nullptr, 0};
      gg.rowdiv_out = fn_a; gg.range_events = ev;
      GemmArgs g2p{h256, 128, reinterpret_cast<const float*>(W2hw), 128, b2, out_a, 256, N, 128, 128, 0, nullptr, nullptr, 0};
      g2p.range_events = ev;
      ChainArgs c{};
      c.A = zc_b; c.W1 = Whw; c.ldw1 = 576; c.b1 = bg; c.W2 = W2hw; c.ldw2 = 128; c.b2 = b2; c.C = out_b; c.ldc = 256; c.M = N; c.K1 = 576;
      c.fn_out = fn_b; c.X = og; c.ldx = 144; c.Kp = 144; c.Wp = Wp2w; c.zc = zc_b; c.z2 = z2_b; c.range_events = ev;
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
      // linear1_ng -> linear2_ng: K 160, hidden 128
      float* ong = dev((size_t)N * 160, 2.0f);
      float* Wn = dev(128 * 160, 0.2f); unsigned* Wnw = words(Wn, 128 * 160);
      GemmArgs n1{ong, 160, reinterpret_cast<const float*>(Wnw), 160, b1, h256, 128, N, 128, 160, EPI_RELU, nullptr, nullptr, 0};
      n1.range_events = ev;
      ChainArgs cn{};
      cn.A = ong; cn.lda = 160; cn.W1 = Wnw; cn.ldw1 = 160; cn.b1 = b1; cn.W2 = W2hw; cn.ldw2 = 128; cn.b2 = b2; cn.C = out_b; cn.ldc = 256; cn.M = N; cn.K1 = 160;
      cn.range_events = ev;
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
  // ---- (3) row-wise stores of transposed tiles (EPI_TR) on the wide plain products ------------------------------------------
  {
    constexpr auto kRowdiv = k_gemm3<EPI_ROWDIV, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
    constexpr auto kRowdivT = k_gemm3<EPI_ROWDIV | EPI_TR, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
    constexpr auto kPlainT = k_gemm3<EPI_TR, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
    constexpr auto kReluT = k_gemm3<EPI_RELU | EPI_TR, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
    raise(kRowdiv, kLds128); raise(kRowdivT, kLds128); raise(kPlainT, kLds128); raise(kReluT, kLds128);
    float* Wq = dev(768 * 256, 0.2f); unsigned* Wqw = words(Wq, 768 * 256);
    float* bq = dev(768, 1.0f);
    float *q_a, *q_b;
    CK(hipMalloc(&q_a, (size_t)N3 * 256 * 4)); CK(hipMalloc(&q_b, (size_t)N3 * 256 * 4));
    GemmArgs q{cat, 256, reinterpret_cast<const float*>(Wqw), 256, bq, q_a, 768, N, 768, 256, EPI_ROWDIV, fn, nullptr, 0};
    q.range_events = ev;
    GemmArgs qt = q; qt.C = q_b;
    hipLaunchKernelGGL(kRowdiv, dim3(tiles * 6), dim3(512), kLds128, 0, q);
    hipLaunchKernelGGL(kRowdivT, dim3(tiles * 6), dim3(512), kLds128, 0, qt);
    CK(hipDeviceSynchronize());
    report("qkv: row-wise stores", fetch(q_a, (size_t)N * 768), fetch(q_b, (size_t)N * 768), 768, 768);
    printf("  qkv (N 768, K 256, / fn): column-wise stores %.1f us | row-wise stores of transposed tiles %.1f us\n",
           time_us([&] { hipLaunchKernelGGL(kRowdiv, dim3(tiles * 6), dim3(512), kLds128, 0, q); }),
           time_us([&] { hipLaunchKernelGGL(kRowdivT, dim3(tiles * 6), dim3(512), kLds128, 0, qt); }));
    float* g = dev((size_t)N3 * 128, 2.0f);
    float* Wu = dev(256 * 128, 0.2f); unsigned* Wuw = words(Wu, 256 * 128);
    const int tiles3 = (N3 + 127) / 128;
    GemmArgs u{g, 128, reinterpret_cast<const float*>(Wuw), 128, nullptr, q_a, 256, N3, 256, 128, 0, nullptr, nullptr, 0};
    u.range_events = ev;
    GemmArgs ut = u; ut.C = q_b;
    hipLaunchKernelGGL(kSplitPlain, dim3(tiles3 * 2), dim3(512), kLds128, 0, u);
    hipLaunchKernelGGL(kPlainT, dim3(tiles3 * 2), dim3(512), kLds128, 0, ut);
    CK(hipDeviceSynchronize());
    report("U: row-wise stores", fetch(q_a, (size_t)N3 * 256), fetch(q_b, (size_t)N3 * 256), 256, 256);
    printf("  U (3 x nodes, N 256, K 128): column-wise stores %.1f us | row-wise stores %.1f us\n",
           time_us([&] { hipLaunchKernelGGL(kSplitPlain, dim3(tiles3 * 2), dim3(512), kLds128, 0, u); }),
           time_us([&] { hipLaunchKernelGGL(kPlainT, dim3(tiles3 * 2), dim3(512), kLds128, 0, ut); }));
    GemmArgs l3{cat, 256, reinterpret_cast<const float*>(W1w), 256, b1, q_a, 256, N, 256, 256, EPI_RELU, nullptr, nullptr, 0};
    l3.range_events = ev;
    GemmArgs l3t = l3; l3t.C = q_b;
    hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, l3);
    hipLaunchKernelGGL(kReluT, dim3(tiles * 2), dim3(512), kLds128, 0, l3t);
    CK(hipDeviceSynchronize());
    report("l3: row-wise stores", fetch(q_a, (size_t)N * 256), fetch(q_b, (size_t)N * 256), 256, 256);
    printf("  l3 (N 256, K 256, ReLU): column-wise stores %.1f us | row-wise stores %.1f us\n",
           time_us([&] { hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, l3); }),
           time_us([&] { hipLaunchKernelGGL(kReluT, dim3(tiles * 2), dim3(512), kLds128, 0, l3t); }));
  }
  unsigned hev = 0; CK(hipMemcpy(&hev, ev, 4, hipMemcpyDeviceToHost));
  printf("range events: %u\n", hev);
  return 0;
}
