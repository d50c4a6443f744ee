// tools/gemm_lab.hip -- diagnostic only: the SET actor's GEMM kernels (gemm_f32.h) on the shapes of one forward:
// exact-f32 MFMA kernel vs the split-precision bf16x6 kernel in several tile configurations, with the error of each against
// a float64 host reference on sampled outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_lab tools/gemm_lab.hip && /tmp/gemm_lab
#include "../sgrl_amd/csrc/gemm_f32.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace sgrl_gemm;

template <class K>
static float timeit(K k, int tiles, int threads, int lds, const GemmArgs& a, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3(tiles), dim3(threads), lds, 0, a);
  hipEventRecord(t0, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(tiles), dim3(threads), lds, 0, a);
  hipEventRecord(t1, 0);
  hipEventSynchronize(t1);
  float ms = 0;
  hipEventElapsedTime(&ms, t0, t1);
  if (hipGetLastError() != hipSuccess) printf("  launch error!\n");
  return ms / reps;
}
template <int F, int WM, int WN, int TM, int TN, int BKT, int PF>
static float run2(const GemmArgs& a, int reps) {
  using Cfg = TileCfg<WM, WN, TM, TN, BKT>;
  const int tiles = ((a.M + Cfg::kBM - 1) / Cfg::kBM) * ((a.N + Cfg::kBN - 1) / Cfg::kBN);
  return timeit(k_gemm2<F, WM, WN, TM, TN, BKT, PF>, tiles, Cfg::kThreads, Cfg::kLdsBytes, a, reps);
}
template <int F, int WM, int WN, int TM, int TN, int BKT = 32, int PF = 1, bool PLA = false, bool PLW = false, bool LATE = false, int ABL = 0, int NPL = 3, bool SKEW = false, int WORDS = 0>
static float run3(const GemmArgs& a, int reps) {
  using Cfg = TileCfg3<WM, WN, TM, TN, BKT, NPL>;
  const int tiles = ((a.M + Cfg::kBM - 1) / Cfg::kBM) * ((a.N + Cfg::kBN - 1) / Cfg::kBN);
  return timeit(k_gemm3<F, WM, WN, TM, TN, BKT, PF, PLA, PLW, LATE, ABL, false, NPL, SKEW, WORDS>, tiles, Cfg::kThreads, Cfg::kLdsBytes, a, reps);
}

template <int F, int BKT, int PF>
static float run4(const GemmArgs& a, int reps) {
  using Cfg = TileCfg3<4, 2, 1, 2, BKT>;
  const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
  return timeit(k_gemm4<F, BKT, PF>, tiles, 1024, Cfg::kLdsBytes, a, reps);
}

int main(int argc, char** argv) {
  const int N0 = argc > 1 ? atoi(argv[1]) : 35840;
  struct Shape { const char* name; int M, N, K, flags; };
  const Shape shapes[] = {
      {"qkv    (rowdiv)", N0, 768, 256, EPI_ROWDIV}, {"l4     (rowdiv)", N0, 1024, 256, EPI_ROWDIV},
      {"lg1    (relu)  ", N0, 256, 544, EPI_RELU},   {"l3     (relu)  ", N0, 256, 256, EPI_RELU},
      {"vg     (plain) ", 3 * N0, 256, 128, 0},      {"gout   (acc2)  ", 3 * N0, 128, 256, EPI_ACC2},
      {"lg2    (plain) ", N0, 128, 256, 0},          {"proj64 (plain) ", 3 * N0, 64, 128, 0},
  };
  size_t maxA = (size_t)3 * N0 * 544, maxC = (size_t)3 * N0 * 1024;
  float *A, *W, *C, *C2, *bias, *rd;
  hipMalloc(&A, maxA * 4); hipMalloc(&W, 1024 * 544 * 4); hipMalloc(&C, maxC * 4); hipMalloc(&C2, maxC * 4);
  hipMalloc(&bias, 1024 * 4); hipMalloc(&rd, (size_t)3 * N0 * 4);
  std::vector<float> hA(maxA), hW(1024 * 544), hb(1024), hrd((size_t)3 * N0);
  {
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    // wide dynamic range: magnitudes over ~6 decades, as Gram entries / activations have
    for (auto& v : hA) { const float m = rnd(); v = m * std::pow(10.0f, 6.0f * rnd()); }
    for (auto& v : hW) v = rnd() * 0.2f;
    for (auto& v : hb) v = rnd();
    for (auto& v : hrd) v = 1.5f + rnd();
    hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(rd, hrd.data(), hrd.size() * 4, hipMemcpyHostToDevice);
  }
  hipMemset(C2, 0, maxC * 4);
  unsigned short *Apl, *Wpl;
  hipMalloc(&Apl, maxA * 2 * 3); hipMalloc(&Wpl, (size_t)1024 * 544 * 2 * 3);
  const int reps = 20;
  if (argc > 2 && argv[2][0] == 'p') {   // pitch test: linear4 / l3 / qkv shapes with the A rows 1024 B apart vs padded pitches
    for (int si : {1, 3, 0}) {
      const Shape& sh = shapes[si];
      for (int lda : {256, 264, 272, 288, 320}) {
        GemmArgs a{A, lda, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N};
        const float ms = si == 3 ? run3<EPI_RELU, 4, 2, 1, 2, 16, 2>(a, 20) : run3<EPI_ROWDIV, 4, 2, 1, 2, 16, 2>(a, 20);
        printf("%s lda %3d (pitch %4d B): %.1f us\n", sh.name, lda, lda * 4, ms * 1e3);
      }
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'e') {   // linear4 with the equivariant epilogue vs the plain store
    const Shape& sh = shapes[1];
    float *zq, *tout;
    hipMalloc(&zq, (size_t)sh.M * 96 * 4); hipMalloc(&tout, (size_t)sh.M * 96 * 4);
    hipMemcpy(zq, A, (size_t)sh.M * 96 * 4, hipMemcpyDeviceToDevice);
    GemmArgs a{A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N};
    a.zq = zq; a.tout = tout;
    const float t0 = run3<EPI_ROWDIV, 4, 2, 1, 2, 16, 2>(a, 20);
    const float t1 = run3<EPI_ROWDIV | EPI_EQUIV, 4, 2, 1, 2, 16, 2>(a, 20);
    const float t2 = run3<EPI_ROWDIV | EPI_EQUIV, 4, 2, 1, 2, 16, 2, false, false, false, 1>(a, 20);
    printf("l4 plain store %.1f us | equivariant epilogue %.1f us | same without staging in the loop %.1f us\n", t0 * 1e3, t1 * 1e3, t2 * 1e3);
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'a') {   // ablations of the 8-wave split kernel on the linear4 / lg1 shapes
    for (int si : {1, 2}) {
      const Shape& sh = shapes[si];
      GemmArgs a{A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N};
      const float t0 = run3<0, 4, 2, 1, 2, 16, 2>(a, 20);
      const float t1 = run3<0, 4, 2, 1, 2, 16, 2, false, false, false, 1>(a, 20);
      const float t2 = run3<0, 4, 2, 1, 2, 16, 2, false, false, false, 2>(a, 20);
      const float t3 = run3<0, 4, 2, 1, 2, 16, 2, false, false, false, 3>(a, 20);
      printf("%s full %.1f us | MFMA + operand reads + barrier only %.1f | staging only (load, split, LDS store, barrier) %.1f | full without split arithmetic %.1f\n",
             sh.name, t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3);
      const float h0 = run3<0, 4, 2, 1, 2, 16, 2, false, false, false, 0, 2>(a, 20);
      const float h1 = run3<0, 4, 2, 1, 2, 16, 2, false, false, false, 1, 2>(a, 20);
      const float h2 = run3<0, 4, 2, 1, 2, 16, 2, false, false, false, 2, 2>(a, 20);
      printf("%s f16x3: full %.1f us | MFMA + operand reads + barrier only %.1f | staging only %.1f\n", sh.name, h0 * 1e3, h1 * 1e3, h2 * 1e3);
      const float g0 = run3<0, 4, 2, 1, 2, 32, 1, false, false, false, 0, 2>(a, 20);
      const float g1 = run3<0, 4, 2, 1, 2, 32, 1, false, false, false, 1, 2>(a, 20);
      const float g2 = run3<0, 4, 2, 1, 2, 32, 1, false, false, false, 2, 2>(a, 20);
      printf("%s f16x3 bk32 pf1: full %.1f us | MFMA + operand reads + barrier only %.1f | staging only %.1f\n", sh.name, g0 * 1e3, g1 * 1e3, g2 * 1e3);
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 's') {   // phase offset between the two blocks of a CU
    for (int si : {0, 1, 3}) {
      const Shape& sh = shapes[si];
      GemmArgs a{A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N};
      printf("%s f16x3 bk16 pf2, phase sleep (x64 clocks):", sh.name);
      for (int ps : {0, 4, 8, 12, 16, 24, 32, 65536 + 4, 65536 + 8, 65536 + 12, 65536 + 16, 65536 + 24, 65536 + 32}) {
        a.phase_sleep = ps;
        const float ms = si == 3 ? run3<EPI_RELU, 4, 2, 1, 2, 16, 2, false, false, false, 0, 2>(a, 20) : run3<EPI_ROWDIV, 4, 2, 1, 2, 16, 2, false, false, false, 0, 2>(a, 20);
        printf(" %d:%.1f", ps, ms * 1e3);
      }
      printf(" us\n");
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'h') {   // two-piece f16 x 3 form against the bf16 x 6 form and the exact-f32 kernel: time and error vs float64
    unsigned* ev; hipMalloc(&ev, 4); hipMemset(ev, 0, 4);
    for (const Shape& sh : shapes) {
      const double gf = 2.0 * sh.M * sh.N * sh.K;
      GemmArgs a{A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N};
      a.range_events = ev;
      auto err = [&]() {
        std::vector<float> h((size_t)sh.M * sh.N);
        hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, sum = 0;
        unsigned s = 777;
        for (int q = 0; q < 4000; q++) {
          s = s * 1664525u + 1013904223u; const int m = (s >> 4) % sh.M;
          s = s * 1664525u + 1013904223u; const int n = (s >> 4) % sh.N;
          double ref = hb[n], mag = fabs((double)hb[n]);
          for (int k = 0; k < sh.K; k++) { const double p = (double)hA[(size_t)m * sh.K + k] * hW[(size_t)n * sh.K + k]; ref += p; mag += fabs(p); }
          if (sh.flags == EPI_RELU) ref = ref > 0 ? ref : 0;
          if (sh.flags == EPI_ROWDIV) { ref *= (double)(1.0f / hrd[m]); mag *= (double)(1.0f / hrd[m]); }
          const double e = fabs((double)h[(size_t)m * sh.N + n] - ref) / mag;
          if (e > worst) worst = e;
          sum += e;
        }
        printf(" err max %.1e mean %.1e", worst, sum / 4000);
      };
      printf("%s M %6d N %4d K %3d\n", sh.name, sh.M, sh.N, sh.K);
      hipLaunchKernelGGL(k_encode_words, dim3(2048), dim3(256), 0, 0, A, reinterpret_cast<unsigned*>(Apl), (long long)sh.M * sh.K, ev);
      hipLaunchKernelGGL(k_encode_words, dim3(512), dim3(256), 0, 0, W, reinterpret_cast<unsigned*>(Wpl), (long long)sh.N * sh.K, ev);
      GemmArgs aww = a; aww.W = reinterpret_cast<const float*>(Wpl);
      GemmArgs aaw = aww; aaw.A = reinterpret_cast<const float*>(Apl);
#define RUNH(F)                                                                                                              \
      { float ms;                                                                                                            \
        ms = run2<F, 4, 2, 1, 2, 32, 2>(a, reps); printf("   exact f32 mfma    %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 4, 2, 1, 2, 16, 2>(a, reps); printf("   bf16x6 bk16 pf2   %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 4, 2, 1, 2, 16, 2, false, false, false, 0, 2>(a, reps); printf("   f16x3  bk16 pf2   %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 4, 2, 1, 2, 16, 1, false, false, false, 0, 2>(a, reps); printf("   f16x3  bk16 pf1   %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 4, 2, 1, 2, 32, 1, false, false, false, 0, 2>(a, reps); printf("   f16x3  bk32 pf1   %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 4, 2, 1, 2, 32, 2, false, false, false, 0, 2>(a, reps); printf("   f16x3  bk32 pf2   %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 4, 2, 1, 2, 16, 2, false, false, false, 0, 2, true, 2>(aww, reps); printf("   f16x3  W words SKEW 128x128/8w   %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 2, 2, 1, 2, 16, 2, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words  64x128/4w        %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 2, 2, 1, 2, 32, 1, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words  64x128/4w bk32pf1 %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 2, 2, 1, 2, 32, 2, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words  64x128/4w bk32pf2 %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 4, 1, 1, 2, 16, 2, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words 128x64/4w         %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 2, 4, 1, 1, 16, 2, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words  64x128/8w(32x32) %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); }
      switch (sh.flags) {
        case 0: RUNH(0); break;
        case EPI_RELU: RUNH(EPI_RELU); break;
        case EPI_ROWDIV: RUNH(EPI_ROWDIV); break;
        case EPI_ACC2: RUNH(EPI_ACC2); break;
      }
    }
    unsigned hev = 0; hipMemcpy(&hev, ev, 4, hipMemcpyDeviceToHost);
    printf("range events (threads that clamped): %u\n", hev);
    return 0;
  }
  if (argc > 2) {   // profiling mode: one configuration on the linear4 shape, a handful of launches
    const Shape& sh = shapes[1];
    GemmArgs a{A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N};
    hipLaunchKernelGGL(k_encode_words, dim3(512), dim3(256), 0, 0, W, reinterpret_cast<unsigned*>(Wpl), (long long)sh.N * sh.K, (unsigned*)nullptr);
    GemmArgs aww = a; aww.W = reinterpret_cast<const float*>(Wpl);
    const float ms = run3<EPI_ROWDIV, 4, 2, 1, 2, 16, 2, false, false, false, 0, 2, true, 2>(aww, 5);
    printf("profile mode: l4 f16x3 W words SKEW bk16 pf2: %.1f us\n", ms * 1e3);
    return 0;
  }
  for (const Shape& sh : shapes) {
    const double gf = 2.0 * sh.M * sh.N * sh.K;
    printf("%s M %6d N %4d K %3d :", sh.name, sh.M, sh.N, sh.K);
    GemmArgs a{A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N};
    // pre-split copies of both operands
    const long long aplane = (long long)sh.M * sh.K, wplane = (long long)sh.N * sh.K;
    hipLaunchKernelGGL(k_split_planes, dim3(2048), dim3(256), 0, 0, A, sh.K, sh.M, sh.K, Apl, sh.K, aplane);
    hipLaunchKernelGGL(k_split_planes, dim3(512), dim3(256), 0, 0, W, sh.K, sh.N, sh.K, Wpl, sh.K, wplane);
    GemmArgs aw = a; aw.W = reinterpret_cast<const float*>(Wpl); aw.w_plane = wplane;
    GemmArgs aaw = aw; aaw.A = reinterpret_cast<const float*>(Apl); aaw.a_plane = aplane;
    // relative error (max |got - ref| / (sum_k |a_k w_k| + |b|)) against float64 on sampled outputs, plain GEMM value
    auto err = [&]() {
      std::vector<float> h((size_t)sh.M * sh.N);
      hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
      double worst = 0;
      unsigned s = 777;
      for (int q = 0; q < 4000; q++) {
        s = s * 1664525u + 1013904223u; const int m = (s >> 4) % sh.M;
        s = s * 1664525u + 1013904223u; const int n = (s >> 4) % sh.N;
        double ref = hb[n], mag = fabs((double)hb[n]);
        for (int k = 0; k < sh.K; k++) { const double p = (double)hA[(size_t)m * sh.K + k] * hW[(size_t)n * sh.K + k]; ref += p; mag += fabs(p); }
        if (sh.flags == EPI_RELU) ref = ref > 0 ? ref : 0;
        if (sh.flags == EPI_ROWDIV) { ref *= (double)(1.0f / hrd[m]); mag *= (double)(1.0f / hrd[m]); }
        const double e = fabs((double)h[(size_t)m * sh.N + n] - ref) / mag;
        if (e > worst) worst = e;
      }
      return worst;
    };
    float m[8];
    double e[8];
#define RUNALL(F)                                                                                            \
    m[0] = run3<F, 4, 2, 1, 2, 16, 2>(a, reps); e[0] = err(); m[1] = run4<F, 16, 3>(a, reps); e[1] = err();  \
    m[2] = run4<F, 16, 2>(a, reps); e[2] = err(); m[3] = run4<F, 16, 4>(a, reps); e[3] = err();  \
    m[4] = run4<F, 32, 2>(a, reps); e[4] = err(); m[5] = run4<F, 32, 3>(a, reps); e[5] = err();  \
    m[6] = run4<F, 16, 1>(a, reps); e[6] = err(); m[7] = run4<F, 32, 1>(a, reps); e[7] = err()
    switch (sh.flags) {
      case 0: RUNALL(0); break;
      case EPI_RELU: RUNALL(EPI_RELU); break;
      case EPI_ROWDIV: RUNALL(EPI_ROWDIV); break;
      case EPI_ACC2: RUNALL(EPI_ACC2); break;
    }
    const char* nm[8] = {"x6 8w bk16pf2", "ws bk16pf3", "ws bk16pf2", "ws bk16pf4", "ws bk32pf2", "ws bk32pf3", "ws bk16pf1", "ws bk32pf1"};
    for (int i = 0; i < 8; i++) printf(" | %s %6.1f us %5.1f TF err %.1e", nm[i], m[i] * 1e3, gf / (m[i] * 1e-3) / 1e12, e[i]);
    printf("\n");
  }
  return 0;
}
