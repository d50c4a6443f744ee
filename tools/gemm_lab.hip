// tools/gemm_lab.hip -- diagnostic only: the SET actor's GEMM kernels (gemm_f32.h) on the shapes of one forward:
// exact-f32 MFMA kernel vs the split-precision bf16x6 kernel in several tile configurations, with the error of each against
// a float64 host reference on sampled outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_lab tools/gemm_lab.hip && /tmp/gemm_lab
// ARCHIVE of the round-3 lab (profiles/r3_gemm_lab*.txt): it is written against the round-3 gemm_f32.h (clamped words,
// range events; `git show 6977511:sgrl_amd/csrc/gemm_f32.h`) and does not build against the row-scaled header of round 4.
// The round-4 lab is tools/chain_lab.hip.
#include "../sgrl_amd/csrc/gemm_f32.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace sgrl_gemm;

// ---- round-3 experiment (not in the product): staging by LDS-DMA from f16 planes -----------------------------------------
namespace sgrl_gemm {
// k_gemm6: the two-piece f16 x 3 product with BOTH operands given as f16 PLANES (h and l', each [rows][ld] halves, `a_plane` /
// `w_plane` halves apart) and staged by LDS-DMA (global_load_lds_dwordx4: no staging registers, no split / permute arithmetic in
// the loop).  128 x 128 tile, 8 waves (4 x 2, one 32 x 64 patch each), 16-wide k-tiles in a ring of kStages6 LDS stages: the
// DMA of tile kt + kStages6 - 1 is issued while tile kt is multiplied, one raw barrier per k-tile, counted vmcnt waits.
// LDS image of a stage: [A h | A l' | W h | W l'], each 128 rows x 32 B, lane-linear (a DMA wave-instruction fills 32 rows); the
// two 16-byte k-halves of rows 8..15 (mod 16) are swapped -- done on the per-lane SOURCE address -- so that the ds_read_b128
// of a half-wave (32 rows, one k-half) touches every bank group once.
constexpr int kStageBytes6 = 4 * 128 * 32;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(g)),
                                   reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(l))), 16, 0, 0);
}
template <int FLAGS, int WORDS = 0, int kStages6 = 3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_gemm6(GemmArgs a) {
  static_assert(!(FLAGS & (EPI_EQUIV | EPI_LN | EPI_ZSPLIT)), "plain epilogues only");
  constexpr bool CWD = (WORDS & 4) != 0;
  constexpr float kCorW = 1.f / kF16LowScale;
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];     // (aligned: static LDS precedes it)
  char* lds = reinterpret_cast<char*>(gemm_lds);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (a.N + 127) / 128;
  int bid;
  {
    const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    bid = (x < rem) ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * 128, n0 = tile_n * 128;
  // this wave's two DMA pieces per k-tile: region = wave / 2 (A h, A l', W h, W l'), rows 32 * sub .. + 31, sub = 2 (wave & 1) + {0, 1}
  const unsigned short* src[2];
  int dst[2];
  {
    const int region = wave >> 1;
    const unsigned short* base = region < 2 ? reinterpret_cast<const unsigned short*>(a.A) + (region & 1) * a.a_plane
                                            : reinterpret_cast<const unsigned short*>(a.W) + (region & 1) * a.w_plane;
    const int ld = region < 2 ? a.lda : a.ldw, row0 = region < 2 ? m0 : n0, rows = region < 2 ? a.M : a.N;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int sub = 2 * (wave & 1) + q, r = 32 * sub + (lane >> 1), slot = lane & 1;
      const int khalf = slot ^ ((r >> 3) & 1);
      src[q] = base + (size_t)min(row0 + r, rows - 1) * ld + 8 * khalf;
      dst[q] = region * 4096 + sub * 1024;
    }
  }
  f32x16 acc[2], cor[2];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) { acc[j][e] = 0.f; cor[j][e] = 0.f; }
  const int nk = a.K / 16;
  auto issue = [&](int kt) {
    char* sb = lds + (kt % kStages6) * kStageBytes6;
    glds16(src[0] + kt * 16, sb + dst[0]);
    glds16(src[1] + kt * 16, sb + dst[1]);
  };
#pragma unroll
  for (int p = 0; p < kStages6 - 1; p++) if (p < nk) issue(p);
  const int arow = wm * 32 + li, aoff = arow * 32 + 16 * (lh ^ ((arow >> 3) & 1));
  int boff[2];
#pragma unroll
  for (int j = 0; j < 2; j++) { const int r = wn * 64 + j * 32 + li; boff[j] = 8192 + r * 32 + 16 * (lh ^ ((r >> 3) & 1)); }
  for (int kt = 0; kt < nk; kt++) {
    // tile kt has landed once at most the pieces of the tiles issued after it are outstanding (two per tile)
    if (kt + kStages6 - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kStages6 - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + kStages6 - 1 < nk) issue(kt + kStages6 - 1);
    const char* base = lds + (kt % kStages6) * kStageBytes6;
    const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + aoff));
    const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + 4096 + aoff));
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + boff[j]));
      const f16x8 bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + 4096 + boff[j]));
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
      cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cor[j], 0, 0, 0);
      cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cor[j], 0, 0, 0);
    }
  }
  float rmax2 = 0.f;
  const int mb = m0 + wm * 32 + 4 * lh;
  float rdiv[16];
  if (FLAGS & EPI_ROWDIV) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int m = mb + (e & 3) + 8 * (e >> 2);
      rdiv[e] = 1.0f / ((m < a.M) ? a.rowdiv[m] : 1.f);
    }
  }
#pragma unroll
  for (int tj = 0; tj < 2; tj++) {
    const int n = n0 + wn * 64 + tj * 32 + li;
    if (n >= a.N) continue;
    const float bvv = a.bias ? a.bias[n] : 0.f;
    float old2[16];
    if (FLAGS & EPI_ACC2) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        old2[e] = (m < a.M) ? a.C2[(size_t)m * a.ldc2 + n] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int m = mb + (e & 3) + 8 * (e >> 2);
      if (m >= a.M) continue;
      float v = acc[tj][e] + cor[tj][e] * kCorW + bvv;
      if (FLAGS & EPI_RELU) v = fmaxf(v, 0.f);
      if (FLAGS & EPI_ROWDIV) v = v * rdiv[e];
      if (a.phase_sleep == 777 && v != 12345.678f) continue;   // ablation: no C stores
      if (CWD) reinterpret_cast<unsigned*>(a.C)[(size_t)m * a.ldc + n] = enc_word(v, rmax2);
      else a.C[(size_t)m * a.ldc + n] = v;
      if (FLAGS & EPI_ACC2) a.C2[(size_t)m * a.ldc2 + n] = old2[e] + v;
    }
  }
  if (CWD && rmax2 > kF16Lim && a.range_events) atomicAdd(a.range_events, 1u);
}

// f32 [rows][ld_src] -> the two f16 planes h | l' of the two-piece form ([rows][ld_dst] halves each, `plane` halves apart)
__global__ __launch_bounds__(256) void k_split_planes2(const float* __restrict__ src, int ld_src, int rows, int cols,
                                                       unsigned short* __restrict__ dst, int ld_dst, long long plane) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)rows * cols; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float x = __builtin_amdgcn_fmed3f(src[(size_t)r * ld_src + c], -kF16Lim, kF16Lim);
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * kF16LowScale);
    dst[(size_t)r * ld_dst + c] = __builtin_bit_cast(unsigned short, h);
    dst[plane + (size_t)r * ld_dst + c] = __builtin_bit_cast(unsigned short, l);
  }
}

}  // namespace sgrl_gemm

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

template <int F, int S>
static float run6(const GemmArgs& a, int reps) {
  const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
  return timeit(k_gemm6<F, 0, S>, tiles, 512, S * kStageBytes6, a, reps);
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
  unsigned short *Apl, *Wpl, *Apl2, *Wpl2;
  hipMalloc(&Apl, maxA * 2 * 3); hipMalloc(&Wpl, (size_t)1024 * 544 * 2 * 3);
  hipMalloc(&Apl2, maxA * 2 * 2 + 4096); hipMalloc(&Wpl2, (size_t)1024 * 544 * 2 * 2 + 4096);
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
      // f16 planes of both operands for the LDS-DMA kernel
      const long long apl2 = (long long)sh.M * sh.K, wpl2 = (long long)sh.N * sh.K;
      hipLaunchKernelGGL(k_split_planes2, dim3(2048), dim3(256), 0, 0, A, sh.K, sh.M, sh.K, Apl2, sh.K, apl2);
      hipLaunchKernelGGL(k_split_planes2, dim3(512), dim3(256), 0, 0, W, sh.K, sh.N, sh.K, Wpl2, sh.K, wpl2);
      GemmArgs a6 = a; a6.A = reinterpret_cast<const float*>(Apl2); a6.W = reinterpret_cast<const float*>(Wpl2); a6.a_plane = apl2; a6.w_plane = wpl2;
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
        ms = run3<F, 2, 4, 1, 1, 16, 2, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words  64x128/8w(32x32) %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 2, 2, 2, 2, 16, 2, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words 128x128/4w (64x64 per wave) %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 2, 2, 2, 2, 16, 1, false, false, false, 0, 2, false, 2>(aww, reps); printf("   f16x3  W words 128x128/4w (64x64 per wave) pf1 %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run3<F, 2, 4, 2, 1, 16, 2, false, false, false, 0, 2, true, 2>(aww, reps); printf("   f16x3  W words 128x128/8w SKEW (64x32 per wave) %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run6<F, 3>(a6, reps); printf("   gemm6  planes by LDS-DMA, 3 stages  %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run6<F, 4>(a6, reps); printf("   gemm6  planes by LDS-DMA, 4 stages  %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        ms = run6<F, 5>(a6, reps); printf("   gemm6  planes by LDS-DMA, 5 stages  %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); \
        { GemmArgs a7 = a6; a7.phase_sleep = 777; ms = run6<F, 3>(a7, reps); printf("   gemm6  3 stages, NO C stores (ablation) %6.1f us\n", ms * 1e3); } \
        ms = run6<F, 8>(a6, reps); printf("   gemm6  planes by LDS-DMA, 8 stages (one block per CU) %6.1f us %5.1f TF", ms * 1e3, gf / (ms * 1e-3) / 1e12); err(); printf("\n"); }
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
