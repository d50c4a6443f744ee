// tools/wdirect_lab.hip -- diagnostic only: a product kernel whose WEIGHT fragments never touch LDS.
//
// The two-piece products of gemm_f32.h / chain_f16.h stage both operands through LDS every 16-wide k-tile: per workgroup and k-tile
// 48 KB of LDS reads + 16..20 KB of writes against 6 matrix instructions per wave -- at 1 KB of LDS read per v_mfma_f32_32x32x16_f16
// the LDS pipe (128 B/clk/CU) is exactly as busy as the matrix pipe would be at peak, and every weight word costs a byte permute
// on the way in.  Here the weights are stored FRAGMENT-MAJOR in memory (one 1 KB block per 32-row tile, 16-wide k-tile and piece:
// lane (li, lh) owns the eight f16 of row li at k = 8 lh .. 8 lh + 7, exactly the matrix instruction's operand), so a wave loads its
// weight operand with ONE coalesced 16-byte load per lane and piece, straight into registers, two k-tiles ahead; only the
// activations (which need the split, and which all eight waves share) go through LDS.  A 64-row workgroup, wave w = hidden tile w
// x all 64 rows: per workgroup and k-tile 32 KB of LDS reads + 4 KB of writes + 16 KB of weight loads, no permutes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/wdirect_lab.exe tools/wdirect_lab.hip && tools/wdirect_lab.exe [nodes]
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

// words [rows][K] (h low half, l' high half) -> fragment planes: uint4 frag[((ht * nk + kt) * 2 + piece) * 64 + lane]
__global__ void k_words_to_frags(const unsigned* __restrict__ w, unsigned short* __restrict__ f, int rows, int K) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * K) return;
  const int r = idx / K, k = idx % K, nk = K / 16;
  const int ht = r >> 5, li = r & 31, kt = k >> 4, lh = (k >> 3) & 1, j = k & 7;
  const unsigned word = w[idx];
  const size_t base = ((size_t)(ht * nk + kt) * 2) * 64 * 8;
  f[base + (size_t)(lh * 32 + li) * 8 + j] = (unsigned short)(word & 0xFFFFu);
  f[base + 64 * 8 + (size_t)(lh * 32 + li) * 8 + j] = (unsigned short)(word >> 16);
}

// C[M, 32 * gridDim.y * 8 ...]: one workgroup = 64 rows x 256 hidden (8 waves x 32), relu(A . W^T + b)
template <int PFW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_wd(const float* __restrict__ A, int lda, const uint4* __restrict__ Wf,
                                                                                     const float* __restrict__ ws, const float* __restrict__ bias,
                                                                                     float* __restrict__ C, int ldc, int M, int K) {
  constexpr int R = 64, RB = 48, kPlaneA = R * RB, kStage = 2 * kPlaneA;
  constexpr float kCorW = 1.f / kF16LowScale;
  __shared__ __attribute__((aligned(16))) char lds[2 * kStage];
  __shared__ float rs_sh[R];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * R;
  const int kq = t & 3, r4 = (t >> 2) & 63;
  const bool stage_a = t < 256;
  const float* arow_g = A + (size_t)min(m0 + r4, M - 1) * lda + 4 * kq;
  const int nk = K / 16;
  auto quad_max = [&](float m) -> float { m = fmaxf(m, __shfl_xor(m, 1, 64)); return fmaxf(m, __shfl_xor(m, 2, 64)); };
  f32x16 acc[2], cor[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) { acc[i][e] = 0.f; cor[i][e] = 0.f; }
  const uint4* wbase = Wf + ((size_t)wave * nk) * 2 * 64 + lane;
  uint4 wh[PFW], wl[PFW];
  float4 ra[2];
  float asc = 1.f;
  auto wload = [&](int slot, int kt) { wh[slot] = wbase[(size_t)(2 * kt) * 64]; wl[slot] = wbase[(size_t)(2 * kt + 1) * 64]; };
  auto aload = [&](int slot, int kt) { if (stage_a) ra[slot] = *reinterpret_cast<const float4*>(arow_g + 16 * kt); };
  auto astore = [&](int slot, int st) {
    if (!stage_a) return;
    const float4 v = make_float4(ra[slot].x * asc, ra[slot].y * asc, ra[slot].z * asc, ra[slot].w * asc);
    unsigned h0, l0, h1, l1;
    split2h(v.x, v.y, h0, l0);
    split2h(v.z, v.w, h1, l1);
    char* p = lds + st * kStage + r4 * RB + 8 * kq;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + kPlaneA) = make_uint2(l0, l1);
  };
  // the lab scales a row by its EXACT maximum (one more pass over the row out of L2; the product kernels use the sampled estimate)
  if (stage_a) {
    float tm = 0.f;
    for (int kt = 0; kt < nk; kt++) {
      const float4 x = *reinterpret_cast<const float4*>(arow_g + 16 * kt);
      tm = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), tm); tm = fmaxf(fmaxf(fabsf(x.z), fabsf(x.w)), tm);
    }
    asc = pow2_scale(quad_max(tm), kScaleExact);
    if (kq == 0) rs_sh[r4] = pow2_inv(asc);
  }
  aload(0, 0);
#pragma unroll
  for (int s = 0; s < PFW; s++) if (s < nk) wload(s, s);
  astore(0, 0);
  if (nk > 1) aload(1, 1);
  __syncthreads();
  const int aoff = li * RB + 16 * lh;
  auto body = [&](int kt, int ws_slot, int as_slot) {
    const int st = kt & 1;
    // the next A tile: registers -> the other stage (its readers passed the barrier of the previous iteration); then the one after
    if (kt + 1 < nk) astore(as_slot ^ 1, st ^ 1);
    if (kt + 2 < nk) aload(as_slot, kt + 2);
    const char* base = lds + st * kStage + aoff;
    const f16x8 bh = __builtin_bit_cast(f16x8, wh[ws_slot]), bl = __builtin_bit_cast(f16x8, wl[ws_slot]);
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
      const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + rt * 32 * RB));
      const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + kPlaneA + rt * 32 * RB));
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[rt], 0, 0, 0);     // registers = hidden, lanes = rows
      cor[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, cor[rt], 0, 0, 0);
      cor[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, cor[rt], 0, 0, 0);
    }
    if (kt + PFW < nk) wload(ws_slot, kt + PFW);
    __syncthreads();
  };
  static_assert(PFW == 2 || PFW == 4, "prefetch depth of the weight fragments");
  int kt = 0;
  if (PFW == 2) {
    for (; kt + 1 < nk; kt += 2) { body(kt, 0, 0); body(kt + 1, 1, 1); }
    if (kt < nk) body(kt, 0, 0);
  } else {
    for (; kt + 3 < nk; kt += 4) { body(kt, 0, 0); body(kt + 1, 1, 1); body(kt + 2, 2, 0); body(kt + 3, 3, 1); }     // (the lab's K are multiples of 64)
  }
#pragma unroll
  for (int rt = 0; rt < 2; rt++) {
    const int rl = 32 * rt + li, m = m0 + rl;
    const float rsn = rs_sh[rl];
    if (m < M) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int hid = 32 * wave + 8 * g + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4*>(bias + hid), w4 = *reinterpret_cast<const float4*>(ws + hid);
        *reinterpret_cast<float4*>(C + (size_t)m * ldc + hid) =
            make_float4(fmaxf((acc[rt][4 * g + 0] + cor[rt][4 * g + 0] * kCorW) * (rsn * w4.x) + b4.x, 0.f),
                        fmaxf((acc[rt][4 * g + 1] + cor[rt][4 * g + 1] * kCorW) * (rsn * w4.y) + b4.y, 0.f),
                        fmaxf((acc[rt][4 * g + 2] + cor[rt][4 * g + 2] * kCorW) * (rsn * w4.z) + b4.z, 0.f),
                        fmaxf((acc[rt][4 * g + 3] + cor[rt][4 * g + 3] * kCorW) * (rsn * w4.w) + b4.w, 0.f));
      }
    }
  }
}

constexpr auto kSplitRelu = k_gemm3<EPI_RELU, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
constexpr int kLds128 = TileCfg3<4, 2, 1, 2, 16, 2>::kLdsBytes;

static unsigned g_seed = 12345;
static float rnd() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffff) / 65536.0f - 0.5f; }
static float* dev(size_t n, float scale) {
  std::vector<float> h(n);
  for (auto& v : h) v = rnd() * scale;
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 35840;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kSplitRelu), hipFuncAttributeMaxDynamicSharedMemorySize, kLds128));
  const int tiles = (N + 127) / 128, blocks = (N + 63) / 64;
  printf("weights-direct lab: %d rows, hidden 256 (%d workgroups of 64 rows; the tile kernel: %d workgroups of 128 x 128)\n", N, blocks, tiles * 2);
  for (int K : {256, 576}) {
    float* A = dev((size_t)N * K, 2.0f);
    float* W = dev((size_t)256 * K, 0.2f);
    float* b = dev(256, 1.0f);
    unsigned* Ww; float* Wsc;
    CK(hipMalloc(&Ww, (size_t)256 * K * 4)); CK(hipMalloc(&Wsc, 256 * 4));
    EncMat m{0, 256, K, 0, 1};
    EncMat* dm; CK(hipMalloc(&dm, sizeof(EncMat))); CK(hipMemcpy(dm, &m, sizeof(EncMat), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_encode_rows, dim3(64), dim3(256), 0, 0, W, Ww, Wsc, dm, 1, 256);
    unsigned short* Wf; CK(hipMalloc(&Wf, (size_t)256 * K * 4));
    hipLaunchKernelGGL(k_words_to_frags, dim3((256 * K + 255) / 256), dim3(256), 0, 0, Ww, Wf, 256, K);
    float *Ca, *Cb;
    CK(hipMalloc(&Ca, (size_t)N * 256 * 4)); CK(hipMalloc(&Cb, (size_t)N * 256 * 4));
    GemmArgs g{A, K, reinterpret_cast<const float*>(Ww), K, b, Ca, 256, N, 256, K, EPI_RELU, nullptr, nullptr, 0};
    g.wscale = Wsc;
    auto tile = [&] { hipLaunchKernelGGL(kSplitRelu, dim3(tiles * 2), dim3(512), kLds128, 0, g); };
    auto wd2 = [&] { hipLaunchKernelGGL((k_wd<2>), dim3(blocks), dim3(512), 0, 0, A, K, reinterpret_cast<const uint4*>(Wf), Wsc, b, Cb, 256, N, K); };
    auto wd4 = [&] { hipLaunchKernelGGL((k_wd<4>), dim3(blocks), dim3(512), 0, 0, A, K, reinterpret_cast<const uint4*>(Wf), Wsc, b, Cb, 256, N, K); };
    tile(); wd2();
    CK(hipDeviceSynchronize());
    std::vector<float> ha((size_t)N * 256), hb((size_t)N * 256);
    CK(hipMemcpy(ha.data(), Ca, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), Cb, hb.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, mag = 0;
    for (size_t i = 0; i < ha.size(); i++) { worst = fmax(worst, fabs((double)ha[i] - hb[i])); mag = fmax(mag, fabs((double)ha[i])); }
    const float t0 = time_us(tile), t2 = time_us(wd2), t4 = time_us(wd4);
    const double gf = 2.0 * N * 256.0 * K * 1e-9;
    printf("  K = %3d: tile kernel (both operands through LDS) %.1f us (%.0f TF) | weights direct, 2 k-tiles ahead %.1f us (%.0f TF), 4 ahead %.1f us (%.0f TF) | max |diff| %.2e of %.2e\n",
           K, t0, gf / t0 * 1e3, t2, gf / t2 * 1e3, t4, gf / t4 * 1e3, worst, mag);
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(Ww)); CK(hipFree(Wf)); CK(hipFree(Ca)); CK(hipFree(Cb));
  }
  return 0;
}
