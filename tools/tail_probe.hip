// tools/tail_probe.hip -- diagnostic: how much of a SET GEMM launch is the partial last wave of tiles?  The split-precision
// kernel (gemm_f32.h) on the l3 (N 256) / qkv (N 768) / l4 (N 1024) shapes at row counts around the bench's 35 840 nodes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o /tmp/tail_probe tools/tail_probe.hip && /tmp/tail_probe
#include "../sgrl_amd/csrc/gemm_f32.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace sgrl_gemm;

template <class K>
static float timeit(K k, int tiles, int threads, int lds, const GemmArgs& a, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k, dim3(tiles), dim3(threads), lds, 0, a);
  hipEventRecord(t0, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(tiles), dim3(threads), lds, 0, a);
  hipEventRecord(t1, 0);
  hipEventSynchronize(t1);
  float ms = 0;
  hipEventElapsedTime(&ms, t0, t1);
  return ms / reps;
}

int main() {
  const int Mmax = 73728;
  float *A, *W, *C, *bias, *rd;
  hipMalloc(&A, (size_t)Mmax * 256 * 4); hipMalloc(&W, 1024 * 256 * 4); hipMalloc(&C, (size_t)Mmax * 1024 * 4);
  hipMalloc(&bias, 4096); hipMalloc(&rd, (size_t)Mmax * 4);
  std::vector<float> h((size_t)Mmax * 256);
  unsigned s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), 1024 * 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), 4096, hipMemcpyHostToDevice);
  for (auto& v : h) v += 2.f;
  hipMemcpy(rd, h.data(), (size_t)Mmax * 4, hipMemcpyHostToDevice);
  using Cfg = TileCfg3<4, 2, 1, 2, 16>;
  for (int N : {256, 768, 1024}) {
    for (int M : {16384, 24576, 28672, 32768, 33792, 34816, 35840, 36864, 40960, 49152, 65536, 66560, 71680}) {
      GemmArgs a{A, 256, W, 256, bias, C, N, M, N, 256, EPI_RELU, rd, nullptr, 0};
      const int tiles = ((M + 127) / 128) * (N / 128);
      const float ms = timeit(k_gemm3<EPI_RELU, 4, 2, 1, 2, 16, 2>, tiles, Cfg::kThreads, Cfg::kLdsBytes, a, 20);
      printf("N %4d M %6d tiles %5d (%.2f waves of 512): %7.1f us  %.3f ns/row\n", N, M, tiles, tiles / 512.0, ms * 1e3, ms * 1e6 / M);
    }
  }
  return 0;
}
