// tools/micro/mfma_valu_overlap.hip -- diagnostic: do VALU instructions issue in the shadow of matrix instructions on gfx950?
// Each wave loops over (NM independent v_mfma_f32_32x32x16_f16, NV f32 FMAs on other registers); time per iteration for
//   (a) matrix only, (b) VALU only, (c) both in one wave, interleaved by the compiler or not, (d) waves that do only one kind
// sharing a SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap.exe tools/micro/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, int MODE>   // MODE 0: every wave does both; 1: even waves matrix, odd waves VALU (same totals per pair)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[2];
  for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e - 3.f); }
  float v[8];
  for (int e = 0; e < 8; e++) v[e] = threadIdx.x + e;
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || (wave & 4) == 0, do_v = MODE == 0 || (wave & 4) != 0;   // waves w and w + 4 share a SIMD
  const int nm = MODE == 0 ? NM : 2 * NM, nv = MODE == 0 ? NV : 2 * NV;
  for (int it = 0; it < iters; it++) {
    if (do_m)
#pragma unroll
      for (int q = 0; q < nm; q++) acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q & 1], 0, 0, 0);
    if (do_v)
#pragma unroll
      for (int q = 0; q < nv; q++) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
  }
  float s = 0;
  for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[j][e];
  for (int e = 0; e < 8; e++) s += v[e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// MODE 2: the same totals as MODE 0, but interleaved INSIDE every wave's instruction stream: one matrix instruction, NV / NM FMAs, ...
template <int NM, int NV>
__global__ __launch_bounds__(512) void k_il(float* out, int iters) {
  f32x16 acc[2];
  for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e - 3.f); }
  float v[8];
  for (int e = 0; e < 8; e++) v[e] = threadIdx.x + e;
  constexpr int PER = NV / NM;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < NM; q++) {
      acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q & 1], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < PER; r++) v[(q * PER + r) & 7] = __builtin_fmaf(v[(q * PER + r) & 7], 1.0001f, 0.5f);
    }
#pragma unroll
    for (int q = 0; q < NM; q++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
    }
  }
  float s = 0;
  for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[j][e];
  for (int e = 0; e < 8; e++) s += v[e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <class K> static float run(K kern, float* out, int iters) {
  hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
  hipLaunchKernelGGL(kern, dim3(512), dim3(512), 0, 0, out, 10);
  hipEventRecord(t0, 0);
  hipLaunchKernelGGL(kern, dim3(512), dim3(512), 0, 0, out, iters);    // 2 blocks of 8 waves per CU: 4 waves per SIMD
  hipEventRecord(t1, 0); hipEventSynchronize(t1);
  float ms; hipEventElapsedTime(&ms, t0, t1);
  return ms * 1e6f / iters;   // ns per iteration
}
int main() {
  float* out; hipMalloc(&out, 512 * 512 * 4);
  const int it = 20000;
  printf("ns per iteration, 4 waves per SIMD (per wave and iteration: 6 mfma 32x32x16 f16 / 48 f32 fma)\n");
  printf("matrix only        : %.1f\n", run(k<6, 0, 0>, out, it));
  printf("valu only          : %.1f\n", run(k<0, 48, 0>, out, it));
  printf("both, every wave   : %.1f\n", run(k<6, 48, 0>, out, it));
  printf("split by wave      : %.1f   (waves 0-3 matrix x2, waves 4-7 valu x2: same totals per SIMD)\n", run(k<6, 48, 1>, out, it));
  printf("interleaved in-wave: %.1f   (1 mfma, 8 fma, 1 mfma, 8 fma, ... in every wave's stream)\n", run(k_il<6, 48>, out, it));
  printf("interleaved, 5/gap : %.1f   (1 mfma, 5 fma, ...: 30 fma per iteration)\n", run(k_il<6, 30>, out, it));
  printf("interleaved, 4/gap : %.1f   (24 fma per iteration)\n", run(k_il<6, 24>, out, it));
  printf("valu only x2       : %.1f\n", run(k<0, 96, 0>, out, it));
  printf("matrix only x2     : %.1f\n", run(k<12, 0, 0>, out, it));
  return 0;
}
