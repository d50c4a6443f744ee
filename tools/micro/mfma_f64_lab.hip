// mfma_f64_lab.hip -- FP64 matrix cores for the dense products of one dynamics evaluation.
//
// One wavefront per workgroup, operands in LDS exactly as k_env_step keeps them (packed lower-triangular L^-1,
// constraint rows Y[r][ldy]).  Timed in-kernel with s_memtime, two regimes: one wave alone on its SIMD (grid = #CUs)
// and the engine's residency (8 single-wave workgroups per CU, LDS padded to the engine's slab).
//   rate      : v_mfma_f64_16x16x4_f64 issue interval (4 independent accumulators) and dependent-accumulator latency
//   trmm_valu : Y <- L^-1 Y, the engine's register form (wave_hip.h trmm_rows_reg: two lanes per right-hand side)
//   trmm_mfma : the same product on v_mfma_f64_16x16x4_f64 (2 x 2 output tiles, triangular k range)
//   aff_valu  : A_FF = Y_F Y_F' (packed lower triangle), the engine's lane-per-entry form
//   aff_mfma  : the same on the matrix cores
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/mfma_f64_lab.exe tools/micro/mfma_f64_lab.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../sgrl_amd/csrc/wave_hip.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NV = 24, LDY = 25, NRHS = 25, NF = 24;
constexpr int REPS = 64;

extern __shared__ double lds[];

__device__ __forceinline__ int tri(int i) { return i * (i + 1) / 2; }

// ---- rate --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rate(double* out, long long* cyc) {
  const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < 256; r++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  v4d d = {0, 0, 0, 0};
#pragma unroll 1
  for (int r = 0; r < 1024; r++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
  long long t2 = __builtin_readcyclecounter();
  // dependent FMA chain for comparison
  double f = a;
#pragma unroll 1
  for (int r = 0; r < 1024; r++) f = fma(f, b, a);
  long long t3 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d[0] + f;
  if (threadIdx.x == 0) { cyc[4 * blockIdx.x] = t1 - t0; cyc[4 * blockIdx.x + 1] = t2 - t1; cyc[4 * blockIdx.x + 2] = t3 - t2; }
}

// ---- operands ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fill(double* T, double* Y, int* flist, unsigned seed) {
  const int lane = threadIdx.x;
  for (int p = lane; p < NV * (NV + 1) / 2; p += 64) T[p] = 0.01 * ((p * 2654435761u + seed) % 1000) - 5.0;
  for (int p = lane; p < (NRHS + 7) * LDY; p += 64) Y[p] = 0.01 * ((p * 40503u + seed * 7) % 1000) - 5.0;
  for (int p = lane; p < 32; p += 64) flist[p] = (p * 5) % NF;        // a permutation of 0..23 for p < 24
  if (lane < 32) flist[lane] = lane < NF ? (lane * 5) % NF : 0;
  __syncthreads();
}

// ---- trmm ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void trmm_mfma(const double* T, double* Y, int n, int nrhs) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  // B[k = c][col = r] = Y[r][c]: this lane's entries for the two right-hand-side tiles and the six k-steps
  double b0[6], b1[6], a0[4], a1[6];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const int c = 4 * s + hi;
    const double y0 = Y[lo * LDY + (c < n ? c : 0)];
    const double y1 = Y[(16 + lo < nrhs ? 16 + lo : 0) * LDY + (c < n ? c : 0)];
    b0[s] = c < n ? y0 : 0.0;
    b1[s] = (c < n && 16 + lo < nrhs) ? y1 : 0.0;
  }
  // A[row = d][k = c] = T[d][c] for c <= d
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const int c = 4 * s + hi;
    if (s < 4) { const int d = lo; const double t = T[tri(d) + (c <= d ? c : 0)]; a0[s] = c <= d ? t : 0.0; }
    const int d = 16 + lo;
    const double t = T[tri(d < n ? d : 0) + ((c <= d && d < n) ? c : 0)];
    a1[s] = (c <= d && d < n) ? t : 0.0;
  }
  v4d c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll
  for (int s = 0; s < 6; s++) {
    if (s < 4) {
      c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b0[s], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b1[s], c01, 0, 0, 0);
    }
    c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b0[s], c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b1[s], c11, 0, 0, 0);
  }
  // D[row = hi + 4 i][col = lo] -> Y[r = lo (+16)][d = hi + 4 i (+16)]
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int d = hi + 4 * i;
    Y[lo * LDY + d] = c00[i];
    if (16 + lo < nrhs) Y[(16 + lo) * LDY + d] = c01[i];
    if (16 + d < n) {
      Y[lo * LDY + 16 + d] = c10[i];
      if (16 + lo < nrhs) Y[(16 + lo) * LDY + 16 + d] = c11[i];
    }
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_trmm(double* out, long long* cyc, int n, int nrhs) {
  double* T = lds; double* Y = lds + 320; int* flist = (int*)(lds + 320 + 32 * LDY);
  fill(T, Y, flist, blockIdx.x);
  sgrl::HipWaveT<24> w;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REPS; r++) {
    if (MODE == 0) w.trmm_rows(nrhs, n, T, Y, LDY);
    else trmm_mfma(T, Y, n, nrhs);
    // keep the values bounded: rescale the rows (same work in both modes)
    for (int p = threadIdx.x; p < nrhs * LDY; p += 64) Y[p] = Y[p] * 1e-3 + 0.25;
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int p = threadIdx.x; p < nrhs * LDY; p += 64) s += Y[p];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// one application, result written out for the correctness check
template <int MODE>
__global__ __launch_bounds__(64) void k_trmm_once(double* out, int n, int nrhs) {
  double* T = lds; double* Y = lds + 320; int* flist = (int*)(lds + 320 + 32 * LDY);
  fill(T, Y, flist, 3);
  sgrl::HipWaveT<24> w;
  if (MODE == 0) w.trmm_rows(nrhs, n, T, Y, LDY); else trmm_mfma(T, Y, n, nrhs);
  for (int p = threadIdx.x; p < nrhs * LDY; p += 64) out[p] = (p % LDY) < n ? Y[p] : 0.0;
}

// ---- A_FF ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_decode(int p, int* a_out, int* b_out) {
  int a = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
  a += ((a + 1) * (a + 2) / 2 <= p) ? 1 : 0;
  a -= (a * (a + 1) / 2 > p) ? 1 : 0;
  *a_out = a; *b_out = p - a * (a + 1) / 2;
}
__device__ __forceinline__ void aff_valu(const double* Y, const int* flist, double* C, int nf, int nv) {
  for (int p = threadIdx.x; p < nf * (nf + 1) / 2; p += 64) {
    int i, j;
    tri_decode(p, &i, &j);
    const double* yi = Y + flist[i] * LDY;
    const double* yj = Y + flist[j] * LDY;
    double a = 0;
    int d = 0;
    for (; d + 4 <= nv; d += 4) {
      const double a0 = yi[d], a1 = yi[d + 1], a2 = yi[d + 2], a3 = yi[d + 3];
      const double b0 = yj[d], b1 = yj[d + 1], b2 = yj[d + 2], b3 = yj[d + 3];
      a += a0 * b0; a += a1 * b1; a += a2 * b2; a += a3 * b3;
    }
    for (; d < nv; d++) a += yi[d] * yj[d];
    C[p] = a;
  }
  __syncthreads();
}
__device__ __forceinline__ void aff_mfma(const double* Y, const int* flist, double* C, int nf, int nv) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int f0 = flist[lo], f1 = flist[16 + lo];      // flist padded to 32 entries
  double y0[6], y1[6];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const int d = 4 * s + hi;
    const double v0 = Y[f0 * LDY + (d < nv ? d : 0)], v1 = Y[f1 * LDY + (d < nv ? d : 0)];
    y0[s] = (d < nv && lo < nf) ? v0 : 0.0;
    y1[s] = (d < nv && 16 + lo < nf) ? v1 : 0.0;
  }
  v4d c00 = {0, 0, 0, 0}, c10 = c00, c11 = c00;
#pragma unroll
  for (int s = 0; s < 6; s++) {
    c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[s], y0[s], c00, 0, 0, 0);
    if (nf > 16) {
      c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[s], y0[s], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[s], y1[s], c11, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r = hi + 4 * i, c = lo;        // D[row r][col c]
    if (r < nf && c <= r) C[tri(r) + c] = c00[i];
    if (16 + r < nf) {
      C[tri(16 + r) + c] = c10[i];
      if (c <= r) C[tri(16 + r) + 16 + c] = c11[i];
    }
  }
  __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_aff(double* out, long long* cyc, int nf, int nv) {
  double* T = lds; double* Y = lds + 320; int* flist = (int*)(lds + 320 + 32 * LDY); double* C = lds + 320 + 32 * LDY + 16;
  fill(T, Y, flist, blockIdx.x);
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REPS; r++) {
    if (MODE == 0) aff_valu(Y, flist, C, nf, nv); else aff_mfma(Y, flist, C, nf, nv);
    if (threadIdx.x < nv) Y[threadIdx.x] += C[threadIdx.x] * 1e-6;     // dependence between repetitions
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  for (int p = threadIdx.x; p < nf * (nf + 1) / 2; p += 64) out[blockIdx.x * 320 + p] = C[p];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double median(std::vector<long long>& v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.gcnArchName, ncu);
  const int lds_small = (320 + 32 * LDY + 16 + 320) * 8;
  double* out; long long* cyc;
  CHECK(hipMalloc(&out, sizeof(double) * 8192 * 320));
  CHECK(hipMalloc(&cyc, sizeof(long long) * 8192 * 4));
  std::vector<long long> h(8192 * 4);
  {
    hipLaunchKernelGGL(k_rate, dim3(ncu), dim3(64), 0, 0, out, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * ncu * 4, hipMemcpyDeviceToHost));
    std::vector<long long> a, b, c;
    for (int i = 0; i < ncu; i++) { a.push_back(h[4 * i]); b.push_back(h[4 * i + 1]); c.push_back(h[4 * i + 2]); }
    printf("rate: v_mfma_f64_16x16x4_f64 issue interval %.1f cycles (4 independent accumulators), dependent-accumulator %.1f cycles; "
           "dependent v_fma_f64 %.1f cycles\n", median(a) / 1024.0, median(b) / 1024.0, median(c) / 1024.0);
  }
  // correctness of the MFMA forms against the engine's forms
  {
    for (int n : {24, 21, 18, 12}) {
      const int nrhs = n + 1;
      std::vector<double> r0(32 * LDY), r1(32 * LDY);
      hipLaunchKernelGGL(k_trmm_once<0>, dim3(1), dim3(64), lds_small, 0, out, n, nrhs);
      CHECK(hipMemcpy(r0.data(), out, sizeof(double) * nrhs * LDY, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(k_trmm_once<1>, dim3(1), dim3(64), lds_small, 0, out, n, nrhs);
      CHECK(hipMemcpy(r1.data(), out, sizeof(double) * nrhs * LDY, hipMemcpyDeviceToHost));
      double worst = 0, big = 0;
      for (int p = 0; p < nrhs * LDY; p++) { worst = fmax(worst, fabs(r0[p] - r1[p])); big = fmax(big, fabs(r0[p])); }
      printf("trmm n=%d: max |valu - mfma| = %.3e (max |value| %.3e)\n", n, worst, big);
    }
    for (int nf : {24, 16, 10}) {
      std::vector<double> r0(320), r1(320);
      hipLaunchKernelGGL(k_aff<0>, dim3(1), dim3(64), lds_small, 0, out, cyc, nf, NV);
      CHECK(hipMemcpy(r0.data(), out, sizeof(double) * 320, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(k_aff<1>, dim3(1), dim3(64), lds_small, 0, out, cyc, nf, NV);
      CHECK(hipMemcpy(r1.data(), out, sizeof(double) * 320, hipMemcpyDeviceToHost));
      double worst = 0, big = 0;
      for (int p = 0; p < nf * (nf + 1) / 2; p++) { worst = fmax(worst, fabs(r0[p] - r1[p])); big = fmax(big, fabs(r0[p])); }
      printf("A_FF nf=%d: max |valu - mfma| = %.3e (max |value| %.3e)\n", nf, worst, big);
    }
  }
  // timing: alone (one workgroup per CU) and at the engine's residency (8 per CU: LDS request 20 KB)
  for (int regime = 0; regime < 2; regime++) {
    const int grid = regime == 0 ? ncu : ncu * 8 * 4;
    const int ldsb = regime == 0 ? lds_small : 20344;
    auto run = [&](auto kern, int a1, int a2, const char* name) {
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64), ldsb, 0, out, cyc, a1, a2);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost));
      std::vector<long long> v(h.begin(), h.begin() + grid);
      printf("  %-34s %8.0f cycles per application\n", name, median(v) / REPS);
    };
    printf("%s:\n", regime == 0 ? "one wave per CU" : "8 workgroups per CU (engine residency), 4 rounds");
    run(k_trmm<0>, 24, 25, "trmm valu  n=24 nrhs=25");
    run(k_trmm<1>, 24, 25, "trmm mfma  n=24 nrhs=25");
    run(k_trmm<0>, 18, 19, "trmm valu  n=18 nrhs=19");
    run(k_trmm<1>, 18, 19, "trmm mfma  n=18 nrhs=19");
    run(k_trmm<0>, 12, 13, "trmm valu  n=12 nrhs=13");
    run(k_trmm<1>, 12, 13, "trmm mfma  n=12 nrhs=13");
    run(k_aff<0>, 24, 24, "A_FF valu  nf=24 nv=24");
    run(k_aff<1>, 24, 24, "A_FF mfma  nf=24 nv=24");
    run(k_aff<0>, 16, 24, "A_FF valu  nf=16 nv=24");
    run(k_aff<1>, 16, 24, "A_FF mfma  nf=16 nv=24");
    run(k_aff<0>, 10, 18, "A_FF valu  nf=10 nv=18");
    run(k_aff<1>, 10, 18, "A_FF mfma  nf=10 nv=18");
  }
  return 0;
}
