// tools/solver_probe.hip -- diagnostic only: cycle counts of the wave-level dense solvers of wave_hip.h in isolation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/solver_probe tools/solver_probe.hip && /tmp/solver_probe [n] [nrhs]
#include "../sgrl_amd/csrc/wave_hip.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void k_probe(int n, int nrhs, int reps, const double* M0, unsigned long long* out) {
  extern __shared__ double S[];
  sgrl::HipWave w;
  const int tri = n * (n + 1) / 2, ldy = n | 1;
  double* P = S;
  double* dinv = P + tri;
  double* Y = dinv + 64;
  double* x = Y + (nrhs + 1) * ldy;
  unsigned long long t_chol = 0, t_trsm = 0, t_trsv = 0, t_inv = 0;
  for (int r = 0; r < reps; r++) {
    for (int i = w.lane; i < tri; i += 64) P[i] = M0[i];
    for (int i = w.lane; i < nrhs * ldy; i += 64) Y[i] = 0.001 * (i % 17) - 0.003;
    if (w.lane < n) x[w.lane] = 0.01 * w.lane;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    (void)dinv;   // (the in-place register Cholesky this slot used to time was superseded by chol_inv_packed)
    long long t1 = __builtin_readcyclecounter();
    w.trsm_lower_rows(nrhs, n, P, dinv, Y, ldy);
    long long t2 = __builtin_readcyclecounter();
    w.trsv_upper(n, P, dinv, x);
    long long t3 = __builtin_readcyclecounter();
    t_chol += t1 - t0; t_trsm += t2 - t1; t_trsv += t3 - t2;
    for (int i = w.lane; i < tri; i += 64) P[i] = M0[i];
    __syncthreads();
    long long t4 = __builtin_readcyclecounter();
    w.chol_inv_packed(n, P, 1e-15);
    long long t5 = __builtin_readcyclecounter();
    t_inv += t5 - t4;
  }
  if (w.lane == 0 && blockIdx.x == 0) { out[0] = t_chol / reps; out[1] = t_trsm / reps; out[2] = t_trsv / reps; out[5] = t_inv / reps; }
  if (w.lane == 0 && blockIdx.x == 1) out[4] = (unsigned long long)(Y[3] * 1e9) + (unsigned long long)(x[1] * 1e9);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 24, nrhs = argc > 2 ? atoi(argv[2]) : 16, blocks = argc > 3 ? atoi(argv[3]) : 1536;
  const int tri = n * (n + 1) / 2;
  std::vector<double> M(tri);
  for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) M[i * (i + 1) / 2 + j] = (i == j) ? 2.0 + 0.1 * i : 0.3 / (1 + i - j);
  double* dM; unsigned long long* dout;
  hipMalloc(&dM, tri * 8); hipMalloc(&dout, 64);
  hipMemcpy(dM, M.data(), tri * 8, hipMemcpyHostToDevice); hipMemset(dout, 0, 64);
  const size_t lds = 8 * (tri + 64 + (nrhs + 1) * (n | 1) + 64) + (argc > 4 ? atoi(argv[4]) : 0);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64), lds, 0, n, nrhs, 50, dM, dout);
  hipDeviceSynchronize();
  unsigned long long o[8];
  hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
  printf("n %d nrhs %d blocks %d lds %zu: chol %llu cycles, trsm %llu, trsv_upper %llu, chol+inverse %llu  (fallback flag %llu)\n", n, nrhs, blocks, lds, o[0], o[1], o[2], o[5], o[3]);
  return 0;
}
