// tools/micro/halfwave_prims.hip -- device check of the gfx950 encodings behind wave_hip.h HipHalfPrim (the half-wave primitives of
// csrc/wave_half.h): row_newbcast, quad_perm exchange, the half reductions through v_permlane16_swap, the ballot halves -- with
// both halves active and with ONE half masked off (the state the two environments' divergent branches run in).  The algorithms
// built on the primitives are checked on the CPU (tests/test_engine_pair_emu.py); this is the part the CPU cannot see.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sgrl_amd/csrc -o tools/micro/halfwave_prims.exe tools/micro/halfwave_prims.hip && tools/micro/halfwave_prims.exe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#define SGRL_CONST_AS __attribute__((address_space(4)))
#define SGRL_ITAB_AS __attribute__((address_space(3)))
#define SGRL_FTAB_AS __attribute__((address_space(4)))
#include "wave_hip.h"

using Prim = sgrl::HipHalfPrim<sgrl::DimsAny>;

// out[test][lane]; mode 0: both halves run, 1: only half 0, 2: only half 1 (divergent branch on the half id)
__global__ void k(const double* in, double* out, unsigned* bal, int mode) {
  Prim p;
  const int t = threadIdx.x;
  const double x = in[t];
  auto body = [&]() {
    for (int j = 0; j < 16; j++) out[j * 64 + t] = Prim::bcast16(x, j);
    out[16 * 64 + t] = Prim::xor1(x);
    out[17 * 64 + t] = Prim::half_sum(x);
    out[18 * 64 + t] = Prim::half_max(x);
    bal[t] = p.half_ballot(((t * 7) % 5) < 2);
  };
  if (mode == 0) body();
  else if (mode == 1) { if (p.half == 0) body(); }
  else { if (p.half == 1) body(); }
}

int main() {
  std::vector<double> in(64), out(19 * 64);
  std::vector<unsigned> bal(64);
  for (int i = 0; i < 64; i++) in[i] = std::sin(1.0 + i) * 100.0 + i;
  double *din, *dout; unsigned* dbal;
  hipMalloc(&din, 64 * 8); hipMalloc(&dout, 19 * 64 * 8); hipMalloc(&dbal, 64 * 4);
  hipMemcpy(din, in.data(), 64 * 8, hipMemcpyHostToDevice);
  int bad = 0;
  for (int mode = 0; mode < 3; mode++) {
    hipMemset(dout, 0xff, 19 * 64 * 8); hipMemset(dbal, 0xff, 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dbal, mode);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(out.data(), dout, 19 * 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(bal.data(), dbal, 64 * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < 64; t++) {
      const int h = t >> 5, l = t & 31;
      if ((mode == 1 && h == 1) || (mode == 2 && h == 0)) continue;      // masked half: nothing written
      for (int j = 0; j < 16; j++) {
        const double want = in[(t & ~15) | j];                            // lane j of the lane's own 16-lane row
        if (out[j * 64 + t] != want) { if (bad++ < 10) printf("mode %d bcast16 j=%d lane %d: %g != %g\n", mode, j, t, out[j * 64 + t], want); }
      }
      if (out[16 * 64 + t] != in[t ^ 1]) { if (bad++ < 10) printf("mode %d xor1 lane %d\n", mode, t); }
      double s = 0, mx = -1e300;
      // the device association: butterflies inside each row, then row totals added
      double rows[2];
      for (int r = 0; r < 2; r++) {
        double v[16];
        for (int i = 0; i < 16; i++) v[i] = in[32 * h + 16 * r + i];
        double a[16];
        for (int i = 0; i < 16; i++) a[i] = v[i] + v[i ^ 1];
        for (int i = 0; i < 16; i++) v[i] = a[i] + a[i ^ 2];
        for (int i = 0; i < 16; i++) a[i] = v[i] + v[(i & ~7) | (7 - (i & 7))];
        for (int i = 0; i < 16; i++) v[i] = a[i] + a[15 - i];
        rows[r] = v[l & 15];
      }
      s = rows[0] + rows[1];
      for (int i = 0; i < 32; i++) mx = std::fmax(mx, in[32 * h + i]);
      if (std::fabs(out[17 * 64 + t] - s) > 1e-9) { if (bad++ < 10) printf("mode %d half_sum lane %d: %.17g != %.17g\n", mode, t, out[17 * 64 + t], s); }
      if (out[18 * 64 + t] != mx) { if (bad++ < 10) printf("mode %d half_max lane %d: %g != %g\n", mode, t, out[18 * 64 + t], mx); }
      unsigned wb = 0;
      for (int i = 0; i < 32; i++) if ((((32 * h + i) * 7) % 5) < 2) wb |= 1u << i;
      if (bal[t] != wb) { if (bad++ < 10) printf("mode %d ballot lane %d: %08x != %08x\n", mode, t, bal[t], wb); }
    }
  }
  printf(bad ? "halfwave_prims: %d MISMATCHES\n" : "halfwave_prims: all primitives ok (both halves, half 0 alone, half 1 alone)\n", bad);
  return bad ? 1 : 0;
}
