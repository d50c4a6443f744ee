#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* t, double a, double b, int n) {
  __shared__ double sm[256];
  const int lane = threadIdx.x;
  sm[lane] = a + lane; sm[lane + 64] = b; __syncthreads();
  double x = a + lane;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) { x = __builtin_fma(x, b, a); x = __builtin_fma(x, b, a); x = __builtin_fma(x, b, a); x = __builtin_fma(x, b, a); }
  long long t1 = __builtin_readcyclecounter();
  // readlane -> mul -> fma chain
  double y = a + lane;
  for (int i = 0; i < n; i++) {
    const int l = i & 31;
    const int lo = __builtin_amdgcn_readlane(__double2loint(y), l), hi = __builtin_amdgcn_readlane(__double2hiint(y), l);
    const double s = __hiloint2double(hi, lo);
    y = __builtin_fma(s, b, y);
  }
  long long t2 = __builtin_readcyclecounter();
  // LDS round trip chain: write then read by neighbour lane
  double z = a;
  for (int i = 0; i < n; i++) { sm[lane] = z; __syncthreads(); z = sm[(lane + 1) & 63] + 1.0; __syncthreads(); }
  long long t3 = __builtin_readcyclecounter();
  // dependent f32 fma chain
  float f = (float)a + lane; const float fb = (float)b, fa = (float)a;
  for (int i = 0; i < n; i++) { f = __builtin_fmaf(f, fb, fa); f = __builtin_fmaf(f, fb, fa); f = __builtin_fmaf(f, fb, fa); f = __builtin_fmaf(f, fb, fa); }
  long long t4 = __builtin_readcyclecounter();
  // rsqrt chain
  double r = a + 2.0;
  for (int i = 0; i < n; i++) { r = rsqrt(r) + 1.5; }
  long long t5 = __builtin_readcyclecounter();
  // division chain
  double d = a + 2.0;
  for (int i = 0; i < n; i++) { d = 1.0 / d + 1.5; }
  long long t6 = __builtin_readcyclecounter();
  out[lane] = x + y + z + f + r + d;
  if (lane == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; t[5] = t6 - t5; }
}
int main() {
  double* out; unsigned long long* t; hipMalloc(&out, 512); hipMalloc(&t, 64);
  const int n = 1000;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 0.5, 0.999, n);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 0.5, 0.999, n);
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
  printf("dependent f64 fma: %.1f cycles/op; readlane+fma step: %.1f; LDS write->read round trip: %.1f; f32 fma: %.1f; rsqrt+add: %.1f; div+add: %.1f (s_memtime units)\n",
         h[0] / (4.0 * n), h[1] / (double)n, h[2] / (double)n, h[3] / (4.0 * n), h[4] / (double)n, h[5] / (double)n);
  // wall-clock calibration of s_memtime
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 0.5, 0.999, 100000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
  unsigned long long tot = h[0] + h[1] + h[2] + h[3] + h[4] + h[5];
  printf("calibration: %llu counter units in %.3f ms -> %.1f MHz\n", tot, ms, tot / (ms * 1e3));
  return 0;
}
