// tools/micro/lds_pattern_lab.hip -- diagnostic: LDS cycles of the staging stores (ds_write_b64, thread = (row, quarter of a 16-f16
// k-tile row)) and of the matrix-operand reads (ds_read_b128, lane = (row li, k-half lh)) of the two-piece product kernels, for
//   (a) the layout in use: rows of 48 bytes (32 + 16 pad)            -- reads conflict-free, stores 2-way on a quarter of the banks
//   (b) rows of 32 bytes, the two 16-byte halves swapped in rows 4..7 of every eight (XOR swizzle): both conflict-free, 1/3 less LDS
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_pattern_lab.exe tools/micro/lds_pattern_lab.hip && tools/micro/lds_pattern_lab.exe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int t = threadIdx.x, lane = t & 63, li = lane & 31, lh = lane >> 5;
  const int kq = t & 3, r4 = t >> 2;                       // staging thread: row r4 (0..63), quarter kq
  int woff, roff;
  if (MODE == 0) { woff = r4 * 48 + 8 * kq; roff = li * 48 + 16 * lh; }
  else { woff = r4 * 32 + ((((kq >> 1) ^ (r4 >> 2)) & 1) * 16) + (kq & 1) * 8; roff = li * 32 + (((lh ^ (li >> 2)) & 1) * 16); }
  uint2 v = make_uint2(t, t * 3);
  uint4 acc = make_uint4(0, 0, 0, 0);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) { *reinterpret_cast<uint2*>(lds + woff + u * 4096) = v; asm volatile("" ::: "memory"); }
  }
  __syncthreads();
  long long t1 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      asm volatile("" ::: "memory");
      const uint4 x = *reinterpret_cast<const uint4*>(lds + roff + u * 4096);
      acc.x += x.x; acc.y ^= x.y; acc.z += x.z; acc.w ^= x.w;
    }
  }
  __syncthreads();
  long long t2 = __builtin_readcyclecounter();
  if (t == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
  if (acc.x == 0x12345u) out[2] = acc.y + acc.z + acc.w;
}

int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  const int iters = 2000;
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
    }
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%s: %.1f ticks per ds_write_b64 x 4 waves, %.1f ticks per ds_read_b128 x 4 waves (s_memtime ticks; ratios matter)\n",
           mode == 0 ? "rows of 48 B (in use)      " : "rows of 32 B, XOR swizzled", (double)h[0] / (iters * 8), (double)h[1] / (iters * 8));
  }
  return 0;
}
