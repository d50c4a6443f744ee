// tools/gemm_probe.hip -- diagnostic only: times the SET actor's MFMA GEMM on the shapes of one forward.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/gemm_probe tools/gemm_probe.hip && gpurun_out/gemm_probe
#include "../sgrl_amd/csrc/set_actor.hip"

#include <cstdio>
#include <cstdlib>

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
  hipMemset(A, 0, maxA * 4); hipMemset(W, 0, 1024 * 544 * 4); hipMemset(bias, 0, 4096); hipMemset(C2, 0, maxC * 4);
  {
    std::vector<float> one((size_t)3 * N0, 1.f);
    hipMemcpy(rd, one.data(), one.size() * 4, hipMemcpyHostToDevice);
  }
  sgrl_set* s = nullptr;
  if (sgrl_set_create(&s) != 0) { printf("create failed: %s\n", sgrl_set_last_error()); return 1; }
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  for (const Shape& sh : shapes) {
    for (int w = 0; w < 3; w++) launch_gemm(0, A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N);
    hipEventRecord(t0, 0);
    const int reps = 20;
    for (int r = 0; r < reps; r++) launch_gemm(0, A, sh.K, W, sh.K, bias, C, sh.N, sh.M, sh.N, sh.K, sh.flags, rd, C2, sh.N);
    hipEventRecord(t1, 0);
    hipEventSynchronize(t1);
    float ms = 0;
    hipEventElapsedTime(&ms, t0, t1);
    ms /= reps;
    printf("%s M %6d N %4d K %3d : %7.1f us  %6.1f TFLOP/s\n", sh.name, sh.M, sh.N, sh.K, ms * 1e3,
           2.0 * sh.M * sh.N * sh.K / (ms * 1e-3) / 1e12);
  }
  return 0;
}
