// stream_pick.h -- choose helper streams that REALLY run concurrently with a given stream (gfx950 / ROCm 7).
//
// HIP multiplexes its streams onto a handful of hardware queues (4 by default); which queue a new stream lands on depends on
// every stream the process created before -- PyTorch's pools, RCCL's channels after init_process_group("nccl").  Two streams on
// one hardware queue execute strictly one after the other: the SET forward's side stream then overlaps nothing (measured:
// forward 2.95 -> 3.27 ms as soon as an RCCL process group exists: tools/diag/queue_ids.py), and launch groups of the step
// kernel run back to back.  HIP has no query for the mapping, so it is measured: two single-wave spin kernels, one per stream,
// take one spin time when the queues differ and two when they are the same.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

namespace sgrl_streams {

static __global__ void k_spin(long long ticks) {          // wall_clock64: the 100 MHz constant counter
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

constexpr long long kSpinTicks = 8000;              // 80 us

// true: `a` and `b` execute concurrently.  Synchronises both streams; never call while either is being captured.
inline bool overlap(hipStream_t a, hipStream_t b) {
  hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
      hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return true;                                    // cannot measure: keep what we have
  }
  float best = 1e9f;
  for (int rep = 0; rep < 2; rep++) {               // first pass also pays the kernel's code upload
    (void)hipEventRecord(e0, a);
    (void)hipStreamWaitEvent(b, e0, 0);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, kSpinTicks);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, kSpinTicks);
    (void)hipEventRecord(eb, b);
    (void)hipStreamWaitEvent(a, eb, 0);
    (void)hipEventRecord(e1, a);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(eb);
  (void)hipGetLastError();
  const float spin_ms = (float)kSpinTicks / 100000.f;   // 100 ticks per microsecond
  return best < 1.6f * spin_ms;
}

// A non-blocking stream that overlaps every stream in `with`: `have` if it already does (or is the best there is), else a new
// one (then `have`, if not null, is destroyed).  At most `tries` candidates are created; the rejected ones are kept alive until
// the search ends so that the runtime does not hand the same queue out again.
inline hipStream_t pick(const std::vector<hipStream_t>& with, hipStream_t have, int tries = 8) {
  auto good = [&](hipStream_t s) {
    for (hipStream_t w : with)
      if (!overlap(w, s)) return false;
    return true;
  };
  if (have && good(have)) return have;
  std::vector<hipStream_t> rejected;
  hipStream_t found = nullptr;
  for (int t = 0; t < tries && !found; t++) {
    hipStream_t c = nullptr;
    if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
    if (good(c)) found = c; else rejected.push_back(c);
  }
  for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
  if (!found) return have;                          // every queue taken: sequential execution is still correct
  if (have) (void)hipStreamDestroy(have);
  return found;
}

inline bool enabled() {
  return true;
}

}  // namespace sgrl_streams
