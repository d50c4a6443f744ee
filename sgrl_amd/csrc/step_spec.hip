// step_spec.hip -- ONE fixed-dimension kernel of the step (engine_kernel.h DimsFixed / FamilyRun): the dimension sets of one
// morphology family as compile-time constants, the workgroup's morphology picks its instance.  Compiled once per family by
// sgrl_amd/_lib.py build(), with
//   -DSGRL_SPEC_ID=<k> -DSGRL_SPEC_WAVES=<waves per SIMD> -DSGRL_SPEC_FAMILY=DimsFixed<nbody,njnt,nq,nv,nu,ngeom,npair,integrator,frame_skip,max_rows,solver>,DimsFixed<...>,...
// (the list lives in spec_table.inc, generated from sgrl_amd/assets/models by the same function).  Exports one launcher;
// engine.hip picks it for the environments whose morphology header matches one of the dimension sets field for field, and keeps
// the generic kernel for everything else (custom XMLs, custom row caps).  gfx950 only.
#include "engine_kernel.h"

#ifndef SGRL_SPEC_ID
#error "compile with -DSGRL_SPEC_ID=<k> -DSGRL_SPEC_WAVES=<w> -DSGRL_SPEC_FAMILY=DimsFixed<...>,..."
#endif
#define SGRL_CAT2(a, b) a##b
#define SGRL_CAT(a, b) SGRL_CAT2(a, b)

namespace {
using namespace sgrl_engine_dev;

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SGRL_SPEC_WAVES, SGRL_SPEC_WAVES))) void k_env_step_spec(BatchArgs a, StepOut out) {
  const int env = __builtin_amdgcn_readfirstlane(a.block_env[blockIdx.x]);
  const int mate = __builtin_amdgcn_readfirstlane(a.block_mate[blockIdx.x]);
  const int slot = __builtin_amdgcn_readfirstlane(a.morphs[__builtin_amdgcn_readfirstlane(a.env_morph[env])].slot);
  FamilyRun<0, SGRL_SPEC_FAMILY>::run(slot, env, mate, a, out);
}
}  // namespace

// launch `n_wg` single-wave workgroups with `lds` bytes of dynamic LDS on `stream`; returns the hipError_t of the launch
extern "C" int SGRL_CAT(sgrl_spec_launch_, SGRL_SPEC_ID)(int n_wg, int lds, void* stream, const void* batch_args, const void* step_out) {
  static int lds_limit[64];
  static bool init = false;
  if (!init) { for (int& v : lds_limit) v = 48 * 1024; init = true; }
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& limit = lds_limit[dev >= 0 && dev < 64 ? dev : 0];
  if (lds > limit) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_env_step_spec), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    limit = lds;
  }
  hipLaunchKernelGGL(k_env_step_spec, dim3(n_wg), dim3(64), lds, (hipStream_t)stream, *reinterpret_cast<const BatchArgs*>(batch_args),
                     *reinterpret_cast<const StepOut*>(step_out));
  return (int)hipGetLastError();
}
