// engine_kernel.h -- device side of the rollout engine shared by engine.hip (generic kernels: dimensions read from the
// morphology header at run time) and step_spec.hip (one translation unit per morphology DIMENSION SET of the shipped assets:
// the same kernel with body / joint / dof / pair counts, integrator and row cap as compile-time constants).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#ifndef SGRL_ITAB_GLOBAL
#define SGRL_ITAB_GLOBAL 0
#endif
#ifndef SGRL_STAGE_FLOATS
#define SGRL_STAGE_FLOATS 0   // 1: also stage the float model tables in LDS (costs ~6 KB of slab per workgroup)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// address spaces of the model view (include/sgrl_model.h): header copies in constant memory (uniform -> s_load),
// int tables in LDS, float tables in constant memory (or LDS when staged)
#define SGRL_CONST_AS __attribute__((address_space(4)))
// SGRL_ITAB_GLOBAL=1 (a per-family compile flag, sgrl_amd/_lib.py ITAB_GLOBAL): the int tables are read where they lie (constant
// address space: L2 / scalar cache) instead of from an LDS copy -- 2.4-3.4 KB less slab for the families whose resident count hangs
// on it; the host computes the same layout (engine.hip: n_int = 0 for such a family's members)
#if SGRL_ITAB_GLOBAL
#define SGRL_ITAB_AS __attribute__((address_space(4)))
#else
#define SGRL_ITAB_AS __attribute__((address_space(3)))
#endif
#if SGRL_STAGE_FLOATS
#define SGRL_FTAB_AS __attribute__((address_space(3)))
#else
#define SGRL_FTAB_AS __attribute__((address_space(4)))
#endif
#endif
#include "../../include/sgrl.h"
#include "step_body.h"
#include "wave_hip.h"
#include "wave_half.h"

namespace sgrl_engine_dev {

using sgrl::DimsAny;
using sgrl::DimsFixed;

struct MorphDev {
  const int32_t* ib;
  const double* fb;
  int32_t slot;      // which instance of its family's fixed-dimension kernel serves this morphology (-1: generic kernel only)
  int32_t pad_;
};

struct BatchArgs {
  const MorphDev* morphs;     // [n_morph]
  const int32_t* env_morph;   // [n_env]
  const int32_t* block_env;   // [n_env] workgroup -> env, most expensive morphologies first (tail balance)
  const int32_t* block_mate;  // fixed-dimension step kernels: [n_workgroups] the SECOND environment of a workgroup that steps two
                              // environments of one light morphology (wave_half.h), -1 for one environment; null elsewhere
  double* rec;                // [n_env * stride]
  int32_t* cnt;               // [n_env * 4]
  double* scratch;            // [n_env * scratch_stride] HBM slabs for the constraint solves with more rows than the LDS arrays hold
  int scratch_stride;         // doubles per environment: max over the batch's morphologies of slab_doubles(max_rows, ldy)
  int stride;
  int n_env;
  int obs_max_len, action_max_len;
  uint64_t seed;
  uint32_t env_id_base;
  int max_episode_steps;
};

struct StepOut {
  const float* actions;
  float* obs32;
  double* obs64;
  float* reward;
  double* reward64;
  uint8_t* done;
  float* dist;
  uint8_t* truncated;
  int auto_reset;
};


#ifdef SGRL_PHASE_PROF
__device__ unsigned long long g_phase_prof[16 * 65536];
__device__ __forceinline__ unsigned long long* sgrl_phase_prof_buf() { return g_phase_prof; }
#endif

extern __shared__ double sgrl_lds[];

// Stage the morphology tables (a few KB, shared by all envs of the morphology, L2 resident) into this workgroup's
// LDS and point the model view at the copy: every table lookup of the ~16 dynamics evaluations then costs an LDS
// access instead of an L2 round trip.
template <class D>
__device__ __forceinline__ void setup(const BatchArgs& a, int env, SgrlModelView* m, sgrl::Layout* o, double** S, int32_t** I) {
  const int mi = __builtin_amdgcn_readfirstlane(a.env_morph[env]);
  const MorphDev md = a.morphs[mi];
  // The header is wave-uniform, but it arrives through vector loads (global memory the kernel also writes), so the
  // compiler would keep every table offset derived from it as a per-lane value (~150 registers).  readfirstlane makes
  // the sizes scalar: the whole layout / model view then lives in SGPRs.
  int32_t hdr[SGRL_NHDR];
#pragma unroll
  for (int k = 0; k < SGRL_NHDR; k++) hdr[k] = __builtin_amdgcn_readfirstlane(md.ib[k]);
  D::apply(hdr);          // a fixed-dimension instance: the counts become compile-time constants from here on
  int n_int, n_f64;
  sgrl_model_blob_sizes(hdr, &n_int, &n_f64);
  // integer tables (paths, masks, parents: walked in inner loops) are staged in LDS; the float tables are read once
  // per evaluation per lane and stay in L2 -- the 6 KB they would cost in LDS buy a fifth workgroup per CU instead
  sgrl::make_layout(hdr, o, (D::kFixed && SGRL_ITAB_GLOBAL) ? 0 : n_int, SGRL_STAGE_FLOATS ? n_f64 : 0);
  double* s = sgrl_lds;
  int32_t* ii = reinterpret_cast<int32_t*>(sgrl_lds + o->s_total);
  const int lane = threadIdx.x;
  if (SGRL_STAGE_FLOATS) for (int k = lane; k < n_f64; k += 64) s[o->model_f + k] = md.fb[k];
#if !SGRL_ITAB_GLOBAL
  for (int k = lane; k < n_int; k += 64) ii[o->model_i + k] = md.ib[k];
#endif
  __syncthreads();
  // the view: sizes from the scalar header, header constants through constant-memory pointers (scalar loads), int
  // tables from the LDS copy
#if SGRL_STAGE_FLOATS
  sgrl_ftab_t ftab = (sgrl_ftab_t)(s + o->model_f);
#else
  sgrl_ftab_t ftab = (sgrl_ftab_t)md.fb;
#endif
#if SGRL_ITAB_GLOBAL
  sgrl_model_view_dims(hdr, (sgrl_hdr_t)md.ib, (sgrl_fhdr_t)md.fb, (sgrl_itab_t)md.ib, ftab, m);
#else
  sgrl_model_view_dims(hdr, (sgrl_hdr_t)md.ib, (sgrl_fhdr_t)md.fb, (sgrl_itab_t)(ii + o->model_i), ftab, m);
#endif
  *S = s;
  *I = ii;
}

__device__ __forceinline__ sgrl::StepIO make_io(const BatchArgs& a, const StepOut& out, int env) {
  sgrl::StepIO io;
  io.rec = a.rec + (size_t)env * a.stride;
  io.cnt = a.cnt + (size_t)env * 4;
  io.action = out.actions ? out.actions + (size_t)env * a.action_max_len : nullptr;
  io.obs32 = out.obs32 ? out.obs32 + (size_t)env * a.obs_max_len : nullptr;
  io.obs64 = out.obs64 ? out.obs64 + (size_t)env * a.obs_max_len : nullptr;
  io.reward = out.reward ? out.reward + env : nullptr;
  io.reward64 = out.reward64 ? out.reward64 + env : nullptr;
  io.done = out.done ? out.done + env : nullptr;
  io.dist = out.dist ? out.dist + env : nullptr;
  io.truncated = out.truncated ? out.truncated + env : nullptr;
  io.obs_max_len = a.obs_max_len;
  io.scratch = a.scratch ? a.scratch + (size_t)env * a.scratch_stride : nullptr;
  io.seed = a.seed;
  io.env_id = a.env_id_base + (uint32_t)env;
  io.max_episode_steps = a.max_episode_steps;
  io.auto_reset = out.auto_reset;
  return io;
}


// the step of one environment by one wavefront (kernel body shared by every instance)
template <class D, class W>
__device__ __forceinline__ void env_step_wave(const BatchArgs& a, const StepOut& out, int env) {
  SgrlModelView m; sgrl::Layout o; double* S; int32_t* I;
  setup<D>(a, env, &m, &o, &S, &I);
  W w;
#ifdef SGRL_PHASE_PROF
  w.prof = sgrl_phase_prof_buf() + 16 * (size_t)(env & 65535);
  const long long t_begin = __builtin_readcyclecounter();
#endif
  const sgrl::StepIO io = make_io(a, out, env);
  sgrl::env_step(w, m, o, S, I, io);
#ifdef SGRL_PHASE_PROF
  if (w.lane == 0) w.prof[15] += (unsigned long long)(__builtin_readcyclecounter() - t_begin);
#endif
}

// The step of TWO environments of one morphology by one wavefront (wave_half.h): lanes 0..31 environment `env_a` on the first LDS
// slab, lanes 32..63 `env_b` on the second, one shared copy of the int tables behind both (Layout::pair_stride).  The model view,
// the layout and every dimension stay wave-uniform; the slab pointers and the StepIO are per lane.
template <class D, class W>
__device__ __forceinline__ void env_step_pair(const BatchArgs& a, const StepOut& out, int env_a, int env_b) {
  static_assert(!SGRL_STAGE_FLOATS && !SGRL_ITAB_GLOBAL, "the pair layout shares the LDS copy of the int tables only");
  const int mi = __builtin_amdgcn_readfirstlane(a.env_morph[env_a]);     // the host pairs environments of ONE morphology
  const MorphDev md = a.morphs[mi];
  int32_t hdr[SGRL_NHDR];
#pragma unroll
  for (int k = 0; k < SGRL_NHDR; k++) hdr[k] = __builtin_amdgcn_readfirstlane(md.ib[k]);
  D::apply(hdr);
  int n_int, n_f64;
  sgrl_model_blob_sizes(hdr, &n_int, &n_f64);
  sgrl::Layout o;
  sgrl::make_layout(hdr, &o, n_int, 0, true);
  int32_t* const ia = reinterpret_cast<int32_t*>(sgrl_lds + o.s_total);
  for (int k = threadIdx.x; k < n_int; k += 64) ia[o.model_i + k] = md.ib[k];
  __syncthreads();
  SgrlModelView m;
  sgrl_model_view_dims(hdr, (sgrl_hdr_t)md.ib, (sgrl_fhdr_t)md.fb, (sgrl_itab_t)(ia + o.model_i), (sgrl_ftab_t)md.fb, &m);
  W w;
  double* const S = sgrl_lds + w.half * o.pair_stride;
  int32_t* const I = ia + 2 * w.half * o.pair_stride;
  const sgrl::StepIO io = make_io(a, out, w.half ? env_b : env_a);
  sgrl::env_step(w, m, o, S, I, io);
}

// one kernel, several dimension sets: the workgroup's morphology says which instance runs (wave-uniform branch); a workgroup
// with a mate (>= 0) steps two environments on the half-wave instance of its set
template <int I, class... Ds> struct FamilyRun {
  __device__ static __forceinline__ void run(int, int, int, const BatchArgs&, const StepOut&) {}
};
template <int I, class D, class... Rest> struct FamilyRun<I, D, Rest...> {
  __device__ static __forceinline__ void run(int slot, int env, int mate, const BatchArgs& a, const StepOut& out) {
    if (slot == I) {
      if constexpr (D::kPair) {
        if (mate >= 0) { env_step_pair<D, sgrl::HalfWaveT<D::kNv, sgrl::HipHalfPrim<D>>>(a, out, env, mate); return; }
      }
      env_step_wave<D, sgrl::HipWaveT<(D::kNv <= 24 ? D::kNv : 24), D>>(a, out, env);
      return;
    }
    FamilyRun<I + 1, Rest...>::run(slot, env, mate, a, out);
  }
};

}  // namespace sgrl_engine_dev
