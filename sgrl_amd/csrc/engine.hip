// engine.hip -- HIP kernels + C ABI (include/sgrl.h) of the batched rollout engine.  gfx950 only.
//
// Launch geometry: one 64-thread workgroup (= one wavefront) per environment; dynamic LDS = the largest
// per-environment slab over the morphologies in the batch (step_body.h Layout).  Per-environment persistent
// state is an array of records in HBM: one wave reads and writes one contiguous record (coalesced), the
// observation row [env][obs_max_len] is written once with consecutive lanes on consecutive floats.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifndef SGRL_STAGE_FLOATS
#define SGRL_STAGE_FLOATS 0   // 1: also stage the float model tables in LDS (costs ~6 KB of slab per workgroup)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// address spaces of the model view (include/sgrl_model.h): header copies in constant memory (uniform -> s_load),
// int tables in LDS, float tables in constant memory (or LDS when staged)
#define SGRL_CONST_AS __attribute__((address_space(4)))
#define SGRL_ITAB_AS __attribute__((address_space(3)))
#if SGRL_STAGE_FLOATS
#define SGRL_FTAB_AS __attribute__((address_space(3)))
#else
#define SGRL_FTAB_AS __attribute__((address_space(4)))
#endif
#endif
#include "../../include/sgrl.h"
#include "step_body.h"
#include "stream_pick.h"
#include "wave_hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(SGRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                \
  } while (0)

struct MorphDev {
  const int32_t* ib;
  const double* fb;
};

struct BatchArgs {
  const MorphDev* morphs;     // [n_morph]
  const int32_t* env_morph;   // [n_env]
  const int32_t* block_env;   // [n_env] workgroup -> env, most expensive morphologies first (tail balance)
  double* rec;                // [n_env * stride]
  int32_t* cnt;               // [n_env * 4]
  double* scratch;            // [n_env * kScratchDoubles] HBM slabs for the rare > 32-row constraint solves
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


extern __shared__ double sgrl_lds[];

// Stage the morphology tables (a few KB, shared by all envs of the morphology, L2 resident) into this workgroup's
// LDS and point the model view at the copy: every table lookup of the ~16 dynamics evaluations then costs an LDS
// access instead of an L2 round trip.
__device__ __forceinline__ void setup(const BatchArgs& a, int env, SgrlModelView* m, sgrl::Layout* o, double** S, int32_t** I) {
  const int mi = __builtin_amdgcn_readfirstlane(a.env_morph[env]);
  const MorphDev md = a.morphs[mi];
  // The header is wave-uniform, but it arrives through vector loads (global memory the kernel also writes), so the
  // compiler would keep every table offset derived from it as a per-lane value (~150 registers).  readfirstlane makes
  // the sizes scalar: the whole layout / model view then lives in SGPRs.
  int32_t hdr[SGRL_NHDR];
#pragma unroll
  for (int k = 0; k < SGRL_NHDR; k++) hdr[k] = __builtin_amdgcn_readfirstlane(md.ib[k]);
#ifdef SGRL_FIX_DIMS_W7
  // diagnostic build only (tools/diag/variant_probe.py): the dimensions of 3d_walker_7_full as compile-time constants, to price
  // what a per-morphology specialisation of the kernel (layout offsets as immediates, no scalar-register spills) would buy
  hdr[SGRL_H_NBODY] = 8; hdr[SGRL_H_NJNT] = 19; hdr[SGRL_H_NQ] = 25; hdr[SGRL_H_NV] = 24; hdr[SGRL_H_NU] = 18;
  hdr[SGRL_H_NGEOM] = 8; hdr[SGRL_H_NPAIR] = 7; hdr[SGRL_H_INTEGRATOR] = 1; hdr[SGRL_H_FRAME_SKIP] = 4; hdr[SGRL_H_MAX_ROWS] = 48;
  hdr[SGRL_H_SOLVER] = 1;
#endif
  int n_int, n_f64;
  sgrl_model_blob_sizes(hdr, &n_int, &n_f64);
  // integer tables (paths, masks, parents: walked in inner loops) are staged in LDS; the float tables are read once
  // per evaluation per lane and stay in L2 -- the 6 KB they would cost in LDS buy a fifth workgroup per CU instead
  sgrl::make_layout(hdr, o, n_int, SGRL_STAGE_FLOATS ? n_f64 : 0);
  double* s = sgrl_lds;
  int32_t* ii = reinterpret_cast<int32_t*>(sgrl_lds + o->s_total);
  const int lane = threadIdx.x;
  if (SGRL_STAGE_FLOATS) for (int k = lane; k < n_f64; k += 64) s[o->model_f + k] = md.fb[k];
  for (int k = lane; k < n_int; k += 64) ii[o->model_i + k] = md.ib[k];
  __syncthreads();
  // the view: sizes from the scalar header, header constants through constant-memory pointers (scalar loads), int
  // tables from the LDS copy
#if SGRL_STAGE_FLOATS
  sgrl_ftab_t ftab = (sgrl_ftab_t)(s + o->model_f);
#else
  sgrl_ftab_t ftab = (sgrl_ftab_t)md.fb;
#endif
  sgrl_model_view_from((sgrl_hdr_t)md.ib, (sgrl_fhdr_t)md.fb, (sgrl_itab_t)(ii + o->model_i), ftab, m);
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
  io.scratch = a.scratch ? a.scratch + (size_t)env * sgrl::kScratchDoubles : nullptr;
  io.seed = a.seed;
  io.env_id = a.env_id_base + (uint32_t)env;
  io.max_episode_steps = a.max_episode_steps;
  io.auto_reset = out.auto_reset;
  return io;
}

#ifdef SGRL_PHASE_PROF
__device__ unsigned long long g_phase_prof[16 * 65536];
#endif

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_env_step(BatchArgs a, StepOut out) {
  const int env = __builtin_amdgcn_readfirstlane(a.block_env[blockIdx.x]);
  SgrlModelView m; sgrl::Layout o; double* S; int32_t* I;
  setup(a, env, &m, &o, &S, &I);
  sgrl::HipWave w;
#ifdef SGRL_PHASE_PROF
  w.prof = g_phase_prof + 16 * (size_t)(env & 65535);
  const long long t_begin = __builtin_readcyclecounter();
#endif
  const sgrl::StepIO io = make_io(a, out, env);
  sgrl::env_step(w, m, o, S, I, io);
#ifdef SGRL_PHASE_PROF
  if (w.lane == 0) w.prof[15] += (unsigned long long)(__builtin_readcyclecounter() - t_begin);
#endif
}

// The step kernel for the LIGHT morphologies (nv <= kLightNv: walker_2/3/4, hopper_3/4): the same source with the register
// solvers of the larger dof counts compiled out and a 168-register budget, i.e. three waves per SIMD = 12 workgroups per CU
// instead of 8.  Their slabs (7..13 KB) leave the LDS half empty at 8 per CU, and the kernel is bound by the latency of its
// dependent chains: throughput follows the number of resident environments.
constexpr int kLightNv = 15;
constexpr int kLightPerCu = 12;
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_env_step_light(BatchArgs a, StepOut out) {
  const int env = __builtin_amdgcn_readfirstlane(a.block_env[blockIdx.x]);
  SgrlModelView m; sgrl::Layout o; double* S; int32_t* I;
  setup(a, env, &m, &o, &S, &I);
  sgrl::HipWaveT<kLightNv> w;
  const sgrl::StepIO io = make_io(a, out, env);
  sgrl::env_step(w, m, o, S, I, io);
}

__global__ __launch_bounds__(64) void k_env_reset(BatchArgs a, StepOut out) {
  const int env = __builtin_amdgcn_readfirstlane(a.block_env[blockIdx.x]);
  SgrlModelView m; sgrl::Layout o; double* S; int32_t* I;
  setup(a, env, &m, &o, &S, &I);
  sgrl::HipWave w;
  const sgrl::StepIO io = make_io(a, out, env);
  sgrl::env_reset(w, m, o, S, I, io, true);
}

__global__ __launch_bounds__(64) void k_env_refresh(BatchArgs a, StepOut out) {
  const int env = __builtin_amdgcn_readfirstlane(a.block_env[blockIdx.x]);
  SgrlModelView m; sgrl::Layout o; double* S; int32_t* I;
  setup(a, env, &m, &o, &S, &I);
  sgrl::HipWave w;
  const sgrl::StepIO io = make_io(a, out, env);
  sgrl::env_refresh(w, m, o, S, I, io);
}

}  // namespace

struct sgrl_engine {
  int n_morph = 0, n_env = 0, stride = 0, lds_bytes = 0;
  int obs_max_len = 0, action_max_len = 0;
  std::vector<int32_t*> d_ib;
  std::vector<double*> d_fb;
  MorphDev* d_morphs = nullptr;
  int32_t* d_env_morph = nullptr;
  int32_t* d_block_env = nullptr;
  double* d_rec = nullptr;
  int32_t* d_cnt = nullptr;
  double* d_scratch = nullptr;
  BatchArgs args{};
  // Launch groups: morphologies whose LDS slab admits the same number of workgroups per CU share one launch (its
  // dynamic LDS = the group's largest slab), so a 23-morphology mix is not dragged to the occupancy of its biggest
  // member.  Groups run concurrently on their own streams, forked from / joined to the caller's stream by events.
  struct Group { int first = 0, count = 0, lds = 0; bool light = false; hipStream_t stream = nullptr; hipEvent_t done = nullptr; };
  std::vector<Group> groups;
  bool groups_checked = false;     // group streams measured to sit on distinct hardware queues (stream_pick.h)
  hipEvent_t fork = nullptr;
  std::vector<int> morph_lds;
};

namespace {
template <class K>
int launch_groups(sgrl_engine* e, K kernel_in, const StepOut& out, hipStream_t user, bool is_step = false) {
  auto pick = [&](const sgrl_engine::Group& g) { return (is_step && g.light) ? k_env_step_light : kernel_in; };
  if (e->groups.size() == 1) {
    hipLaunchKernelGGL(pick(e->groups[0]), dim3(e->n_env), dim3(64), e->groups[0].lds, user, e->args, out);
  } else {
    if (!e->groups_checked && sgrl_streams::enabled()) {
      // first multi-group launch: every group's stream on its own hardware queue, as far as the runtime has them
      // (stream_pick.h; the caller's stream only waits meanwhile, so it may share a queue with a group)
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(user, &cap);
      if (cap == hipStreamCaptureStatusNone) {
        std::vector<hipStream_t> taken;
        for (auto& g : e->groups) {
          if (!taken.empty()) g.stream = sgrl_streams::pick(taken, g.stream, 6);
          taken.push_back(g.stream);
        }
        e->groups_checked = true;
      }
    }
    if (hipEventRecord(e->fork, user) != hipSuccess) return fail(SGRL_ERR_HIP, "hipEventRecord(fork) failed");
    for (auto& g : e->groups) {
      BatchArgs a = e->args;
      a.block_env = e->args.block_env + g.first;
      if (hipStreamWaitEvent(g.stream, e->fork, 0) != hipSuccess) return fail(SGRL_ERR_HIP, "hipStreamWaitEvent(fork) failed");
      hipLaunchKernelGGL(pick(g), dim3(g.count), dim3(64), g.lds, g.stream, a, out);
      if (hipEventRecord(g.done, g.stream) != hipSuccess || hipStreamWaitEvent(user, g.done, 0) != hipSuccess)
        return fail(SGRL_ERR_HIP, "cannot join a launch group back to the caller's stream");
    }
  }
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(SGRL_ERR_HIP, std::string("kernel launch failed (") + hipGetErrorName(le) + ": " + hipGetErrorString(le) + ")");
  }
  return SGRL_OK;
}
}  // namespace


// ---- replay push: the rows of ReplayBuffer.add_transition, one launch -----------------------------------------------------
// block[e] = obs[o] | action[a] | next_obs[o] | reward | done | store | morph_id (floats; sgrl.h sgrl_pack_transitions).  A null
// source leaves its columns as they are (the observation half is written before the step overwrites the observation buffer, the
// rest after it).  HBM-bound copy: one workgroup per environment, consecutive lanes on consecutive floats of the row.
struct PackArgs {
  const float* obs; const float* act; const float* nxt; const float* rew; const float* done_f; const uint8_t* done_u8;
  const uint8_t* store; const int64_t* morph; float* block;
  int ld_obs, ld_act, ld_nxt, o, a;
};
__global__ __launch_bounds__(256) void k_pack_transitions(PackArgs p) {
  const size_t e = blockIdx.x;
  const int row = 2 * p.o + p.a + 4;
  float* b = p.block + e * row;
  for (int c = threadIdx.x; c < row; c += 256) {
    if (c < p.o) {
      if (p.obs) b[c] = p.obs[e * p.ld_obs + c];
    } else if (c < p.o + p.a) {
      if (p.act) b[c] = p.act[e * p.ld_act + (c - p.o)];
    } else if (c < 2 * p.o + p.a) {
      if (p.nxt) b[c] = p.nxt[e * p.ld_nxt + (c - p.o - p.a)];
    } else {
      const int k = c - 2 * p.o - p.a;
      if (k == 0) { if (p.rew) b[c] = p.rew[e]; }
      else if (k == 1) { if (p.done_f) b[c] = p.done_f[e]; else if (p.done_u8) b[c] = p.done_u8[e] ? 1.f : 0.f; }
      else if (k == 2) { if (p.store) b[c] = p.store[e] ? 1.f : 0.f; }
      else if (p.morph) b[c] = (float)p.morph[e];
    }
  }
}

// replay ingest: one workgroup per row of the gathered block; the row's morphology picks the ring, slot[] the place in it
__global__ __launch_bounds__(256) void k_ingest_rows(const float* __restrict__ block, int o, int a, const int64_t* __restrict__ slot,
                                                     const sgrl_ring* __restrict__ rings, int n_rings) {
  const size_t r = blockIdx.x;
  const long long sl = slot[r];
  if (sl < 0) return;
  const int row = 2 * o + a + 4;
  const float* b = block + r * row;
  const int k = (int)b[2 * o + a + 3];
  if (k < 0 || k >= n_rings) return;
  const sgrl_ring g = rings[k];
  for (int c = threadIdx.x; c < g.obs_dim; c += 256) {
    g.obs[(size_t)sl * g.obs_dim + c] = b[c];
    g.next_obs[(size_t)sl * g.obs_dim + c] = b[o + a + c];
  }
  for (int c = threadIdx.x; c < g.act_dim; c += 256) g.action[(size_t)sl * g.act_dim + c] = b[o + c];
  if (threadIdx.x == 0) { g.reward[sl] = b[2 * o + a]; g.done[sl] = b[2 * o + a + 1]; }
}

extern "C" {

const char* sgrl_last_error(void) { return g_err.c_str(); }
const char* sgrl_version(void) { return "sgrl-hip 0.1.0 (gfx950)"; }
#ifdef SGRL_PHASE_PROF
int sgrl_phase_prof(unsigned long long* host, int n_env, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_prof), sizeof(unsigned long long) * 16 * (size_t)n_env) != hipSuccess) return -2;
  if (reset) { static unsigned long long z[16 * 65536]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_prof), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#endif

void sgrl_engine_destroy(sgrl_engine* e) {
  if (!e) return;
  for (auto p : e->d_ib) if (p) (void)hipFree(p);
  for (auto p : e->d_fb) if (p) (void)hipFree(p);
  if (e->d_morphs) (void)hipFree(e->d_morphs);
  if (e->d_env_morph) (void)hipFree(e->d_env_morph);
  if (e->d_block_env) (void)hipFree(e->d_block_env);
  if (e->d_rec) (void)hipFree(e->d_rec);
  if (e->d_cnt) (void)hipFree(e->d_cnt);
  if (e->d_scratch) (void)hipFree(e->d_scratch);
  for (auto& g : e->groups) { if (g.stream) (void)hipStreamDestroy(g.stream); if (g.done) (void)hipEventDestroy(g.done); }
  if (e->fork) (void)hipEventDestroy(e->fork);
  delete e;
}

int sgrl_engine_create(int n_morph, const int32_t* const* ib, const int32_t* ib_len, const double* const* fb,
                       const int32_t* fb_len, const int32_t* morph_count, int obs_max_len, int action_max_len,
                       uint64_t seed, uint32_t env_id_base, int max_episode_steps, sgrl_engine** out) {
  if (!out) return fail(SGRL_ERR_ARG, "out is null");
  *out = nullptr;
  if (n_morph <= 0 || !ib || !ib_len || !fb || !fb_len || !morph_count) return fail(SGRL_ERR_ARG, "null or empty morphology list");
  int ndev = 0;
  (void)hipGetLastError();
  const hipError_t dev_err = hipGetDeviceCount(&ndev);
  if (dev_err != hipSuccess || ndev == 0)
    return fail(SGRL_ERR_HIP, std::string("no HIP device visible: libsgrl_hip.so needs an MI355X (there is no CPU fallback) [") +
                                  hipGetErrorString(dev_err) + ", " + std::to_string(ndev) + " devices]");
  sgrl_engine* e = new sgrl_engine();
  e->n_morph = n_morph;
  e->obs_max_len = obs_max_len;
  e->action_max_len = action_max_len;
  std::vector<int32_t> env_morph;
  std::vector<MorphDev> morphs(n_morph);
  e->d_ib.assign(n_morph, nullptr);
  e->d_fb.assign(n_morph, nullptr);
  int rc = SGRL_OK;
  for (int k = 0; k < n_morph && rc == SGRL_OK; k++) {
    SgrlModelView v;
    if (!ib[k] || !fb[k] || ib_len[k] < SGRL_NHDR || sgrl_model_view(ib[k], fb[k], &v) != 0) { rc = fail(SGRL_ERR_MODEL, "bad magic in model blob " + std::to_string(k)); break; }
    if (v.n_int != ib_len[k] || v.n_f64 != fb_len[k]) { rc = fail(SGRL_ERR_MODEL, "model blob " + std::to_string(k) + " has unexpected length"); break; }
    if (7 * v.njnt > 26 * v.npair) { rc = fail(SGRL_ERR_LIMIT, "model " + std::to_string(k) + ": joint rotations / axes do not fit the contact scratch (7 njnt > 26 npair)"); break; }
    if (3 * v.njnt > 10 * v.nbody) { rc = fail(SGRL_ERR_LIMIT, "model " + std::to_string(k) + ": joint positions do not fit the inertia scratch (3 njnt > 10 nbody)"); break; }
    if (v.nv > 64 || v.nbody > 64 || v.npair > 64) { rc = fail(SGRL_ERR_LIMIT, "morphology exceeds 64 dofs/bodies/pairs"); break; }
    if ((v.nv | 1) > sgrl::kSlabLdy || ib[k][SGRL_H_MAX_ROWS] > sgrl::kSlabRows) { rc = fail(SGRL_ERR_LIMIT, "morphology exceeds the HBM constraint slab (nv <= 46, max_rows <= 64)"); break; }
    const int L = v.nbody - 1;
    if (41 * L > obs_max_len || 3 * L > action_max_len) { rc = fail(SGRL_ERR_ARG, "obs_max_len/action_max_len too small for morphology " + std::to_string(k)); break; }
    if (morph_count[k] < 0) { rc = fail(SGRL_ERR_ARG, "negative morph_count"); break; }
    sgrl::Layout o;
    sgrl::make_layout(ib[k], &o, v.n_int, SGRL_STAGE_FLOATS ? v.n_f64 : 0);
    const int bytes = sgrl::layout_bytes(&o);
    if (bytes > 160 * 1024) { rc = fail(SGRL_ERR_LIMIT, "per-environment LDS slab exceeds 160 KiB"); break; }
    if (bytes > e->lds_bytes) e->lds_bytes = bytes;
    e->morph_lds.push_back(bytes);
    const int need = v.nq + v.nv + 4;
    if (need > e->stride) e->stride = need;
    for (int i = 0; i < morph_count[k]; i++) env_morph.push_back(k);
    if (hipMalloc(&e->d_ib[k], sizeof(int32_t) * ib_len[k]) != hipSuccess || hipMalloc(&e->d_fb[k], sizeof(double) * fb_len[k]) != hipSuccess) { rc = fail(SGRL_ERR_HIP, "hipMalloc(model) failed"); break; }
    (void)hipMemcpy(e->d_ib[k], ib[k], sizeof(int32_t) * ib_len[k], hipMemcpyHostToDevice);
    (void)hipMemcpy(e->d_fb[k], fb[k], sizeof(double) * fb_len[k], hipMemcpyHostToDevice);
    morphs[k].ib = e->d_ib[k];
    morphs[k].fb = e->d_fb[k];
  }
  if (rc == SGRL_OK && env_morph.empty()) rc = fail(SGRL_ERR_ARG, "no environments requested");
  if (rc != SGRL_OK) { sgrl_engine_destroy(e); return rc; }
  e->n_env = (int)env_morph.size();
  e->stride = (e->stride + 1) & ~1;
  bool ok = hipMalloc(&e->d_morphs, sizeof(MorphDev) * n_morph) == hipSuccess &&
            hipMalloc(&e->d_env_morph, sizeof(int32_t) * e->n_env) == hipSuccess &&
            hipMalloc(&e->d_block_env, sizeof(int32_t) * e->n_env) == hipSuccess &&
            hipMalloc(&e->d_rec, sizeof(double) * (size_t)e->n_env * e->stride) == hipSuccess &&
            hipMalloc(&e->d_cnt, sizeof(int32_t) * (size_t)e->n_env * 4) == hipSuccess &&
            hipMalloc(&e->d_scratch, sizeof(double) * (size_t)e->n_env * sgrl::kScratchDoubles) == hipSuccess;
  if (!ok) { sgrl_engine_destroy(e); return fail(SGRL_ERR_HIP, "hipMalloc(state) failed"); }
  (void)hipMemcpy(e->d_morphs, morphs.data(), sizeof(MorphDev) * n_morph, hipMemcpyHostToDevice);
  (void)hipMemcpy(e->d_env_morph, env_morph.data(), sizeof(int32_t) * e->n_env, hipMemcpyHostToDevice);
  {
    // dispatch order: group by LDS occupancy class (fewest workgroups per CU first = costliest), inside a group the
    // morphologies with the most dofs first -- workgroups are handed out by index, this balances the tail
    std::vector<int32_t> order(e->n_env);
    for (int i = 0; i < e->n_env; i++) order[i] = i;
    std::vector<int> cost(n_morph), cls(n_morph);
    // A batch made of light morphologies ONLY (few dofs, small slabs: walker_2/3/4, hopper_3/4) runs on k_env_step_light, three
    // waves per SIMD: measured -20 % (8192 x walker_2: 1.96 -> 1.57 ms, walker_4: 2.74 -> 2.20 ms).  In a MIXED batch the light
    // class as a second concurrent dispatch LOSES (walker mix 2.84 -> 3.18 ms, hopper++ 1.59 -> 1.78 ms): 168- and 235-register
    // waves fragment the register file of a SIMD (1 heavy + 1 light instead of 3 light), so the light waves pay for their 31
    // spilled registers without getting the occupancy -- mixed batches therefore stay on the one kernel.  SGRL_LIGHT=0: never.
    bool light_batch = [] { const char* v = getenv("SGRL_LIGHT"); return !(v && v[0] == '0'); }();
    for (int k = 0; k < n_morph; k++) {
      cost[k] = ib[k][SGRL_H_NV];
      // LDS is handed out in 1280-byte granules (measured with tools/occupancy_probe.py: a 27 064-byte slab fits five
      // times into the 160 KB of a CU, not six)
      const int granule = 1280;
      int per_cu = (160 * 1024) / (((e->morph_lds[k] + granule - 1) / granule) * granule);
      cls[k] = per_cu > 8 ? 8 : per_cu;   // register budget (<= 256 VGPRs): two waves per SIMD = 8 workgroups per CU at most
      light_batch = light_batch && ib[k][SGRL_H_NV] <= kLightNv && per_cu >= kLightPerCu;
    }
    if (light_batch) for (int k = 0; k < n_morph; k++) cls[k] = kLightPerCu;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
      const int mx = env_morph[x], my = env_morph[y];
      if (cls[mx] != cls[my]) return cls[mx] < cls[my];
      return cost[mx] > cost[my];
    });
    (void)hipMemcpy(e->d_block_env, order.data(), sizeof(int32_t) * e->n_env, hipMemcpyHostToDevice);
    for (int i = 0; i < e->n_env;) {
      sgrl_engine::Group g;
      g.first = i;
      // A batch whose costliest class already has 6+ workgroups per CU runs that class and the next as ONE dispatch:
      // losing one of seven or eight workgroups on the lighter morphologies costs nothing measurable (walker mix: one
      // dispatch at 7 per CU is as fast as two at 7 and 8).  Batches with heavier members (humanoid 5-6, cheetah 3-4 per
      // CU) keep one dispatch per class: measured 7 % faster on the 23-morphology cwhh batch.
      const int c = cls[env_morph[order[i]]];
      const int c_min = cls[env_morph[order[0]]];
      int top = (c == c_min && c >= 6) ? c + 1 : c;
      // SGRL_GROUP_POLICY=0: never merge, =2: always merge adjacent classes (8 % faster on a cheetah-only batch)
      if (const char* pol = getenv("SGRL_GROUP_POLICY")) top = pol[0] == '0' ? c : (pol[0] == '2' ? c + 1 : top);
      if (top >= kLightPerCu) top = kLightPerCu;            // the light class never merges with a heavier one ...
      else if (top > 8) top = 8;                            // ... nor a heavier one with it
      g.light = c == kLightPerCu;
      while (i < e->n_env && cls[env_morph[order[i]]] <= top) { g.lds = std::max(g.lds, e->morph_lds[env_morph[order[i]]]); i++; }
      // diagnostics only (tools/occupancy_sweep.py): SGRL_LDS_PAD=<bytes> inflates the dynamic LDS request to force fewer
      // resident workgroups per CU
      if (const char* pad = getenv("SGRL_LDS_PAD")) { g.lds += atoi(pad); e->lds_bytes = std::max(e->lds_bytes, g.lds); }
      g.count = i - g.first;
      e->groups.push_back(g);
    }
    if (e->groups.size() > 1) {
      bool sok = hipEventCreateWithFlags(&e->fork, hipEventDisableTiming) == hipSuccess;
      // The group streams are PLAIN non-blocking streams.  Round 1 gave the heavy groups higher stream priorities (their
      // workgroups then all start at once and the light ones fill the gaps: k_env_step alone 5..10 % faster on mixed
      // batches), but on this ROCm (7.2) the mere existence of hipStreamCreateWithPriority streams in a process corrupts
      // LATER hipGraph captures -- work enqueued shortly before a capture is recorded into the graph a second time (the TD3
      // update graphs of train_loop.py replayed at 45..60 ms instead of 13..24 ms; tools/diag/train_time_probe3.py) -- and
      // the priorities also slowed the SET forward that follows the step (humanoid mix: step + forward 7.4 ms with, 6.1 ms
      // without).  SGRL_GROUP_PRIO=1 restores them for engine-only experiments.
      int prio_least = 0, prio_greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);   // numerically lower = higher priority
      const char* want_prio = getenv("SGRL_GROUP_PRIO");
      const bool use_prio = want_prio && want_prio[0] == '1';
      int gi = 0;
      for (auto& g : e->groups) {
        int prio = prio_greatest + gi++;
        if (prio > prio_least) prio = prio_least;
        const hipError_t se = use_prio ? hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, prio)
                                       : hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking);
        sok = sok && se == hipSuccess && hipEventCreateWithFlags(&g.done, hipEventDisableTiming) == hipSuccess;
      }
      if (!sok) { sgrl_engine_destroy(e); return fail(SGRL_ERR_HIP, "cannot create launch-group streams"); }
    }
  }
  {
    std::vector<int32_t> cnt0((size_t)e->n_env * 4, 0);
    for (int i = 0; i < e->n_env; i++) cnt0[4 * (size_t)i + 1] = -1;  // episode = -1: the first reset bumps it to 0
    (void)hipMemcpy(e->d_cnt, cnt0.data(), sizeof(int32_t) * cnt0.size(), hipMemcpyHostToDevice);
  }
  {
    // hipFuncAttributeMaxDynamicSharedMemorySize is kept per kernel AND per device: only ever RAISE it (a second engine with a
    // smaller slab must not lower the limit under the first one's launches), and remember what was raised on WHICH device
    static int g_lds_limit[64];
    static bool g_lds_init = false;
    if (!g_lds_init) { for (int& v : g_lds_limit) v = 48 * 1024; g_lds_init = true; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& limit = g_lds_limit[dev >= 0 && dev < 64 ? dev : 0];
    if (e->lds_bytes > limit) {
      hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_env_step), hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes);
      hipError_t a2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_env_reset), hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes);
      hipError_t a3 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_env_refresh), hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes);
      hipError_t a4 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_env_step_light), hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes);
      if (a1 != hipSuccess || a2 != hipSuccess || a3 != hipSuccess || a4 != hipSuccess) { sgrl_engine_destroy(e); return fail(SGRL_ERR_HIP, "cannot raise the dynamic LDS limit"); }
      limit = e->lds_bytes;
    }
  }
  BatchArgs& a = e->args;
  a.morphs = e->d_morphs; a.env_morph = e->d_env_morph; a.block_env = e->d_block_env; a.rec = e->d_rec; a.cnt = e->d_cnt; a.scratch = e->d_scratch;
  a.stride = e->stride; a.n_env = e->n_env; a.obs_max_len = obs_max_len; a.action_max_len = action_max_len;
  a.seed = seed; a.env_id_base = env_id_base; a.max_episode_steps = max_episode_steps;
  if (hipDeviceSynchronize() != hipSuccess) { sgrl_engine_destroy(e); return fail(SGRL_ERR_HIP, "device error during engine setup"); }
  *out = e;
  return SGRL_OK;
}

int sgrl_num_envs(const sgrl_engine* e) { return e ? e->n_env : SGRL_ERR_ARG; }
int sgrl_record_stride(const sgrl_engine* e) { return e ? e->stride : SGRL_ERR_ARG; }
int sgrl_lds_bytes(const sgrl_engine* e) { return e ? e->lds_bytes : SGRL_ERR_ARG; }
int sgrl_launch_groups(const sgrl_engine* e) { return e ? (int)e->groups.size() : SGRL_ERR_ARG; }

int sgrl_reset(sgrl_engine* e, float* obs, double* obs64, void* stream) {
  if (!e || !obs) return fail(SGRL_ERR_ARG, "sgrl_reset: null engine or obs");
  StepOut out{};
  out.obs32 = obs; out.obs64 = obs64;
  return launch_groups(e, k_env_reset, out, (hipStream_t)stream);
}

int sgrl_step(sgrl_engine* e, const float* actions, float* obs, float* reward, uint8_t* done, float* dist,
              uint8_t* truncated, double* obs64, double* reward64, int auto_reset, void* stream) {
  if (!e || !actions || !obs) return fail(SGRL_ERR_ARG, "sgrl_step: null engine, actions or obs");
  StepOut out{};
  out.actions = actions; out.obs32 = obs; out.obs64 = obs64; out.reward = reward; out.reward64 = reward64;
  out.done = done; out.dist = dist; out.truncated = truncated; out.auto_reset = auto_reset;
  return launch_groups(e, k_env_step, out, (hipStream_t)stream, true);
}

int sgrl_refresh(sgrl_engine* e, float* obs, double* obs64, void* stream) {
  if (!e || !obs) return fail(SGRL_ERR_ARG, "sgrl_refresh: null engine or obs");
  StepOut out{};
  out.obs32 = obs; out.obs64 = obs64;
  return launch_groups(e, k_env_refresh, out, (hipStream_t)stream);
}

int sgrl_get_records(sgrl_engine* e, double* rec, int32_t* cnt) {
  if (!e) return fail(SGRL_ERR_ARG, "null engine");
  HIP_TRY(hipDeviceSynchronize());
  if (rec) HIP_TRY(hipMemcpy(rec, e->d_rec, sizeof(double) * (size_t)e->n_env * e->stride, hipMemcpyDeviceToHost));
  if (cnt) HIP_TRY(hipMemcpy(cnt, e->d_cnt, sizeof(int32_t) * (size_t)e->n_env * 4, hipMemcpyDeviceToHost));
  return SGRL_OK;
}

int sgrl_set_records(sgrl_engine* e, const double* rec, const int32_t* cnt) {
  if (!e) return fail(SGRL_ERR_ARG, "null engine");
  HIP_TRY(hipDeviceSynchronize());
  if (rec) HIP_TRY(hipMemcpy(e->d_rec, rec, sizeof(double) * (size_t)e->n_env * e->stride, hipMemcpyHostToDevice));
  if (cnt) HIP_TRY(hipMemcpy(e->d_cnt, cnt, sizeof(int32_t) * (size_t)e->n_env * 4, hipMemcpyHostToDevice));
  return SGRL_OK;
}

int sgrl_time_steps(sgrl_engine* e, const float* actions, float* obs, float* reward, uint8_t* done, int reps,
                    void* stream, float* ms_out) {
  if (!e || !actions || !obs || !ms_out || reps <= 0) return fail(SGRL_ERR_ARG, "sgrl_time_steps: bad argument");
  hipEvent_t t0, t1;
  HIP_TRY(hipEventCreate(&t0));
  HIP_TRY(hipEventCreate(&t1));
  StepOut out{};
  out.actions = actions; out.obs32 = obs; out.reward = reward; out.done = done; out.auto_reset = 1;
  HIP_TRY(hipEventRecord(t0, (hipStream_t)stream));
  for (int r = 0; r < reps; r++) {
    const int rc = launch_groups(e, k_env_step, out, (hipStream_t)stream, true);
    if (rc != SGRL_OK) return rc;
  }
  HIP_TRY(hipEventRecord(t1, (hipStream_t)stream));
  HIP_TRY(hipEventSynchronize(t1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  HIP_TRY(hipGetLastError());
  *ms_out = ms / reps;
  return SGRL_OK;
}

int sgrl_ingest_rows(const float* block, int n_rows, int obs_len, int act_len, const int64_t* slot, const sgrl_ring* rings,
                     int n_rings, void* stream) {
  if (!block || !slot || !rings || n_rows <= 0 || obs_len <= 0 || act_len <= 0 || n_rings <= 0)
    return fail(SGRL_ERR_ARG, "sgrl_ingest_rows: bad argument");
  hipLaunchKernelGGL(k_ingest_rows, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, block, obs_len, act_len, slot, rings, n_rings);
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(SGRL_ERR_HIP, std::string("k_ingest_rows launch failed (") + hipGetErrorName(le) + ")");
  return SGRL_OK;
}

int sgrl_pack_transitions(const float* obs, int ld_obs, const float* action, int ld_act, const float* next_obs, int ld_next,
                          const float* reward, const float* done_f32, const uint8_t* done_u8, const uint8_t* store,
                          const int64_t* morph_id, float* block, int n_env, int obs_len, int act_len, void* stream) {
  if (!block || n_env < 0 || obs_len <= 0 || act_len <= 0 || (obs && ld_obs < obs_len) || (action && ld_act < act_len) ||
      (next_obs && ld_next < obs_len) || (done_f32 && done_u8))
    return fail(SGRL_ERR_ARG, "sgrl_pack_transitions: bad argument");
  if (n_env == 0) return SGRL_OK;
  PackArgs p{obs, action, next_obs, reward, done_f32, done_u8, store, morph_id, block, ld_obs, ld_act, ld_next, obs_len, act_len};
  hipLaunchKernelGGL(k_pack_transitions, dim3(n_env), dim3(256), 0, (hipStream_t)stream, p);
  HIP_TRY(hipGetLastError());
  return SGRL_OK;
}

}  // extern "C"
