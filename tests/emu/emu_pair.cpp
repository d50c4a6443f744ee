// tests/emu/emu_pair.cpp -- TEST HARNESS ONLY (never loaded by the product).
//
// SIMT emulator of the TWO-ENVIRONMENTS-PER-WAVEFRONT instance of the step kernel (sgrl_amd/csrc/wave_half.h + step_body.h): 64 lane
// fibers (a register-and-stack switch of its own), lanes 0..31 = environment A, 32..63 = environment B, every fiber runs the whole env_step() exactly as a GPU
// lane does -- own copy of the wave object, per-lane slab pointers into ONE shared "LDS" buffer with the pair layout
// (Layout::pair_stride, one shared copy of the int tables behind both slabs), per-lane StepIO.  Cross-lane primitives
// (wave_hip.h HipHalfPrim on the GPU) are rendezvous points of the 32 fibers of a half: a lane publishes its operand, waits for
// its half, reads its source lane.  The two halves never wait for each other -- which is how the hardware behaves when the two
// environments' data-dependent branches diverge -- and a lane that takes a different path from its half shows up as a deadlock,
// which the scheduler reports instead of hanging.  What this leaves to the GPU: the DPP / permlane encodings of the primitives
// themselves (tools/micro/halfwave_prims.hip checks those on the device).

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sgrl_amd/csrc/wave_half.h"

namespace {
constexpr int NL = 32, NF = 64;
constexpr size_t kStack = 512 * 1024;

// A context switch without the two sigprocmask system calls per swapcontext() (a wavefront step is ~10^5 switches): the callee-saved
// registers and the stack pointer, x86-64 System V (what the build container and the GPU box's host are).
struct Ctx { void* sp; };
extern "C" void sgrl_emu_ctx_switch(Ctx* from, Ctx* to);
asm(R"(
.text
.globl sgrl_emu_ctx_switch
.type sgrl_emu_ctx_switch,@function
sgrl_emu_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size sgrl_emu_ctx_switch,.-sgrl_emu_ctx_switch
)");
#if !defined(__x86_64__)
#error "tests/emu/emu_pair.cpp: the fiber switch is written for x86-64"
#endif

struct Sched {
  Ctx main_ctx, ctx[NF];
  std::vector<char> stacks;
  bool finished[NF];
  int cur = -1;
  // per half: rendezvous state + exchange slots
  int arrived[2] = {0, 0};
  unsigned gen[2] = {0, 0};
  double slot[2][NL];
  uint32_t bits[2] = {0, 0};
  unsigned long progress = 0;
  void (*body)(int) = nullptr;
};
Sched* g = nullptr;

void yield_() { const int me = g->cur; sgrl_emu_ctx_switch(&g->ctx[me], &g->main_ctx); }

void half_sync() {
  const int h = g->cur >> 5;
  const unsigned g0 = g->gen[h];
  if (++g->arrived[h] == NL) { g->arrived[h] = 0; g->gen[h]++; g->progress++; }
  while (g->gen[h] == g0) yield_();
}
// every lane publishes x and reads the slot of logical lane src(lane) of its own half
template <class F> double exchange(double x, F src) {
  const int h = g->cur >> 5, l = g->cur & 31;
  g->slot[h][l] = x;
  half_sync();
  const double r = g->slot[h][src(l)];
  half_sync();
  return r;
}

void trampoline() {
  const int idx = g->cur;
  g->body(idx);
  g->finished[idx] = true;
  g->progress++;
  sgrl_emu_ctx_switch(&g->ctx[idx], &g->main_ctx);
  abort();      // a finished fiber is never resumed
}

// runs body(lane) on 64 fibers; returns 0, or -1 if the lanes of a half stopped meeting (divergence inside a half)
int run_wave(void (*body)(int)) {
  Sched s;
  g = &s;
  s.body = body;
  s.stacks.resize(kStack * NF);
  for (int i = 0; i < NF; i++) {
    s.finished[i] = false;
    // first switch into the fiber: six zeroed callee-saved registers, then `ret` into trampoline() with the stack 8 mod 16
    uintptr_t top = (reinterpret_cast<uintptr_t>(s.stacks.data() + kStack * (i + 1))) & ~uintptr_t(15);
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;                                   // return address of trampoline (never used)
    *--sp = reinterpret_cast<void*>(&trampoline);
    for (int k = 0; k < 6; k++) *--sp = nullptr;
    s.ctx[i].sp = sp;
  }
  int rc = 0;
  for (;;) {
    const unsigned long p0 = s.progress;
    bool any = false;
    for (int i = 0; i < NF; i++) {
      if (s.finished[i]) continue;
      any = true;
      s.cur = i;
      sgrl_emu_ctx_switch(&s.main_ctx, &s.ctx[i]);
    }
    if (!any) break;
    if (s.progress == p0) { rc = -1; break; }     // a full round without a rendezvous completing or a lane finishing
  }
  g = nullptr;
  return rc;
}

// the primitive set of wave_half.h, emulated lane for lane with the SAME data movement as the gfx950 encodings
struct EmuHalfPrim {
  static constexpr bool kFixedDims = false;
  template <class M> static int hdr_const(const M& m, int idx) { return m.hdr[idx]; }
  int lane, half;
  EmuHalfPrim() : lane(g->cur & 31), half(g->cur >> 5) {}
  void fence_lane() {}
  template <class T> T fenced(T v) { return v; }
  static void sync() { half_sync(); }
  // DPP row_newbcast:j -- lane j of each 16-lane row to its row (the second row of the half sees ITS lane j, as on the device)
  static double bcast16(double x, int j) { return exchange(x, [j](int l) { return (l & 16) | j; }); }
  static double xor1(double x) { return exchange(x, [](int l) { return l ^ 1; }); }
  template <class Op> static double reduce(double x, Op op) {
    x = op(x, exchange(x, [](int l) { return l ^ 1; }));                           // quad_perm [1,0,3,2]
    x = op(x, exchange(x, [](int l) { return l ^ 2; }));                           // quad_perm [2,3,0,1]
    x = op(x, exchange(x, [](int l) { return (l & ~7) | (7 - (l & 7)); }));        // row_half_mirror
    x = op(x, exchange(x, [](int l) { return (l & ~15) | (15 - (l & 15)); }));     // row_mirror
    const double e = exchange(x, [](int) { return 0; }), o = exchange(x, [](int) { return 16; });   // v_permlane16_swap
    return op(e, o);
  }
  static double half_sum(double x) { return reduce(x, [](double a, double b) { return a + b; }); }
  static double half_max(double x) { return reduce(x, [](double a, double b) { return std::fmax(a, b); }); }
  uint32_t half_ballot(bool p) const {
    const int h = g->cur >> 5;
    if (lane == 0) g->bits[h] = 0;
    half_sync();
    if (p) g->bits[h] |= 1u << lane;
    half_sync();
    const uint32_t r = g->bits[h];
    half_sync();
    return r;
  }
};
using EmuHalfWave = sgrl::HalfWaveT<15, EmuHalfPrim>;

// arguments of the call in flight (fibers read them)
struct Call {
  int op;
  SgrlModelView m;
  sgrl::Layout o;
  double* lds;
  sgrl::StepIO io[2];
} c;

void lane_body(int) {
  EmuHalfWave w;
  double* S = c.lds + w.half * c.o.pair_stride;
  int32_t* I = reinterpret_cast<int32_t*>(c.lds + c.o.s_total) + 2 * w.half * c.o.pair_stride;
  const sgrl::StepIO io = c.io[w.half];
  if (c.op == 0) sgrl::env_reset(w, c.m, c.o, S, I, io, false);
  else if (c.op == 1) sgrl::env_step(w, c.m, c.o, S, I, io);
  else sgrl::env_refresh(w, c.m, c.o, S, I, io);
}
}  // namespace

extern "C" {

int sgrl_emu_pair_layout_bytes(const int32_t* ib, const double* fb) {
  SgrlModelView v;
  if (sgrl_model_view(ib, fb, &v)) return -1;
  sgrl::Layout o; sgrl::make_layout(ib, &o, v.n_int, 0, true);
  return sgrl::layout_bytes(&o);
}

// op 0 = reset, 1 = step, 2 = refresh for TWO environments of one morphology (arrays of two pointers / values each).
// Returns 0; -1 bad model; -2 the lanes of a half diverged; -3 an access left its slab (guard words touched).
int sgrl_emu_pair_env(int op, const int32_t* ib, const double* fb, double* const* rec, int32_t* const* cnt, const float* const* action,
                      float* const* obs32, double* const* obs64, int obs_max_len, uint64_t seed, const uint32_t* env_id,
                      int max_episode_steps, int auto_reset, double* const* reward64, uint8_t* const* done, float* const* dist,
                      uint8_t* const* truncated) {
  SgrlModelView v;
  if (sgrl_model_view(ib, fb, &v)) return -1;
  sgrl::Layout o; sgrl::make_layout(ib, &o, v.n_int, 0, true);
  const int bytes = sgrl::layout_bytes(&o);
  const int nd = bytes / 8, guard = 64;
  std::vector<double> lds(nd + 2 * guard);
  // poison: NaN in the double part of both slabs, -12345 in the int parts, a pattern in the guard words around the buffer
  uint64_t pat = 0x7ff8dead0000beefull;
  for (auto& x : lds) std::memcpy(&x, &pat, 8);
  double* base = lds.data() + guard;
  for (int h = 0; h < 2; h++) {
    int32_t* I = reinterpret_cast<int32_t*>(base + o.s_total) + 2 * h * o.pair_stride;
    for (int k = 0; k < 2 * (o.pair_stride - o.s_total); k++) I[k] = -12345;
  }
  // ONE copy of the int tables behind both slabs, as the kernel stages it
  int32_t* ia = reinterpret_cast<int32_t*>(base + o.s_total);
  for (int k = 0; k < v.n_int; k++) ia[o.model_i + k] = ib[k];
  sgrl_model_view_dims(ib, ib, fb, ia + o.model_i, fb, &c.m);
  c.op = op; c.o = o; c.lds = base;
  static std::vector<double> scratch[2];
  for (int h = 0; h < 2; h++) {
    scratch[h].resize(sgrl::kScratchDoublesMax);
    sgrl::StepIO& io = c.io[h];
    io.rec = rec[h]; io.cnt = cnt[h]; io.action = action ? action[h] : nullptr; io.obs32 = obs32[h]; io.obs64 = obs64[h];
    io.reward = nullptr; io.done = done[h]; io.dist = dist[h]; io.truncated = truncated[h]; io.reward64 = reward64[h];
    io.obs_max_len = obs_max_len; io.scratch = scratch[h].data();
    io.seed = seed; io.env_id = env_id[h]; io.max_episode_steps = max_episode_steps; io.auto_reset = auto_reset;
  }
  if (run_wave(lane_body) != 0) return -2;
  for (int k = 0; k < guard; k++) {
    uint64_t a, b;
    std::memcpy(&a, &lds[k], 8); std::memcpy(&b, &lds[guard + nd + k], 8);
    if (a != pat || b != pat) return -3;
  }
  // the shared tables must come out as they went in
  for (int k = 0; k < v.n_int; k++) if (ia[o.model_i + k] != ib[k]) return -3;
  return 0;
}
}
