// train_gemm.hip -- the dense products of the TD3 update's linear layers for gfx950 (C ABI: include/sgrl_train.h).
//
// One kernel, three operand layouts:   C[m][n] = sum_k a(m, k) * b(n, k)
//   forward   y  = x . w^T      a = x  [M][K]  (contraction contiguous)   b = w [N][K]  (contraction contiguous)
//   dgrad     dx = g . w        a = g  [M][N]  (contraction contiguous)   b = w [N][K]  (contraction = ROW index)
//   wgrad     dw = g^T . x      a = g  [M][N]  (contraction = ROW index)  b = x [M][K]  (contraction = ROW index)
// The update's products are SMALL (700..4 200 rows, 30..1 024 columns, contractions up to 4 200) and each is on the critical
// path of the backward pass, so the kernel is built for LATENCY, not for arithmetic rate: 32 x 32 output tiles (a 700 x 256
// product is 176 workgroups instead of the vendor libraries' 3..12), and k-tiles 128 deep -- every thread has eight
// independent 16-byte loads in flight per step and a 256-long contraction is two steps, not sixteen.  Tiles are staged
// k-major in LDS (an operand whose contraction index is contiguous is transposed on its way in); the arithmetic is the
// float32 matrix instruction v_mfma_f32_32x32x2_f32 (exact float32 products, float32 accumulation): each of the workgroup's
// four waves multiplies its quarter of every k-tile into a full 32 x 32 accumulator (one LDS dword per operand and MFMA: a
// VALU 2 x 2 micro-tile version of this kernel was LDS-bandwidth bound at 4x the time), the four partial tiles meet in LDS
// in wave order, and the next k-tile's global loads are issued before the current tile's arithmetic.  A weight gradient
// with a long contraction and a small output (30 x 128 over 2 100 rows: four tiles) is split along the contraction over
// blockIdx.z: partial tiles go to scratch (agent-scope stores / loads, no cache write-back: st_agent below) and the LAST
// workgroup to arrive at a tile adds them in split order and applies the epilogue -- no floating-point atomics, the result
// is bit-reproducible.  g may be masked on the
// fly by the forward's output (ReLU backward), and the weight gradient's workgroups of the first column tile also produce
// the bias gradient (column sums of g).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "../../include/sgrl.h"
#include "../../include/sgrl_train.h"

namespace {

thread_local std::string g_train_err;
int tfail(int code, const std::string& msg) { g_train_err = msg; return code; }
// after a launch: the runtime's last error, named in the message (a stale error of an earlier call reads differently from a
// refused launch, e.g. into an invalidated stream capture)
bool launched(const char* what, int* rc) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return true;
  *rc = tfail(SGRL_ERR_HIP, std::string(what) + " (" + hipGetErrorName(e) + ": " + hipGetErrorString(e) + ")");
  return false;
}

constexpr int BT = 32;          // tile edge
constexpr int BKF = 128;        // k-tile depth of the forward / input-gradient products (BKW for the weight gradient)
constexpr int BKW = 128;        // weight gradient: 256-deep tiles (half the serial steps, 240 registers) measured 3-5 % slower
constexpr int LDP = 36;         // LDS row pitch (floats): rows 16-byte aligned, 8-byte aligned pairs

typedef float f32x16 __attribute__((ext_vector_type(16)));
static_assert(BKF * LDP >= 4 * 32 * 33, "the partial tiles reuse the A operand tile's LDS");

struct SArgs {
  const float* A; int lda;
  const float* Amask; int ldmask;   // null, or: a(m, k) counts only where Amask (same layout as A) is > 0
  const float* B; int ldb;
  const float* bias; int relu;
  const float* rowdiv;              // forward and input gradient: C[m][:] /= rowdiv[m]; weight gradient: the A operand's row k is divided by rowdiv[k]
  float* C; int ldc;
  float* db;                        // wgrad only: column sums of the (masked) A operand = rows of C
  int M, N, K;
  int kper;                         // contraction length per split (multiple of BK); gridDim.z splits
  float* ws; unsigned* counters;    // split scratch: [split][tile][TILE_WS] floats, one counter per tile (zero between calls)
  // forward only (both optional, last so that the other products' initialisers leave them null):
  const float* addend; int ldadd;   // C[m][n] += addend[m][n] after the epilogue (forward: a residual stream, g + linear5(...);
                                    // input gradient: accumulate onto what is already in C -- sgrl_linear_backward_acc)
  const float* tail; int ntail;     // C[m][N + j] = tail[m][j], j < ntail: columns appended to the product's N (z = [proj(x) | gdir],
                                    // c = [inv | ng]): the last column tile's spare columns, then column tiles of their own
  const float* omask; int ldomask;  // C[m][n] = 0 where omask[m][n] <= 0 (same layout as C): the ReLU mask of the layer BELOW applied
                                    // in this input gradient's epilogue (dx = (g . w) masked by x > 0), once per output element,
                                    // instead of on the fly in every k-tile of that layer's two backward products
};
constexpr int TILE_WS = BT * BT + BT;
constexpr int64_t kWsTiles = 8192;      // (split, tile) slots of the scratch buffer (a 64 x 64 partial tile takes four)
constexpr int kCounters = 4096;

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// Partial tiles of a split contraction travel between workgroups WITHOUT cache maintenance: agent-scope relaxed stores / loads
// (write-through / bypass of the XCD's non-coherent L2, `sc1`), a wait for the stores' completion and a barrier before the arrival
// counter -- instead of two device-scope fences per workgroup, each of which writes back / invalidates a whole L2 and serialises
// when hundreds of workgroups fence at once (LAB_LOG round 5).
// The protocol leans on this target: `sc1` stores are written through to the level every XCD sees and have completed when
// s_waitcnt vmcnt(0) returns; `sc1` loads bypass the reader's L2.  Nothing in the HIP memory model promises that elsewhere.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "train_gemm.hip: the fence-free partial-tile exchange is written for gfx942 / gfx950 (sc1 write-through); use __threadfence() elsewhere"
#endif
// store side: wait for this thread's stores, keep the compiler from sinking them below the barrier (a workgroup-scope release
// fence costs no cache maintenance)
__device__ __forceinline__ void publish_partials() {
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
// load side: nothing the last workgroup reads may be hoisted above its arrival
__device__ __forceinline__ void acquire_partials() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


// four consecutive elements of a row starting at column c (valid columns: c + j < cmax), zero filled
__device__ __forceinline__ float4 load4(const float* row, int c, int cmax, bool vec_ok) {
  if (vec_ok && c + 3 < cmax) return *reinterpret_cast<const float4*>(row + c);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cmax) v.x = row[c];
  if (c + 1 < cmax) v.y = row[c + 1];
  if (c + 2 < cmax) v.z = row[c + 2];
  if (c + 3 < cmax) v.w = row[c + 3];
  return v;
}
__device__ __forceinline__ float4 relu_mask(float4 v, float4 y) {
  v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f; v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
  return v;
}

// AT / BTR: the operand's contraction index is its ROW index (tile rows = k, staged as is: thread = (k, four columns));
// otherwise its row index is the output index and the contraction runs along the row (thread = (output row, four k), rows
// on adjacent lanes so that the transposing LDS stores fall on distinct banks).
// One output tile (bx = column tile, by = row tile, bz = contraction split of nz, on a grid of gx x gy tiles).
// RM (round 5): an operand whose contraction index is contiguous is staged ROW-major (pitch BK + 4 floats) instead of being
// transposed on its way into LDS: its global loads are coalesced (32 lanes = 512 contiguous bytes of one row; the transposing
// form reads 32 rows x 16 bytes per instruction), its LDS stores are one conflict-free 16-byte store per load instead of four
// 4-byte stores, and a lane fetches FOUR matrix instructions' worth of its row with one 16-byte LDS read (lane (i, h) of group g
// takes k = 8 g + 4 h + j for instruction j: both operands follow the same order, so the contraction is complete).
template <bool AT, bool BTR, int BK, bool RM = true>
__device__ __forceinline__ void sgemm_tile(const SArgs& a, float (*As)[LDP], float (*Bs)[LDP], int* s_last_p, int bx, int by,
                                           int bz, int gx, int gy, int nz) {
  constexpr int NLD = BT * BK / 4 / 256;   // float4 loads per thread, operand and k-tile
  constexpr int Q = BK / 4;                // float4s per staged row of a row-major tile
  constexpr int PR = BK + 4;               // its pitch: 16-byte aligned rows, rows i and i + 1 four banks apart
  static_assert(BT * PR <= BK * LDP, "a row-major tile fits the k-major tile's LDS");
  float* const Ar = &As[0][0];
  float* const Br = &Bs[0][0];
  int& s_last = *s_last_p;
  const int t = threadIdx.x;
  const int m0 = by * BT, n0 = bx * BT;
  const int k_begin = bz * a.kper, k_end = min(a.K, k_begin + a.kper);
  const bool a_vec = (a.lda & 3) == 0 && aligned16(a.A) && (!a.Amask || ((a.ldmask & 3) == 0 && aligned16(a.Amask)));
  const bool b_vec = (a.ldb & 3) == 0 && aligned16(a.B);
  if (!AT && !BTR && a.tail && n0 >= a.N) {      // a column tile past the product: its slice of the appended block, copied
    const int m = m0 + (t >> 3), c0 = n0 + 4 * (t & 7);
    if (m < a.M)
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (c0 + j < a.N + a.ntail) a.C[(size_t)m * a.ldc + c0 + j] = a.tail[(size_t)m * a.ntail + (c0 + j - a.N)];
    return;
  }
  const bool want_db = AT && a.db != nullptr && bx == 0;
  float4 dbp = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ra[2][NLD], rb[2][NLD];               // global loads run TWO k-tiles ahead of the arithmetic

  // INTERIOR tiles of the forward / input-gradient products (the tile's 32 rows / columns inside the operand, the k-tile inside the
  // contraction, 16-byte aligned rows, no mask) load through one base pointer per thread and operand, set up once: the general path
  // below spends ~25 vector instructions per 16-byte load on 64-bit addresses, bounds and zero fills, and the counters showed the
  // vector ALU busier than the matrix pipe (profiles/r5_sgemm_pmc.txt).  Element (k-tile at k0, load i) of thread t lies at
  // base + k0 * kmul + i * imul.  Not in the weight-gradient instance: the second path costs it 16 registers and its third wave
  // per SIMD (55 -> 82 us per grouped launch), and an instance with ONLY this path gains 10 % in isolation but nothing in the update
  // (its groups of twelve are rarely all regular; grouping them by kind adds launches: 8.03 -> 8.25 ms).
  const bool a_in = !AT && a_vec && !a.Amask && m0 + BT <= a.M;
  const bool b_in = !AT && b_vec && n0 + BT <= a.N;
  const size_t b_kmul = BTR ? (size_t)a.ldb : 1;
  const size_t a_imul = RM ? (size_t)(256 / Q) * a.lda : 32;
  const size_t b_imul = BTR ? (size_t)32 * a.ldb : (RM ? (size_t)(256 / Q) * a.ldb : 32);
  const float* const a_base = RM ? a.A + (size_t)(m0 + t / Q) * a.lda + 4 * (t % Q) : a.A + (size_t)(m0 + (t & 31)) * a.lda + 4 * (t >> 5);
  const float* const b_base = BTR ? a.B + (size_t)(t >> 3) * a.ldb + n0 + 4 * (t & 7)
                                  : (RM ? a.B + (size_t)(n0 + t / Q) * a.ldb + 4 * (t % Q) : a.B + (size_t)(n0 + (t & 31)) * a.ldb + 4 * (t >> 5));

  auto load_tiles = [&](auto slot_c, int k0) __attribute__((always_inline)) {     // slot as a compile-time constant: the staging arrays stay in registers
    constexpr int slot = decltype(slot_c)::value;
    const bool k_in = k0 + BK <= k_end;
    if (a_in && k_in) {
      const float* pa = a_base + k0;
#pragma unroll
      for (int i = 0; i < NLD; i++) ra[slot][i] = *reinterpret_cast<const float4*>(pa + i * a_imul);
    }
    if (b_in && k_in) {
      const float* pb = b_base + (size_t)k0 * b_kmul;
#pragma unroll
      for (int i = 0; i < NLD; i++) rb[slot][i] = *reinterpret_cast<const float4*>(pb + i * b_imul);
    }
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int idx = t + 256 * i;
      if (a_in && k_in) {
      } else if (AT) {                             // A[k][m]: k = idx / 8, columns m0 + 4 (idx % 8)
        const int k = k0 + (idx >> 3), c = m0 + 4 * (idx & 7);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k_end) {
          v = load4(a.A + (size_t)k * a.lda, c, a.M, a_vec);
          if (a.Amask) v = relu_mask(v, load4(a.Amask + (size_t)k * a.ldmask, c, a.M, a_vec));
          if (a.rowdiv) { const float f = 1.f / a.rowdiv[k]; v.x *= f; v.y *= f; v.z *= f; v.w *= f; }   // wgrad: g = dy / fn, row k (one reciprocal per load)
        }
        ra[slot][i] = v;
        if (want_db) { dbp.x += v.x; dbp.y += v.y; dbp.z += v.z; dbp.w += v.w; }
      } else {                              // A[m][k]: m = idx % 32, k = k0 + 4 (idx / 32); row-major staging: m = idx / Q, k = k0 + 4 (idx % Q)
        const int m = m0 + (RM ? idx / Q : (idx & 31)), c = k0 + 4 * (RM ? idx % Q : (idx >> 5));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < a.M) {
          v = load4(a.A + (size_t)m * a.lda, c, k_end, a_vec);
          if (a.Amask) v = relu_mask(v, load4(a.Amask + (size_t)m * a.ldmask, c, k_end, a_vec));
        }
        ra[slot][i] = v;
      }
      if (b_in && k_in) {
      } else if (BTR) {
        const int k = k0 + (idx >> 3), c = n0 + 4 * (idx & 7);
        rb[slot][i] = k < k_end ? load4(a.B + (size_t)k * a.ldb, c, a.N, b_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int n = n0 + (RM ? idx / Q : (idx & 31)), c = k0 + 4 * (RM ? idx % Q : (idx >> 5));
        rb[slot][i] = n < a.N ? load4(a.B + (size_t)n * a.ldb, c, k_end, b_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_tiles = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int idx = t + 256 * i;
      const float4 va = ra[slot][i], vb = rb[slot][i];
      if (AT) *reinterpret_cast<float4*>(&As[idx >> 3][4 * (idx & 7)]) = va;
      else if (RM) *reinterpret_cast<float4*>(Ar + (idx / Q) * PR + 4 * (idx % Q)) = va;
      else { const int k = 4 * (idx >> 5), m = idx & 31; As[k][m] = va.x; As[k + 1][m] = va.y; As[k + 2][m] = va.z; As[k + 3][m] = va.w; }
      if (BTR) *reinterpret_cast<float4*>(&Bs[idx >> 3][4 * (idx & 7)]) = vb;
      else if (RM) *reinterpret_cast<float4*>(Br + (idx / Q) * PR + 4 * (idx % Q)) = vb;
      else { const int k = 4 * (idx >> 5), n = idx & 31; Bs[k][n] = vb.x; Bs[k + 1][n] = vb.y; Bs[k + 2][n] = vb.z; Bs[k + 3][n] = vb.w; }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;             // MFMA operand lane: (row / column li, k parity lh)
  auto body = [&](int k0, auto slot_c) __attribute__((always_inline)) {
    store_tiles(slot_c);
    __syncthreads();
    if (k0 + 2 * BK < k_end) load_tiles(slot_c, k0 + 2 * BK);   // in flight underneath this tile's and the next tile's arithmetic
    const int kn = min(BK, k_end - k0);              // rows kn .. BK - 1 of the tiles are zero (load_tiles): harmless
    const int kw = wave * (BK / 4);
    if constexpr (RM && !(AT && BTR)) {
#pragma unroll
      for (int g8 = 0; g8 < BK / 4; g8 += 8)
        if (kw + g8 < kn) {                          // wave-uniform: skip the all-zero tail of a short contraction
          const int kq = kw + g8 + 4 * lh;           // this lane's four contraction indices of the group
          float av[4], bv[4];
          if (AT) {
#pragma unroll
            for (int j = 0; j < 4; j++) av[j] = As[kq + j][li];
          } else {
            const float4 q = *reinterpret_cast<const float4*>(Ar + li * PR + kq);
            av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w;
          }
          if (BTR) {
#pragma unroll
            for (int j = 0; j < 4; j++) bv[j] = Bs[kq + j][li];
          } else {
            const float4 q = *reinterpret_cast<const float4*>(Br + li * PR + kq);
            bv[0] = q.x; bv[1] = q.y; bv[2] = q.z; bv[3] = q.w;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 4; kk += 2)
        if (kw + kk < kn)                              // wave-uniform: skip the all-zero tail of a short contraction
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kw + kk + lh][li], Bs[kw + kk + lh][li], acc, 0, 0, 0);
    }
    __syncthreads();
  };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  load_tiles(S0, k_begin);
  if (k_begin + BK < k_end) load_tiles(S1, k_begin + BK);
  {
    int k0 = k_begin;
    for (; k0 + BK < k_end; k0 += 2 * BK) { body(k0, S0); body(k0 + BK, S1); }
    if (k0 < k_end) body(k0, S0);
  }
  float dbv = 0.f;         // bias gradient: this thread summed four columns of the k rows it staged; fold the 32 k-row groups
  if (want_db) {
    float* red = &Bs[0][0];                                   // [32 groups][32 columns], the tiles are dead now
    *reinterpret_cast<float4*>(red + (t >> 3) * BT + 4 * (t & 7)) = dbp;
    __syncthreads();
    if (t < BT) {
#pragma unroll
      for (int r = 0; r < 32; r++) dbv += red[r * BT + t];
    }
  }
  // the four waves' partial tiles meet in LDS (C layout of the instruction: column = lane & 31, row = (e & 3) + 8 (e >> 2)
  // + 4 (lane >> 5)); thread (row = t / 8, four columns) then sums them in wave order and applies the epilogue
  float* part = &As[0][0];                                    // [4][32][33] floats <= the two operand tiles (want_db is done with As)
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; e++) part[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * 33 + li] = acc[e];
  __syncthreads();
  const int r = t >> 3, c0 = 4 * (t & 7);
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
    v[j] = (part[(0 * 32 + r) * 33 + c0 + j] + part[(1 * 32 + r) * 33 + c0 + j]) +
           (part[(2 * 32 + r) * 33 + c0 + j] + part[(3 * 32 + r) * 33 + c0 + j]);
  const int splits = nz;
  if (splits > 1) {
    const int ntiles = gx * gy, tile = by * gx + bx;
    float* mine = a.ws + ((size_t)bz * ntiles + tile) * TILE_WS;
#pragma unroll
    for (int j = 0; j < 4; j++) st_agent(mine + r * BT + c0 + j, v[j]);
    if (want_db && t < BT) st_agent(mine + BT * BT + t, dbv);
    publish_partials();                            // this thread's stores have reached the coherent level ...
    __syncthreads();                               // ... and so have the workgroup's, before its arrival is counted
    if (t == 0) s_last = atomicAdd(&a.counters[tile], 1u) == (unsigned)(splits - 1);
    __syncthreads();
    if (!s_last) return;
    acquire_partials();
    v[0] = v[1] = v[2] = v[3] = 0.f;
    dbv = 0.f;
    for (int sp = 0; sp < splits; sp++) {                      // fixed order: the sum does not depend on who arrived when
      const float* p = a.ws + ((size_t)sp * ntiles + tile) * TILE_WS;
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] += ld_agent(p + r * BT + c0 + j);
      if (want_db && t < BT) dbv += ld_agent(p + BT * BT + t);
    }
    if (t == 0) a.counters[tile] = 0;                         // left zero for the next call on this stream
  }
  if (want_db && t < BT && m0 + t < a.M) a.db[m0 + t] = dbv;
  const int m = m0 + r;
  if (m >= a.M) return;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int n = n0 + c0 + j;
    if (n >= a.N) {
      if (!AT && !BTR && a.tail && n < a.N + a.ntail) a.C[(size_t)m * a.ldc + n] = a.tail[(size_t)m * a.ntail + (n - a.N)];
      continue;
    }
    float o = v[j] + (a.bias ? a.bias[n] : 0.f);
    if (a.relu) o = fmaxf(o, 0.f);
    if (!AT && a.rowdiv) o = o / a.rowdiv[m];          // forward: y / fn; input gradient: (dy / fn) . w = (dy . w) / fn, row by row
    if (!AT && BTR && a.omask && !(a.omask[(size_t)m * a.ldomask + n] > 0.f)) o = 0.f;
    // forward: a residual stream; input gradient: the gradient other consumers of the same input have already left in C
    // (addend == C: every element is read and written by this one thread)
    if (!AT && a.addend) o += a.addend[(size_t)m * a.ldadd + n];
    a.C[(size_t)m * a.ldc + n] = o;
  }
}

// Workgroups b and b + 8 of a dispatch share an XCD (round-robin): renumber so that each XCD works on a contiguous run of output
// tiles -- the column tiles of one row tile then re-read the same operand rows out of ONE L2 instead of pulling them into eight.
// Speed only, never correctness.  id = linear workgroup index in a grid of n tiles, column tiles fastest.
__device__ __forceinline__ int xcd_tile(int id, int n) {
  const int per = n >> 3, rem = n & 7, x = id & 7, i = id >> 3;
  return (x < rem) ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
}
template <bool AT, bool BTR, int BK, bool RM>
__global__ __launch_bounds__(256) void k_sgemm(SArgs a) {
  __shared__ __attribute__((aligned(16))) float As[BK][LDP];
  __shared__ __attribute__((aligned(16))) float Bs[BK][LDP];
  __shared__ int s_last;
  const int tile = xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  sgemm_tile<AT, BTR, BK, RM>(a, As, Bs, &s_last, tile % gridDim.x, tile / gridDim.x, blockIdx.z, gridDim.x, gridDim.y, gridDim.z);
}

// Twin launch: TWO independent products of the same shape (the layers of the twin critics, reference SECritic.py: critic1 /
// critic2 are two TransformerModels applied to the same batch) in one grid -- blockIdx.z picks the argument set.  Half the
// launches of a critic pass; forward and input-gradient products only (their contractions are never split).
struct SArgs2 { SArgs a[2]; };
template <bool BTR, bool RM>
__global__ __launch_bounds__(256) void k_sgemm_twin(SArgs2 p) {
  __shared__ __attribute__((aligned(16))) float As[BKF][LDP];
  __shared__ __attribute__((aligned(16))) float Bs[BKF][LDP];
  __shared__ int s_last;
  const SArgs a = p.a[blockIdx.z];              // uniform: scalar loads from the kernel-argument segment
  const int tile = xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  sgemm_tile<false, BTR, BKF, RM>(a, As, Bs, &s_last, tile % gridDim.x, tile / gridDim.x, 0, gridDim.x, gridDim.y, 1);
}

// The two products of a layer's backward pass -- input gradient and weight (+ bias) gradient -- need the same dy and nothing
// from each other: ONE launch, the first nd workgroups take the tiles of the input gradient, the rest those of the weight
// gradient (and its contraction splits).  Half the launches of the update's backward GEMMs, and each small product no longer
// waits for the other to drain.
struct BwdArgs { SArgs d; SArgs w; int nd, dgx, dgy, wgx, wgy, wnz; };
template <bool RM>
__global__ __launch_bounds__(256) void k_sgemm_bwd(BwdArgs p) {
  __shared__ __attribute__((aligned(16))) float As[BKF][LDP];
  __shared__ __attribute__((aligned(16))) float Bs[BKF][LDP];
  __shared__ int s_last;
  static_assert(BKW == BKF, "the fused backward kernel shares one pair of LDS tiles");
  const int id = blockIdx.x;
  if (id < p.nd) {
    sgemm_tile<false, true, BKF, RM>(p.d, As, Bs, &s_last, id % p.dgx, id / p.dgx, 0, p.dgx, p.dgy, 1);
  } else {
    const int r = id - p.nd, per = p.wgx * p.wgy;
    sgemm_tile<true, true, BKW>(p.w, As, Bs, &s_last, (r % per) % p.wgx, (r % per) / p.wgx, r / per, p.wgx, p.wgy, p.wnz);
  }
}

// Weight (+ bias) gradients of up to kGroup layers in ONE launch: they are needed only when the optimizer steps, not by the
// backward pass itself, so the autograd functions can postpone them (train_ops.py) and the serial chain of a backward pass
// shrinks to its input-gradient products.  first[g] = first workgroup of problem g.
// A REGULAR weight gradient (output dimensions multiples of 64, 16-byte aligned rows, no mask; the host checks) on 64 x 64 output
// tiles: the workgroup's four waves own a 32 x 32 quadrant each and walk the WHOLE k-tile (64 contraction rows) -- no partial tiles
// to merge in LDS, and every staged float feeds two matrix instructions instead of one (the 32 x 32 tile loads 256 bytes per
// contraction row for 1 024 multiply-adds, this one 512 for 4 096).  Tiles k-major in LDS without padding, the column index rotated
// by 32 on odd rows: lane (i, h) of a matrix instruction reads row k + h, so the two half-waves fall on the two halves of the banks.
// The contraction is split over workgroups; the partial tiles travel through scratch WITHOUT cache maintenance: agent-scope relaxed
// stores / loads (write-through / bypass of the XCD's L2) around the tile's counter, instead of the two device-scope fences of the
// small tile's protocol, which write back and invalidate a whole L2 each and serialise when hundreds of workgroups fence at once
// (twelve 256 x 256 gradients, contraction cut four ways: 141 us with fences).  The last workgroup of a tile adds the partial tiles
// in split order: bit-reproducible.
constexpr int T64 = 64;
constexpr int kSlots64 = 4;                       // 32 x 32 scratch slots one 64 x 64 partial tile (+ its 64 bias sums) takes
static_assert(kSlots64 * TILE_WS >= T64 * T64 + T64, "a 64 x 64 partial tile fits four scratch slots");
static_assert(2 * T64 * T64 <= 2 * BKW * LDP, "the 64 x 64 tiles fit the small tile's LDS");
__device__ __forceinline__ void wgrad_tile64(const SArgs& a, float* As, float* Bs, int* s_last_p, int bx, int by, int bz, int gx,
                                             int gy, int nz) {
  int& s_last = *s_last_p;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int m0 = by * T64, n0 = bx * T64;         // output rows (columns of A = dy), output columns (columns of B = x)
  const int k_begin = bz * a.kper, k_end = min(a.K, k_begin + a.kper);
  const bool want_db = a.db != nullptr && bx == 0;
  const int lr = t >> 4, lc = 4 * (t & 15);       // this thread's slot of a k-tile: rows lr + 16 i, four columns at lc
  const float* const pa = a.A + (size_t)lr * a.lda + m0 + lc;
  const float* const pb = a.B + (size_t)lr * a.ldb + n0 + lc;
  float4 ra[2][4], rb[2][4];
  float4 dbp = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_tiles = [&](auto slot_c, int k0) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = k0 + 16 * i;                   // + lr: inside pa / pb
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), u = v;
      if (k + lr < k_end) {
        v = *reinterpret_cast<const float4*>(pa + (size_t)k * a.lda);
        u = *reinterpret_cast<const float4*>(pb + (size_t)k * a.ldb);
        if (a.rowdiv) { const float f = 1.f / a.rowdiv[k + lr]; v.x *= f; v.y *= f; v.z *= f; v.w *= f; }
      }
      if (want_db) { dbp.x += v.x; dbp.y += v.y; dbp.z += v.z; dbp.w += v.w; }
      ra[slot][i] = v; rb[slot][i] = u;
    }
  };
  auto store_tiles = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = lr + 16 * i, c = (lc + 32 * (r & 1)) & 63;
      *reinterpret_cast<float4*>(As + r * T64 + c) = ra[slot][i];
      *reinterpret_cast<float4*>(Bs + r * T64 + c) = rb[slot][i];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
  const int wr = wave >> 1, wc = wave & 1;
  const int ca = (32 * wr + li + 32 * lh) & 63, cb = (32 * wc + li + 32 * lh) & 63;   // rows kk + lh have parity lh (kk even)
  auto body = [&](int k0, auto slot_c) __attribute__((always_inline)) {
    store_tiles(slot_c);
    __syncthreads();
    if (k0 + 2 * T64 < k_end) load_tiles(slot_c, k0 + 2 * T64);
    const int kn = min(T64, k_end - k0);
#pragma unroll
    for (int kk = 0; kk < T64; kk += 2)
      if (kk < kn) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kk + lh) * T64 + ca], Bs[(kk + lh) * T64 + cb], acc, 0, 0, 0);
    __syncthreads();
  };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  load_tiles(S0, k_begin);
  if (k_begin + T64 < k_end) load_tiles(S1, k_begin + T64);
  {
    int k0 = k_begin;
    for (; k0 + T64 < k_end; k0 += 2 * T64) { body(k0, S0); body(k0 + T64, S1); }
    if (k0 < k_end) body(k0, S0);
  }
  // bias gradient: the 16 row groups' partial column sums meet in LDS (the tiles are dead)
  float dbv = 0.f;
  if (want_db) {
    *reinterpret_cast<float4*>(As + lr * T64 + lc) = dbp;
    __syncthreads();
    if (t < T64) {
#pragma unroll
      for (int r = 0; r < 16; r++) dbv += As[r * T64 + t];
    }
  }
  // this wave's quadrant, C layout of the instruction: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const int splits = nz;
  if (splits > 1) {
    const int ntiles = gx * gy, tile = by * gx + bx;
    float* mine = a.ws + ((size_t)bz * ntiles + tile) * (kSlots64 * TILE_WS);
#pragma unroll
    for (int e = 0; e < 16; e++) st_agent(mine + (32 * wr + (e & 3) + 8 * (e >> 2) + 4 * lh) * T64 + 32 * wc + li, acc[e]);
    if (want_db && t < T64) st_agent(mine + T64 * T64 + t, dbv);
    publish_partials();                            // every store of this thread has reached the coherent level ...
    __syncthreads();                               // ... and so have the workgroup's, before its arrival is counted
    if (t == 0) s_last = atomicAdd(&a.counters[tile], 1u) == (unsigned)(splits - 1);
    __syncthreads();
    if (!s_last) return;
    acquire_partials();
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    dbv = 0.f;
    for (int sp = 0; sp < splits; sp++) {                      // fixed order: the sum does not depend on who arrived when
      const float* q = a.ws + ((size_t)sp * ntiles + tile) * (kSlots64 * TILE_WS);
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] += ld_agent(q + (32 * wr + (e & 3) + 8 * (e >> 2) + 4 * lh) * T64 + 32 * wc + li);
      if (want_db && t < T64) dbv += ld_agent(q + T64 * T64 + t);
    }
    if (t == 0) a.counters[tile] = 0;
  }
  if (want_db && t < T64) a.db[m0 + t] = dbv;
#pragma unroll
  for (int e = 0; e < 16; e++)
    a.C[(size_t)(m0 + 32 * wr + (e & 3) + 8 * (e >> 2) + 4 * lh) * a.ldc + n0 + 32 * wc + li] = acc[e];
}

constexpr int kGroup = 12;
constexpr int kGroupNoSplitTiles = 256;       // sgrl_linear_wgrad_group: launches with at least this many output tiles do not split
struct GroupArgs { SArgs w[kGroup]; int first[kGroup + 1]; int gx[kGroup], gy[kGroup], nz[kGroup]; int big[kGroup]; int n; };
__global__ __launch_bounds__(256) void k_sgemm_wgroup(GroupArgs p) {
  __shared__ __attribute__((aligned(16))) float As[BKW][LDP];
  __shared__ __attribute__((aligned(16))) float Bs[BKW][LDP];
  __shared__ int s_last;
  int g = 0;
  while (g + 1 < p.n && (int)blockIdx.x >= p.first[g + 1]) g++;
  const SArgs a = p.w[g];                       // g is uniform: scalar loads from the kernel-argument segment
  const int gx = p.gx[g], gy = p.gy[g], nz = p.nz[g];
  const int r = blockIdx.x - p.first[g], per = gx * gy;
  if (p.big[g]) wgrad_tile64(a, &As[0][0], &Bs[0][0], &s_last, (r % per) % gx, (r % per) / gx, r / per, gx, gy, nz);
  else sgemm_tile<true, true, BKW>(a, As, Bs, &s_last, (r % per) % gx, (r % per) / gx, r / per, gx, gy, nz);
}

// db alone (no weight gradient requested): column sums of the masked g, 64 columns per workgroup
__global__ __launch_bounds__(256) void k_colsum(const float* g, int ldg, const float* mask, int ldm, float* db, int M, int N) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r0 = threadIdx.x >> 6;
  float s = 0.f;
  if (c < N)
    for (int m = r0; m < M; m += 4) {
      const float v = g[(size_t)m * ldg + c];
      s += (!mask || mask[(size_t)m * ldm + c] > 0.f) ? v : 0.f;
    }
  red[r0][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && c < N) db[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// drowdiv[m] = -(sum_n dy[m][n] y[m][n]) / rowdiv[m]   (y = (x w^T + b) / rowdiv: d/d rowdiv of the forward); one wave per row
// (blockIdx.y = 1: the second argument set -- the twin critics' pair of layers in one launch)
__global__ __launch_bounds__(256) void k_rowdot(const float* dy, int lddy, const float* y, int ldy, const float* rowdiv, float* out,
                                                int M, int N, const float* dy1, const float* y1, const float* rowdiv1, float* out1) {
  if (blockIdx.y) { dy = dy1; y = y1; rowdiv = rowdiv1; out = out1; }
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  float s = 0.f;
  for (int n = lane; n < N; n += 64) s += dy[(size_t)m * lddy + n] * y[(size_t)m * ldy + n];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) out[m] = -s / rowdiv[m];
}

// Gram invariants of a node's three 32-vectors (reference SEActor.py:94-98 / subequivariant_attentions.py:38-44):
//   G = Z' Z  (32 x 32, stored flat, 1024 values)   and   fn = ||G||_F + 1.     One 256-thread workgroup per node.
__global__ __launch_bounds__(256) void k_gram_fwd(const float* __restrict__ z, float* gram, float* fn, int M) {
  __shared__ float zs[96];
  __shared__ float red[4];
  const int m = blockIdx.x, t = threadIdx.x;
  if (t < 96) zs[t] = z[(size_t)m * 96 + t];
  __syncthreads();
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int o = t + 256 * i, a = o >> 5, c = o & 31;
    const float g = zs[a] * zs[c] + zs[32 + a] * zs[32 + c] + zs[64 + a] * zs[64 + c];
    gram[(size_t)m * 1024 + o] = g;
    sq += g * g;
  }
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  if ((t & 63) == 0) red[t >> 6] = sq;
  __syncthreads();
  if (t == 0) fn[m] = sqrtf((red[0] + red[1]) + (red[2] + red[3])) + 1.0f;
}
// dz = Z (D + D'),  D = dG + (dfn / ||G||) G   (||G|| = fn - 1; no norm term where it is zero)
__global__ __launch_bounds__(256) void k_gram_bwd(const float* __restrict__ z, const float* __restrict__ dgram,
                                                  const float* __restrict__ dfn, const float* __restrict__ fn, float* dz, int M) {
  __shared__ float zs[96];
  __shared__ float D[32][33];
  const int m = blockIdx.x, t = threadIdx.x;
  if (t < 96) zs[t] = z[(size_t)m * 96 + t];
  __syncthreads();
  const float nrm = fn[m] - 1.0f;
  const float coef = (dfn && nrm > 0.f) ? dfn[m] / nrm : 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int o = t + 256 * i, a = o >> 5, c = o & 31;
    const float g = zs[a] * zs[c] + zs[32 + a] * zs[32 + c] + zs[64 + a] * zs[64 + c];
    D[a][c] = (dgram ? dgram[(size_t)m * 1024 + o] : 0.f) + coef * g;
  }
  __syncthreads();
  if (t < 96) {
    const int sx = t >> 5, a = t & 31;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 32; c++) acc += (D[a][c] + D[c][a]) * zs[32 * sx + c];
    dz[(size_t)m * 96 + t] = acc;
  }
}

// The same invariants on the LOWER TRIANGLE of the symmetric Z'Z only, packed k = a (a + 1) / 2 + b (a >= b): 528 values instead of
// 1 024.  A linear layer on vec(Z'Z) sees every off-diagonal entry twice, so  W vec(G) = W' tri(G)  with W'[n][k] = W[n][a 32 + b] +
// W[n][b 32 + a] (a > b), W[n][a 33] (a = b): the layer's three products (forward, input gradient, weight gradient) contract over /
// produce 528 columns instead of 1 024 (what the rollout's weight pack does for its blocked 576: csrc/set_actor.hip k_pack FOLD).
constexpr int TRI = 528;
__device__ __forceinline__ void tri_ab(int k, int* a, int* b) {
  int r = (int)((sqrtf(8.f * k + 1.f) - 1.f) * 0.5f);
  while ((r + 1) * (r + 2) / 2 <= k) r++;
  while (r * (r + 1) / 2 > k) r--;
  *a = r; *b = k - r * (r + 1) / 2;
}
__global__ __launch_bounds__(256) void k_gram_tri_fwd(const float* __restrict__ z, float* gram, float* fn, int M) {
  __shared__ float zs[96];
  __shared__ float red[4];
  const int m = blockIdx.x, t = threadIdx.x;
  if (t < 96) zs[t] = z[(size_t)m * 96 + t];
  __syncthreads();
  float sq = 0.f;
  for (int k = t; k < TRI; k += 256) {
    int a, c;
    tri_ab(k, &a, &c);
    const float g = zs[a] * zs[c] + zs[32 + a] * zs[32 + c] + zs[64 + a] * zs[64 + c];
    gram[(size_t)m * TRI + k] = g;
    sq += (a == c ? 1.f : 2.f) * g * g;
  }
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  if ((t & 63) == 0) red[t >> 6] = sq;
  __syncthreads();
  if (t == 0) fn[m] = sqrtf((red[0] + red[1]) + (red[2] + red[3])) + 1.0f;
}
// dz = Z S,  S symmetric: S[a][b] = S[b][a] = dtri[k(a, b)] + 2 c G[a][b] (a > b), S[a][a] = 2 dtri[k(a, a)] + 2 c G[a][a],  c = dfn / ||G||
__global__ __launch_bounds__(256) void k_gram_tri_bwd(const float* __restrict__ z, const float* __restrict__ dtri,
                                                      const float* __restrict__ dfn, const float* __restrict__ fn, float* dz, int M) {
  __shared__ float zs[96];
  __shared__ float S[32][33];
  const int m = blockIdx.x, t = threadIdx.x;
  if (t < 96) zs[t] = z[(size_t)m * 96 + t];
  __syncthreads();
  const float nrm = fn[m] - 1.0f;
  const float coef = (dfn && nrm > 0.f) ? dfn[m] / nrm : 0.f;
  for (int k = t; k < TRI; k += 256) {
    int a, c;
    tri_ab(k, &a, &c);
    const float g = zs[a] * zs[c] + zs[32 + a] * zs[32 + c] + zs[64 + a] * zs[64 + c];
    const float d = dtri ? dtri[(size_t)m * TRI + k] : 0.f;
    const float v = (a == c ? 2.f * d : d) + 2.f * coef * g;
    S[a][c] = v;
    S[c][a] = v;
  }
  __syncthreads();
  if (t < 96) {
    const int sx = t >> 5, a = t & 31;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 32; c++) acc += S[a][c] * zs[32 * sx + c];
    dz[(size_t)m * 96 + t] = acc;
  }
}
// W [rows][1024] -> W' [rows][528] (fold) and dW' [rows][528] -> dW [rows][1024] (unfold: both mirror entries get dW'[k]) for up to
// kFoldMax matrices per launch (blockIdx.y picks one): all the invariant layers of a network at once
constexpr int kFoldMax = 16;
struct FoldArgs { const float* src[kFoldMax]; float* dst[kFoldMax]; int rows[kFoldMax]; };
__global__ __launch_bounds__(256) void k_fold_sym(FoldArgs p) {
  const float* src = p.src[blockIdx.y];
  float* dst = p.dst[blockIdx.y];
  const int n = p.rows[blockIdx.y] * TRI;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = i / TRI, k = i - r * TRI;
    int a, b;
    tri_ab(k, &a, &b);
    const float* row = src + (size_t)r * 1024;
    dst[i] = a == b ? row[a * 33] : row[a * 32 + b] + row[b * 32 + a];
  }
}
__global__ __launch_bounds__(256) void k_unfold_sym(FoldArgs p) {
  const float* src = p.src[blockIdx.y];
  float* dst = p.dst[blockIdx.y];
  const int n = p.rows[blockIdx.y] * 1024;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = i >> 10, o = i & 1023, x = o >> 5, y = o & 31;
    const int a = x > y ? x : y, b = x > y ? y : x;
    dst[i] = src[(size_t)r * TRI + a * (a + 1) / 2 + b];
  }
}

// Limb attention of one environment (reference subequivariant_attentions.py:90-151 between the projections): H = 2 heads of
// 128 channels, L <= 14 limbs.  qkv [B, L, 768] = (q | k | v) as the stacked projection leaves them (q is scaled by `scale`
// here); the vector values vg[j][s][h][d] are vgp [B, L, 3, 252] (d < 126: the projected part, 126 per head) and the node's
// gravity / direction pair gdir [B, L, 3, 2] (d = 126, 127) -- the concatenation is never built; bias [2, L, L] or null.
//   w = softmax_j(scale q_i . k_j + bias);   o[i][c] = sum_j w[h(c)][i][j] v[j][c];   og[i][s][c] = sum_j w[h(c)][i][j] vg[j][s][c]
// One 128-thread workgroup per (environment, head) -- round 6; rounds 3-5 ran one 256-thread workgroup per environment and went to
// memory up to ten times in a row (the backward: four operand passes, then q, k, do and the three dog slices again).  A workgroup is
// bound by those round trips, not by its arithmetic: EVERY global load of a kernel is now issued in one batch at its top (14 x 10
// registers at most), and the backward keeps its eight operand slices in LDS for both of its halves.  Scores by a pair of lanes per
// (i, j) entry (half of the 128 channels each) from LDS rows of pitch 129 (the rows of different limbs start on different banks),
// outputs by a thread per channel.  w is kept for the backward.
constexpr int AL = 14, AQ = 129;
__device__ __forceinline__ float vg_at(const float* vgp, const float* gdir, size_t node, int sx, int h, int d) {
  return d < 126 ? vgp[(node * 3 + sx) * 252 + h * 126 + d] : gdir[(node * 3 + sx) * 2 + (d - 126)];
}
__global__ __launch_bounds__(128) void k_attn_fwd(const float* __restrict__ qkv, const float* __restrict__ vgp,
                                                  const float* __restrict__ gdir, const float* __restrict__ bias, float scale,
                                                  float* wout, float* o, float* og, int L) {
  __shared__ float qs[AL * AQ], ks[AL * AQ];
  __shared__ float sc[AL * AL];
  const int b = blockIdx.x >> 1, h = blockIdx.x & 1, d = threadIdx.x, c = 128 * h + d;
  const size_t n0 = (size_t)b * L;
  float qv[AL], kv[AL], vv[4][AL];
#pragma unroll
  for (int i = 0; i < AL; i++)
    if (i < L) {
      const float* row = qkv + (n0 + i) * 768 + c;
      qv[i] = row[0]; kv[i] = row[256]; vv[0][i] = row[512];
#pragma unroll
      for (int sx = 0; sx < 3; sx++) vv[1 + sx][i] = vg_at(vgp, gdir, n0 + i, sx, h, d);
    }
#pragma unroll
  for (int i = 0; i < AL; i++)
    if (i < L) { qs[i * AQ + d] = qv[i] * scale; ks[i * AQ + d] = kv[i]; }
  __syncthreads();
  const int half = d & 1;
  for (int e = d >> 1; e < L * L; e += 64) {        // (both lanes of a pair run the same trips)
    const float* a = qs + (e / L) * AQ + 64 * half;
    const float* k = ks + (e % L) * AQ + 64 * half;
    float acc = 0.f;
#pragma unroll 8
    for (int x = 0; x < 64; x++) acc += a[x] * k[x];
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) sc[e] = bias ? acc + bias[h * L * L + e] : acc;
  }
  __syncthreads();
  if (d < L) {
    float* row = sc + d * L;
    float mx = row[0];
    for (int j = 1; j < L; j++) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < L; j++) { row[j] = expf(row[j] - mx); sum += row[j]; }
    for (int j = 0; j < L; j++) row[j] = row[j] / sum;
  }
  __syncthreads();
  for (int idx = d; idx < L * L; idx += 128) wout[((size_t)b * 2 + h) * L * L + idx] = sc[idx];
#pragma unroll
  for (int i = 0; i < AL; i++)
    if (i < L) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < AL; j++)
        if (j < L) {
          const float wij = sc[i * L + j];
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] += wij * vv[q][j];
        }
      o[(n0 + i) * 256 + c] = acc[0];
#pragma unroll
      for (int sx = 0; sx < 3; sx++) og[((n0 + i) * 3 + sx) * 256 + c] = acc[1 + sx];
    }
}
// Backward: dw = do v' + sum_s dog_s vg_s' (per head), ds = w (dw - sum_j w dw), dq = scale ds k, dk = ds' (scale q), dv = w' do,
// dvg = w' dog.  dqkv [B, L, 768] is written packed; dvg goes to dvgp [B, L, 3, 252] and, per head, dgdh [B, L, 3, 2 heads, 2];
// ds [B, 2, L, L] is written out as well (its sum over the environments is the gradient of the relation bias).
// Dynamic LDS (attn_bwd_lds_bytes): A [4][L][AQ] = do | dog_0..2, V [4][L][AQ] = v | vg_0..2 (this head's channels), w, dw [L][L].
__host__ __device__ constexpr int attn_bwd_lds_bytes(int L) { return (8 * L * AQ + 2 * L * L) * (int)sizeof(float); }
__global__ __launch_bounds__(128) void k_attn_bwd(const float* __restrict__ qkv, const float* __restrict__ vgp,
                                                  const float* __restrict__ gdir, float scale, const float* __restrict__ win,
                                                  const float* __restrict__ dout, const float* __restrict__ dog, float* dqkv,
                                                  float* dvgp, float* dgdh, float* ds_out, int L) {
  extern __shared__ __attribute__((aligned(16))) float attn_lds[];
  float* A = attn_lds;
  float* V = A + 4 * L * AQ;
  float* w = V + 4 * L * AQ;
  float* dw = w + L * L;
  const int b = blockIdx.x >> 1, h = blockIdx.x & 1, d = threadIdx.x, c = 128 * h + d;
  const size_t n0 = (size_t)b * L;
  float qv[AL], kv[AL], ra[4][AL], rv[4][AL];
#pragma unroll
  for (int i = 0; i < AL; i++)
    if (i < L) {
      const float* row = qkv + (n0 + i) * 768 + c;
      qv[i] = row[0]; kv[i] = row[256]; rv[0][i] = row[512];
      ra[0][i] = dout[(n0 + i) * 256 + c];
#pragma unroll
      for (int sx = 0; sx < 3; sx++) {
        ra[1 + sx][i] = dog[((n0 + i) * 3 + sx) * 256 + c];
        rv[1 + sx][i] = vg_at(vgp, gdir, n0 + i, sx, h, d);
      }
    }
  for (int idx = d; idx < L * L; idx += 128) w[idx] = win[((size_t)b * 2 + h) * L * L + idx];
#pragma unroll
  for (int i = 0; i < AL; i++)
    if (i < L) {
#pragma unroll
      for (int q = 0; q < 4; q++) { A[(q * L + i) * AQ + d] = ra[q][i]; V[(q * L + i) * AQ + d] = rv[q][i]; }
    }
  __syncthreads();
  const int half = d & 1;
  for (int e = d >> 1; e < L * L; e += 64) {        // dw[i][j]: a pair of lanes per entry, half of the channels each, all four slices
    const int i = e / L, j = e % L;
    float acc = 0.f;
    for (int q = 0; q < 4; q++) {
      const float* a = A + (q * L + i) * AQ + 64 * half;
      const float* v = V + (q * L + j) * AQ + 64 * half;
#pragma unroll 8
      for (int x = 0; x < 64; x++) acc += a[x] * v[x];
    }
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) dw[e] = acc;
  }
  __syncthreads();
  if (d < L) {                                       // softmax backward, one thread per row i; dw becomes ds in place
    float* wr = w + d * L;
    float* dr = dw + d * L;
    float dot = 0.f;
    for (int j = 0; j < L; j++) dot += wr[j] * dr[j];
    for (int j = 0; j < L; j++) dr[j] = wr[j] * (dr[j] - dot);
  }
  __syncthreads();
  for (int idx = d; idx < L * L; idx += 128) ds_out[((size_t)b * 2 + h) * L * L + idx] = dw[idx];
  // dq[i] = scale sum_j ds[i][j] k[j];  dk[j] = sum_i ds[i][j] (scale q[i])
#pragma unroll
  for (int i = 0; i < AL; i++)
    if (i < L) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < AL; j++) if (j < L) acc += dw[i * L + j] * kv[j];
      dqkv[(n0 + i) * 768 + c] = acc * scale;
    }
#pragma unroll
  for (int j = 0; j < AL; j++)
    if (j < L) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < AL; i++) if (i < L) acc += dw[i * L + j] * (qv[i] * scale);
      dqkv[(n0 + j) * 768 + 256 + c] = acc;
    }
  // dv[j] = sum_i w[i][j] do[i];  dvg[j][s] = sum_i w[i][j] dog[i][s] -- do / dog are still in this thread's registers
#pragma unroll
  for (int j = 0; j < AL; j++)
    if (j < L) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < AL; i++)
        if (i < L) {
          const float wij = w[i * L + j];
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] += wij * ra[q][i];
        }
      dqkv[(n0 + j) * 768 + 512 + c] = acc[0];
#pragma unroll
      for (int sx = 0; sx < 3; sx++) {
        if (d < 126) dvgp[((n0 + j) * 3 + sx) * 252 + h * 126 + d] = acc[1 + sx];
        else dgdh[(((n0 + j) * 3 + sx) * 2 + h) * 2 + (d - 126)] = acc[1 + sx];
      }
    }
}

// Equivariant contraction of a node's three 32-vectors with its 32 x 32 matrix (reference SEActor.py:108-110, 262-264):
//   t[s][c] = sum_a z[s][a] mat[a][c]       backward:  dz[s][a] = sum_c dt[s][c] mat[a][c],  dmat[a][c] = sum_s z[s][a] dt[s][c]
// One 128-thread half-workgroup per node (two nodes per workgroup).
__global__ __launch_bounds__(256) void k_zmat_fwd(const float* __restrict__ z, const float* __restrict__ mat, float* t, int M) {
  __shared__ float zs[2][96];
  const int half = threadIdx.x >> 7, u = threadIdx.x & 127, m = blockIdx.x * 2 + half;
  if (m < M && u < 96) zs[half][u] = z[(size_t)m * 96 + u];
  __syncthreads();
  if (m >= M || u >= 96) return;
  const int sx = u >> 5, c = u & 31;
  const float* mm = mat + (size_t)m * 1024 + c;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 32; a++) acc += zs[half][32 * sx + a] * mm[32 * a];
  t[(size_t)m * 96 + u] = acc;
}
__global__ __launch_bounds__(256) void k_zmat_bwd(const float* __restrict__ z, const float* __restrict__ mat,
                                                  const float* __restrict__ dt, float* dz, float* dmat, int M) {
  __shared__ float zs[2][96], ds[2][96];
  const int half = threadIdx.x >> 7, u = threadIdx.x & 127, m = blockIdx.x * 2 + half;
  if (m < M && u < 96) { zs[half][u] = z[(size_t)m * 96 + u]; ds[half][u] = dt[(size_t)m * 96 + u]; }
  __syncthreads();
  if (m >= M) return;
  const float* mm = mat + (size_t)m * 1024;
#pragma unroll
  for (int i = 0; i < 8; i++) {                     // dmat: 1024 entries, eight per thread
    const int o = u + 128 * i, a = o >> 5, c = o & 31;
    dmat[(size_t)m * 1024 + o] = zs[half][a] * ds[half][c] + zs[half][32 + a] * ds[half][32 + c] + zs[half][64 + a] * ds[half][64 + c];
  }
  if (u < 96) {                                      // dz[s][a] = sum_c dt[s][c] mat[a][c]
    const int sx = u >> 5, a = u & 31;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 32; c++) acc += ds[half][32 * sx + c] * mm[32 * a + c];
    dz[(size_t)m * 96 + u] = acc;
  }
}

// Residual + LayerNorm over 128 columns (reference SEActor.py:90-91, 113-114, 164: norm(ng + update)), forward and backward in
// ONE launch each -- the update's rows are few (700 .. 4 200), so a norm is launch-bound: torch's add + native_layer_norm and its
// three backward kernels are five launches per site (eleven for the twin critics' unbind / stack form).  `nets` networks are
// stacked along the row axis, `rows` rows each; network i normalises with w[i], b[i].  One wavefront per row, two columns per lane.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__global__ __launch_bounds__(256) void k_add_ln_fwd(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ w0,
                                                    const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
                                                    float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd, int rows, int total, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= total) return;
  const bool second = row >= rows;
  const float* w = second ? w1 : w0;
  const float* b = second ? b1 : b0;
  float2 v = *reinterpret_cast<const float2*>(x + (size_t)row * 128 + 2 * lane);
  if (res) { const float2 r = *reinterpret_cast<const float2*>(res + (size_t)row * 128 + 2 * lane); v.x += r.x; v.y += r.y; }
  const float mean = wave_sum(v.x + v.y) * (1.f / 128.f);
  const float d0 = v.x - mean, d1 = v.y - mean;
  const float var = wave_sum(d0 * d0 + d1 * d1) * (1.f / 128.f);
  const float r = 1.0f / sqrtf(var + eps);
  const float2 xh = make_float2(d0 * r, d1 * r);
  const float2 wv = *reinterpret_cast<const float2*>(w + 2 * lane), bv = *reinterpret_cast<const float2*>(b + 2 * lane);
  *reinterpret_cast<float2*>(y + (size_t)row * 128 + 2 * lane) = make_float2(xh.x * wv.x + bv.x, xh.y * wv.y + bv.y);
  if (xhat) {
    *reinterpret_cast<float2*>(xhat + (size_t)row * 128 + 2 * lane) = xh;
    if (lane == 0) rstd[row] = r;
  }
}
// workgroups [0, nrow_blocks): dx of four rows each; the rest: (network, 16 columns) -> dw = sum_rows dy xhat, db = sum_rows dy, the
// rows summed in a fixed order (sixteen interleaved row groups, then the groups in order): bit-reproducible, no atomics.  (Rounds 4-5:
// 32 columns x eight groups with eight rows in flight -- at the update's 1 792 rows a thread then made 28 dependent trips to memory,
// 12.4 us per launch; sixteen groups with sixteen rows in flight make 7.)
constexpr int kLnCols = 16, kLnGroups = 16, kLnFlight = 16;
__global__ __launch_bounds__(256) void k_add_ln_bwd(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                    const float* __restrict__ w0, const float* __restrict__ w1, float* __restrict__ dx,
                                                    float* dw0, float* db0, float* dw1, float* db1, int rows, int total, int nrow_blocks) {
  __shared__ float red[2][kLnGroups][kLnCols];
  if ((int)blockIdx.x < nrow_blocks) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= total) return;
    const float* w = row >= rows ? w1 : w0;
    const float2 g0 = *reinterpret_cast<const float2*>(dy + (size_t)row * 128 + 2 * lane);
    const float2 wv = *reinterpret_cast<const float2*>(w + 2 * lane);
    const float2 xh = *reinterpret_cast<const float2*>(xhat + (size_t)row * 128 + 2 * lane);
    const float gx = g0.x * wv.x, gy = g0.y * wv.y;
    const float c1 = wave_sum(gx + gy) * (1.f / 128.f);
    const float c2 = wave_sum(gx * xh.x + gy * xh.y) * (1.f / 128.f);
    const float r = rstd[row];
    *reinterpret_cast<float2*>(dx + (size_t)row * 128 + 2 * lane) = make_float2(r * (gx - c1 - xh.x * c2), r * (gy - c1 - xh.y * c2));
    return;
  }
  constexpr int kPer = 128 / kLnCols;          // workgroups per network
  const int q = blockIdx.x - nrow_blocks, net = q / kPer, col = threadIdx.x % kLnCols, c = kLnCols * (q % kPer) + col, grp = threadIdx.x / kLnCols;
  float* dw = net ? dw1 : dw0;
  float* db = net ? db1 : db0;
  if (!dw && !db) return;
  float aw = 0.f, ab = 0.f;
  const size_t base = (size_t)net * rows * 128 + c;
  int r = grp;
  for (; r + kLnGroups * (kLnFlight - 1) < rows; r += kLnGroups * kLnFlight) {       // kLnFlight of this group's rows in flight; summed in the same order as one at a time
    float g[kLnFlight], xh[kLnFlight];
#pragma unroll
    for (int u = 0; u < kLnFlight; u++) { g[u] = dy[base + (size_t)(r + kLnGroups * u) * 128]; xh[u] = xhat[base + (size_t)(r + kLnGroups * u) * 128]; }
#pragma unroll
    for (int u = 0; u < kLnFlight; u++) { aw += g[u] * xh[u]; ab += g[u]; }
  }
  for (; r < rows; r += kLnGroups) {
    const float g = dy[base + (size_t)r * 128];
    aw += g * xhat[base + (size_t)r * 128];
    ab += g;
  }
  red[0][grp][col] = aw;
  red[1][grp][col] = ab;
  __syncthreads();
  if (grp == 0) {
    float sw = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < kLnGroups; k++) { sw += red[0][k][col]; sb += red[1][k][col]; }
    if (dw) dw[c] = sw;
    if (db) db[c] = sb;
  }
}

// Three traversal-index embeddings concatenated (reference SEActor.py:18-31: ConcatPositionalEmbedding, widths 42 | 42 | 44 of 128):
//   out[l][c] = W_t[idx[t][l]][c - off_t],  t = the table column c falls into;   backward: dW_t[r][j] = sum_{l: idx[t][l] == r} dout[l][off_t + j]
// (written densely for all `rows` rows of each table, limbs summed in order: bit-reproducible).  One launch each instead of three
// index_select + a concatenation / three times (embedding_dense_backward + fill + copy).
struct Embed3 { const long long* idx; const float* w[3]; float* dw[3]; int n[3]; int L, rows, ld; };
__global__ __launch_bounds__(128) void k_embed3_fwd(Embed3 e, float* __restrict__ out) {
  const int l = blockIdx.x, c = threadIdx.x;
  if (c >= e.ld) return;
  const int t = c < e.n[0] ? 0 : (c < e.n[0] + e.n[1] ? 1 : 2);
  const int off = t == 0 ? 0 : (t == 1 ? e.n[0] : e.n[0] + e.n[1]);
  const long long r = e.idx[(size_t)t * e.L + l];
  out[(size_t)l * e.ld + c] = e.w[t][(size_t)r * e.n[t] + (c - off)];
}
__global__ __launch_bounds__(128) void k_embed3_bwd(Embed3 e, const float* __restrict__ dout) {
  const int r = blockIdx.x, c = threadIdx.x;                 // one workgroup per table row, one thread per output column
  if (c >= e.ld) return;
  const int t = c < e.n[0] ? 0 : (c < e.n[0] + e.n[1] ? 1 : 2);
  const int off = t == 0 ? 0 : (t == 1 ? e.n[0] : e.n[0] + e.n[1]);
  if (!e.dw[t]) return;
  float s = 0.f;
  for (int l = 0; l < e.L; l++)
    if (e.idx[(size_t)t * e.L + l] == r) s += dout[(size_t)l * e.ld + c];
  e.dw[t][(size_t)r * e.n[t] + (c - off)] = s;
}

// Optimizer steps over a TABLE of tensors (reference agent.py:161-177: clip_grad_norm_ + Adam.step per network; common/functional.py:
// 7-10: the soft target update).  torch's multi-tensor kernels take a few dozen tensors per launch through their argument
// block: gradient clipping + Adam + soft update of the ~400 parameter tensors of actor and critics are ~50 launches of ~18 us per
// TD3 update.  Here the tensors' addresses sit in a device table (6 x int64 per tensor: param, grad, exp_avg, exp_avg_sq, step,
// numel), the work is cut into chunks of kOptChunk elements ((tensor, first element) pairs), one workgroup per chunk:
//   k_opt_sqnorm   partial[c] = sum of grad^2 over chunk c
//   k_opt_total    total = sum_c partial[c] in chunk order (one workgroup; bit-reproducible); every tensor's step += 1
//   k_opt_adam     g *= min(1, max_norm / (sqrt(total) + 1e-6))  (written back, as clip_grad_norm_ does);  Adam as torch's fused kernel
//                  computes it: m = lerp(m, g, 1 - b1), v = b2 v + (1 - b2) g g, p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   k_opt_lerp     dst = dst (1 - tau) + tau src   (table rows: dst, src, -, -, -, numel)
constexpr int kOptChunk = 1024;
struct OptRow { float* p; float* g; float* m; float* v; float* step; long long n; };
__global__ __launch_bounds__(256) void k_opt_sqnorm(const OptRow* __restrict__ tab, const int2* __restrict__ chunks, float* __restrict__ partial) {
  __shared__ float red[4];
  const int2 c = chunks[blockIdx.x];
  const OptRow r = tab[c.x];
  const long long end = min(r.n, (long long)c.y + kOptChunk);
  float s = 0.f;
  for (long long i = c.y + threadIdx.x; i < end; i += 256) { const float g = r.g[i]; s += g * g; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void k_opt_total(const float* __restrict__ partial, int n_chunks, const OptRow* __restrict__ tab, int n_tensors,
                                                   float* __restrict__ total) {
  __shared__ float red[256];
  if (partial) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n_chunks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) *total = red[0];
  }
  for (int i = threadIdx.x; i < n_tensors; i += 256) *tab[i].step += 1.0f;
}
// (the hyper-parameters are doubles and mix into the float arithmetic exactly where torch's fused kernel lets them: fused_adam_utils.cuh)
__global__ __launch_bounds__(256) void k_opt_adam(const OptRow* __restrict__ tab, const int2* __restrict__ chunks, double lr, double b1, double b2, double eps,
                                                  const float* __restrict__ total, float max_norm) {
  const int2 c = chunks[blockIdx.x];
  const OptRow r = tab[c.x];
  const long long end = min(r.n, (long long)c.y + kOptChunk);
  const float coef = total ? fminf(1.0f, max_norm / (sqrtf(*total) + 1e-6f)) : 1.0f;
  const double t = (double)*r.step;                               // already advanced by k_opt_total
  const float bc1 = (float)(1.0 - pow(b1, t));
  const float bc2_sqrt = sqrtf((float)(1.0 - pow(b2, t)));
  const float step_size = (float)(lr / (double)bc1);
  for (long long i = c.y + threadIdx.x; i < end; i += 256) {
    float g = r.g[i];
    if (total) { g *= coef; r.g[i] = g; }
    const float m = (float)(b1 * (double)r.m[i] + (1.0 - b1) * (double)g);
    const float v = (float)(b2 * (double)r.v[i] + (1.0 - b2) * (double)g * (double)g);
    r.m[i] = m; r.v[i] = v;
    const float denom = (float)((double)(sqrtf(v) / bc2_sqrt) + eps);
    r.p[i] -= step_size * m / denom;
  }
}
__global__ __launch_bounds__(256) void k_opt_lerp(const OptRow* __restrict__ tab, const int2* __restrict__ chunks, float tau) {
  const int2 c = chunks[blockIdx.x];
  const OptRow r = tab[c.x];
  const long long end = min(r.n, (long long)c.y + kOptChunk);
  const float keep = 1.0f - tau;
  for (long long i = c.y + threadIdx.x; i < end; i += 256) r.p[i] = r.p[i] * keep + tau * r.g[i];
}

// Only the weight gradient (AT) splits its contraction, and only where that pays.  (These thresholds date from rounds 2-4, when the
// last-workgroup reduction went through two device-scope fences that cost ~10-15 us on the eight-XCD chip -- a 256 x 256 gradient over
// 700 rows was 20 us unsplit, 27 us in three splits; a 30 x 128 gradient over 2 100 rows 51 us unsplit, 17 us in six.  Since the
// partial tiles travel as agent-scope stores / loads (st_agent above) a split costs a few microseconds; the rule was kept.)  So: a
// handful of tiles, or a contraction of a dozen k-tiles and more.  At least two k-tiles per split, never more (split, tile)
// slots than the scratch holds.
// grid of one product: tiles + the contraction split (weight gradient only, see above); fills a.kper / a.ws / a.counters
// SGRL_TRAIN_RM=0 (probe): the transposing staging of rounds 2-4 instead of the row-major one
bool row_major_staging() {
  constexpr bool on = true;
  return on;
}
template <bool AT>
int plan(SArgs& a, float* ws, int* tn, int* tm, int* splits_out, int64_t ws_slot0 = 0, int counter0 = 0, int force_splits = 0) {
  constexpr int BK = AT ? BKW : BKF;
  *tm = (a.M + BT - 1) / BT; *tn = (a.N + (!AT && a.tail ? a.ntail : 0) + BT - 1) / BT;   // + the column tiles of an appended block
  const int ntiles = *tm * *tn;
  if (*tm > 65535) return tfail(SGRL_ERR_LIMIT, "train gemm: too many row tiles");
  const int ktiles = (a.K + BK - 1) / BK;
  int splits = 1;
  if (AT && ws && ktiles >= 4 && (ntiles <= 8 || (ktiles >= 12 && ntiles <= 128)) && counter0 + ntiles <= kCounters)
    splits = std::max(1, std::min({(256 + ntiles - 1) / ntiles, ktiles / 2, (int)((kWsTiles - ws_slot0) / ntiles)}));
  if (AT && ws && force_splits > 1 && counter0 + ntiles <= kCounters)       // the grouped launch evens out its contractions (below)
    splits = std::max(1, std::min({force_splits, ktiles / 2, (int)((kWsTiles - ws_slot0) / ntiles)}));
  a.kper = ((ktiles + splits - 1) / splits) * BK;
  *splits_out = (a.K + a.kper - 1) / a.kper;                  // no empty splits
  a.ws = ws ? ws + ws_slot0 * TILE_WS : nullptr;      // this product's own (split, tile) slots and counters
  a.counters = ws ? reinterpret_cast<unsigned*>(ws + kWsTiles * TILE_WS) + counter0 : nullptr;
  return SGRL_OK;
}
template <bool AT, bool BTR>
int launch(SArgs a, float* ws, hipStream_t st) {
  constexpr int BK = AT ? BKW : BKF;
  int tn, tm, splits;
  const int rc = plan<AT>(a, ws, &tn, &tm, &splits);
  if (rc != SGRL_OK) return rc;
  if (row_major_staging()) hipLaunchKernelGGL((k_sgemm<AT, BTR, BK, true>), dim3(tn, tm, splits), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_sgemm<AT, BTR, BK, false>), dim3(tn, tm, splits), dim3(256), 0, st, a);
  { int lrc = SGRL_OK; if (!launched("train gemm: kernel launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}
int launch_bwd(SArgs d, SArgs w, float* ws, hipStream_t st) {
  BwdArgs p;
  int dsplits;
  int rc = plan<false>(d, nullptr, &p.dgx, &p.dgy, &dsplits);
  if (rc != SGRL_OK) return rc;
  rc = plan<true>(w, ws, &p.wgx, &p.wgy, &p.wnz);
  if (rc != SGRL_OK) return rc;
  p.d = d; p.w = w; p.nd = p.dgx * p.dgy;
  if (row_major_staging()) hipLaunchKernelGGL(k_sgemm_bwd<true>, dim3(p.nd + p.wgx * p.wgy * p.wnz), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(k_sgemm_bwd<false>, dim3(p.nd + p.wgx * p.wgy * p.wnz), dim3(256), 0, st, p);
  { int lrc = SGRL_OK; if (!launched("train gemm: backward kernel launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

}  // namespace

extern "C" {

const char* sgrl_train_last_error(void) { return g_train_err.c_str(); }

int64_t sgrl_train_ws_floats(void) { return kWsTiles * TILE_WS + kCounters; }

int sgrl_linear_forward(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowdiv, float* y,
                        int ldy, int M, int N, int K, int relu, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < K || ldy < N)
    return tfail(SGRL_ERR_ARG, "sgrl_linear_forward: bad argument");
  SArgs a{x, ldx, nullptr, 0, w, ldw, bias, relu ? 1 : 0, rowdiv, y, ldy, nullptr, M, N, K, 0, nullptr, nullptr};
  return launch<false, false>(a, nullptr, (hipStream_t)stream);
}

int sgrl_linear_forward_fused(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowdiv,
                              const float* addend, int ldadd, const float* tail, int ntail, float* y, int ldy, int M, int N, int K,
                              int relu, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < K || ldy < N + (tail ? ntail : 0) || (addend && ldadd < N) ||
      (tail && ntail <= 0))
    return tfail(SGRL_ERR_ARG, "sgrl_linear_forward_fused: bad argument");
  SArgs a{x, ldx, nullptr, 0, w, ldw, bias, relu ? 1 : 0, rowdiv, y, ldy, nullptr, M, N, K, 0, nullptr, nullptr, addend, ldadd, tail, ntail};
  return launch<false, false>(a, nullptr, (hipStream_t)stream);
}

int sgrl_linear_backward(const float* dy, int lddy, const float* y, int ldyo, int relu, const float* rowdiv, const float* x,
                         int ldx, const float* w, int ldw, float* dx, int lddx, float* dw, int lddw, float* db,
                         float* drowdiv, int M, int N, int K, float* ws, void* stream) {
  return sgrl_linear_backward_xrelu(dy, lddy, y, ldyo, relu, rowdiv, x, ldx, w, ldw, dx, lddx, dw, lddw, db, drowdiv, M, N, K, 0, ws, stream);
}

int sgrl_linear_backward_xrelu(const float* dy, int lddy, const float* y, int ldyo, int relu, const float* rowdiv, const float* x,
                               int ldx, const float* w, int ldw, float* dx, int lddx, float* dw, int lddw, float* db,
                               float* drowdiv, int M, int N, int K, int x_relu, float* ws, void* stream) {
  return sgrl_linear_backward_acc(dy, lddy, y, ldyo, relu, rowdiv, x, ldx, w, ldw, dx, lddx, dw, lddw, db, drowdiv, M, N, K, x_relu, 0, ws, stream);
}

int sgrl_linear_backward_acc(const float* dy, int lddy, const float* y, int ldyo, int relu, const float* rowdiv, const float* x,
                             int ldx, const float* w, int ldw, float* dx, int lddx, float* dw, int lddw, float* db,
                             float* drowdiv, int M, int N, int K, int x_relu, int acc_dx, float* ws, void* stream) {
  if (x_relu && dx && (!x || ldx < K)) return tfail(SGRL_ERR_ARG, "sgrl_linear_backward_xrelu: x_relu needs the layer's input");
  if (!dy || M <= 0 || N <= 0 || K <= 0 || lddy < N || ((relu || drowdiv) && (!y || ldyo < N)) || (drowdiv && !rowdiv) ||
      (relu && rowdiv))
    return tfail(SGRL_ERR_ARG, "sgrl_linear_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const float* mask = relu ? y : nullptr;
  if (drowdiv) {
    hipLaunchKernelGGL(k_rowdot, dim3((M + 3) / 4), dim3(256), 0, st, dy, lddy, y, ldyo, rowdiv, drowdiv, M, N,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    { int lrc = SGRL_OK; if (!launched("k_rowdot launch failed", &lrc)) return lrc; }
  }
  if (dx && (!w || ldw < K || lddx < K)) return tfail(SGRL_ERR_ARG, "sgrl_linear_backward: bad weight / dx argument");
  if (dw && (!x || ldx < K || lddw < K)) return tfail(SGRL_ERR_ARG, "sgrl_linear_backward: bad input / dw argument");
  // dx[M][K] = g[M][N] . w[N][K]: contraction N, contiguous in g, the row index of w
  SArgs ad{dy, lddy, mask, ldyo, w, ldw, nullptr, 0, rowdiv, dx, lddx, nullptr, M, K, N, 0, nullptr, nullptr};
  if (x_relu) { ad.omask = x; ad.ldomask = ldx; }
  if (acc_dx && dx) { ad.addend = dx; ad.ldadd = lddx; }
  // dw[N][K] = g^T . x: contraction M, the row index of both operands; db rides along
  SArgs aw{dy, lddy, mask, ldyo, x, ldx, nullptr, 0, rowdiv, dw, lddw, db, N, K, M, 0, nullptr, nullptr};
  if (dx && dw) {
    const int rc = launch_bwd(ad, aw, ws, st);
    if (rc != SGRL_OK) return rc;
  } else if (dx) {
    const int rc = launch<false, true>(ad, nullptr, st);
    if (rc != SGRL_OK) return rc;
  } else if (dw) {
    const int rc = launch<true, true>(aw, ws, st);
    if (rc != SGRL_OK) return rc;
  } else if (db) {
    if (rowdiv) return tfail(SGRL_ERR_ARG, "sgrl_linear_backward: db without dw is not offered together with rowdiv");
    hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64), dim3(256), 0, st, dy, lddy, mask, ldyo, db, M, N);
    { int lrc = SGRL_OK; if (!launched("k_colsum launch failed", &lrc)) return lrc; }
  }
  return SGRL_OK;
}

int sgrl_linear_forward_twin(const float* x0, const float* x1, int ldx, const float* w0, const float* w1, int ldw, const float* b0,
                             const float* b1, const float* rd0, const float* rd1, float* y0, float* y1, int ldy, int M, int N, int K,
                             int relu, void* stream) {
  if (!x0 || !x1 || !w0 || !w1 || !y0 || !y1 || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < K || ldy < N || (!b0) != (!b1) ||
      (!rd0) != (!rd1))
    return tfail(SGRL_ERR_ARG, "sgrl_linear_forward_twin: bad argument");
  SArgs2 p;
  p.a[0] = SArgs{x0, ldx, nullptr, 0, w0, ldw, b0, relu ? 1 : 0, rd0, y0, ldy, nullptr, M, N, K, 0, nullptr, nullptr};
  p.a[1] = SArgs{x1, ldx, nullptr, 0, w1, ldw, b1, relu ? 1 : 0, rd1, y1, ldy, nullptr, M, N, K, 0, nullptr, nullptr};
  int tn, tm, splits;
  for (int i = 0; i < 2; i++) { const int rc = plan<false>(p.a[i], nullptr, &tn, &tm, &splits); if (rc != SGRL_OK) return rc; }
  if (row_major_staging()) hipLaunchKernelGGL((k_sgemm_twin<false, true>), dim3(tn, tm, 2), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((k_sgemm_twin<false, false>), dim3(tn, tm, 2), dim3(256), 0, (hipStream_t)stream, p);
  { int lrc = SGRL_OK; if (!launched("k_sgemm_twin launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_linear_forward_twin_fused(const float* x0, const float* x1, int ldx, const float* w0, const float* w1, int ldw, const float* b0,
                                   const float* b1, const float* rd0, const float* rd1, const float* add0, const float* add1, int ldadd,
                                   const float* tail0, const float* tail1, int ntail, float* y0, float* y1, int ldy, int M, int N, int K,
                                   int relu, void* stream) {
  if (!x0 || !x1 || !w0 || !w1 || !y0 || !y1 || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < K || (!b0) != (!b1) || (!rd0) != (!rd1) ||
      (!add0) != (!add1) || (!tail0) != (!tail1) || ldy < N + (tail0 ? ntail : 0) || (add0 && ldadd < N) ||
      (tail0 && ntail <= 0))
    return tfail(SGRL_ERR_ARG, "sgrl_linear_forward_twin_fused: bad argument");
  SArgs2 p;
  p.a[0] = SArgs{x0, ldx, nullptr, 0, w0, ldw, b0, relu ? 1 : 0, rd0, y0, ldy, nullptr, M, N, K, 0, nullptr, nullptr, add0, ldadd, tail0, ntail};
  p.a[1] = SArgs{x1, ldx, nullptr, 0, w1, ldw, b1, relu ? 1 : 0, rd1, y1, ldy, nullptr, M, N, K, 0, nullptr, nullptr, add1, ldadd, tail1, ntail};
  int tn, tm, splits;
  for (int i = 0; i < 2; i++) { const int rc = plan<false>(p.a[i], nullptr, &tn, &tm, &splits); if (rc != SGRL_OK) return rc; }
  if (row_major_staging()) hipLaunchKernelGGL((k_sgemm_twin<false, true>), dim3(tn, tm, 2), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((k_sgemm_twin<false, false>), dim3(tn, tm, 2), dim3(256), 0, (hipStream_t)stream, p);
  { int lrc = SGRL_OK; if (!launched("k_sgemm_twin launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_linear_dgrad_twin(const float* dy0, const float* dy1, int lddy, const float* y0, const float* y1, int ldyo, int relu,
                           const float* rd0, const float* rd1, const float* w0, const float* w1, int ldw, float* dx0, float* dx1,
                           int lddx, float* drd0, float* drd1, int M, int N, int K, void* stream) {
  return sgrl_linear_dgrad_twin_xrelu(dy0, dy1, lddy, y0, y1, ldyo, relu, rd0, rd1, w0, w1, ldw, dx0, dx1, lddx, drd0, drd1, nullptr, nullptr,
                                      0, M, N, K, stream);
}

int sgrl_linear_dgrad_twin_xrelu(const float* dy0, const float* dy1, int lddy, const float* y0, const float* y1, int ldyo, int relu,
                                 const float* rd0, const float* rd1, const float* w0, const float* w1, int ldw, float* dx0, float* dx1,
                                 int lddx, float* drd0, float* drd1, const float* x0, const float* x1, int ldx, int M, int N, int K,
                                 void* stream) {
  return sgrl_linear_dgrad_twin_acc(dy0, dy1, lddy, y0, y1, ldyo, relu, rd0, rd1, w0, w1, ldw, dx0, dx1, lddx, drd0, drd1, x0, x1, ldx, M, N, K,
                                    0, stream);
}

int sgrl_linear_dgrad_twin_acc(const float* dy0, const float* dy1, int lddy, const float* y0, const float* y1, int ldyo, int relu,
                               const float* rd0, const float* rd1, const float* w0, const float* w1, int ldw, float* dx0, float* dx1,
                               int lddx, float* drd0, float* drd1, const float* x0, const float* x1, int ldx, int M, int N, int K,
                               int acc_dx, void* stream) {
  if ((!x0) != (!x1) || (x0 && ldx < K)) return tfail(SGRL_ERR_ARG, "sgrl_linear_dgrad_twin_xrelu: bad input-mask argument");
  if (!dy0 || !dy1 || !w0 || !w1 || !dx0 || !dx1 || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldw < K || lddx < K ||
      ((relu || drd0) && (!y0 || !y1 || ldyo < N)) || (!rd0) != (!rd1) || (!drd0) != (!drd1) || (drd0 && !rd0) || (relu && rd0))
    return tfail(SGRL_ERR_ARG, "sgrl_linear_dgrad_twin: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (drd0) {
    hipLaunchKernelGGL(k_rowdot, dim3((M + 3) / 4, 2), dim3(256), 0, st, dy0, lddy, y0, ldyo, rd0, drd0, M, N, dy1, y1, rd1, drd1);
    { int lrc = SGRL_OK; if (!launched("k_rowdot launch failed", &lrc)) return lrc; }
  }
  SArgs2 p;
  p.a[0] = SArgs{dy0, lddy, relu ? y0 : nullptr, ldyo, w0, ldw, nullptr, 0, rd0, dx0, lddx, nullptr, M, K, N, 0, nullptr, nullptr};
  p.a[1] = SArgs{dy1, lddy, relu ? y1 : nullptr, ldyo, w1, ldw, nullptr, 0, rd1, dx1, lddx, nullptr, M, K, N, 0, nullptr, nullptr};
  if (x0) { p.a[0].omask = x0; p.a[1].omask = x1; p.a[0].ldomask = p.a[1].ldomask = ldx; }
  if (acc_dx) { p.a[0].addend = dx0; p.a[1].addend = dx1; p.a[0].ldadd = p.a[1].ldadd = lddx; }
  int tn, tm, splits;
  for (int i = 0; i < 2; i++) { const int rc = plan<false>(p.a[i], nullptr, &tn, &tm, &splits); if (rc != SGRL_OK) return rc; }
  if (row_major_staging()) hipLaunchKernelGGL((k_sgemm_twin<true, true>), dim3(tn, tm, 2), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_sgemm_twin<true, false>), dim3(tn, tm, 2), dim3(256), 0, st, p);
  { int lrc = SGRL_OK; if (!launched("k_sgemm_twin (input gradient) launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_gram_forward(const float* z, float* gram, float* fn, int M, void* stream) {
  if (!z || !gram || !fn || M <= 0) return tfail(SGRL_ERR_ARG, "sgrl_gram_forward: bad argument");
  hipLaunchKernelGGL(k_gram_fwd, dim3(M), dim3(256), 0, (hipStream_t)stream, z, gram, fn, M);
  { int lrc = SGRL_OK; if (!launched("k_gram_fwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_gram_backward(const float* z, const float* dgram, const float* dfn, const float* fn, float* dz, int M, void* stream) {
  if (!z || !fn || !dz || M <= 0 || (!dgram && !dfn)) return tfail(SGRL_ERR_ARG, "sgrl_gram_backward: bad argument");
  hipLaunchKernelGGL(k_gram_bwd, dim3(M), dim3(256), 0, (hipStream_t)stream, z, dgram, dfn, fn, dz, M);
  { int lrc = SGRL_OK; if (!launched("k_gram_bwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_gram_tri_forward(const float* z, float* tri, float* fn, int M, void* stream) {
  if (!z || !tri || !fn || M <= 0) return tfail(SGRL_ERR_ARG, "sgrl_gram_tri_forward: bad argument");
  hipLaunchKernelGGL(k_gram_tri_fwd, dim3(M), dim3(256), 0, (hipStream_t)stream, z, tri, fn, M);
  { int lrc = SGRL_OK; if (!launched("k_gram_tri_fwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_gram_tri_backward(const float* z, const float* dtri, const float* dfn, const float* fn, float* dz, int M, void* stream) {
  if (!z || !fn || !dz || M <= 0 || (!dtri && !dfn)) return tfail(SGRL_ERR_ARG, "sgrl_gram_tri_backward: bad argument");
  hipLaunchKernelGGL(k_gram_tri_bwd, dim3(M), dim3(256), 0, (hipStream_t)stream, z, dtri, dfn, fn, dz, M);
  { int lrc = SGRL_OK; if (!launched("k_gram_tri_bwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_sym_fold(int n, const float* const* w, float* const* wtri, const int* rows, int unfold, void* stream) {
  if (n <= 0 || n > kFoldMax || !w || !wtri || !rows) return tfail(SGRL_ERR_ARG, "sgrl_sym_fold: bad argument (at most 16 matrices per call)");
  FoldArgs p;
  int most = 0;
  for (int i = 0; i < n; i++) {
    if (!w[i] || !wtri[i] || rows[i] <= 0) return tfail(SGRL_ERR_ARG, "sgrl_sym_fold: bad matrix " + std::to_string(i));
    p.src[i] = unfold ? wtri[i] : w[i];
    p.dst[i] = unfold ? const_cast<float*>(w[i]) : wtri[i];
    p.rows[i] = rows[i];
    most = std::max(most, rows[i]);
  }
  const int per = most * (unfold ? 1024 : TRI);
  if (unfold) hipLaunchKernelGGL(k_unfold_sym, dim3((per + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(k_fold_sym, dim3((per + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, p);
  { int lrc = SGRL_OK; if (!launched("k_fold_sym launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_attention_forward(const float* qkv, const float* vgp, const float* gdir, const float* bias, float scale, float* w,
                           float* o, float* og, int B, int L, void* stream) {
  if (!qkv || !vgp || !gdir || !w || !o || !og || B <= 0 || L < 1 || L > AL) return tfail(SGRL_ERR_ARG, "sgrl_attention_forward: bad argument");
  hipLaunchKernelGGL(k_attn_fwd, dim3(2 * B), dim3(128), 0, (hipStream_t)stream, qkv, vgp, gdir, bias, scale, w, o, og, L);
  { int lrc = SGRL_OK; if (!launched("k_attn_fwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_attention_backward(const float* qkv, const float* vgp, const float* gdir, float scale, const float* w, const float* d_o,
                            const float* d_og, float* dqkv, float* dvgp, float* dgdh, float* ds, int B, int L, void* stream) {
  if (!qkv || !vgp || !gdir || !w || !d_o || !d_og || !dqkv || !dvgp || !dgdh || !ds || B <= 0 || L < 1 || L > AL)
    return tfail(SGRL_ERR_ARG, "sgrl_attention_backward: bad argument");
  static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 attn_bwd_lds_bytes(AL)) == hipSuccess;
  if (!lds_ok) return tfail(SGRL_ERR_HIP, "sgrl_attention_backward: the kernel's dynamic LDS size was refused");
  hipLaunchKernelGGL(k_attn_bwd, dim3(2 * B), dim3(128), attn_bwd_lds_bytes(L), (hipStream_t)stream, qkv, vgp, gdir, scale, w, d_o, d_og, dqkv, dvgp,
                     dgdh, ds, L);
  { int lrc = SGRL_OK; if (!launched("k_attn_bwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_linear_wgrad_group(int n, const sgrl_wgrad_desc* d, float* ws, void* stream) {
  if (n <= 0 || !d) return tfail(SGRL_ERR_ARG, "sgrl_linear_wgrad_group: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += kGroup) {
    GroupArgs p;
    p.n = std::min(kGroup, n - i0);
    int64_t slot = 0;
    int counter = 0, blocks = 0;
    // Contraction splits exist to give a LONE small product enough workgroups.  A launch whose 32 x 32-tile products already put a
    // workgroup on every CU without splitting runs them unsplit (a rule from the time a split cost two device-scope fences per
    // workgroup: twelve 64 x 128 gradients over 2 100 rows took 73 us in one launch, 14.7 us each alone; kept with the fence-free
    // exchange, where the 64 x 64 tile below takes the regular products and cuts their contractions anyway).
    int unsplit_tiles = 0;
    for (int g = 0; g < p.n; g++) unsplit_tiles += ((d[i0 + g].N + BT - 1) / BT) * ((d[i0 + g].K + BT - 1) / BT);
    float* const ws_eff = unsplit_tiles >= kGroupNoSplitTiles ? nullptr : ws;
    // ... but such a launch lasts as long as its LONGEST contraction: at the update batch of 256 the three-vector channels contract
    // over 5 376 rows (42 k-tiles) beside 1 792-row products (14).  A product at least twice as long as the group's shortest is cut
    // into that many splits, so that every workgroup of the launch walks about the same number of k-tiles (these products have few
    // output tiles).
    int kt_min = 1 << 30;
    for (int g = 0; g < p.n; g++) kt_min = std::min(kt_min, (d[i0 + g].M + BKW - 1) / BKW);
    constexpr bool big_on = true;       // (round 5's probe switch SGRL_TRAIN_WGRAD64 is gone: the 64 x 64 tile is the product path)
    for (int g = 0; g < p.n; g++) {
      const sgrl_wgrad_desc& q = d[i0 + g];
      if (!q.dy || !q.x || !q.dw || q.M <= 0 || q.N <= 0 || q.K <= 0 || q.lddy < q.N || q.ldx < q.K || q.lddw < q.K ||
          (q.relu && (!q.y || q.ldy < q.N)) || (q.relu && q.rowdiv))
        return tfail(SGRL_ERR_ARG, "sgrl_linear_wgrad_group: bad descriptor " + std::to_string(i0 + g));
      SArgs a{q.dy, q.lddy, q.relu ? q.y : nullptr, q.ldy, q.x, q.ldx, nullptr, 0, q.rowdiv, q.dw, q.lddw, q.db, q.N, q.K, q.M,
              0, nullptr, nullptr};
      const int kt = (q.M + BKW - 1) / BKW;
      p.big[g] = 0;
      // regular products on 64 x 64 tiles (wgrad_tile64), their contraction cut into pieces of about seven 64-row k-tiles
      if (big_on && ws && !q.relu && q.N % T64 == 0 && q.K % T64 == 0 && (q.lddy & 3) == 0 && (q.ldx & 3) == 0 &&
          (reinterpret_cast<uintptr_t>(q.dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(q.x) & 15) == 0) {
        const int tiles = (q.N / T64) * (q.K / T64), kt64 = (q.M + T64 - 1) / T64;
        // every workgroup of the launch walks about the same short chain of k-tiles whatever its product's size: in twelve EQUAL
        // products fewer, longer pieces win for the large outputs (1 024 x 256 unsplit: 127 us against 177), but in the update's mixed
        // groups the longest chain is the launch (7.31 ms per update with pieces of seven k-tiles, 7.69 with ~64 workgroups per product)
        constexpr int per_split = 7;
        const int splits = std::max(1, std::min(kt64 / per_split, 16));
        if (slot + (int64_t)tiles * splits * kSlots64 <= kWsTiles && counter + tiles <= kCounters) {
          p.big[g] = 1;
          p.gx[g] = q.K / T64; p.gy[g] = q.N / T64;
          a.kper = ((kt64 + splits - 1) / splits) * T64;
          p.nz[g] = (q.M + a.kper - 1) / a.kper;
          a.ws = ws + slot * TILE_WS;
          a.counters = reinterpret_cast<unsigned*>(ws + kWsTiles * TILE_WS) + counter;
          slot += (int64_t)tiles * p.nz[g] * kSlots64; counter += tiles;
        }
      }
      if (!p.big[g]) {
        // since a split no longer costs fences, every 32 x 32-tile product of a group is cut into pieces of about seven k-tiles as
        // well: 7.28 -> 7.13 ms per update, same results to rounding (tests/test_wgrad_stress_gpu.py).  Round 5 parked this behind
        // SGRL_W32_KT because config 5's take-off had turned out to hang on rounding-level differences; round 6's take-off table
        // (profiles/r6_takeoff) showed that it does so on EVERY arithmetic, the vendor libraries' included: the default since.
        constexpr int kt32 = 7;
        int even = (!ws_eff && ws && kt_min >= 4 && kt >= 2 * kt_min) ? kt / kt_min : 0;
        if (kt32 > 0 && ws && kt >= 2 * kt32) even = kt / kt32;
        const int rc = plan<true>(a, even ? ws : ws_eff, &p.gx[g], &p.gy[g], &p.nz[g], slot, counter, even);
        if (rc != SGRL_OK) return rc;
        if (p.nz[g] > 1) { slot += (int64_t)p.gx[g] * p.gy[g] * p.nz[g]; counter += p.gx[g] * p.gy[g]; }
      }
      p.w[g] = a;
      p.first[g] = blocks;
      blocks += p.gx[g] * p.gy[g] * p.nz[g];
    }
    p.first[p.n] = blocks;
    hipLaunchKernelGGL(k_sgemm_wgroup, dim3(blocks), dim3(256), 0, st, p);
    { int lrc = SGRL_OK; if (!launched("k_sgemm_wgroup launch failed", &lrc)) return lrc; }
  }
  return SGRL_OK;
}

int sgrl_zmat_forward(const float* z, const float* mat, float* t, int M, void* stream) {
  if (!z || !mat || !t || M <= 0) return tfail(SGRL_ERR_ARG, "sgrl_zmat_forward: bad argument");
  hipLaunchKernelGGL(k_zmat_fwd, dim3((M + 1) / 2), dim3(256), 0, (hipStream_t)stream, z, mat, t, M);
  { int lrc = SGRL_OK; if (!launched("k_zmat_fwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_zmat_backward(const float* z, const float* mat, const float* dt, float* dz, float* dmat, int M, void* stream) {
  if (!z || !mat || !dt || !dz || !dmat || M <= 0) return tfail(SGRL_ERR_ARG, "sgrl_zmat_backward: bad argument");
  hipLaunchKernelGGL(k_zmat_bwd, dim3((M + 1) / 2), dim3(256), 0, (hipStream_t)stream, z, mat, dt, dz, dmat, M);
  { int lrc = SGRL_OK; if (!launched("k_zmat_bwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_add_ln_forward(const float* x, const float* res, const float* w0, const float* b0, const float* w1, const float* b1, float* y,
                        float* xhat, float* rstd, int rows, int nets, float eps, void* stream) {
  if (!x || !w0 || !b0 || !y || rows <= 0 || (nets != 1 && nets != 2) || (nets == 2 && (!w1 || !b1)) || ((xhat == nullptr) != (rstd == nullptr)))
    return tfail(SGRL_ERR_ARG, "sgrl_add_ln_forward: bad argument");
  const int total = rows * nets;
  hipLaunchKernelGGL(k_add_ln_fwd, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, res, w0, b0, w1, b1, y, xhat, rstd, rows, total, eps);
  { int lrc = SGRL_OK; if (!launched("k_add_ln_fwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_add_ln_backward(const float* dy, const float* xhat, const float* rstd, const float* w0, const float* w1, float* dx, float* dw0,
                         float* db0, float* dw1, float* db1, int rows, int nets, void* stream) {
  if (!dy || !xhat || !rstd || !w0 || !dx || rows <= 0 || (nets != 1 && nets != 2) || (nets == 2 && !w1))
    return tfail(SGRL_ERR_ARG, "sgrl_add_ln_backward: bad argument");
  const int total = rows * nets, nrow_blocks = (total + 3) / 4;
  const bool params = dw0 || db0 || dw1 || db1;
  hipLaunchKernelGGL(k_add_ln_bwd, dim3(nrow_blocks + (params ? (128 / kLnCols) * nets : 0)), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, w0, w1, dx,
                     dw0, db0, dw1, db1, rows, total, nrow_blocks);
  { int lrc = SGRL_OK; if (!launched("k_add_ln_bwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_embed3_forward(const long long* idx, const float* w0, const float* w1, const float* w2, int n0, int n1, int n2, float* out, int L,
                        void* stream) {
  if (!idx || !w0 || !w1 || !w2 || !out || L <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0 || n0 + n1 + n2 > 128)
    return tfail(SGRL_ERR_ARG, "sgrl_embed3_forward: bad argument");
  Embed3 e{idx, {w0, w1, w2}, {nullptr, nullptr, nullptr}, {n0, n1, n2}, L, 0, n0 + n1 + n2};
  hipLaunchKernelGGL(k_embed3_fwd, dim3(L), dim3(128), 0, (hipStream_t)stream, e, out);
  { int lrc = SGRL_OK; if (!launched("k_embed3_fwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_embed3_backward(const long long* idx, const float* dout, float* dw0, float* dw1, float* dw2, int n0, int n1, int n2, int L, int rows,
                         void* stream) {
  if (!idx || !dout || L <= 0 || rows <= 0 || n0 <= 0 || n1 <= 0 || n2 <= 0 || n0 + n1 + n2 > 128)
    return tfail(SGRL_ERR_ARG, "sgrl_embed3_backward: bad argument");
  Embed3 e{idx, {nullptr, nullptr, nullptr}, {dw0, dw1, dw2}, {n0, n1, n2}, L, rows, n0 + n1 + n2};
  hipLaunchKernelGGL(k_embed3_bwd, dim3(rows), dim3(128), 0, (hipStream_t)stream, e, dout);
  { int lrc = SGRL_OK; if (!launched("k_embed3_bwd launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_optim_chunk(void) { return kOptChunk; }

int sgrl_optim_clip_adam(const void* table, int n_tensors, const void* chunks, int n_chunks, double lr, double beta1, double beta2, double eps,
                         float max_norm, float* scratch, void* stream) {
  if (!table || !chunks || n_tensors <= 0 || n_chunks <= 0 || (max_norm > 0.f && !scratch))
    return tfail(SGRL_ERR_ARG, "sgrl_optim_clip_adam: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const OptRow* tab = reinterpret_cast<const OptRow*>(table);
  const int2* ch = reinterpret_cast<const int2*>(chunks);
  const bool clip = max_norm > 0.f;
  if (clip) hipLaunchKernelGGL(k_opt_sqnorm, dim3(n_chunks), dim3(256), 0, st, tab, ch, scratch + 1);
  hipLaunchKernelGGL(k_opt_total, dim3(1), dim3(256), 0, st, clip ? scratch + 1 : (const float*)nullptr, n_chunks, tab, n_tensors, scratch);
  hipLaunchKernelGGL(k_opt_adam, dim3(n_chunks), dim3(256), 0, st, tab, ch, lr, beta1, beta2, eps, clip ? scratch : (const float*)nullptr, max_norm);
  { int lrc = SGRL_OK; if (!launched("optimizer kernels: launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

int sgrl_optim_lerp(const void* table, const void* chunks, int n_chunks, float tau, void* stream) {
  if (!table || !chunks || n_chunks <= 0) return tfail(SGRL_ERR_ARG, "sgrl_optim_lerp: bad argument");
  hipLaunchKernelGGL(k_opt_lerp, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const OptRow*>(table),
                     reinterpret_cast<const int2*>(chunks), tau);
  { int lrc = SGRL_OK; if (!launched("k_opt_lerp launch failed", &lrc)) return lrc; }
  return SGRL_OK;
}

}  // extern "C"
