// chain_f16.h -- back-to-back products of the SET actor in ONE kernel (two-piece f16 x 3 form of gemm_f32.h):
//
//     out[M, 128] = epi2( relu(A[M, K1] . W1[HID, K1]^T + b1) . W2[128, HID]^T + b2 ),      HID = 256 or 128
//
// (reference SEActor.py:93-123 linear1 -> ReLU -> linear2 and linear_g1 -> ReLU -> linear_g2, SEActor.py:256-276 the head's
// pairs).  A workgroup owns 64 rows from the first operand load to the last store:
//   prologue (PROJ)  Z = X[3 rows per node, Kp] . Wp^T -- the site's stacked 30-column projections -- written to zc / z2
//   phase 1   the WHOLE HID-wide intermediate of its rows in accumulators (8 waves = 2 row groups x 4 hidden groups, tiles kept
//             TRANSPOSED: lane = row, registers = hidden index), operands staged per 16-wide k-tile exactly as k_gemm3 does;
//             the first operand is either loaded (SRC 0) or GENERATED from Z as the blocked Gram triangle (SRC 1, K1 = 576)
//   hand-off  bias + ReLU + the two-piece split happen in REGISTERS; the f16 planes of the intermediate go to LDS in 64-column
//             slices (two slice buffers in the bytes phase 1 used for its stages) -- the intermediate never reaches memory and
//             is split once
//   phase 2   64 x 128 outputs (32 x 32 per wave, transposed again), k running over the slices, only W2 is staged
//   epilogue  plain store | row division | residual + LayerNorm in place (lane = row: row sums are register sums)
// 60 KB of LDS and 128 registers: two workgroups per CU, as the single products.
// Range (gemm_f32.h, pow2_scale): every operand row is scaled by a power of two before it is split -- weights by k_encode_rows,
// the Gram rows by ||Z'Z||_F, loaded rows (A, X) by a first-tile estimate with a workgroup-uniform repeat on the exact maxima
// if the estimate fell short, the intermediate by its exact row maximum (lane = row: a register maximum, one LDS exchange
// between the four hidden groups) -- and every epilogue multiplies the inverse powers back.
#pragma once
#include "gemm_f32.h"

namespace sgrl_gemm {

// -DSGRL_CHAIN_PROF (tools/chain_lab.hip only): s_memtime stamps at the phase boundaries of k_chain, wave 0 of every workgroup adds its
// deltas to g_chain_prof[8] (0 prologue | 1 phase-1 set-up (Gram norm, first stage) | 2 phase-1 k-loop | 3 hand-off | 4 phase 2 |
// 5 epilogue | 6 whole kernel | 7 workgroups)
#ifdef SGRL_CHAIN_PROF
__device__ unsigned long long g_chain_prof[8];
#define SGRL_CSTAMP(id) do { const long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_chain_prof[id], (unsigned long long)(now_ - cp_last)); cp_last = __builtin_readcyclecounter(); } while (0)
#else
#define SGRL_CSTAMP(id)
#endif

struct ChainArgs {
  const float* A; int lda;            // SRC 0: [M, K1] float32; SRC 1: Z [M, 3, 32] (= zc when PROJ writes it)
  const unsigned* W1; int ldw1;       // pre-split words (enc_word) [HID][ldw1]
  const float* b1;
  const unsigned* W2; int ldw2;       // pre-split words [128][ldw2]
  const float* b2;
  float* C; int ldc;                  // plain epilogue: C[m][0:128]
  int M, K1;
  const float* rowdiv;                // EPI_ROWDIV: value / rowdiv[m]
  float* fn_out;                      // SRC 1: receives ||Z'Z||_F + 1 per row
  float* ln_io; int ln_ld;            // EPI_LN: residual stream, rewritten in place
  const float* ln_w; const float* ln_b;
  // PROJ: X [3 M, Kp] (row stride ldx), Wp words [64][Kp] (rows 0..29 -> zc columns 0..29, rows 32..61 -> z2 columns 0..29)
  const float* X; int ldx; int Kp;
  const unsigned* Wp;
  float* zc; float* z2;
  // inverse row scales of the three weight operands (k_encode_rows): ws1 [HID], ws2 [128], wsp [64]
  const float* ws1; const float* ws2; const float* wsp;
  float* ln_out;                      // EPI_LN: where the normalised rows go (row stride ln_ld); null = in place
  // EPI_EQUIV (second product N = 1024 in the c * 32 + q order, W2 [1024][HID], b2 [1024], ws2 one scale per 32-row block): the
  // [M, 1024] result -- a 32 x 32 matrix per row -- is contracted with zq [M, 3, 32] on the fly, only tout [M, 3, 32] is stored
  const float* zq; float* tout;
  // ... and, with g != null, the update of the vector stream that consumes tout (reference SEActor.py:89, 108-114) rides on the same
  // workgroup, which holds all 32 columns of its rows' tout:  g[r][:] += g1[r][:] + tout[r][0:32] . W5[0:128][0:32]^T  for its 3 x 64
  // rows r = 3 m + s (W5 as row-scaled words, ws5 its inverse scales); outg != null: the new g also into outg[r][8:136] (row
  // stride outg_ld, zero K-padding columns 136 .. outg_ld - 1)
  float* g; const float* g1; const unsigned* W5; const float* ws5; float* outg; int outg_ld;
  const float* xsc;                   // lab form only (DBG & 8): X arrives as row-scaled words, xsc [3 M] undoes the rows' scales
};

constexpr int kChainRows = 64;
constexpr int kChainLds = 61440;
// EPI_EQUIV: two 32-column sub-slice buffers | two W2 stages | the block's z rows (pitch 100 floats) | the bias b2 [1024]
constexpr int kChainEqSub = 2 * kChainRows * 80, kChainEqW2 = 2 * kChainEqSub, kChainEqZ = kChainEqW2 + 2 * 2 * 128 * kTileRB16,
              kChainEqB = kChainEqZ + kChainRows * 100 * 4, kChainEqLds = kChainEqB + 1024 * 4;

// SRC: 0 loaded operand, 1 Gram operand.  PROJ: 0 none, 1 one projection (zc), 2 two (zc and z2).
// DBG (tools/chain_lab.hip only; the results are wrong): 1 = no sub-slice hand-over inside the passes, 2 = no contraction epilogue,
// 4 = no W2 staging inside the passes -- what each part of the equivariant phase costs; 8 (PROJ kernels; results RIGHT, but the
// forward has no producer that writes such an operand): the prologue's X arrives as row-scaled words + ChainArgs::xsc -- no estimate,
// no vote, no split in its k-loop (what the site kernel gains when its producer emits the split form)
template <int SRC, int HID, int EPI2, int PROJ, int DBG = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_chain(ChainArgs a) {
  static_assert(HID == 256 || HID == 128, "hidden width 256 or 128");
  static_assert(PROJ == 0 || SRC == 1, "the projection prologue feeds the Gram operand");
  constexpr int R = kChainRows, RB = kTileRB16;  // a k-tile row: 16 f16, halves swizzled (gemm_f32.h lds_wr / lds_rd: stores and reads conflict-free)
  constexpr int TN = HID / 128;                 // 32-wide hidden tiles per wave in phase 1
  constexpr int kPlaneA = R * RB, kPlaneW = HID * RB, kStage1 = 2 * (kPlaneA + kPlaneW);
  constexpr int SLP = 144;                      // slice row: 64 f16 + 16 B pad
  constexpr int kSlicePlane = R * SLP, kSliceBuf = 2 * kSlicePlane;
  constexpr int kW2Plane = 128 * RB, kW2Stage = 2 * kW2Plane, kW2Base = 2 * kSliceBuf;
  static_assert(2 * kStage1 <= kChainLds && kW2Base + 2 * kW2Stage <= kChainLds, "LDS image");
  static_assert(!(EPI2 & EPI_EQUIV) || (HID == 256 && !(EPI2 & EPI_LN) && kChainEqLds >= kChainLds && kChainEqLds <= 80 * 1024), "equivariant second product");
  constexpr float kCorW = 1.f / kF16LowScale;
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];     // (aligned: static LDS precedes it)
  char* lds = reinterpret_cast<char*>(gemm_lds);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;      // row group | hidden group (phase 1) / column group (phase 2)
  const int m0 = blockIdx.x * R;
  const int kq = t & 3, r4 = t >> 2;            // staging: float4 kq of tile row r4 (+ 128 i)
  __shared__ float rs_sh[R];                    // 1 / scale of the workgroup's A rows (phase 1)
  __shared__ float as_sh[SRC == 1 ? R : 1];     // Gram operand: the scale itself
  __shared__ float px_sh[PROJ ? 192 : 1];       // 1 / scale of its X rows (projection prologue)
  __shared__ float hm_sh[R * 4];                // row maxima of the intermediate, per hidden group
  __shared__ unsigned redo_sh;
  if (t == 0) redo_sh = 0;
#ifdef SGRL_CHAIN_PROF
  long long cp_last = __builtin_readcyclecounter();
  const long long cp_begin = cp_last;
#endif
  auto quad_max = [&](float m) -> float { m = fmaxf(m, __shfl_xor(m, 1, 64)); return fmaxf(m, __shfl_xor(m, 2, 64)); };
  // split a float4 of a row scaled by sc (a power of two) into the two f16 planes; mx keeps the largest SCALED magnitude: one
  // beyond the f16 range (possible under an estimated scale only) makes the workgroup repeat the phase with the exact maxima
  auto put = [&](char* plane0, int plane_stride, int row, const float4& vin, float sc, float* mx) {
    const float4 v = make_float4(vin.x * sc, vin.y * sc, vin.z * sc, vin.w * sc);
    if (mx) { *mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), *mx); *mx = fmaxf(fmaxf(fabsf(v.z), fabsf(v.w)), *mx); }
    unsigned h0, l0, h1, l1;
    split2h(v.x, v.y, h0, l0);
    split2h(v.z, v.w, h1, l1);
    char* p = plane0 + lds_wr<RB>(row, kq);
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(l0, l1);
  };
  auto put_words = [&](char* plane0, int plane_stride, int row, const float4& v) {
    const unsigned w0 = __float_as_uint(v.x), w1 = __float_as_uint(v.y), w2 = __float_as_uint(v.z), w3 = __float_as_uint(v.w);
    char* p = plane0 + lds_wr<RB>(row, kq);
    *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u));
    *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u));
  };

  // ---- prologue: the site's projections ----------------------------------------------------------------------------------
  if (PROJ) {
    constexpr int kPXA = 192 * RB, kPXW = 64 * RB, kPStage = 2 * (kPXA + kPXW);
    static_assert(2 * kPStage <= kChainLds, "LDS image of the projection prologue");
    const int rows3 = 3 * a.M, x0 = 3 * m0;
    const float* xr0 = a.X + (size_t)min(x0 + r4, rows3 - 1) * a.ldx + 4 * kq;
    const float* xr1 = a.X + (size_t)min(x0 + 128 + (r4 & 63), rows3 - 1) * a.ldx + 4 * kq;     // threads 0..255
    const unsigned* wr = a.Wp + (size_t)(r4 & 63) * a.Kp + 4 * kq;                               // threads 256..511
    const bool lowh = t < 256;
    float4 x0v, x1v;
    auto pload = [&](int k0) {
      x0v = *reinterpret_cast<const float4*>(xr0 + k0);
      x1v = lowh ? *reinterpret_cast<const float4*>(xr1 + k0) : *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wr) + k0);
    };
    float psc[2] = {1.f, 1.f}, pmx[2] = {0.f, 0.f};     // scale / true maximum of X rows r4 and (threads 0..255) 128 + r4
    constexpr bool XW = (DBG & 8) != 0;
    auto pstore = [&](int st) {
      char* base = lds + st * kPStage;
      if (XW) put_words(base, kPXA, r4, x0v);
      else put(base, kPXA, r4, x0v, psc[0], &pmx[0]);
      if (lowh) { if (XW) put_words(base, kPXA, 128 + r4, x1v); else put(base, kPXA, 128 + r4, x1v, psc[1], &pmx[1]); }
      else put_words(base + 2 * kPXA, kPXW, r4 & 63, x1v);
    };
    f32x16 pacc[PROJ ? PROJ : 1], pcor[PROJ ? PROJ : 1];
    const int nkp = a.Kp / 16;
    const int paoff = lds_rd<RB>(32 * wave + li, lh);
    const int pboff = 2 * kPXA + lds_rd<RB>(li, lh);
    for (int attempt = 0;; attempt++) {
#pragma unroll
      for (int j = 0; j < PROJ; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) { pacc[j][e] = 0.f; pcor[j][e] = 0.f; }
      // samples of each row's middle (staging thread 0 of the row) and last k-tile (thread 1), four values each, in flight with tile 0
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
      if (attempt == 0 && kq < 2 && !XW) {
        const int ks = (kq == 0 ? (nkp >> 1) : nkp - 1) * 16 - 4 * kq;
        s0 = *reinterpret_cast<const float4*>(xr0 + ks);
        if (lowh) s1 = *reinterpret_cast<const float4*>(xr1 + ks);
      }
      pload(0);
      if (attempt == 0 && !XW) {              // estimates of the rows' magnitudes (gemm_f32.h, pow2_scale)
        auto m4 = [](const float4& v, float m) { m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), m); return fmaxf(fmaxf(fabsf(v.z), fabsf(v.w)), m); };
        float e0 = m4(s0, m4(x0v, 0.f)), e1 = lowh ? m4(s1, m4(x1v, 0.f)) : 0.f;
        psc[0] = pow2_scale(quad_max(e0), kScaleEstimate);
        psc[1] = pow2_scale(quad_max(e1), kScaleEstimate);
      }
      if (XW) { if (kq == 0) { px_sh[r4] = a.xsc[min(x0 + r4, rows3 - 1)]; if (lowh) px_sh[128 + r4] = a.xsc[min(x0 + 128 + (r4 & 63), rows3 - 1)]; } }
      else if (kq == 0) { px_sh[r4] = pow2_inv(psc[0]); if (lowh) px_sh[128 + r4] = pow2_inv(psc[1]); }
      pstore(0);
      __syncthreads();
      if (nkp > 1) pload(16);
      for (int kt = 0; kt < nkp; kt++) {
        const int st = kt & 1;
        if (kt + 1 < nkp) pstore(st ^ 1);
        if (kt + 2 < nkp) pload((kt + 2) * 16);
        if (wave < 6) {
          const char* base = lds + st * kPStage;
          const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + paoff));
          const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + kPXA + paoff));
#pragma unroll
          for (int j = 0; j < PROJ; j++) {
            const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + pboff + 32 * j * RB));
            const f16x8 bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + kPXW + pboff + 32 * j * RB));
            pacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, pacc[j], 0, 0, 0);
            pcor[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, pcor[j], 0, 0, 0);
            pcor[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, pcor[j], 0, 0, 0);
          }
        }
        if (kt == nkp - 1 && (!(pmx[0] <= kF16Lim) || !(pmx[1] <= kF16Lim))) redo_sh = 1u;   // the vote rides on the last barrier
        __syncthreads();
      }
      if (attempt == 1) break;
      if (redo_sh == 0u) break;               // (block-uniform) the usual exit
      if (t == 0) atomicAdd(&g_scale_redos, 1u);
      {                                       // the rare path: exact maxima of this thread's X rows (read again), then once more
        float t0 = 0.f, t1 = 0.f;
        for (int kt = 0; kt < nkp; kt++) {
          const float4 x = *reinterpret_cast<const float4*>(xr0 + 16 * kt);
          t0 = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), t0); t0 = fmaxf(fmaxf(fabsf(x.z), fabsf(x.w)), t0);
          if (lowh) {
            const float4 y = *reinterpret_cast<const float4*>(xr1 + 16 * kt);
            t1 = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), t1); t1 = fmaxf(fmaxf(fabsf(y.z), fabsf(y.w)), t1);
          }
        }
        psc[0] = pow2_scale(quad_max(t0), kScaleExact);
        psc[1] = pow2_scale(quad_max(t1), kScaleExact);
        pmx[0] = 0.f; pmx[1] = 0.f;
      }
      __syncthreads();
      if (t == 0) redo_sh = 0;                // phase 1 votes again
    }
    if (wave < 6 && li < 30) {
#pragma unroll
      for (int j = 0; j < PROJ; j++) {
        float* dst = j == 0 ? a.zc : a.z2;
        const float wsn = a.wsp ? a.wsp[32 * j + li] : 1.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int rl = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * lh, row = x0 + rl;
          if (row < rows3) dst[(size_t)row * 32 + li] = (pacc[j][e] + pcor[j][e] * kCorW) * (px_sh[rl] * wsn);
        }
      }
    }
    __syncthreads();        // the block's Z rows are in memory (same CU: visible to its other waves) and the LDS is free
  }
  SGRL_CSTAMP(0);

  // ---- phase 1 ------------------------------------------------------------------------------------------------------------
  f32x16 acc[TN], cor[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) { acc[j][e] = 0.f; cor[j][e] = 0.f; }
  const bool stage_a = t < 256;                 // waves 0..3 stage the 64 A rows (one float4 each)
  const int arow = min(m0 + (r4 & 63), a.M - 1);
  const float* arow_g = SRC == 1 ? a.A + (size_t)arow * 96 : a.A + (size_t)arow * a.lda + 4 * kq;
  const unsigned* wrow_g[TN];
#pragma unroll
  for (int i = 0; i < TN; i++) wrow_g[i] = a.W1 + (size_t)(r4 + 128 * i) * a.ldw1 + 4 * kq;
  // global loads run PF k-tiles ahead of the LDS stage they are written to: two for a loaded first operand (it streams from HBM), one
  // for the Gram form (only W is loaded, out of L2; the eight registers are what keeps its loop free of spills)
  constexpr int PF = SRC == 1 ? 1 : 2;
  float4 ra[PF], rw[PF][TN];
  float asc = 1.f, amx = 0.f;                   // scale of this staging thread's A row | its largest magnitude so far
  float gza[3];
  float4 gzb[3];
  int ga = 0, gb = 0;
  auto gload_gram = [&]() {
#pragma unroll
    for (int sx = 0; sx < 3; sx++) gzb[sx] = *reinterpret_cast<const float4*>(arow_g + 32 * sx + 4 * gb);
    if (gb == 0) {
#pragma unroll
      for (int sx = 0; sx < 3; sx++) gza[sx] = arow_g[32 * sx + 4 * ga + kq] * as_sh[r4 & 63];      // the Gram entries scaled by the row's power of two (kept in LDS: eight reads per workgroup, no register)
    }
    if (++gb > ga) { ga++; gb = 0; }
  };
  auto gload = [&](int slot, int k0) {
    if (SRC == 0 && stage_a) ra[slot] = *reinterpret_cast<const float4*>(arow_g + k0);
#pragma unroll
    for (int i = 0; i < TN; i++) rw[slot][i] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wrow_g[i]) + k0);
  };
  auto sstore = [&](int slot, int st) {
    char* base = lds + st * kStage1;
    if (stage_a) {
      if (SRC == 1) {
        const float za0 = gza[0], za1 = gza[1], za2 = gza[2];
        const float4 b0 = gzb[0], b1 = gzb[1], b2 = gzb[2];
        put(base, kPlaneA, r4, make_float4(za0 * b0.x + za1 * b1.x + za2 * b2.x, za0 * b0.y + za1 * b1.y + za2 * b2.y,
                                           za0 * b0.z + za1 * b1.z + za2 * b2.z, za0 * b0.w + za1 * b1.w + za2 * b2.w), 1.f, nullptr);
      } else {
        put(base, kPlaneA, r4, ra[slot], asc, &amx);
      }
    }
#pragma unroll
    for (int i = 0; i < TN; i++) put_words(base + 2 * kPlaneA, kPlaneW, r4 + 128 * i, rw[slot][i]);
  };
  if (SRC == 1 && stage_a) {
    // fn[m] = ||Z'Z||_F + 1 = ||Z Z'||_F + 1 (gemm_f32.h): six 32-term dot products, a quarter per staging thread of the row;
    // the norm bounds every entry of the row's Gram matrix and is the scale of the generated operand
    float c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float4 x = *reinterpret_cast<const float4*>(arow_g + 8 * kq + 4 * h);
      const float4 y = *reinterpret_cast<const float4*>(arow_g + 32 + 8 * kq + 4 * h);
      const float4 z = *reinterpret_cast<const float4*>(arow_g + 64 + 8 * kq + 4 * h);
      c[0] += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      c[1] += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
      c[2] += z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w;
      c[3] += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      c[4] += x.x * z.x + x.y * z.y + x.z * z.z + x.w * z.w;
      c[5] += y.x * z.x + y.y * z.y + y.z * z.z + y.w * z.w;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      c[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[k]), 0xB1, 0xF, 0xF, true));
      c[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[k]), 0x4E, 0xF, 0xF, true));
    }
    const float frob = sqrtf((c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + 2.f * (c[3] * c[3] + c[4] * c[4] + c[5] * c[5]));
    if (kq == 0 && m0 + r4 < a.M && a.fn_out) a.fn_out[m0 + r4] = frob + 1.0f;
    const float gs = pow2_scale(frob, kScaleExact);
    if (kq == 0) { rs_sh[r4] = pow2_inv(gs); as_sh[r4] = gs; }
  }
  if (SRC == 1) __syncthreads();                // as_sh is read by the row's four staging threads
  const int nk = a.K1 / 16;
  const int aoff = lds_rd<RB>(wm * 32 + li, lh);
  const int boff = 2 * kPlaneA + lds_rd<RB>(wn * 32 * TN + li, lh);
  const bool late = wave >= 4;                  // waves w and w + 4 share a SIMD: opposite phase order (gemm_f32.h SKEW)
  auto body = [&](int kt, int slot) {
    const int st = kt & 1;
    if (!late) {
      if (kt + 1 < nk) sstore(slot, st ^ 1);
      if (SRC == 1 && stage_a && kt + 2 < nk) gload_gram();
      if (kt + 1 + PF < nk) gload(slot, (kt + 1 + PF) * 16);
    }
    const char* base = lds + st * kStage1;
    const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + aoff));
    const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + kPlaneA + aoff));
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + boff + 32 * j * RB));
      const f16x8 bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(base + kPlaneW + boff + 32 * j * RB));
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[j], 0, 0, 0);     // transposed: registers = hidden, lanes = rows
      cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, cor[j], 0, 0, 0);
      cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, cor[j], 0, 0, 0);
    }
    if (late) {
      if (kt + 1 < nk) sstore(slot, st ^ 1);
      if (kt + 1 + PF < nk) gload(slot, (kt + 1 + PF) * 16);
    }
    if (SRC == 0 && kt == nk - 1 && !(amx <= kF16Lim)) redo_sh = 1u;       // the vote on repeating rides on the last barrier
    __syncthreads();
  };
  for (int attempt = 0;; attempt++) {
    if (SRC == 1 && stage_a) { ga = 0; gb = 0; gload_gram(); }
    float4 sma = make_float4(0.f, 0.f, 0.f, 0.f);      // a sample of the row's middle (staging thread 0) / last (thread 1) k-tile, in flight with tile 0
    if (SRC == 0 && stage_a && attempt == 0 && kq < 2) sma = *reinterpret_cast<const float4*>(arow_g - 4 * kq + (kq == 0 ? (nk >> 1) : nk - 1) * 16);
    gload(0, 0);
    if (SRC == 0 && stage_a) {
      if (attempt == 0) {                       // estimate from three sampled k-tiles of the row (gemm_f32.h, pow2_scale)
        auto m4 = [](const float4& v, float m) { m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), m); return fmaxf(fmaxf(fabsf(v.z), fabsf(v.w)), m); };
        asc = pow2_scale(quad_max(m4(sma, m4(ra[0], 0.f))), kScaleEstimate);
      }
      if (kq == 0) rs_sh[r4] = pow2_inv(asc);
    }
    sstore(0, 0);
    __syncthreads();
    SGRL_CSTAMP(1);
    if (SRC == 1 && stage_a && nk > 1) gload_gram();
    if (nk > 1) gload(0, 16);
    if (PF == 2 && nk > 2) gload(PF - 1, 32);
    if (PF == 1) {
      for (int kt = 0; kt < nk; kt++) body(kt, 0);
    } else {
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) { body(kt, 0); body(kt + 1, PF - 1); }
      if (kt < nk) body(kt, 0);
    }
    if (SRC != 0 || attempt == 1) break;
    if (redo_sh == 0u) break;                   // (block-uniform) the usual exit: the estimate held
    if (t == 0) atomicAdd(&g_scale_redos, 1u);
    {                                           // the rare path: the exact maximum of this thread's A row (read again), then once more
      float tm = 0.f;
      if (stage_a)
        for (int k2 = 0; k2 < nk; k2++) {
          const float4 x = *reinterpret_cast<const float4*>(arow_g + 16 * k2);
          tm = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), tm); tm = fmaxf(fmaxf(fabsf(x.z), fabsf(x.w)), tm);
        }
      asc = pow2_scale(quad_max(tm), kScaleExact);
    }
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) { acc[j][e] = 0.f; cor[j][e] = 0.f; }
  }

  SGRL_CSTAMP(2);
  // ---- hand-off: undo the operand scales, bias + ReLU, the row's exact maximum, split in registers ------------------------------
  uint2 Hh[TN][4], Hl[TN][4];
  float hs;                                     // this lane's row (wm * 32 + li): scale of the intermediate
  {
    const float rsn = rs_sh[wm * 32 + li];
    float hv[TN][16];
    float hmax = 0.f;
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int hid = wn * 32 * TN + 32 * j + 8 * g + 4 * lh;
        const float4 b4 = a.b1 ? *reinterpret_cast<const float4*>(a.b1 + hid) : make_float4(0, 0, 0, 0);
        const float4 w4 = a.ws1 ? *reinterpret_cast<const float4*>(a.ws1 + hid) : make_float4(1, 1, 1, 1);
        hv[j][4 * g + 0] = fmaxf((acc[j][4 * g + 0] + cor[j][4 * g + 0] * kCorW) * (rsn * w4.x) + b4.x, 0.f);
        hv[j][4 * g + 1] = fmaxf((acc[j][4 * g + 1] + cor[j][4 * g + 1] * kCorW) * (rsn * w4.y) + b4.y, 0.f);
        hv[j][4 * g + 2] = fmaxf((acc[j][4 * g + 2] + cor[j][4 * g + 2] * kCorW) * (rsn * w4.z) + b4.z, 0.f);
        hv[j][4 * g + 3] = fmaxf((acc[j][4 * g + 3] + cor[j][4 * g + 3] * kCorW) * (rsn * w4.w) + b4.w, 0.f);
        hmax = fmaxf(fmaxf(hmax, fmaxf(hv[j][4 * g + 0], hv[j][4 * g + 1])), fmaxf(hv[j][4 * g + 2], hv[j][4 * g + 3]));
      }
    hmax = fmaxf(hmax, __shfl_xor(hmax, 32, 64));
    if (lh == 0) hm_sh[(wm * 32 + li) * 4 + wn] = hmax;
    __syncthreads();
    const float4 q = *reinterpret_cast<const float4*>(hm_sh + (wm * 32 + li) * 4);
    hs = pow2_scale(fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)), kScaleExact);
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        unsigned h0, l0, h1, l1;
        split2h(hv[j][4 * g + 0] * hs, hv[j][4 * g + 1] * hs, h0, l0);
        split2h(hv[j][4 * g + 2] * hs, hv[j][4 * g + 3] * hs, h1, l1);
        Hh[j][g] = make_uint2(h0, h1);
        Hl[j][g] = make_uint2(l0, l1);
      }
  }
  const int my_slice = HID == 256 ? wn : (wn >> 1);
  auto write_slice = [&](int buf) {
    char* sb = lds + buf * kSliceBuf + (wm * 32 + li) * SLP;
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int off = 2 * ((HID == 256 ? 32 * j : 32 * (wn & 1)) + 8 * g + 4 * lh);
        *reinterpret_cast<uint2*>(sb + off) = Hh[j][g];
        *reinterpret_cast<uint2*>(sb + kSlicePlane + off) = Hl[j][g];
      }
  };

  if constexpr ((EPI2 & EPI_EQUIV) != 0) {
    // ---- phase 2, equivariant form: 1024 outputs per row in eight passes of 128 (four 32 x 32 matrices blocks c = 4 p + wn) ----------
    // The planes of the intermediate stay in REGISTERS (Hh / Hl) and are handed to the other waves again in every pass, through two
    // 32-column sub-slice buffers: k-steps 2 u, 2 u + 1 read sub-slice u from buffer u & 1 while its owners write sub-slice u + 1
    // into the other one.  Only W2 is staged (the pass's 128 rows, one k-step at a time); z rows and the bias wait in LDS.
    constexpr int kSubPlane = R * 80, kSubBuf = kChainEqSub;
    float* zs = reinterpret_cast<float*>(lds + kChainEqZ);
    float* bs = reinterpret_cast<float*>(lds + kChainEqB);
    const float* w2base = reinterpret_cast<const float*>(a.W2 + (size_t)r4 * a.ldw2 + 4 * kq);
    auto w2ptr = [&](int gs) { return w2base + (size_t)(gs >> 4) * 128 * a.ldw2 + (gs & 15) * 16; };
    auto store2 = [&](const float4& v, int st) { put_words(lds + kChainEqW2 + st * (2 * 128 * RB), 128 * RB, r4, v); };
    auto write_sub = [&](int u, int buf) {      // sub-slice u = hidden 32 u .. 32 u + 31: tile u & 1 of the waves with wn == u >> 1
      if (wn == (u >> 1)) {
        char* sb = lds + buf * kSubBuf + (wm * 32 + li) * 80;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int off = 2 * (8 * g + 4 * lh);
          *reinterpret_cast<uint2*>(sb + off) = (u & 1) ? Hh[1][g] : Hh[0][g];
          *reinterpret_cast<uint2*>(sb + kSubPlane + off) = (u & 1) ? Hl[1][g] : Hl[0][g];
        }
      }
    };
    for (int idx = t; idx < R * 24; idx += 512) {                 // z rows of the block (rows beyond M: the last valid one)
      const int r = idx / 24, q4 = idx % 24;
      *reinterpret_cast<float4*>(zs + r * 100 + 4 * q4) = *reinterpret_cast<const float4*>(a.zq + (size_t)min(m0 + r, a.M - 1) * 96 + 4 * q4);
    }
    for (int idx = t; idx < 256; idx += 512) *reinterpret_cast<float4*>(bs + 4 * idx) = a.b2 ? *reinterpret_cast<const float4*>(a.b2 + 4 * idx) : make_float4(0, 0, 0, 0);
    float4 rw2[2];
    rw2[0] = *reinterpret_cast<const float4*>(w2ptr(0));
    write_sub(0, 0);
    store2(rw2[0], 0);
    __syncthreads();
    rw2[0] = *reinterpret_cast<const float4*>(w2ptr(1));
    rw2[1] = *reinterpret_cast<const float4*>(w2ptr(2));
    const int m = m0 + wm * 32 + li;
    const bool ok = m < a.M;
    const float rd = (EPI2 & EPI_ROWDIV) ? 1.0f / a.rowdiv[ok ? m : a.M - 1] : 1.f;
    const float hsi = pow2_inv(hs);
    const int hoff = (wm * 32 + li) * 80 + 16 * lh;
    const int w2off = kChainEqW2 + lds_rd<RB>(wn * 32 + li, lh);
    for (int p = 0; p < 8; p++) {
      f32x16 acc2, cor2;
#pragma unroll
      for (int e = 0; e < 16; e++) { acc2[e] = 0.f; cor2[e] = 0.f; }
#pragma unroll
      for (int kt = 0; kt < 16; kt++) {
        const int gs = 16 * p + kt, st = kt & 1;
        if (!(DBG & 4)) {
          if (gs + 1 < 128) store2(rw2[st], st ^ 1);
          if (gs + 3 < 128) rw2[st] = *reinterpret_cast<const float4*>(w2ptr(gs + 3));
        }
        if (!(DBG & 1) && (kt & 1) == 0) write_sub(((kt >> 1) + 1) & 7, ((kt >> 1) + 1) & 1);
        const char* hb = lds + ((kt >> 1) & 1) * kSubBuf + hoff + 32 * (kt & 1);
        const f16x8 hh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(hb));
        const f16x8 hl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(hb + kSubPlane));
        const char* wb = lds + w2off + st * (2 * 128 * RB);
        const f16x8 wh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wb));
        const f16x8 wl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wb + 128 * RB));
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh, acc2, 0, 0, 0);       // registers = q (W2 rows c * 32 + q), lanes = rows
        cor2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl, cor2, 0, 0, 0);
        cor2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh, cor2, 0, 0, 0);
        __syncthreads();
      }
      if (DBG & 2) { if (ok && lh == 0 && acc2[0] + cor2[0] == 12345.f) a.tout[m] = 1.f; continue; }
      // contraction of this pass's block c with the row's three z vectors (16 register FMAs per s, one exchange between the halves)
      const int c = 4 * p + wn;
      const float un = hsi * (a.ws2 ? a.ws2[c * 32] : 1.f);
      float ts[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int q0 = 8 * g + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4*>(bs + c * 32 + q0);
        const float v0 = (acc2[4 * g + 0] + cor2[4 * g + 0] * kCorW) * un + b4.x, v1 = (acc2[4 * g + 1] + cor2[4 * g + 1] * kCorW) * un + b4.y;
        const float v2 = (acc2[4 * g + 2] + cor2[4 * g + 2] * kCorW) * un + b4.z, v3 = (acc2[4 * g + 3] + cor2[4 * g + 3] * kCorW) * un + b4.w;
#pragma unroll
        for (int sx = 0; sx < 3; sx++) {
          const float4 z4 = *reinterpret_cast<const float4*>(zs + (wm * 32 + li) * 100 + sx * 32 + q0);
          ts[sx] += z4.x * v0 + z4.y * v1 + z4.z * v2 + z4.w * v3;
        }
      }
#pragma unroll
      for (int sx = 0; sx < 3; sx++) {
        float tv = ts[sx] * rd;
        tv += __shfl_xor(tv, 32, 64);
        if (ok && lh == 0) a.tout[(size_t)m * 96 + sx * 32 + c] = tv;
      }
    }
    if (a.g == nullptr) return;                 // (uniform over the launch)
    // ---- the vector stream's update for the block's 192 rows: a [192, 32] x [32, 128] product on the matrix cores ------------------
    __syncthreads();                            // every wave's tout rows are in memory (same CU) and the LDS is free
    {
      constexpr int kTP = 192 * 80, kWB = 2 * kTP, kWP = 128 * 80;        // T planes | W5 planes, rows of 32 f16 + 16 B pad
      static_assert(kWB + 2 * kWP + 192 * 4 <= kChainEqLds, "LDS image of the update");
      float* ts_sh = reinterpret_cast<float*>(lds + kWB + 2 * kWP);       // 1 / scale of the 192 T rows (sub-slices, stages, z rows: all dead)
      const int rows3 = 3 * a.M, x0 = 3 * m0;
#pragma unroll
      for (int i = 0; i < 3; i++) {             // T rows: eight staging threads per row, exact row maximum, split, store
        const int idx = t + 512 * i, r = idx >> 3, q = idx & 7;
        const float4 v = *reinterpret_cast<const float4*>(a.tout + (size_t)min(x0 + r, rows3 - 1) * 32 + 4 * q);
        float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        const float sc = pow2_scale(mx, kScaleExact);
        if (q == 0) ts_sh[r] = pow2_inv(sc);
        unsigned h0, l0, h1, l1;
        split2h(v.x * sc, v.y * sc, h0, l0);
        split2h(v.z * sc, v.w * sc, h1, l1);
        char* pp = lds + r * 80 + 8 * q;
        *reinterpret_cast<uint2*>(pp) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(pp + kTP) = make_uint2(l0, l1);
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {             // W5 words [128][32]
        const int idx = t + 512 * i, r = idx >> 3, q = idx & 7;
        const uint4 w = *reinterpret_cast<const uint4*>(a.W5 + (size_t)r * 32 + 4 * q);
        char* pp = lds + kWB + r * 80 + 8 * q;
        *reinterpret_cast<uint2*>(pp) = make_uint2(__builtin_amdgcn_perm(w.y, w.x, 0x05040100u), __builtin_amdgcn_perm(w.w, w.z, 0x05040100u));
        *reinterpret_cast<uint2*>(pp + kWP) = make_uint2(__builtin_amdgcn_perm(w.y, w.x, 0x07060302u), __builtin_amdgcn_perm(w.w, w.z, 0x07060302u));
      }
      __syncthreads();
      const int ct = wave & 3;                  // this wave's 32 output columns; its three row tiles: (wave >> 2) + 2 i
      const float4 w5s[4] = {*reinterpret_cast<const float4*>(a.ws5 + ct * 32 + 4 * lh), *reinterpret_cast<const float4*>(a.ws5 + ct * 32 + 8 + 4 * lh),
                             *reinterpret_cast<const float4*>(a.ws5 + ct * 32 + 16 + 4 * lh), *reinterpret_cast<const float4*>(a.ws5 + ct * 32 + 24 + 4 * lh)};
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int rt = (wave >> 2) + 2 * i, rl = 32 * rt + li, row = x0 + rl;
        f32x16 ua, uc;
#pragma unroll
        for (int e = 0; e < 16; e++) { ua[e] = 0.f; uc[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          const char* tb = lds + rl * 80 + 32 * ks + 16 * lh;
          const char* wb5 = lds + kWB + (ct * 32 + li) * 80 + 32 * ks + 16 * lh;
          const f16x8 th = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(tb)), tl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(tb + kTP));
          const f16x8 wh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wb5)), wl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wb5 + kWP));
          ua = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, th, ua, 0, 0, 0);          // registers = output columns, lanes = rows
          uc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, tl, uc, 0, 0, 0);
          uc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, th, uc, 0, 0, 0);
        }
        if (row < rows3) {
          const float rs = ts_sh[rl];
          float* grow = a.g + (size_t)row * 128 + ct * 32 + 4 * lh;
          const float* g1row = a.g1 + (size_t)row * 128 + ct * 32 + 4 * lh;
#pragma unroll
          for (int gq = 0; gq < 4; gq++) {
            const float4 x = *reinterpret_cast<const float4*>(grow + 8 * gq), y = *reinterpret_cast<const float4*>(g1row + 8 * gq);
            const float4 u = w5s[gq];
            const float4 gn = make_float4(x.x + (y.x + (ua[4 * gq + 0] + uc[4 * gq + 0] * kCorW) * (rs * u.x)), x.y + (y.y + (ua[4 * gq + 1] + uc[4 * gq + 1] * kCorW) * (rs * u.y)),
                                          x.z + (y.z + (ua[4 * gq + 2] + uc[4 * gq + 2] * kCorW) * (rs * u.z)), x.w + (y.w + (ua[4 * gq + 3] + uc[4 * gq + 3] * kCorW) * (rs * u.w)));
            *reinterpret_cast<float4*>(grow + 8 * gq) = gn;
            if (a.outg) *reinterpret_cast<float4*>(a.outg + (size_t)row * a.outg_ld + 8 + ct * 32 + 4 * lh + 8 * gq) = gn;
          }
          if (a.outg && ct == 3 && lh == 1) {   // the zero K-padding columns of the read-out operand (136 .. outg_ld - 1)
            for (int cpad = 136; cpad < a.outg_ld; cpad += 4) *reinterpret_cast<float4*>(a.outg + (size_t)row * a.outg_ld + cpad) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
    return;
  }
  SGRL_CSTAMP(3);
  // ---- phase 2 ------------------------------------------------------------------------------------------------------------
  constexpr int nk2 = HID / 16;
  f32x16 acc2, cor2;
#pragma unroll
  for (int e = 0; e < 16; e++) { acc2[e] = 0.f; cor2[e] = 0.f; }
  const float* w2row = reinterpret_cast<const float*>(a.W2 + (size_t)r4 * a.ldw2 + 4 * kq);
  float4 rw2[2];
  auto store2 = [&](int slot, int st) { put_words(lds + kW2Base + st * kW2Stage, kW2Plane, r4, rw2[slot]); };
  rw2[0] = *reinterpret_cast<const float4*>(w2row);
  if (my_slice < 2) write_slice(my_slice);      // (the last barrier of phase 1 has freed the stage bytes)
  store2(0, 0);
  __syncthreads();
  rw2[0] = *reinterpret_cast<const float4*>(w2row + 16);
  if (nk2 > 2) rw2[1] = *reinterpret_cast<const float4*>(w2row + 32);
  const int hoff = (wm * 32 + li) * SLP + 16 * lh;
  const int w2off = kW2Base + lds_rd<RB>(wn * 32 + li, lh);
  auto body2 = [&](int kt, int slot) {
    const int st = kt & 1;
    if (kt + 1 < nk2) store2(slot, st ^ 1);
    if (kt + 3 < nk2) rw2[slot] = *reinterpret_cast<const float4*>(w2row + (kt + 3) * 16);
    if (HID == 256) {                           // slices 2 / 3 replace slices 0 / 1 once their last reader has passed its barrier
      if (kt == 4 && my_slice == 2) write_slice(0);
      if (kt == 8 && my_slice == 3) write_slice(1);
    }
    const char* hb = lds + ((kt >> 2) & 1) * kSliceBuf + hoff + 32 * (kt & 3);
    const f16x8 hh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(hb));
    const f16x8 hl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(hb + kSlicePlane));
    const char* wb = lds + w2off + st * kW2Stage;
    const f16x8 wh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wb));
    const f16x8 wl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wb + kW2Plane));
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh, acc2, 0, 0, 0);           // transposed: registers = columns, lanes = rows
    cor2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl, cor2, 0, 0, 0);
    cor2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh, cor2, 0, 0, 0);
    __syncthreads();
  };
#pragma unroll
  for (int kt = 0; kt < nk2; kt += 2) { body2(kt, 0); body2(kt + 1, 1); }

  SGRL_CSTAMP(4);
  // ---- epilogue: lane = row m, registers = columns 32 wn + 8 g + 4 lh + (0..3) ----------------------------------------------
  const int m = m0 + wm * 32 + li;
  const bool ok = m < a.M;
  const float rd = (EPI2 & EPI_ROWDIV) ? 1.0f / a.rowdiv[ok ? m : a.M - 1] : 1.f;
  const float hsi = pow2_inv(hs);
  float v[16];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const float4 b4 = a.b2 ? *reinterpret_cast<const float4*>(a.b2 + wn * 32 + 8 * g + 4 * lh) : make_float4(0, 0, 0, 0);
    const float4 w4 = a.ws2 ? *reinterpret_cast<const float4*>(a.ws2 + wn * 32 + 8 * g + 4 * lh) : make_float4(1, 1, 1, 1);
    v[4 * g + 0] = ((acc2[4 * g + 0] + cor2[4 * g + 0] * kCorW) * (hsi * w4.x) + b4.x) * rd;
    v[4 * g + 1] = ((acc2[4 * g + 1] + cor2[4 * g + 1] * kCorW) * (hsi * w4.y) + b4.y) * rd;
    v[4 * g + 2] = ((acc2[4 * g + 2] + cor2[4 * g + 2] * kCorW) * (hsi * w4.z) + b4.z) * rd;
    v[4 * g + 3] = ((acc2[4 * g + 3] + cor2[4 * g + 3] * kCorW) * (hsi * w4.w) + b4.w) * rd;
  }
  if (EPI2 & EPI_LN) {
    float* red = gemm_lds;                      // [64 rows][4 column groups]; the LDS is idle (last barrier of phase 2)
    float* rrow = a.ln_io + (size_t)(ok ? m : a.M - 1) * a.ln_ld + wn * 32 + 4 * lh;
    float* orow = a.ln_out ? a.ln_out + (size_t)(ok ? m : a.M - 1) * a.ln_ld + wn * 32 + 4 * lh : rrow;
    float part = 0.f;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const float4 x = *reinterpret_cast<const float4*>(rrow + 8 * g);
      v[4 * g + 0] += x.x; v[4 * g + 1] += x.y; v[4 * g + 2] += x.z; v[4 * g + 3] += x.w;
      part += (v[4 * g + 0] + v[4 * g + 1]) + (v[4 * g + 2] + v[4 * g + 3]);
    }
    auto row_sum = [&](float p) -> float {      // over the row's 128 columns: the other half-wave, then the four column groups
      p += __shfl_xor(p, 32, 64);
      __syncthreads();
      if (lh == 0) red[(wm * 32 + li) * 4 + wn] = p;
      __syncthreads();
      const float4 q = *reinterpret_cast<const float4*>(red + (wm * 32 + li) * 4);
      return (q.x + q.y) + (q.z + q.w);
    };
    const float mu = row_sum(part) * (1.f / 128.f);
    part = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) { v[e] -= mu; part += v[e] * v[e]; }
    const float inv = 1.0f / sqrtf(row_sum(part) * (1.f / 128.f) + 1e-5f);
    if (ok) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const float4 lw = *reinterpret_cast<const float4*>(a.ln_w + wn * 32 + 8 * g + 4 * lh);
        const float4 lb = *reinterpret_cast<const float4*>(a.ln_b + wn * 32 + 8 * g + 4 * lh);
        *reinterpret_cast<float4*>(orow + 8 * g) = make_float4(v[4 * g + 0] * inv * lw.x + lb.x, v[4 * g + 1] * inv * lw.y + lb.y,
                                                                v[4 * g + 2] * inv * lw.z + lb.z, v[4 * g + 3] * inv * lw.w + lb.w);
      }
    }
    return;
  }
  if (ok) {
    float* crow = a.C + (size_t)m * a.ldc + wn * 32 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; g++) *reinterpret_cast<float4*>(crow + 8 * g) = make_float4(v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
  }
  SGRL_CSTAMP(5);
#ifdef SGRL_CHAIN_PROF
  if (threadIdx.x == 0) { atomicAdd(&g_chain_prof[6], (unsigned long long)(__builtin_readcyclecounter() - cp_begin)); atomicAdd(&g_chain_prof[7], 1ull); }
#endif
}

}  // namespace sgrl_gemm
