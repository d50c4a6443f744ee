// gemm_f32.h -- the SET actor's f32 MFMA GEMM for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)
//
// v_mfma_f32_32x32x2_f32 (exact f32 arithmetic: results track the reference's f32 PyTorch path).  Row-major LDS tiles
// [rows][BK + 4] so that both the global->LDS staging (one ds_write_b128 per float4) and the operand fetch (one
// ds_read_b128 = FOUR k-steps of a lane's operand) move 16 bytes per instruction: lane (row, half) of the MFMA owns the
// contiguous k range [8 * half, 8 * half + 8) of its row -- the matrix instruction sums over k, so which pair of k values
// an MFMA step covers is free as long as A and W use the same assignment.  Row stride 20 dwords: the 16 lanes of every
// ds_read_b128 lane group fall on distinct 4-bank groups (20 r mod 64 is injective on 16 consecutive rows).
#pragma once
#include <hip/hip_runtime.h>

namespace sgrl_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// EPI_ZSPLIT (the stacked 3-vector projections): output column n < 30 goes to C[m][n], column 32 <= n < 62 to
// C2[m][n - 32], both with row stride 32 (columns 30 / 31 of those rows hold the gravity / direction pair, written once per
// forward by k_embed); the zero-padding columns 30, 31, 62, 63 of the stacked operand are not stored
// EPI_LN (k_gemm3, N = 128 = one column tile): the result is not stored; it is the update of a residual stream that is layer-
// normalised in place: ln_io[m][:] = LayerNorm(ln_io[m][:] + value[m][:]) * ln_w + ln_b over the 128 columns of the row
// EPI_TR (k_gemm3): the tiles are accumulated TRANSPOSED (lane = row, registers = columns) and stored row-wise, 16 bytes per lane
// and store instruction (plain / ReLU / row-division epilogues; one divisor per lane instead of sixteen)
enum { EPI_RELU = 1, EPI_ROWDIV = 2, EPI_ACC2 = 4, EPI_EQUIV = 8, EPI_ZSPLIT = 16, EPI_LN = 32, EPI_TR = 64 };

struct GemmArgs {
  const float* A; int lda;
  const float* W; int ldw;
  const float* bias;
  float* C; int ldc;
  int M, N, K;
  int flags;
  const float* rowdiv;  // [M]: C = (A.W^T + b) / rowdiv[m]
  float* C2; int ldc2;  // EPI_ACC2: C2[m][n] += value
  // split-precision kernel only: an operand may arrive PRE-SPLIT as three bf16 planes (h | m | l, each [rows][ld] bf16,
  // `plane` bf16 elements apart) instead of f32 -- A/W then point at plane h and lda/ldw count bf16 elements
  long long a_plane = 0, w_plane = 0;
  // EPI_EQUIV (split-precision kernel, N = 1024 with the output columns ordered c * 32 + a): the [M, 1024] result -- one
  // 32 x 32 matrix mat[m][a][c] per row m -- is NOT stored; the epilogue contracts it on the fly with zq[m][s][a] (three
  // 32-vectors per row) and stores only tout[m][s][c] = sum_a zq[m][s][a] * mat[m][a][c]  ([M, 3, 32] floats)
  const float* zq = nullptr;
  float* tout = nullptr;
  float* ln_io = nullptr; int ln_ld = 0;          // EPI_LN: residual stream (read and rewritten), row stride
  const float* ln_w = nullptr; const float* ln_b = nullptr;
  float* rowdiv_out = nullptr;   // GRAM: receives ||Z'Z||_F + 1 per row (from the blocks of the first column tile)
  int phase_sleep = 0;                // k_gemm3: blocks of odd dispatch rounds start this many x 64 clocks late (see the kernel)
  // two-piece f16 form: W arrives as words of its rows PRE-SCALED by powers of two (k_encode_rows); wscale[n] undoes row n's scale
  const float* wscale = nullptr;
  // lab form only (k_gemm3 WORDS = 3, tools/chain_lab.hip): A arrives as row-scaled words too; ascale[m] undoes row m's scale
  const float* ascale = nullptr;
};

// BKT = k extent of an LDS tile (16 or 32); row stride BKT + 4 floats (conflict-free ds_read_b128, see above).
// PF = register prefetch depth: global loads run PF k-tiles ahead of the LDS stage they are written to (PF = 2 doubles the
// latency budget of a load at the price of one more set of staging registers).
template <int WM, int WN, int TM, int TN, int BKT>
struct TileCfg {
  static constexpr int kThreads = 64 * WM * WN;
  static constexpr int kBM = 32 * TM * WM, kBN = 32 * TN * WN;
  static constexpr int kSK = BKT + 4;
  static constexpr int kLdsBytes = 2 * (kBM + kBN) * kSK * (int)sizeof(float);
};

// WM x WN waves, each computing a (32 TM) x (32 TN) patch of the block tile.
template <int FLAGS, int WM, int WN, int TM, int TN, int BKT = 16, int PF = 1>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm2(GemmArgs a) {
  using Cfg = TileCfg<WM, WN, TM, TN, BKT>;
  constexpr int T = Cfg::kThreads, BMT = Cfg::kBM, BNT = Cfg::kBN, SK = Cfg::kSK;
  constexpr int QPR = BKT / 4;                // float4 per tile row
  constexpr int RPP = T / QPR;                // tile rows covered per staging pass
  static_assert(BMT % RPP == 0 && BNT % RPP == 0, "tile rows must be a multiple of the staging pass");
  static_assert(PF == 1 || PF == 2, "prefetch depth 1 or 2");
  constexpr int NPA = BMT / RPP, NPW = BNT / RPP;
  constexpr int KH = BKT / 2;                 // k values per MFMA half (lanes 0..31 | 32..63)
  constexpr int NQ = KH / 4;                  // float4 per lane per operand row per tile
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];     // (aligned: static LDS precedes it)
  float* As = gemm_lds;                        // [2][BMT][SK]
  float* Ws = gemm_lds + 2 * BMT * SK;         // [2][BNT][SK]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.N + BNT - 1) / BNT;
  // blocks b and b+8 share an XCD (round-robin dispatch): renumber so that each XCD works on a contiguous run of tiles
  // (the column tiles of one row tile re-read the same A rows out of ONE L2); speed only, never correctness
  int bid;
  {
    const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    bid = (x < rem) ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BMT, n0 = tile_n * BNT;
  const int kq = t % QPR, r0 = t / QPR;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  float4 ra[PF][NPA], rw[PF][NPW];
  auto gload = [&](int slot, int k0) {
#pragma unroll
    for (int i = 0; i < NPA; i++) {
      const int m = m0 + r0 + RPP * i;
      ra[slot][i] = (m < a.M) ? *reinterpret_cast<const float4*>(a.A + (size_t)m * a.lda + k0 + 4 * kq) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NPW; i++) {
      const int n = n0 + r0 + RPP * i;
      rw[slot][i] = (n < a.N) ? *reinterpret_cast<const float4*>(a.W + (size_t)n * a.ldw + k0 + 4 * kq) : make_float4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int slot, int st) {
#pragma unroll
    for (int i = 0; i < NPA; i++) *reinterpret_cast<float4*>(As + (st * BMT + r0 + RPP * i) * SK + 4 * kq) = ra[slot][i];
#pragma unroll
    for (int i = 0; i < NPW; i++) *reinterpret_cast<float4*>(Ws + (st * BNT + r0 + RPP * i) * SK + 4 * kq) = rw[slot][i];
  };
  const int nk = a.K / BKT;
  gload(0, 0);
  sstore(0, 0);
  __syncthreads();
  // registers hold tiles 1 .. PF (slot of tile k = (k - 1) % PF)
  if (nk > 1) gload(0, BKT);
  if (PF == 2 && nk > 2) gload(1, 2 * BKT);
  const int li = lane & 31, lh = lane >> 5;
  const float* arow = As + (wm * 32 * TM + li) * SK + KH * lh;
  const float* brow = Ws + (wn * 32 * TN + li) * SK + KH * lh;
  auto body = [&](int kt, int slot) {      // slot = register slot holding tile kt + 1 (compile-time constant per call site)
    const int st = kt & 1;
    // push tile kt+1 into the idle LDS stage FIRST so the writes drain underneath this tile's MFMAs, then start fetching
    // tile kt+1+PF into the register slot just freed
    if (kt + 1 < nk) sstore(slot, st ^ 1);
    if (kt + 1 + PF < nk) gload(slot, (kt + 1 + PF) * BKT);
    float4 av[TM][NQ], bv[TN][NQ];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int q = 0; q < NQ; q++) av[i][q] = *reinterpret_cast<const float4*>(arow + (st * BMT + 32 * i) * SK + 4 * q);
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int q = 0; q < NQ; q++) bv[j][q] = *reinterpret_cast<const float4*>(brow + (st * BNT + 32 * j) * SK + 4 * q);
#pragma unroll
    for (int h = 0; h < NQ; h++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
          const float x = q == 0 ? av[i][h].x : (q == 1 ? av[i][h].y : (q == 2 ? av[i][h].z : av[i][h].w));
#pragma unroll
          for (int j = 0; j < TN; j++) {
            const float y = q == 0 ? bv[j][h].x : (q == 1 ? bv[j][h].y : (q == 2 ? bv[j][h].z : bv[j][h].w));
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  };
  if (PF == 1) {
    for (int kt = 0; kt < nk; kt++) body(kt, 0);
  } else {
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) { body(kt, 0); body(kt + 1, 1); }
    if (kt < nk) body(kt, 0);
  }
  // epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // FLAGS is a compile-time constant: no per-element branches, the 16 row divisors of a tile are fetched together.
#pragma unroll
  for (int ti = 0; ti < TM; ti++) {
    const int mb = m0 + wm * 32 * TM + ti * 32 + 4 * lh;
    float rdiv[16];
    if (FLAGS & EPI_ROWDIV) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        rdiv[e] = 1.0f / ((m < a.M) ? a.rowdiv[m] : 1.f);     // one reciprocal per row, applied by multiplication below
      }
    }
#pragma unroll
    for (int tj = 0; tj < TN; tj++) {
      const int n = n0 + wn * 32 * TN + tj * 32 + li;
      if (n >= a.N) continue;
      const float bvv = a.bias ? a.bias[n] : 0.f;
      float old2[16];
      if (FLAGS & EPI_ACC2) {   // all sixteen second-destination reads in flight before the first store
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int m = mb + (e & 3) + 8 * (e >> 2);
          old2[e] = (m < a.M) ? a.C2[(size_t)m * a.ldc2 + n] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m >= a.M) continue;
        float v = acc[ti][tj][e] + bvv;
        if (FLAGS & EPI_RELU) v = fmaxf(v, 0.f);
        if (FLAGS & EPI_ROWDIV) v = v * rdiv[e];
        if (FLAGS & EPI_ZSPLIT) {
          if (n < 30) a.C[(size_t)m * 32 + n] = v;
          else if (n >= 32 && n < 62) a.C2[(size_t)m * 32 + (n - 32)] = v;
          continue;
        }
        a.C[(size_t)m * a.ldc + n] = v;
        if (FLAGS & EPI_ACC2) a.C2[(size_t)m * a.ldc2 + n] = old2[e] + v;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Split-precision variant: the same f32 GEMM carried by the bf16 matrix cores (16x the f32 MFMA rate on gfx950).
//
// Every f32 operand x is split EXACTLY into three bf16 pieces x = h + m + l (8 + 8 + 8 significand bits, by truncation:
// h = top 16 bits of x, m = top 16 bits of x - h, l = x - h - m, each difference exact in f32), and the product a.b is
// rebuilt from the six partial products whose weight reaches 2^-16 relative: hh | hm mh | hl lh mm.  A bf16 x bf16 product
// is exact in f32 and the matrix core accumulates in f32, so what is dropped is ml + lm + ll <= 2^-23 |a||b| per term --
// the size of the rounding error an f32 FMA chain commits per term anyway.  The leading hh products and the five
// correction products accumulate in separate registers and are added once in the epilogue, so the small terms are not
// rounded against the large running sum.  Six v_mfma_f32_32x32x16_bf16 (32 cycles, k = 16) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles, k = 2) per 16 k-steps: 2.67x fewer matrix-pipe cycles.
// tests/test_split_products_gpu.py (every instantiation, through sgrl_set_debug_product) and tools/chain_lab.hip mode r hold the result
// against float64: same error as the exact-f32 kernel (the round-3 lab tools/gemm_lab.hip is an archive that no longer builds).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2);                 // at most 8 significant bits are left: its low 16 bits are zero
}
// two f32 bit patterns -> one dword holding their top halves (bf16 of the first in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// Two-piece form (NPL = 2): x = h + l' / 2^11 with h = f16(x) and l' = f16((x - h) * 2^11), both rounded to nearest.  h carries
// 11 significant bits, the remainder x - h is exact in f32 and at most 2^-11 |x|, l' carries 11 bits of it: |x - h - l'/2^11|
// <= 2^-22 |x|.  a.b is rebuilt from hh + (h l' + l' h) / 2^11 -- THREE v_mfma_f32_32x32x16_f16 instead of six bf16 ones, two
// LDS planes per operand instead of three; dropped: l'l' / 2^22 <= 2^-22 |a||b|.  An f16 x f16 product is exact in f32 and the
// matrix core accumulates in f32.  The scaling of l' keeps the small piece in f16's NORMAL range wherever h is normal: full
// accuracy for 6.1e-5 <= |x| <= 65 000.  RANGE: f16 ends at 65 504 and loses precision below 6.1e-5, float32 does neither -- so no
// operand is split as it stands.  Every ROW of an operand (a row of A = one node's activations, a row of W = one output
// feature's weights) is first multiplied by a power of two that brings its largest magnitude to 2^14 .. 2^15 (exact in f32), the
// pieces are taken of the scaled values, and the epilogue multiplies the result by the two inverse powers (exact again):
//   * weights: k_encode_rows scales each row by its exact maximum when the flat weight buffer is rebuilt (once per forward);
//   * generated Gram rows: by ||Z'Z||_F, which bounds every entry and is computed in the kernel's prologue anyway;
//   * loaded activation rows: by an ESTIMATE -- the largest magnitude in the row's first k-tile and in the first four values of
//     its middle and last k-tiles (the operands of the forward are concatenations such as [invariants | scalars] whose halves
//     differ in size), placed at 2^6:
//     512 x headroom above, 20 octaves of full precision below; the staging threads keep the largest scaled magnitude they
//     split, and a workgroup that did meet a value beyond the headroom repeats its tile with the exact row maxima -- a
//     block-uniform branch for data no sane network state produces, so the common case costs two extra loads per staging
//     thread, two cross-lane maxima and one vote;
//   * intermediates of the fused chains (chain_f16.h): by their exact row maxima, known in registers.
// Elements more than 2^-22 below their row's maximum lose relative (not absolute) accuracy, as they do in any float32 sum
// dominated by the large terms.  Nothing is clamped, nothing is counted: the form has float32's range by construction
// (tests/test_split_products_gpu.py holds it against float64 at operand magnitudes 1e-20 .. 1e8).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float kF16Lim = 65000.f;
constexpr float kF16LowScale = 2048.f;
constexpr int kScaleExact = 14, kScaleEstimate = 6;     // where a row's exact maximum / sampled estimate is placed (powers of two)
// 2^(target - floor(log2 est)) with the exponent kept inside +-100; est <= 0 or NaN -> 2^100.
// A ZERO estimate means "unknown", not "small": an exact maximum of zero is an all-zero row (any scale will do), but a SAMPLED
// estimate of zero (every sampled entry of a sparse post-ReLU row is 0) says nothing about the rest of the row -- split unscaled,
// entries below f16's normal range (6e-5) would lose bits and those below 6e-8 flush to zero where float32 would not.  At 2^100
// every entry a float32 sum could care about (>= 2^-84) lands beyond the f16 range, the overflow vote fires and the tile is
// repeated on the exact row maxima (counted in g_scale_redos); an all-zero row overflows nothing and costs nothing.
__device__ __forceinline__ float pow2_scale(float est, int target) {
  const int e = (int)((__float_as_uint(est) >> 23) & 0xFFu) - 127;
  const int se = est > 0.f ? min(max(target - e, -100), 100) : 100;
  return __uint_as_float((unsigned)(se + 127) << 23);
}
// diagnostics: workgroups that repeated a tile / a phase with exact row maxima since the counter was last cleared (the rare path;
// tests assert that sane inputs never take it: sgrl_set_debug_redos)
__device__ unsigned g_scale_redos;
__device__ __forceinline__ float pow2_inv(float s) { return __uint_as_float(0x7F000000u - __float_as_uint(s)); }   // 1 / 2^k, exact
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(float x0, float x1, unsigned& h, unsigned& l) {
  const f16x2 hv = __builtin_convertvector(f32x2{x0, x1}, f16x2);                        // one v_cvt_pk_f16_f32 (round to nearest)
  const float r0 = __builtin_fmaf((float)hv[0], -1.0f, x0), r1 = __builtin_fmaf((float)hv[1], -1.0f, x1);   // exact
  const f16x2 lv = __builtin_convertvector(f32x2{r0 * kF16LowScale, r1 * kF16LowScale}, f16x2);
  h = __builtin_bit_cast(unsigned, hv);
  l = __builtin_bit_cast(unsigned, lv);
}

// An operand of the two-piece form may arrive PRE-SPLIT ("words"): one 32-bit word per element, h in the low half, l' in the
// high half -- the same four bytes as the f32 it stands for, written ONCE by the kernel that produces the value (a GEMM
// epilogue, the attention / equivariant kernels, the weight packer) instead of being split again by every column tile of every
// consumer.  On gfx950 VALU instructions do not issue under matrix instructions (tools/micro/mfma_valu_overlap.hip: the times
// add), so the ~22 VALU operations per four elements of an in-loop split cost matrix time; a word operand costs four byte
// permutes per four elements.
__device__ __forceinline__ unsigned enc_word(float x) {      // x: already scaled into the f16 range by its row's power of two
  const float c = __builtin_amdgcn_fmed3f(x, -kF16Lim, kF16Lim);
  const _Float16 h = (_Float16)c;
  const _Float16 l = (_Float16)((c - (float)h) * kF16LowScale);
  return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
__device__ __forceinline__ float dec_word(unsigned w) {
  const _Float16 h = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)), l = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
  return (float)h + (float)l * (1.f / kF16LowScale);
}

// LDS image of one k-tile row of a plane.  16-wide k-tiles (every product kernel in use): rows of 32 bytes, NO padding, the two
// 16-byte halves of rows 4..7 of every eight swapped (XOR swizzle) -- the matrix operands' ds_read_b128 (lane = row, 16 bytes) and
// the staging threads' ds_write_b64 (four threads per row, 8 bytes each) are BOTH bank-conflict free.  Rounds 2-4 used rows of
// 32 + 16 pad bytes: conflict-free reads, but four consecutive rows' 32-byte windows fold onto 96 bytes of banks -- the stores ran
// 2-way conflicted on a quarter of the banks (tools/micro/lds_pattern_lab.hip: 32.0 -> 24.0 ticks per four stores, reads 33.5 ->
// 34.0; SQ_LDS_BANK_CONFLICT 17-24 % of the LDS cycles of every product kernel) -- and took half again the LDS.
// 32-wide k-tiles (lab configurations only) keep the padded rows.
template <int RB> __device__ __forceinline__ int lds_wr(int row, int kq) {      // a staging thread's 8 bytes: quarter kq of the row
  if (RB == 32) return row * 32 + ((((kq >> 1) ^ (row >> 2)) & 1) << 4) + ((kq & 1) << 3);
  return row * RB + 8 * kq;
}
template <int RB> __device__ __forceinline__ int lds_rd(int row, int lh) {      // a lane's 16-byte operand: k-half lh of the row
  if (RB == 32) return row * 32 + (((lh ^ (row >> 2)) & 1) << 4);
  return row * RB + 16 * lh;
}
constexpr int kTileRB16 = 32;                                                   // row bytes of a 16-wide k-tile plane

template <int WM, int WN, int TM, int TN, int BKT = 32, int NPL = 3>
struct TileCfg3 {
  static constexpr int kThreads = 64 * WM * WN;
  static constexpr int kBM = 32 * TM * WM, kBN = 32 * TN * WN;
  static constexpr int kBK = BKT;
  static constexpr int kRowBytes = BKT == 16 ? kTileRB16 : 2 * kBK + 16;     // lds_wr / lds_rd above
  static constexpr int kPlaneA = kBM * kRowBytes, kPlaneW = kBN * kRowBytes;
  static constexpr int kStageBytes = NPL * (kPlaneA + kPlaneW);
  static constexpr int kLdsBytes = 2 * kStageBytes;
};

// PLA / PLW: operand A / W is given as bf16 planes (GemmArgs::a_plane / w_plane) and is copied to LDS as is
// LATE: the next tile is split and written to LDS AFTER this tile's MFMAs have been issued (the matrix pipe runs them
// while the wave does the VALU / LDS-write work) instead of before
// ABL (diagnostics of the archived round-3 lab, profiles/r2_gemm_lab_ablation.log): 1 = no staging in the loop (LDS keeps tile 0: MFMA + operand reads + barrier only),
// 2 = staging only (no operand reads / MFMA), 3 = staging without the split arithmetic (raw bit copies)
// GRAM: the A operand is GENERATED, not loaded: a.A points at Z [M][3][32] (three 32-vectors per row) and A[m][k] is the
// entry G[a][b] = sum_s Z[m][s][a] Z[m][s][b] of the row's 32 x 32 Gram matrix Z'Z, with k running over the 36 blocks
// (4 a-values x 4 b-values, block row >= block column) of its lower triangle -- one block per 16-wide k-tile, K = 576
// (kGramK; the weight rows are folded onto the same order at pack time, entries above the diagonal inside the eight
// diagonal blocks carry zero weight).  The [M, 576] Gram operand (and its [M, 1024] dense form) never exists in memory:
// a staging thread reads 3 + 12 floats of its row's Z (L2-resident: 384 bytes per row) per k-tile and forms 4 entries.
constexpr int kGramK = 576;
// NPL = 3: bf16 x 6 (exact three-way split, f32's exponent range); NPL = 2: f16 x 3 (two-piece split above)
// SKEW: the upper half of a block's waves runs every k-tile in the LATE order (matrix instructions of tile kt first, then the
// split + LDS store of tile kt + 1), the lower half in the early order.  Waves w and w + 4 of a block share a SIMD, and the
// barrier keeps all waves of a block in lockstep: without the skew both waves of a SIMD (and, started together, those of the
// co-resident block) sit in their VALU phase at the same time and in their matrix phase at the same time -- the phases ADD
// (profiles/r2_gemm_lab_ablation.log: full = matrix part + staging part).  With it one wave's split arithmetic runs under the other's matrix
// instructions by construction.  Both orders keep the same invariant at the barrier (LDS stage kt & 1 = tile kt, register
// slots = tiles kt + 1, kt + 2).
template <int FLAGS, int WM, int WN, int TM, int TN, int BKT = 32, int PF = 1, bool PLA = false, bool PLW = false, bool LATE = false, int ABL = 0, bool GRAM = false, int NPL = 3, bool SKEW = false, int WORDS = 0>
#ifndef SGRL_GEMM_WPE
#define SGRL_GEMM_WPE 4        // lab builds compile variants with another register budget (-DSGRL_GEMM_WPE=6 / 8: profiles/r2_gemm_lab_wpe6.log)
#endif
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(SGRL_GEMM_WPE))) void k_gemm3(GemmArgs a) {
  using Cfg = TileCfg3<WM, WN, TM, TN, BKT, NPL>;
  static_assert(NPL == 3 || (NPL == 2 && !PLA && !PLW && ABL != 3), "two-piece form: f32 operands only");
  // two-piece form: the correction accumulator holds 2^11 x its sum; the accumulators hold the product of the ROW-SCALED operands
  // (see pow2_scale): the epilogues multiply by rs_sh[row] (A) and a.wscale[column] (W), both powers of two
  constexpr float kCorW = NPL == 2 ? 1.f / kF16LowScale : 1.f;
  constexpr bool SCL = NPL == 2;
  auto fin = [&](float hi, float co) -> float { return NPL == 2 ? hi + co * kCorW : hi + co; };
  static_assert(!SKEW || (!LATE && ABL == 0 && WM * WN == 8), "the skew assumes eight waves (w and w + 4 on one SIMD)");
  // WORDS: bit 1 = W arrives as pre-scaled, pre-split words (k_encode_rows); an f32 W of the two-piece form is split as it stands
  // (lab / diagnostics only: no row scaling on that side).  Bit 0 (WORDS = 3, LAB ONLY -- the forward has no producer that writes
  // such an operand): A arrives as words too, with GemmArgs::ascale; nothing of the row-scale estimate / vote / split is left in
  // the k-loop (what a consumer gains when its producer emits the split form)
  static_assert(WORDS == 0 || (NPL == 2 && (WORDS == 2 || (WORDS == 3 && !GRAM && !PLA))), "pre-split words: operands of the two-piece form");
  constexpr bool WWD = (WORDS & 2) != 0, AWD = (WORDS & 1) != 0;
  constexpr int T = Cfg::kThreads, BMT = Cfg::kBM, BNT = Cfg::kBN, RB = Cfg::kRowBytes;
  constexpr int QPR = BKT / 4;                // float4 per tile row
  constexpr int RPP = T / QPR;                // tile rows covered per staging pass
  static_assert(PF == 1 || PF == 2, "prefetch depth 1 or 2");
  constexpr int NPA = (BMT + RPP - 1) / RPP, NPW = (BNT + RPP - 1) / RPP;    // a pass may be partly idle (more threads than float4s)
  constexpr int KS = BKT / 16;                // MFMA k-steps per LDS tile
  static_assert(!GRAM || (BKT == 16 && RPP == BMT && !PLA), "the Gram operand needs one 16-wide k-tile per block pair and one staging pass");
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];     // (aligned: static LDS precedes it)
  char* lds = reinterpret_cast<char*>(gemm_lds);
  __shared__ float rs_sh[SCL ? BMT : 1];      // two-piece form: 1 / (scale of A row r of the tile), for the epilogue
  __shared__ unsigned redo_sh;                // ... and the workgroup vote on repeating the tile with exact row maxima
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (a.N + BNT - 1) / BNT;
  int bid;
  {
    const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    bid = (x < rem) ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BMT, n0 = tile_n * BNT;
  const int kq = t % QPR, r0 = t / QPR;
  float asc[NPA], amx[NPA];                   // scale of this staging thread's A rows | the largest scaled magnitude it has split
#pragma unroll
  for (int i = 0; i < NPA; i++) { asc[i] = 1.f; amx[i] = 0.f; }
  auto quad_max = [&](float m) -> float {     // over the QPR staging threads of a row (adjacent lanes)
#pragma unroll
    for (int o = 1; o < QPR; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
  };
  f32x16 acc[TM][TN], cor[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) { acc[i][j][e] = 0.f; cor[i][j][e] = 0.f; }
  // staging registers: an f32 operand travels as one float4 (16 B) per slot, a pre-split one as three uint2 (24 B)
  float4 ra[PF][NPA], rw[PF][NPW];
  uint2 pa[PF][PLA ? NPA : 1][3], pw[PF][PLW ? NPW : 1][3];
  // GRAM: ONE staging slot for A (its Z values come from L2 and are fetched one k-tile ahead), independent of PF
  float gza[3];                               // Z[m][s][4 A + kq]
  float4 gzb[3];                              // Z[m][s][4 B .. 4 B + 3]
  int ga = 0, gb = 0;                         // block pair (A, B) of the next k-tile to be loaded (tiles are loaded in order)

  const unsigned short* Ap = reinterpret_cast<const unsigned short*>(a.A);
  const unsigned short* Wp = reinterpret_cast<const unsigned short*>(a.W);
  // Rows beyond M / N are CLAMPED to the last valid row instead of predicated: the loop body stays free of branches (the
  // values computed for them are never stored -- the epilogue checks the bounds)
  const float* arow_g[NPA];
  const float* wrow_g[NPW];
#pragma unroll
  for (int i = 0; i < NPA; i++) {
    const int m = min(m0 + min(r0 + RPP * i, BMT - 1), a.M - 1);
    arow_g[i] = GRAM ? a.A + (size_t)m * 96
                     : (PLA ? reinterpret_cast<const float*>(Ap + (size_t)m * a.lda + 4 * kq) : a.A + (size_t)m * a.lda + 4 * kq);
  }
#pragma unroll
  for (int i = 0; i < NPW; i++) {
    const int n = min(n0 + min(r0 + RPP * i, BNT - 1), a.N - 1);
    wrow_g[i] = PLW ? reinterpret_cast<const float*>(Wp + (size_t)n * a.ldw + 4 * kq) : a.W + (size_t)n * a.ldw + 4 * kq;
  }
  auto gload_gram = [&]() {
#pragma unroll
    for (int sx = 0; sx < 3; sx++) gzb[sx] = *reinterpret_cast<const float4*>(arow_g[0] + 32 * sx + 4 * gb);
    if (gb == 0) {                            // a new block row: this thread's a = 4 A + kq changes (uniform branch)
#pragma unroll
      for (int sx = 0; sx < 3; sx++) gza[sx] = arow_g[0][32 * sx + 4 * ga + kq] * asc[0];     // G entries scaled by the row's power of two
    }
    if (++gb > ga) { ga++; gb = 0; }
  };
  auto gload = [&](int slot, int k0) {
#pragma unroll
    for (int i = 0; i < NPA; i++) {
      if (GRAM) {
      } else if (PLA) {
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
          pa[slot][PLA ? i : 0][pl] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(arow_g[i]) + pl * a.a_plane + k0);
      } else {
        ra[slot][i] = *reinterpret_cast<const float4*>(arow_g[i] + k0);
      }
    }
#pragma unroll
    for (int i = 0; i < NPW; i++) {
      if (PLW) {
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
          pw[slot][PLW ? i : 0][pl] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(wrow_g[i]) + pl * a.w_plane + k0);
      } else {
        rw[slot][i] = *reinterpret_cast<const float4*>(wrow_g[i] + k0);
      }
    }
  };
  auto put_planes = [&](char* plane0, int plane_stride, int row, const uint2* v) {
    char* p = plane0 + lds_wr<RB>(row, kq);
    *reinterpret_cast<uint2*>(p) = v[0];
    *reinterpret_cast<uint2*>(p + plane_stride) = v[1];
    *reinterpret_cast<uint2*>(p + 2 * plane_stride) = v[2];
  };
  // four pre-split words -> 8 bytes of h and 8 bytes of l' (two byte permutes each)
  auto put_words = [&](char* plane0, int plane_stride, int row, const float4& v) {
    const unsigned w0 = __float_as_uint(v.x), w1 = __float_as_uint(v.y), w2 = __float_as_uint(v.z), w3 = __float_as_uint(v.w);
    char* p = plane0 + lds_wr<RB>(row, kq);
    *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u));
    *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u));
  };
  // split a float4 into its three bf16 planes and store 8 bytes into each
  auto put = [&](char* plane0, int plane_stride, int row, const float4& vin, float sc = 1.f, float* mx = nullptr) {
    if (NPL == 2) {
      // the row's power of two first (exact).  Nothing is clamped: `mx` keeps the largest SCALED magnitude this thread has split,
      // and a value beyond the f16 range (possible only under an estimated scale) makes the workgroup repeat the tile
      const float4 v = make_float4(vin.x * sc, vin.y * sc, vin.z * sc, vin.w * sc);
      if (mx) { *mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), *mx); *mx = fmaxf(fmaxf(fabsf(v.z), fabsf(v.w)), *mx); }
      unsigned h0, l0, h1, l1;
      split2h(v.x, v.y, h0, l0); split2h(v.z, v.w, h1, l1);
      char* p = plane0 + lds_wr<RB>(row, kq);
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(l0, l1);
      return;
    }
    const float4& v = vin;
    unsigned h[4], m[4], l[4];
    if (ABL == 3) {
      h[0] = __float_as_uint(v.x); h[1] = __float_as_uint(v.y); h[2] = __float_as_uint(v.z); h[3] = __float_as_uint(v.w);
      for (int q = 0; q < 4; q++) { m[q] = h[q] << 3; l[q] = h[q] << 7; }
    } else {
      split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    }
    char* p = plane0 + lds_wr<RB>(row, kq);
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
    *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
    *reinterpret_cast<uint2*>(p + 2 * plane_stride) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
  };
  auto sstore = [&](int slot, int st) {
    char* base = lds + st * Cfg::kStageBytes;
#pragma unroll
    for (int i = 0; i < NPA; i++)
      if (BMT % RPP == 0 || r0 + RPP * i < BMT) {
        if (GRAM) {
          const float za0 = gza[0], za1 = gza[1], za2 = gza[2];
          const float4 b0 = gzb[0], b1 = gzb[1], b2 = gzb[2];
          const float4 gv = make_float4(za0 * b0.x + za1 * b1.x + za2 * b2.x, za0 * b0.y + za1 * b1.y + za2 * b2.y,
                                        za0 * b0.z + za1 * b1.z + za2 * b2.z, za0 * b0.w + za1 * b1.w + za2 * b2.w);
          put(base, Cfg::kPlaneA, r0, gv);      // (scaled through gza; bounded by the row's ||Z'Z||_F)
        } else if (PLA) put_planes(base, Cfg::kPlaneA, r0 + RPP * i, pa[slot][PLA ? i : 0]);
        else if (AWD) put_words(base, Cfg::kPlaneA, r0 + RPP * i, ra[slot][i]);
        else put(base, Cfg::kPlaneA, r0 + RPP * i, ra[slot][i], asc[i], &amx[i]);
      }
#pragma unroll
    for (int i = 0; i < NPW; i++)
      if (BNT % RPP == 0 || r0 + RPP * i < BNT) {
        if (PLW) put_planes(base + NPL * Cfg::kPlaneA, Cfg::kPlaneW, r0 + RPP * i, pw[slot][PLW ? i : 0]);
        else if (WWD) put_words(base + NPL * Cfg::kPlaneA, Cfg::kPlaneW, r0 + RPP * i, rw[slot][i]);
        else put(base + NPL * Cfg::kPlaneA, Cfg::kPlaneW, r0 + RPP * i, rw[slot][i]);
      }
  };
  const int nk = a.K / BKT;
  // The two blocks a CU holds start together and run the same instruction sequence at the same pace: their load, LDS and
  // matrix phases coincide and each unit idles while the others work.  Blocks of the second half of a dispatch round (the
  // second block of each CU: 8 XCDs x 32 CUs = 256 blocks per half) start a fraction of a k-tile later.
  // (phase_sleep bits 16..: which blocks are delayed -- 0: the second half of a 512-block round, 1: every second block of an XCD)
  if ((a.phase_sleep & 0xFFFF) > 0 && (((a.phase_sleep >> 16) == 0 ? (blockIdx.x >> 8) : (blockIdx.x >> 3)) & 1))
    for (int q = 0; q < (a.phase_sleep & 0xFFFF); q++) __builtin_amdgcn_s_sleep(1);
  if (GRAM && (SCL || (a.rowdiv_out && tile_n == 0))) {
    // fn[m] = ||Z'Z||_F + 1 = ||Z Z'||_F + 1: six 32-term dot products of the row's three vectors, a quarter (eight columns)
    // per staging thread of the row, folded over the four adjacent lanes; written by the first column tile only.  The norm
    // bounds every entry of the row's Gram matrix: it is also what the two-piece form scales the generated operand by.
    float c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float4 x = *reinterpret_cast<const float4*>(arow_g[0] + 8 * kq + 4 * h);
      const float4 y = *reinterpret_cast<const float4*>(arow_g[0] + 32 + 8 * kq + 4 * h);
      const float4 z = *reinterpret_cast<const float4*>(arow_g[0] + 64 + 8 * kq + 4 * h);
      c[0] += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      c[1] += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
      c[2] += z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w;
      c[3] += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      c[4] += x.x * z.x + x.y * z.y + x.z * z.z + x.w * z.w;
      c[5] += y.x * z.x + y.y * z.y + y.z * z.z + y.w * z.w;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {             // sum over the quad of lanes: DPP quad_perm [1,0,3,2] then [2,3,0,1]
      c[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[k]), 0xB1, 0xF, 0xF, true));
      c[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[k]), 0x4E, 0xF, 0xF, true));
    }
    const float frob = sqrtf((c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + 2.f * (c[3] * c[3] + c[4] * c[4] + c[5] * c[5]));
    if (kq == 0 && m0 + r0 < a.M && a.rowdiv_out && tile_n == 0) a.rowdiv_out[m0 + r0] = frob + 1.0f;
    if (SCL) {
      asc[0] = pow2_scale(frob, kScaleExact);
      if (kq == 0) rs_sh[r0] = pow2_inv(asc[0]);
    }
  }
  if (SCL && t == 0) redo_sh = 0;
  const int li = lane & 31, lh = lane >> 5;
  const int aoff = lds_rd<RB>(wm * 32 * TM + li, lh);        // this lane's 8 bf16 of k-step 0; k-step 1 (32-wide k-tiles) is 32 bytes on
  const int boff = NPL * Cfg::kPlaneA + lds_rd<RB>(wn * 32 * TN + li, lh);
  const bool late = SKEW ? (wave >= 4) : LATE;                   // wave-uniform
  if (AWD && kq == 0) {
#pragma unroll
    for (int i = 0; i < NPA; i++)
      if (BMT % RPP == 0 || r0 + RPP * i < BMT) rs_sh[r0 + RPP * i] = a.ascale[min(m0 + r0 + RPP * i, a.M - 1)];
  }
  constexpr bool EST = SCL && !GRAM && !PLA && !AWD;  // loaded f32 activation rows: scaled by an estimate, repeated with the exact maxima if it fell short
  auto body = [&](int kt, int slot) {
    const int st = kt & 1;
    if (!late && ABL != 1) {
      if (kt + 1 < nk) sstore(slot, st ^ 1);
      if (GRAM && kt + 2 < nk) gload_gram();
      if (kt + 1 + PF < nk) gload(slot, (kt + 1 + PF) * BKT);
    }
    const char* base = lds + (ABL == 1 ? 0 : st) * Cfg::kStageBytes;
    if (ABL != 2)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      bf16x8 av[TM][3], bv[TN][3];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
          av[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + pl * Cfg::kPlaneA + aoff + 32 * i * RB + 32 * ks));
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
          bv[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + pl * Cfg::kPlaneW + boff + 32 * j * RB + 32 * ks));
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          if (NPL == 2) {
            const f16x8 ah = __builtin_bit_cast(f16x8, av[i][0]), al = __builtin_bit_cast(f16x8, av[i][1]);
            const f16x8 bh = __builtin_bit_cast(f16x8, bv[j][0]), bl = __builtin_bit_cast(f16x8, bv[j][1]);
            if (FLAGS & (EPI_EQUIV | EPI_TR)) {          // transposed tile, as below
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[i][j], 0, 0, 0);
              cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, cor[i][j], 0, 0, 0);
              cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, cor[i][j], 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);   // hh
              cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cor[i][j], 0, 0, 0);   // l'h
              cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cor[i][j], 0, 0, 0);   // hl'
            }
          } else if (FLAGS & (EPI_EQUIV | EPI_TR)) {
            // transposed tile (W rows on the accumulator rows = registers, nodes on the lanes): the epilogue's contraction over
            // the W-row index then runs over REGISTERS of a lane instead of across lanes
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][0], av[i][0], acc[i][j], 0, 0, 0);
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][0], av[i][2], cor[i][j], 0, 0, 0);
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][2], av[i][0], cor[i][j], 0, 0, 0);
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][1], av[i][1], cor[i][j], 0, 0, 0);
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][0], av[i][1], cor[i][j], 0, 0, 0);
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][1], av[i][0], cor[i][j], 0, 0, 0);
          } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[j][0], acc[i][j], 0, 0, 0);   // hh
          cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][2], bv[j][0], cor[i][j], 0, 0, 0);   // lh
          cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[j][2], cor[i][j], 0, 0, 0);   // hl
          cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][1], bv[j][1], cor[i][j], 0, 0, 0);   // mm
          cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][1], bv[j][0], cor[i][j], 0, 0, 0);   // mh
          cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[j][1], cor[i][j], 0, 0, 0);   // hm
          }
        }
    }
    if (late) {
      if (kt + 1 < nk) sstore(slot, st ^ 1);
      if (GRAM && kt + 2 < nk) gload_gram();
      if (kt + 1 + PF < nk) gload(slot, (kt + 1 + PF) * BKT);
    }
    if (EST && kt == nk - 1) {                // every tile has been split by now: the vote on repeating rides on the last barrier
      bool over = false;
#pragma unroll
      for (int i = 0; i < NPA; i++) over = over || !(amx[i] <= kF16Lim);
      if (over) redo_sh = 1u;
    }
    __syncthreads();
  };
  for (int attempt = 0;; attempt++) {
    if (GRAM) { ga = 0; gb = 0; gload_gram(); }
    // two more samples of each row, in flight with tile 0: the first four values of its middle k-tile (staging thread 0 of the row)
    // and of its last one (thread 1) -- 32 bytes per row, not two more tiles: every workgroup of a dispatch round starts at once
    float4 smp[NPA];
    if (EST && attempt == 0) {
#pragma unroll
      for (int i = 0; i < NPA; i++)
        smp[i] = kq < 2 ? *reinterpret_cast<const float4*>(arow_g[i] - 4 * kq + (kq == 0 ? (nk >> 1) : nk - 1) * BKT) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    gload(0, 0);
    if (EST && attempt == 0) {
#pragma unroll
      for (int i = 0; i < NPA; i++) {
        const float4 v0 = ra[0][i], v1 = smp[i];
        float est = fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w)));
        est = fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), est); est = fmaxf(fmaxf(fabsf(v1.z), fabsf(v1.w)), est);
        asc[i] = pow2_scale(quad_max(est), kScaleEstimate);
      }
    }
    if (EST && kq == 0) {
#pragma unroll
      for (int i = 0; i < NPA; i++) if (BMT % RPP == 0 || r0 + RPP * i < BMT) rs_sh[r0 + RPP * i] = pow2_inv(asc[i]);
    }
    sstore(0, 0);
    __syncthreads();
    if (GRAM && nk > 1) gload_gram();
    if (nk > 1) gload(0, BKT);
    if (PF == 2 && nk > 2) gload(1, 2 * BKT);
    if (PF == 1) {
      for (int kt = 0; kt < nk; kt++) body(kt, 0);
    } else {
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) { body(kt, 0); body(kt + 1, 1); }
      if (kt < nk) body(kt, 0);
    }
    if (!EST || attempt == 1) break;
    if (redo_sh == 0u) break;                 // (block-uniform) the usual exit: no scaled value left the f16 range
    if (t == 0) atomicAdd(&g_scale_redos, 1u);
    // the rare path: exact maxima of the rows this thread stages (its quarter of every k-tile, read again), then once more
#pragma unroll
    for (int i = 0; i < NPA; i++) {
      float tm = 0.f;
      for (int kt = 0; kt < nk; kt++) {
        const float4 x = *reinterpret_cast<const float4*>(arow_g[i] + kt * BKT);
        tm = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), tm); tm = fmaxf(fmaxf(fabsf(x.z), fabsf(x.w)), tm);
      }
      asc[i] = pow2_scale(quad_max(tm), kScaleExact);
    }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) { acc[i][j][e] = 0.f; cor[i][j][e] = 0.f; }
  }
  if (FLAGS & EPI_EQUIV) {
    // Equivariant epilogue.  The accumulators hold TRANSPOSED 32 x 32 tiles: lane = node (column li of the tile), registers
    // = W rows q (row (e & 3) + 8 (e >> 2) + 4 lh); one tile = mat[node][q][c] for ONE c (output columns ordered c * 32 + q).
    // tout[node][s][c] = rdiv[node] * sum_q z[node][s][q] * (acc + bias)[q]: 16 register FMAs per lane and s, one exchange
    // between the two half-waves.  The z rows of the block's 128 nodes are staged in the (now idle) LDS, row stride 100
    // floats: 16-byte aligned and conflict-free for ds_read_b128 across nodes.
    static_assert(!(FLAGS & EPI_EQUIV) || TM == 1, "the equivariant epilogue assumes one 32-node row tile per wave");
    float* zs = gemm_lds;
    constexpr int ZS = 100;
    for (int idx = t; idx < BMT * 24; idx += T) {
      const int r = idx / 24, q4 = idx % 24;
      const int m = min(m0 + r, a.M - 1);
      *reinterpret_cast<float4*>(zs + r * ZS + 4 * q4) = *reinterpret_cast<const float4*>(a.zq + (size_t)m * 96 + 4 * q4);
    }
    __syncthreads();
    const int mloc = wm * 32 + li, m = m0 + mloc;
    const bool ok = m < a.M;
    const float rd = (FLAGS & EPI_ROWDIV) ? 1.0f / (ok ? a.rowdiv[m] : 1.f) : 1.f;
    const float rsn = SCL ? rs_sh[mloc] : 1.f;
#pragma unroll
    for (int tj = 0; tj < TN; tj++) {
      const int cidx = (n0 + wn * 32 * TN + tj * 32) >> 5;
      const float un = (SCL && a.wscale) ? rsn * a.wscale[cidx * 32] : rsn;
      float ts[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int q0 = 8 * g4 + 4 * lh;
        const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + cidx * 32 + q0) : make_float4(0, 0, 0, 0);
        // undo the row scales of A (this lane's node) and W (ONE scale per 32-row block c: EncMat::group -- a scalar, not a vector)
        const float v0 = fin(acc[0][tj][4 * g4 + 0], cor[0][tj][4 * g4 + 0]) * un + b4.x;
        const float v1 = fin(acc[0][tj][4 * g4 + 1], cor[0][tj][4 * g4 + 1]) * un + b4.y;
        const float v2 = fin(acc[0][tj][4 * g4 + 2], cor[0][tj][4 * g4 + 2]) * un + b4.z;
        const float v3 = fin(acc[0][tj][4 * g4 + 3], cor[0][tj][4 * g4 + 3]) * un + b4.w;
#pragma unroll
        for (int sx = 0; sx < 3; sx++) {
          const float4 z4 = *reinterpret_cast<const float4*>(zs + mloc * ZS + sx * 32 + q0);
          ts[sx] += z4.x * v0 + z4.y * v1 + z4.z * v2 + z4.w * v3;
        }
      }
#pragma unroll
      for (int sx = 0; sx < 3; sx++) {
        float tv = ts[sx] * rd;
        tv += __shfl_xor(tv, 32, 64);
        if (ok && lh == 0) a.tout[(size_t)m * 96 + sx * 32 + cidx] = tv;
      }
    }
    return;
  }
  if (FLAGS & EPI_TR) {
    // Row-wise stores of transposed tiles: lane = row, registers = columns 8 g + 4 lh + (0..3) of the tile -- one float4 per g
    static_assert(!(FLAGS & EPI_TR) || (TM == 1 && !(FLAGS & (EPI_ACC2 | EPI_ZSPLIT | EPI_LN | EPI_EQUIV))), "EPI_TR: plain / ReLU / row-division epilogues");
    const int m = m0 + wm * 32 + li;
    if (m < a.M) {
      const float rd = (FLAGS & EPI_ROWDIV) ? 1.0f / a.rowdiv[m] : 1.f;
      const float rsn = SCL ? rs_sh[wm * 32 + li] : 1.f;
#pragma unroll
      for (int tj = 0; tj < TN; tj++) {
        const int nb = n0 + wn * 32 * TN + tj * 32 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int n = nb + 8 * g;
          if (n + 3 >= a.N) continue;                  // (N is a multiple of 4 for every product of the forward)
          const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0, 0, 0, 0);
          float4 u4 = make_float4(rsn, rsn, rsn, rsn);
          if (SCL && a.wscale) { const float4 w4 = *reinterpret_cast<const float4*>(a.wscale + n); u4 = make_float4(rsn * w4.x, rsn * w4.y, rsn * w4.z, rsn * w4.w); }
          float4 v = make_float4(fin(acc[0][tj][4 * g + 0], cor[0][tj][4 * g + 0]) * u4.x + b4.x, fin(acc[0][tj][4 * g + 1], cor[0][tj][4 * g + 1]) * u4.y + b4.y,
                                 fin(acc[0][tj][4 * g + 2], cor[0][tj][4 * g + 2]) * u4.z + b4.z, fin(acc[0][tj][4 * g + 3], cor[0][tj][4 * g + 3]) * u4.w + b4.w);
          if (FLAGS & EPI_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
          if (FLAGS & EPI_ROWDIV) v = make_float4(v.x * rd, v.y * rd, v.z * rd, v.w * rd);
          *reinterpret_cast<float4*>(a.C + (size_t)m * a.ldc + n) = v;
        }
      }
    }
    return;
  }
  if (FLAGS & EPI_LN) {
    // Residual + LayerNorm epilogue (N = 128: the block holds whole rows).  A row's 128 values sit in two waves (wn) x two
    // 32-column tiles x 32 lanes: sums run over the lanes of a half-wave (xor shuffles), the two tiles (registers) and the
    // two waves (through the now idle LDS); mean first, then the centred second moment, as k_add_ln does.
    static_assert(!(FLAGS & EPI_LN) || (TM == 1 && TN == 2 && WN == 2 && BNT == 128), "EPI_LN assumes the 128 x 128 tile of 4 x 2 waves");
    float* red = gemm_lds;                       // [128 rows][2 waves]
    const int mb = m0 + wm * 32 + 4 * lh;
    float part[16];                              // the accumulators are reused for the row values (acc[0][tj][e])
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
      const float bvv = a.bias ? a.bias[wn * 64 + tj * 32 + li] : 0.f;
      const float wsn = (SCL && a.wscale) ? a.wscale[wn * 64 + tj * 32 + li] : 1.f;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        float v = fin(acc[0][tj][e], cor[0][tj][e]);
        if (SCL) v *= rs_sh[wm * 32 + 4 * lh + (e & 3) + 8 * (e >> 2)] * wsn;
        v += bvv;
        if (FLAGS & EPI_RELU) v = fmaxf(v, 0.f);
        acc[0][tj][e] = v;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int m = min(mb + (e & 3) + 8 * (e >> 2), a.M - 1);
      const float rd = (FLAGS & EPI_ROWDIV) ? 1.0f / a.rowdiv[m] : 1.f;
      const float* rrow = a.ln_io + (size_t)m * a.ln_ld + wn * 64 + li;
      acc[0][0][e] = rrow[0] + acc[0][0][e] * rd;
      acc[0][1][e] = rrow[32] + acc[0][1][e] * rd;
      part[e] = acc[0][0][e] + acc[0][1][e];
    }
    auto row_sums = [&]() {                      // part[e] -> sum over the row's 128 columns, in every lane
#pragma unroll
      for (int e = 0; e < 16; e++) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) part[e] += __shfl_xor(part[e], off, 32);
      }
      __syncthreads();                           // LDS free (first use: the k-loop's last reads; second: the previous sums)
      if (li == 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) red[(wm * 32 + 4 * lh + (e & 3) + 8 * (e >> 2)) * 2 + wn] = part[e];
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int r = wm * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
        part[e] = red[2 * r] + red[2 * r + 1];
      }
    };
    row_sums();
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const float mu = part[e] * (1.f / 128.f);
      const float d0 = acc[0][0][e] - mu, d1 = acc[0][1][e] - mu;
      acc[0][0][e] = d0; acc[0][1][e] = d1;
      part[e] = d0 * d0 + d1 * d1;
    }
    row_sums();
#pragma unroll
    for (int e = 0; e < 16; e++) part[e] = 1.0f / sqrtf(part[e] * (1.f / 128.f) + 1e-5f);
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
      const int n = wn * 64 + tj * 32 + li;
      const float lw = a.ln_w[n], lb = a.ln_b[n];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m < a.M) {
          const float y = acc[0][tj][e] * part[e] * lw + lb;
          a.ln_io[(size_t)m * a.ln_ld + n] = y;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int ti = 0; ti < TM; ti++) {
    const int mb = m0 + wm * 32 * TM + ti * 32 + 4 * lh;
    float rdiv[16], rsc[16];
    if (FLAGS & EPI_ROWDIV) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        rdiv[e] = 1.0f / ((m < a.M) ? a.rowdiv[m] : 1.f);
      }
    }
    if (SCL) {
#pragma unroll
      for (int e = 0; e < 16; e++) rsc[e] = rs_sh[wm * 32 * TM + ti * 32 + 4 * lh + (e & 3) + 8 * (e >> 2)];
    }
#pragma unroll
    for (int tj = 0; tj < TN; tj++) {
      const int n = n0 + wn * 32 * TN + tj * 32 + li;
      if (n >= a.N) continue;
      const float bvv = a.bias ? a.bias[n] : 0.f;
      const float wsn = (SCL && a.wscale) ? a.wscale[n] : 1.f;
      float old2[16];
      if (FLAGS & EPI_ACC2) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int m = mb + (e & 3) + 8 * (e >> 2);
          old2[e] = (m < a.M) ? a.C2[(size_t)m * a.ldc2 + n] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m >= a.M) continue;
        float v = fin(acc[ti][tj][e], cor[ti][tj][e]);
        if (SCL) v *= rsc[e] * wsn;
        v += bvv;
        if (FLAGS & EPI_RELU) v = fmaxf(v, 0.f);
        if (FLAGS & EPI_ROWDIV) v = v * rdiv[e];
        if (FLAGS & EPI_ZSPLIT) {               // stacked projections: see the enum's comment
          if (n < 30) a.C[(size_t)m * 32 + n] = v;
          else if (n >= 32 && n < 62) a.C2[(size_t)m * 32 + (n - 32)] = v;
          continue;
        }
        a.C[(size_t)m * a.ldc + n] = v;
        if (FLAGS & EPI_ACC2) a.C2[(size_t)m * a.ldc2 + n] = old2[e] + v;
      }
    }
  }
}

// Weights of the two-piece form: every row of every product matrix of the flat weight buffer, scaled by the power of two that
// brings its largest magnitude to 2^14 .. 2^15 and cut into words (once per forward, behind the packer).  One wave per row;
// wsc[first_row + r] receives the INVERSE scale of row r (what the consumer's epilogue multiplies by).
// group = 32: the rows of every aligned block of 32 share one scale (the 32 x 32 matrix blocks of linear4 / linear2_m, whose
// equivariant epilogue then undoes ONE scale per block); group = 1: a scale per row
struct EncMat { long long off; int rows, K, first_row, group; };   // K % 4 == 0, K <= 1024; rows of the matrix contiguous in `w`
__global__ __launch_bounds__(256) void k_encode_rows(const float* __restrict__ w, unsigned* __restrict__ ww, float* __restrict__ wsc,
                                                     const EncMat* __restrict__ mats, int n_mats, int total_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= total_rows) return;
  int mi = 0;
  while (mi + 1 < n_mats && mats[mi + 1].first_row <= row) mi++;          // (wave-uniform; a few dozen matrices)
  const EncMat mt = mats[mi];
  const long long base = mt.off + (long long)(row - mt.first_row) * mt.K;
  const int nq = mt.K >> 2;
  float4 v[4];
  float mx = 0.f;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = lane + 64 * q;
    v[q] = i < nq ? *reinterpret_cast<const float4*>(w + base + 4 * i) : make_float4(0, 0, 0, 0);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (mt.group == 32) {                       // the maximum over the block of 32 rows (8 workgroups of 4 rows, all of this matrix)
    const float* blk = w + mt.off + (long long)((row - mt.first_row) & ~31) * mt.K;
    float bm = 0.f;
    for (int i = lane; i < 8 * mt.K; i += 64) {
      const float4 x = *reinterpret_cast<const float4*>(blk + 4 * i);
      bm = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), bm); bm = fmaxf(fmaxf(fabsf(x.z), fabsf(x.w)), bm);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o, 64));
    mx = bm;
  }
  const float sc = pow2_scale(mx, kScaleExact);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = lane + 64 * q;
    if (i < nq) *reinterpret_cast<uint4*>(ww + base + 4 * i) = make_uint4(enc_word(v[q].x * sc), enc_word(v[q].y * sc), enc_word(v[q].z * sc), enc_word(v[q].w * sc));
  }
  if (lane == 0) wsc[row] = pow2_inv(sc);
}

// ------------------------------------------------------------------------------------------------------------------------
// Wave-specialised form of the split-precision GEMM: 16 waves per block, 8 CONSUMERS (4 x 2, a 32 x 64 patch each: LDS
// operand reads + the six bf16 MFMAs per product block) and 8 PRODUCERS (global loads PF k-tiles ahead, the exact 3-way bf16
// split, LDS stores).  One barrier per k-tile: while the consumers run tile kt out of LDS stage kt & 1, the producers fill
// stage (kt + 1) & 1 -- the matrix pipe of a SIMD (two consumer waves) never waits on a global load or on split arithmetic,
// which sit in the two producer waves sharing that SIMD (VALU and MFMA issue from different waves overlap).
template <int FLAGS, int BKT = 16, int PF = 3>
__global__ __launch_bounds__(1024) void k_gemm4(GemmArgs a) {
  using Cfg = TileCfg3<4, 2, 1, 2, BKT>;        // same 128 x 128 tile / LDS image as the 8-wave k_gemm3
  constexpr int BMT = 128, BNT = 128, RB = Cfg::kRowBytes, TN = 2;
  constexpr int QPR = BKT / 4, RPP = 512 / QPR, NP = 128 / RPP;   // producer staging: NP float4 of A and of W per thread
  constexpr int KS = BKT / 16;
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];     // (aligned: static LDS precedes it)
  char* lds = reinterpret_cast<char*>(gemm_lds);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool producer = wave >= 8;              // wave-uniform
  const int tiles_n = (a.N + BNT - 1) / BNT;
  int bid;
  {
    const int nt = gridDim.x, per = nt >> 3, rem = nt & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    bid = (x < rem) ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BMT, n0 = tile_n * BNT;
  const int nk = a.K / BKT;
  // ---- producer state
  const int pt = t & 511, kq = pt % QPR, r0 = pt / QPR;
  const float* arow_g[NP];
  const float* wrow_g[NP];
  float4 ra[PF][NP], rw[PF][NP];
#pragma unroll
  for (int i = 0; i < NP; i++) {
    arow_g[i] = a.A + (size_t)min(m0 + r0 + RPP * i, a.M - 1) * a.lda + 4 * kq;      // clamped rows: never stored
    wrow_g[i] = a.W + (size_t)min(n0 + r0 + RPP * i, a.N - 1) * a.ldw + 4 * kq;
  }
  auto gload = [&](int slot, int k0) {
#pragma unroll
    for (int i = 0; i < NP; i++) { ra[slot][i] = *reinterpret_cast<const float4*>(arow_g[i] + k0); rw[slot][i] = *reinterpret_cast<const float4*>(wrow_g[i] + k0); }
  };
  auto put = [&](char* plane0, int plane_stride, int row, const float4& v) {
    unsigned h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    char* p = plane0 + lds_wr<RB>(row, kq);
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
    *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
    *reinterpret_cast<uint2*>(p + 2 * plane_stride) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
  };
  auto sstore = [&](int slot, int st) {
    char* base = lds + st * Cfg::kStageBytes;
#pragma unroll
    for (int i = 0; i < NP; i++) { put(base, Cfg::kPlaneA, r0 + RPP * i, ra[slot][i]); put(base + 3 * Cfg::kPlaneA, Cfg::kPlaneW, r0 + RPP * i, rw[slot][i]); }
  };
  // ---- consumer state
  const int cw = wave & 7, wm = cw >> 1, wn = cw & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int aoff = lds_rd<RB>(wm * 32 + li, lh);
  const int boff = 3 * Cfg::kPlaneA + lds_rd<RB>(wn * 64 + li, lh);
  f32x16 acc[TN], cor[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) { acc[j][e] = 0.f; cor[j][e] = 0.f; }

  if (producer) {
    gload(0, 0);
    sstore(0, 0);
#pragma unroll
    for (int q = 0; q < PF; q++) if (1 + q < nk) gload(q, (1 + q) * BKT);      // slot of tile k = (k - 1) % PF
  }
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
    for (int q = 0; q < PF; q++) {
      const int kt = kt0 + q;
      if (kt < nk) {                         // uniform over the block
        const int st = kt & 1;
        if (producer) {
          if (kt + 1 < nk) sstore(q, st ^ 1);                        // tile kt + 1 sits in slot q
          if (kt + 1 + PF < nk) gload(q, (kt + 1 + PF) * BKT);
        } else {
          const char* base = lds + st * Cfg::kStageBytes;
#pragma unroll
          for (int ks = 0; ks < KS; ks++) {
            bf16x8 av[3], bv[TN][3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) av[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + pl * Cfg::kPlaneA + aoff + 32 * ks));
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
              for (int pl = 0; pl < 3; pl++)
                bv[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + pl * Cfg::kPlaneW + boff + 32 * j * RB + 32 * ks));
#pragma unroll
            for (int j = 0; j < TN; j++) {
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[j][0], acc[j], 0, 0, 0);   // hh
              cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bv[j][0], cor[j], 0, 0, 0);   // lh
              cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[j][2], cor[j], 0, 0, 0);   // hl
              cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[j][1], cor[j], 0, 0, 0);   // mm
              cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[j][0], cor[j], 0, 0, 0);   // mh
              cor[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[j][1], cor[j], 0, 0, 0);   // hm
            }
          }
        }
        __syncthreads();
      }
    }
  }
  if (producer) return;
  const int mb = m0 + wm * 32 + 4 * lh;
  float rdiv[16];
  if (FLAGS & EPI_ROWDIV) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int m = mb + (e & 3) + 8 * (e >> 2);
      rdiv[e] = 1.0f / ((m < a.M) ? a.rowdiv[m] : 1.f);
    }
  }
#pragma unroll
  for (int tj = 0; tj < TN; tj++) {
    const int n = n0 + wn * 64 + tj * 32 + li;
    if (n >= a.N) continue;
    const float bvv = a.bias ? a.bias[n] : 0.f;
    float old2[16];
    if (FLAGS & EPI_ACC2) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        old2[e] = (m < a.M) ? a.C2[(size_t)m * a.ldc2 + n] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int m = mb + (e & 3) + 8 * (e >> 2);
      if (m >= a.M) continue;
      float v = (acc[tj][e] + cor[tj][e]) + bvv;
      if (FLAGS & EPI_RELU) v = fmaxf(v, 0.f);
      if (FLAGS & EPI_ROWDIV) v = v * rdiv[e];
      a.C[(size_t)m * a.ldc + n] = v;
      if (FLAGS & EPI_ACC2) a.C2[(size_t)m * a.ldc2 + n] = old2[e] + v;
    }
  }
}

// f32 [rows][ld_src] (cols used: cols) -> three bf16 planes [rows][ld_dst] (`plane` elements apart); columns cols..ld_dst-1 zero
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ src, int ld_src, int rows, int cols,
                                                      unsigned short* dst, int ld_dst, long long plane) {
  const size_t n = (size_t)rows * ld_dst;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / ld_dst), c = (int)(i % ld_dst);
    unsigned h = 0, m = 0, l = 0;
    if (c < cols) split3(src[(size_t)r * ld_src + c], h, m, l);
    dst[i] = (unsigned short)(h >> 16);
    dst[plane + i] = (unsigned short)(m >> 16);
    dst[2 * plane + i] = (unsigned short)(l >> 16);
  }
}

}  // namespace sgrl_gemm
