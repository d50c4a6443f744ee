// set_actor.hip -- SET actor forward for gfx950 (C ABI: include/sgrl_set.h).
//
// All per-node linear layers are batched over the nodes of every morphology (weights are shared) and run as float32
// products on the matrix cores: batches above 2 048 nodes through the split kernels of gemm_f32.h (every operand cut into
// two f16 pieces -- or three bf16 pieces -- whose partial products are exact in f32: float32's error against float64, so
// results track the reference's f32 PyTorch arithmetic), smaller ones through the exact-f32 32 x 32 tile kernels of
// train_gemm.hip; the per-limb 3x32 Gram invariants, the 3x32 . 32x32 equivariant updates, the per-environment
// limb attention (<= 14 keys) and layer norms are wave-reduced VALU kernels.
//
// Node order: environments in batch order, limbs of one environment contiguous.  Buffers (float32, N = nodes):
//   g    [N,3,128]  equivariant stream           cat  [N,256] = [invariants | ng]  (ng lives in cat[:,128:])
//   zc   [N,3,32]   Z (Gram operand Z'Z is generated inside the GEMMs)   fn   [N]   ||Z'Z||_F + 1
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/sgrl.h"
#include "../../include/sgrl_set.h"
#include "../../include/sgrl_train.h"
#include "gemm_f32.h"
#include "chain_f16.h"
#include "stream_pick.h"

namespace {

thread_local std::string g_set_err;
int sfail(int code, const std::string& msg) { g_set_err = msg; return code; }
#define SHIP_TRY(expr)                                                                     \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return sfail(SGRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

constexpr int D = 128;
constexpr int ZD = 32;
// The 32x32 Gram matrix Z'Z is symmetric and is never materialised: the GEMMs that consume it generate their A operand
// from Z on the fly (gemm_f32.h, GRAM), running over the 36 4x4 blocks of the lower triangle (GK = 576 k values, block
// (A, B), B <= A, at k = 16 (A (A + 1) / 2 + B) + 4 (a - 4 A) + (b - 4 B)); the 1024-wide weight rows of those layers are
// folded onto the same order at pack time (k_pack FOLD / sgrl_amd/set_hip.py fold_gram_weight).
constexpr int GK = sgrl_gemm::kGramK;     // 576
constexpr int OGLD = 144;                  // row stride of outg: 136 channels padded to the GEMM K tile

// The f32 MFMA GEMM (C[M,N] = epi(A[M,K] . W[N,K]^T)) lives in gemm_f32.h.
using sgrl_gemm::EPI_ACC2;
using sgrl_gemm::EPI_EQUIV;
using sgrl_gemm::EPI_LN;
using sgrl_gemm::EPI_RELU;
using sgrl_gemm::EPI_ROWDIV;
using sgrl_gemm::EPI_ZSPLIT;
using sgrl_gemm::GemmArgs;
using sgrl_gemm::k_gemm2;

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f32(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// embed: one 128-thread block per kEmbedNodes nodes (thread = channel): a thread keeps its rows of the two encoder weights
// (8 + ngf <= 28 values, gathered at a stride of ngf floats) in registers and walks the block's nodes, whose input rows are
// staged in LDS together -- one gather of the weights per eight nodes instead of one per node (52 -> 42 us at 35 840 nodes)
struct NodeTab {
  const int32_t* node_env;    // [N]
  const int32_t* node_limb;   // [N]
  const int32_t* node_mnode;  // [N] index into the per-morphology node tables (trav)
  const int32_t* trav;        // [3][TM]
  int TM;
};
constexpr int kEmbedNodes = 8;
__global__ __launch_bounds__(128) void k_embed(const float* __restrict__ obs, int obs_ld, const float* __restrict__ action,
                                               int act_ld, int ngf, NodeTab nt, const float* Wge, const float* We,
                                               const float* be, const float* e0, const float* e1, const float* e2, float* g,
                                               float* cat, float* outg, float* outng, float* gdir, float* zc, float* z2, int N) {
  // per-limb input row: 24 geometric values (8 three-vectors) | ngf non-geometric ones -- 17 from the observation and,
  // for the critic (ngf = 20), the limb's 3 action slots appended (reference SECritic.py:80-83)
  __shared__ float os[kEmbedNodes][44];
  const int n0 = blockIdx.x * kEmbedNodes, c = threadIdx.x;
  for (int idx = c; idx < kEmbedNodes * 44; idx += 128) {
    const int q = idx / 44, k = idx % 44, n = n0 + q;
    float v = 0.f;
    if (n < N) {
      const int env = nt.node_env[n], limb = nt.node_limb[n];
      if (k < 41) v = obs[(size_t)env * obs_ld + 41 * limb + k];
      else if (action) v = action[(size_t)env * act_ld + 3 * limb + (k - 41)];
    }
    os[q][k] = v;
  }
  float wg[8], we[20];
#pragma unroll
  for (int j = 0; j < 8; j++) wg[j] = Wge[c * 8 + j];
#pragma unroll
  for (int j = 0; j < 20; j++) we[j] = j < ngf ? We[c * ngf + j] : 0.f;
  const float bias = be[c];
  __syncthreads();
  const float sc = sqrtf(128.f);
  for (int q = 0; q < kEmbedNodes; q++) {
    const int n = n0 + q;
    if (n >= N) break;
    const float* o = os[q];
    const int mn = nt.node_mnode[n];
    for (int s = 0; s < 3; s++) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) v += wg[j] * o[3 * j + s];
      g[((size_t)n * 3 + s) * D + c] = v * sc;
    }
    float v = bias;
#pragma unroll
    for (int j = 0; j < 20; j++) if (j < ngf) v += we[j] * o[24 + j];
    float pos;
    if (c < 42) pos = e0[nt.trav[mn] * 42 + c];
    else if (c < 84) pos = e1[nt.trav[nt.TM + mn] * 42 + (c - 42)];
    else pos = e2[nt.trav[2 * nt.TM + mn] * 44 + (c - 84)];
    cat[(size_t)n * 256 + 128 + c] = v * sc + pos;
    if (c < 24) { const int s = c / 8, j = c % 8; outg[((size_t)n * 3 + s) * OGLD + j] = o[3 * j + s]; }
    if (c < ngf) outng[(size_t)n * 160 + c] = o[24 + c];
    if (c >= ngf && c < 32) outng[(size_t)n * 160 + 128 + c] = 0.f;  // cols 128 + ngf .. 159
    // the gravity / direction pair of the node: for the attention kernel (gdir) and as columns 30 / 31 of the projected
    // vectors zc / z2 (written here once per forward; the projection GEMMs fill columns 0..29 of those rows, EPI_ZSPLIT)
    if (c < 6) {
      const int s = c / 2, e = c % 2;
      const float gv = o[3 * (1 + e) + s];
      gdir[((size_t)n * 3 + s) * 2 + e] = gv;
      zc[((size_t)n * 3 + s) * ZD + 30 + e] = gv;
      z2[((size_t)n * 3 + s) * ZD + 30 + e] = gv;
    }
  }
}

// The two 30-row projections of a site stacked into one zero-padded GEMM operand out[64][Cpad]:
// rows 0..29 = Wp, rows 32..61 = Wq (if any), columns C..Cpad-1 = 0.
__global__ void k_stack_proj(const float* __restrict__ Wp, const float* __restrict__ Wq, int C, int Cpad, float* out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * Cpad; i += gridDim.x * blockDim.x) {
    const int r = i / Cpad, c = i % Cpad;
    float v = 0.f;
    if (c < C) {
      if (r < 30) v = Wp[r * C + c];
      else if (Wq && r >= 32 && r < 62) v = Wq[(r - 32) * C + c];
    }
    out[i] = v;
  }
}

// Live weight pack (sgrl_set_bind_params): every forward rebuilds the flat weight buffer from the parameters' OWN
// storage, so in-place updates of any kind (optimizer steps, `target.data.copy_(...)` soft updates, load_state_dict)
// are seen by the very next forward -- there is no host-side change detection to go stale.  ~46 MB of traffic (27 MB
// read, 19 MB written): a few tens of microseconds per forward.  One block handles one PACK_CHUNK-element run of one segment.
constexpr int PACK_CHUNK = 4096;
constexpr int PACK_CHUNK_MM = 256;      // matrix-product segments: one output (a dot product) per thread
__global__ __launch_bounds__(256) void k_pack(const sgrl_pack_seg* __restrict__ segs, const int2* __restrict__ chunks,
                                              const unsigned short* __restrict__ tri, float* w) {
  const int2 ch = chunks[blockIdx.x];
  const sgrl_pack_seg sg = segs[ch.x];
  const float* s0 = static_cast<const float*>(sg.src0);
  const float* s1 = static_cast<const float*>(sg.src1);
  const int end = min(ch.y + (sg.kind == SGRL_PACK_MATMUL ? PACK_CHUNK_MM : PACK_CHUNK), sg.n);
  for (int i = ch.y + (int)threadIdx.x; i < end; i += 256) {
    float v = 0.f;
    switch (sg.kind) {
      case SGRL_PACK_COPY:      // first `a` elements from src0 (scaled), zeros after (row padding)
        if (i < sg.a) v = s0[i] * sg.scale;
        break;
      case SGRL_PACK_PADCOL: {  // [rows, a] -> [rows, b], zero columns appended
        const int r = i / sg.b, c = i % sg.b;
        if (c < sg.a) v = s0[(size_t)r * sg.a + c];
        break;
      }
      case SGRL_PACK_FOLD: {    // [rows, 1024] acting on vec(G), G symmetric 32x32 -> [rows, 576] on the blocked lower triangle
        const int r = i / GK, k = i % GK;
        const unsigned short t = tri[k];
        if (t != 0xFFFF) {
          const int aa = t >> 8, bb = t & 255;
          const float* row = s0 + (size_t)r * (ZD * ZD);
          v = (aa == bb) ? row[aa * ZD + bb] : row[aa * ZD + bb] + row[bb * ZD + aa];
        }
        break;
      }
      case SGRL_PACK_STACK: {   // out[64][b]: rows 0..29 = src0 [30, a], rows 32..61 = src1 [30, a] (if any), rest zero
        const int r = i / sg.b, c = i % sg.b;
        if (c < sg.a) {
          if (r < 30) v = s0[r * sg.a + c];
          else if (s1 && r >= 32 && r < 62) v = s1[(r - 32) * sg.a + c];
        }
        break;
      }
      case SGRL_PACK_MATMUL: {  // dst [rows, b] = src0 [rows, a] (stride lda) . src1 [a, b] (stride ldb): weight folds
        const int r = i / sg.b, c = i % sg.b;
        const float* ar = s0 + (size_t)r * sg.lda;
        float acc = 0.f;
        int k = 0;
        for (; k + 8 <= sg.a; k += 8) {          // eight independent load pairs in flight (the trip count is a run-time value; 16: no faster)
          float x[8], y[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { x[u] = ar[k + u]; y[u] = s1[(size_t)(k + u) * sg.ldb + c]; }
#pragma unroll
          for (int u = 0; u < 8; u++) acc += x[u] * y[u];
        }
        for (; k < sg.a; k++) acc += ar[k] * s1[(size_t)k * sg.ldb + c];
        v = acc * sg.scale;
        break;
      }
      case SGRL_PACK_PERM32: {  // [1024, a] rows regrouped: dst row c * 32 + q = src0 row q * 32 + c  (32 x 32 matrix per row block)
        const int r = i / sg.a, k = i % sg.a;
        v = s0[(size_t)((r & 31) * 32 + (r >> 5)) * sg.a + k];
        break;
      }
      case SGRL_PACK_SUBMAT: {  // dst [rows, b] = src0 [rows, b] (stride lda)
        const int r = i / sg.b, c = i % sg.b;
        v = s0[(size_t)r * sg.lda + c];
        break;
      }
      default: break;
    }
    w[sg.dst + i] = v;
  }
}

// relation bias per morphology: relb[off + (h*L + i)*L + j] = rel_encoder(relation[i,j])[h]
__global__ void k_relbias(const float* rel, const float* Wr, const float* br, float* relb, const int32_t* m_off,
                          const int32_t* m_L, int n_morph) {
  const int k = blockIdx.x;
  if (k >= n_morph) return;
  const int L = m_L[k], off = m_off[k];
  for (int i = threadIdx.x; i < 2 * L * L; i += blockDim.x) {
    const int h = i / (L * L), ij = i % (L * L);
    const float* r = rel + (size_t)(off / 2) * 3 + ij * 3;  // rel offset = sum L^2 * 3 ; relb offset = sum 2 L^2
    relb[off + i] = Wr[h * 3] * r[0] + Wr[h * 3 + 1] * r[1] + Wr[h * 3 + 2] * r[2] + br[h];
  }
}

// attention: one 256-thread block per environment.  With the output projections folded into the value projections
// (include/sgrl_set.h) the kernel produces the attention block's two outputs directly:
//   delta[i][c]   = b_ng[c] + sum_h sum_j w_h[i,j] v'[j][128 h + c]                               (scalar stream, 128 wide)
//   g1[i][s][c]   = sum_h ( sum_j w_h[i,j] U[j][s][128 h + c]  +  GD[h][c][:] . sum_j w_h[i,j] gdir[j][s][:] )
// (the residual g += g1 of reference SEActor.py:89 is applied by k_equiv together with the feed-forward update of g).
// The scalar stream's residual + norm1 (SEActor.py:90-91) happens here too: ng[i][:] = LayerNorm(ng[i][:] + delta[i][:]),
// in place (row stride ng_ld); delta itself is only written out for the parity probes (delta_dbg != null).
struct EnvTab {
  const int32_t* env_off;   // [n_env] first node
  const int32_t* env_L;     // [n_env]
  const int32_t* env_relb;  // [n_env] offset into relb
};
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, const float* __restrict__ U,
                                                   const float* __restrict__ gdir, const float* relb, EnvTab et,
                                                   int use_bias, const float* __restrict__ b_ng, const float* __restrict__ GD,
                                                   float* delta_dbg, float* g1, float* ng, int ng_ld,
                                                   const float* __restrict__ ln_w, const float* __restrict__ ln_b) {
  constexpr int LMAX = 14;
  __shared__ float sc[2 * LMAX * LMAX];
  __shared__ float gd[2 * LMAX * 3 * 2];        // [h][i][s][e] = sum_j w_h[i,j] gdir[j][s][e]
  __shared__ float dl[LMAX][128];               // delta rows, handed from the column threads to the LayerNorm waves
  const int e = blockIdx.x, t = threadIdx.x;
  const int n0 = et.env_off[e], L = et.env_L[e];
  // scores: eight lanes per (head, i, j) dot product -- each group reads its q / k rows as four coalesced 128-byte
  // segments, partial sums folded with three xor shuffles (all 256 threads busy instead of 2 L^2 <= 98)
  {
    const int sub = t & 7;
    for (int idx = t >> 3; idx < 2 * L * L; idx += 32) {
      const int h = idx / (L * L), i = (idx / L) % L, j = idx % L;
      const float4* q = reinterpret_cast<const float4*>(qkv + (size_t)(n0 + i) * 768 + h * 128);
      const float4* k = reinterpret_cast<const float4*>(qkv + (size_t)(n0 + j) * 768 + 256 + h * 128);
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float4 a = q[sub + 8 * u], b = k[sub + 8 * u];
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      if (sub == 0) sc[idx] = use_bias ? s + relb[et.env_relb[e] + idx] : s;
    }
  }
  __syncthreads();
  if (t < 2 * L) {
    float* row = sc + t * L;
    float mx = row[0];
    for (int j = 1; j < L; j++) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < L; j++) { row[j] = expf(row[j] - mx); sum += row[j]; }
    for (int j = 0; j < L; j++) row[j] = row[j] / sum;
  }
  __syncthreads();
  // attention-weighted gravity / direction columns: 2 heads x L x 3 x 2 <= 168 values
  for (int idx = t; idx < 2 * L * 6; idx += 256) {
    const int h = idx / (L * 6), i = (idx / 6) % L, se = idx % 6;
    const float* w = sc + (h * L + i) * L;
    float s = 0.f;
    for (int j = 0; j < L; j++) s += w[j] * gdir[(size_t)(n0 + j) * 6 + se];
    gd[idx] = s;
  }
  __syncthreads();                               // gd[] complete
  // wave = output quantity (0: scalar stream, 1..3: spatial row quantity - 1), half-wave = head, lane = FOUR output columns:
  // a lane streams its head's value rows as 16-byte loads (one per key limb, the next one in flight while this one is used),
  // keeps the L output rows of its four columns in registers, and the two heads meet over one cross-half exchange.  The row
  // count is a compile-time bound per instance (4 or 8 rows at a time) so that the register arrays stay small.
  {
    const int quantity = t >> 6, ln = t & 63, h = ln >> 5, c4 = (ln & 31) * 4;
    const float* vbase = quantity == 0 ? qkv + (size_t)n0 * 768 + 512 + h * 128 + c4
                                       : U + ((size_t)n0 * 3 + (quantity - 1)) * 256 + h * 128 + c4;     // 768 floats per node either way
    auto body = [&](auto ltc, const int i0) __attribute__((always_inline)) {      // output rows i0 .. i0 + LT - 1
      constexpr int LT = decltype(ltc)::value;
      float4 out[LT];
#pragma unroll
      for (int i = 0; i < LT; i++) out[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* wh = sc + h * L * L;
      if (L <= 8) {          // value rows four at a time (one 16-byte load per key limb in flight, then the products)
#pragma unroll
        for (int jb = 0; jb < 8; jb += 4) {
          if (jb >= L) break;
          float4 v[4];
#pragma unroll
          for (int j = 0; j < 4; j++)
            v[j] = jb + j < L ? *reinterpret_cast<const float4*>(vbase + (size_t)(jb + j) * 768) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (jb + j < L) {
#pragma unroll
              for (int i = 0; i < LT; i++)
                if (i0 + i < L) {
                  const float w = wh[(i0 + i) * L + jb + j];
                  out[i].x += w * v[j].x; out[i].y += w * v[j].y; out[i].z += w * v[j].z; out[i].w += w * v[j].w;
                }
            }
        }
      } else {               // more than 8 key limbs: stream them, the next row in flight while this one is used
        float4 v = *reinterpret_cast<const float4*>(vbase);
        for (int j = 0; j < L; j++) {
          const float4 vn = *reinterpret_cast<const float4*>(vbase + (size_t)(j + 1 < L ? j + 1 : j) * 768);
#pragma unroll
          for (int i = 0; i < LT; i++)
            if (i0 + i < L) {
              const float w = wh[(i0 + i) * L + j];
              out[i].x += w * v.x; out[i].y += w * v.y; out[i].z += w * v.z; out[i].w += w * v.w;
            }
          v = vn;
        }
      }
      if (quantity != 0) {
        const float4 ga = *reinterpret_cast<const float4*>(GD + (h * 128 + c4) * 2);        // (gd0, gd1) of columns c4, c4 + 1
        const float4 gb = *reinterpret_cast<const float4*>(GD + (h * 128 + c4) * 2 + 4);    // ... of columns c4 + 2, c4 + 3
#pragma unroll
        for (int i = 0; i < LT; i++)
          if (i0 + i < L) {
            const float* gg = gd + (h * L + i0 + i) * 6 + 2 * (quantity - 1);
            const float g0 = gg[0], g1v = gg[1];
            out[i].x += ga.x * g0 + ga.y * g1v; out[i].y += ga.z * g0 + ga.w * g1v;
            out[i].z += gb.x * g0 + gb.y * g1v; out[i].w += gb.z * g0 + gb.w * g1v;
          }
      }
      // both heads: exchange across the half-waves; afterwards both halves hold the sums, each stores every other row
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (quantity == 0) bb = *reinterpret_cast<const float4*>(b_ng + c4);
#pragma unroll
      for (int i = 0; i < LT; i++)
        if (i0 + i < L) {
          float4 o = out[i];
          const float px = __shfl_xor(o.x, 32, 64), py = __shfl_xor(o.y, 32, 64), pz = __shfl_xor(o.z, 32, 64), pw = __shfl_xor(o.w, 32, 64);
          // head 0's term first, as the sequential sum over the heads had it
          o = h == 0 ? make_float4(o.x + px, o.y + py, o.z + pz, o.w + pw) : make_float4(px + o.x, py + o.y, pz + o.z, pw + o.w);
          if ((i & 1) == h) {
            if (quantity != 0) {
              *reinterpret_cast<float4*>(g1 + ((size_t)(n0 + i0 + i) * 3 + (quantity - 1)) * 128 + c4) = o;
            } else {
              o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w;
              *reinterpret_cast<float4*>(&dl[i0 + i][c4]) = o;       // to the LayerNorm below, one wave per limb row
              if (delta_dbg) *reinterpret_cast<float4*>(delta_dbg + (size_t)(n0 + i0 + i) * 128 + c4) = o;
            }
          }
        }
    };
    // four output rows at a time: 16 + 32 registers of row and value data keep the kernel at six waves per SIMD (eight rows at a
    // time: 114 registers, four waves); the value rows of the later passes come from L1 / L2
    for (int i0 = 0; i0 < L; i0 += 4) body(std::integral_constant<int, 4>{}, i0);
  }
  __syncthreads();
  const int lane = t & 63;
  const float w0 = ln_w[lane], w1 = ln_w[64 + lane], b0 = ln_b[lane], b1 = ln_b[64 + lane];
  for (int i = t >> 6; i < L; i += 4) {
    float* row = ng + (size_t)(n0 + i) * ng_ld;
    const float v0 = row[lane] + dl[i][lane], v1 = row[64 + lane] + dl[i][64 + lane];
    const float mu = wave_sum_f32(v0 + v1) * (1.f / 128.f);
    const float d0 = v0 - mu, d1 = v1 - mu;
    const float var = wave_sum_f32(d0 * d0 + d1 * d1) * (1.f / 128.f);
    const float inv = 1.0f / sqrtf(var + 1e-5f);
    row[lane] = d0 * inv * w0 + b0;
    row[64 + lane] = d1 * inv * w1 + b1;
  }
}

// y = LayerNorm(x + delta) over 128 channels; one wave per row (2 channels per lane)
__global__ __launch_bounds__(256) void k_add_ln(const float* x, int ldx, const float* delta, int ldd, const float* w,
                                                const float* b, float* out1, int ld1, float* out2, int ld2, int N) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  float v0 = x[(size_t)row * ldx + lane], v1 = x[(size_t)row * ldx + 64 + lane];
  if (delta) { v0 += delta[(size_t)row * ldd + lane]; v1 += delta[(size_t)row * ldd + 64 + lane]; }
  const float mu = wave_sum_f32(v0 + v1) * (1.f / 128.f);
  const float d0 = v0 - mu, d1 = v1 - mu;
  const float var = wave_sum_f32(d0 * d0 + d1 * d1) * (1.f / 128.f);
  const float inv = 1.0f / sqrtf(var + 1e-5f);
  const float y0 = d0 * inv * w[lane] + b[lane], y1 = d1 * inv * w[64 + lane] + b[64 + lane];
  if (out1) { out1[(size_t)row * ld1 + lane] = y0; out1[(size_t)row * ld1 + 64 + lane] = y1; }
  if (out2) { out2[(size_t)row * ld2 + lane] = y0; out2[(size_t)row * ld2 + 64 + lane] = y1; }
}

// g[n][s][:] += g1[n][s][:] + T[n][s][:] . W5^T     8 nodes per 128-thread block, thread = output channel
// (both residual updates of the vector stream in one pass: the attention output g1, reference SEActor.py:89, and the
// equivariant feed-forward term, SEActor.py:108-114, whose 3 x 32 factor T = z3 . mat comes out of the linear4 GEMM's
// epilogue -- the 32 x 32 matrix per node never reaches memory).  With outg != null (last layer) the new g is also written
// into the read-out operand outg[row][8 + c] (reference SEActor.py:254) together with its zero K-padding columns.
__global__ __launch_bounds__(128) void k_equiv(const float* __restrict__ T, const float* __restrict__ W5,
                                               const float* __restrict__ g1, float* g, float* outg, int N) {
  __shared__ float Ts[8 * 96];
  const int t = threadIdx.x, n0 = blockIdx.x * 8;
  const int rows = min(8, N - n0) * 3;            // (node, s) rows of this block
  for (int o = t; o < 8 * 96; o += 128) Ts[o] = (n0 + o / 96 < N) ? T[(size_t)n0 * 96 + o] : 0.f;
  float w[32];
#pragma unroll
  for (int c = 0; c < 32; c++) w[c] = W5[t * 32 + c];
  __syncthreads();
  // eight rows of g and g1 are requested at a time (the kernel is bandwidth-bound: 16 loads in flight per thread at seven waves
  // per SIMD), the products run while they arrive
#pragma unroll
  for (int rb = 0; rb < 24; rb += 8) {
    if (rb >= rows) break;
    float gv[8], g1v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const size_t row = (size_t)n0 * 3 + (rb + r < rows ? rb + r : 0);
      gv[r] = g[row * D + t];
      g1v[r] = g1[row * D + t];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (rb + r >= rows) break;
      const float* tt = Ts + (rb + r) * 32;
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < 32; c++) v += tt[c] * w[c];
      const size_t row = (size_t)n0 * 3 + rb + r;
      const float gn = gv[r] + (g1v[r] + v);
      g[row * D + t] = gn;
      if (outg) {
        outg[row * OGLD + 8 + t] = gn;
        if (t < OGLD - 136) outg[row * OGLD + 136 + t] = 0.f;
      }
    }
  }
}

// head: vec[s] = T[n][s][:] . wdec (T = Zh . mat from the linear2_m GEMM's epilogue); action_k = max_action * tanh(sum_s
// axis_k[s] * vec[s]); 32 lanes per node
__global__ __launch_bounds__(128) void k_head_out(const float* __restrict__ T, const float* __restrict__ wdec,
                                                  const float* __restrict__ obs, int obs_ld, NodeTab nt, float* act, int act_ld,
                                                  float max_action, int N) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 5), c = threadIdx.x & 31;
  if (n >= N) return;
  float vec[3];
  const float wd = wdec[c];
  for (int s = 0; s < 3; s++) {
    float v = T[(size_t)n * 96 + s * 32 + c] * wd;
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
    vec[s] = v;
  }
  if (c < 3) {
    const int env = nt.node_env[n], limb = nt.node_limb[n];
    const float* o = obs + (size_t)env * obs_ld + 41 * limb + 3 * (5 + c);
    const float a = o[0] * vec[0] + o[1] * vec[1] + o[2] * vec[2];
    act[(size_t)env * act_ld + 3 * limb + c] = max_action * tanhf(a);
  }
}


// head with decoder_g folded through linear2_m (include/sgrl_set.h): m2[n][q] = (Wf . h[n] + bf)[q] / fn[n] comes out of a 32-wide
// product; vec[s] = z[n][s][:] . m2[n][:]; action_k = max_action * tanh(sum_s axis_k[s] * vec[s]); 32 lanes per node
__global__ __launch_bounds__(128) void k_head_out2(const float* __restrict__ m2, const float* __restrict__ z, const float* __restrict__ obs,
                                                   int obs_ld, NodeTab nt, float* act, int act_ld, float max_action, int N) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 5), c = threadIdx.x & 31;
  if (n >= N) return;
  float vec[3];
  const float mq = m2[(size_t)n * 32 + c];
  for (int s = 0; s < 3; s++) {
    float v = z[(size_t)n * 96 + s * 32 + c] * mq;
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
    vec[s] = v;
  }
  if (c < 3) {
    const int env = nt.node_env[n], limb = nt.node_limb[n];
    const float* o = obs + (size_t)env * obs_ld + 41 * limb + 3 * (5 + c);
    const float a = o[0] * vec[0] + o[1] * vec[1] + o[2] * vec[2];
    act[(size_t)env * act_ld + 3 * limb + c] = max_action * tanhf(a);
  }
}

// critic head: q[env][limb] = (w . c[n] + b) / fn[n]   (reference SEActor.py:279-281 with output_size = 1); one wave per node
__global__ __launch_bounds__(256) void k_q_head(const float* __restrict__ c, const float* __restrict__ w, const float* __restrict__ b,
                                                const float* __restrict__ fn, NodeTab nt, float* q, int q_ld, int N) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  const float* row = c + (size_t)n * 256;
  float v = row[lane] * w[lane] + row[64 + lane] * w[64 + lane] + row[128 + lane] * w[128 + lane] + row[192 + lane] * w[192 + lane];
  v = wave_sum_f32(v);
  if (lane == 0) q[(size_t)nt.node_env[n] * q_ld + nt.node_limb[n]] = (v + b[0]) / fn[n];
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// One batch structure (sgrl_set_graph): the device tables of a (morphologies x env counts) configuration.
struct GraphCfg {
  std::vector<int32_t> key_i;   // n_morph | L[] | count[] | trav[]
  std::vector<float> key_f;     // rel[]
  int n_env = 0, N = 0, n_morph = 0, TM = 0, Lmax = 0;
  int32_t *d_node_env = nullptr, *d_node_limb = nullptr, *d_node_mnode = nullptr, *d_trav = nullptr;
  int32_t *d_env_off = nullptr, *d_env_L = nullptr, *d_env_relb = nullptr, *d_m_off = nullptr, *d_m_L = nullptr;
  float *d_rel = nullptr, *d_relb = nullptr;
  uint64_t last_use = 0;
  void release() {
    void* ptrs[] = {d_node_env, d_node_limb, d_node_mnode, d_trav, d_env_off, d_env_L, d_env_relb, d_m_off, d_m_L, d_rel, d_relb};
    for (void* q : ptrs) if (q) (void)hipFree(q);
  }
};

struct sgrl_set {
  const float* w = nullptr;
  int64_t off[SGRL_SET_NW];
  bool have_w = false, have_graph = false;
  // current batch structure (copied out of the cache entry by use_cfg)
  int n_env = 0, N = 0, n_morph = 0, TM = 0, Lmax = 0;
  int32_t *d_node_env = nullptr, *d_node_limb = nullptr, *d_node_mnode = nullptr, *d_trav = nullptr;
  int32_t *d_env_off = nullptr, *d_env_L = nullptr, *d_env_relb = nullptr, *d_m_off = nullptr, *d_m_L = nullptr;
  float *d_rel = nullptr, *d_relb = nullptr;
  std::vector<GraphCfg*> cfgs;
  uint64_t use_clock = 0;
  // bumped whenever device memory a captured hipGraph may have baked a pointer into is FREED: a batch structure evicted from
  // the cache, the flat weight buffers of a rebinding, a regrown workspace (sgrl_set_generation; td3.GraphedUpdates compares it)
  int64_t generation = 0;
  // workspace: shared by all batch structures, grows only; carved for the current N
  float* ws = nullptr;
  int64_t ws_floats = 0;
  int carved_N = 0;
  float* cat_cur = nullptr;    // which of cat / cat2 holds [inv | ng] after the last forward (the fused form alternates: run_forward)
  float *g, *cat, *cat2, *zc, *fn, *h256, *qkv, *vg, *g1, *z2, *mat, *t256, *t256b, *t128a, *t128b, *delta,
      *outg, *outng, *gdir;
  // stacked projection weights of the 7 proj+gram sites: static weights -> own buffer rebuilt on the forward stream after
  // sgrl_set_weights; live weights -> part of the flat buffer, rebuilt by k_pack with everything else
  float* wstack = nullptr;
  const float* site_ptr[SGRL_SET_NSITES];
  // sgrl_set_hold_weights: the caller promises the bound parameters do not change while the hold lasts -- the flat buffer (and its
  // row-scaled words) built by the first forward after the promise serves the following ones
  bool hold = false, packed_ok = false;
  int packed_form = 0;
  hipEvent_t ev_pack = nullptr;
  const float* l2mf_w = nullptr;   // live weights: decoder_g folded through linear2_m [32, 256] and its bias [32] (null: unfolded head)
  const float* l2mf_b = nullptr;
  unsigned short* d_tri = nullptr;
  bool stack_dirty = true;
  bool stack_critic = false;   // mode the stacked operands were built for
  int stop_after = -1;         // parity probes: leave run_forward after this stage (sgrl_set_debug_stop_after)
  int small_nodes = -1;        // batches of at most this many nodes take the small-batch products; -1: SGRL_SET_SMALL_NODES / default
  int gemm_form = 0;           // SGRL_SET_FORM_* of the tile products; 0: SGRL_SET_GEMM / default (sgrl_set_gemm_form)
  // two-piece products: the product matrices of the flat buffer, cut into row-scaled words by k_encode_rows behind every k_pack
  sgrl_gemm::EncMat* d_enc = nullptr;   // [n_enc] (offset, rows, K, first row)
  int n_enc = 0, enc_rows = 0;
  float* wsc = nullptr;                 // [enc_rows] inverse row scales
  std::vector<std::pair<int64_t, int>> enc_index;   // flat offset of a matrix -> its first row in wsc
  // live weights (sgrl_set_bind_params)
  bool live = false;
  float* wflat = nullptr;
  unsigned* wwords = nullptr;  // wflat as pre-split words (two-piece products)
  int64_t wflat_floats = 0;
  sgrl_pack_seg* d_segs = nullptr;
  int2* d_chunks = nullptr;
  int n_chunks = 0;
  // side stream for the GEMM chains that do not depend on each other (their epilogues / tile tails overlap)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_l3 = nullptr;
  std::vector<hipStream_t> side_ok_for;   // caller streams `side` has been measured to overlap (stream_pick.h)
  int side_picks = 0;                     // measurements spent so far (bounded: a caller hopping between streams must not pay forever)
  const float* W(int slot) const { return w + off[slot]; }
  const float* WL(int layer, int k) const { return w + off[SGRL_SET_NGLOBAL + layer * SGRL_SET_NLAYER + k]; }
};

namespace {

constexpr int64_t kPerNodeFloats = 384 /*g*/ + 256 + 256 /*cat, cat2*/ + 96 /*zc*/ + 1 /*fn*/ + 256 /*h256*/ + 768 /*qkv*/ +
                                   768 /*vg*/ + 384 /*g1*/ + 96 /*z2*/ + 96 /*T (was mat: 1024)*/ + 256 + 256 /*t256, t256b*/ +
                                   128 + 128 + 128 /*t128a,b,delta*/ + 3 * OGLD /*outg*/ + 160 /*outng*/ + 6 /*gdir*/;
constexpr int kWsArrays = 19;

int64_t ws_floats_for(int64_t N) { return kPerNodeFloats * N + 32 * kWsArrays; }

void free_graphs(sgrl_set* s) {
  for (GraphCfg* c : s->cfgs) { c->release(); delete c; }
  s->cfgs.clear();
  if (s->ws) (void)hipFree(s->ws);
  s->ws = nullptr; s->ws_floats = 0; s->carved_N = 0;
  s->have_graph = false;
}

template <class T>
int upload(T** dst, const std::vector<T>& v) {
  if (hipMalloc(dst, sizeof(T) * (v.size() ? v.size() : 1)) != hipSuccess) return -1;
  if (!v.empty() && hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
  return 0;
}

// point the handle at a cached batch structure; (re)carve the shared workspace for its node count
int use_cfg(sgrl_set* s, GraphCfg* c) {
  const int64_t N = c->N;
  const int64_t need = ws_floats_for(N);
  if (need > s->ws_floats) {
    if (s->ws) { (void)hipFree(s->ws); s->generation++; }       // hipFree waits for the device: no kernel still reads the old block
    s->ws = nullptr; s->ws_floats = 0; s->carved_N = 0;
    if (hipMalloc(&s->ws, sizeof(float) * need) != hipSuccess) return sfail(SGRL_ERR_HIP, "device allocation failed (SET workspace)");
    s->ws_floats = need;
  }
  if (s->carved_N != c->N) {
    float* p = s->ws;
    auto take = [&](int64_t n) { float* r = p; p += (n + 31) & ~int64_t(31); return r; };
    s->g = take(384 * N); s->cat = take(256 * N); s->cat2 = take(256 * N); s->zc = take(96 * N); s->fn = take(N);
    s->h256 = take(256 * N); s->qkv = take(768 * N); s->vg = take(768 * N);
    s->g1 = take(384 * N); s->z2 = take(96 * N); s->mat = take(96 * N); s->t256 = take(256 * N); s->t256b = take(256 * N); s->t128a = take(128 * N);
    s->t128b = take(128 * N); s->delta = take(128 * N); s->outg = take(3 * OGLD * N); s->outng = take(160 * N); s->gdir = take(6 * N);
    if (p - s->ws > s->ws_floats) return sfail(SGRL_ERR_LIMIT, "workspace carve-up overflow");
    s->carved_N = c->N;
  }
  s->n_env = c->n_env; s->N = c->N; s->n_morph = c->n_morph; s->TM = c->TM; s->Lmax = c->Lmax;
  s->d_node_env = c->d_node_env; s->d_node_limb = c->d_node_limb; s->d_node_mnode = c->d_node_mnode; s->d_trav = c->d_trav;
  s->d_env_off = c->d_env_off; s->d_env_L = c->d_env_L; s->d_env_relb = c->d_env_relb; s->d_m_off = c->d_m_off; s->d_m_L = c->d_m_L;
  s->d_rel = c->d_rel; s->d_relb = c->d_relb;
  c->last_use = ++s->use_clock;
  s->have_graph = true;
  return SGRL_OK;
}

// Tile configurations (measured on the shapes of one forward: profiles/r2_gemm_lab_*.log, r3_gemm_*_lab.txt -- the lab of those rounds, tools/gemm_lab.hip, is an archive; the live labs are tools/chain_lab.hip and tools/wdirect_lab.hip).  Default: the split-precision kernel
// k_gemm3 in its two-piece form (three f16 MFMAs per product block, weights pre-split by k_pack; float32 result -- its
// error against float64 is BELOW that of the exact-f32 MFMA chain, gemm_f32.h) on 128 x 128 tiles with 8 waves, the stacked
// projections on 128 x 64 tiles (kProjH); SGRL_SET_GEMM=bf16x6 / sgrl_set_gemm_form select the three-piece bf16 form.  The
// exact-f32 kernel k_gemm2 serves K not a multiple of 32 and, with SGRL_SET_GEMM=f32 in the environment, everything (A/B
// comparisons).
//   kNarrow  f32  128 x  64 tile,  4 waves (32 x 64 each), BK 16
//   kWide    f32  128 x 128 tile,  8 waves (32 x 64 each), BK 32, loads two k-tiles ahead (N >= 512)
//   kSquare  f32  128 x 128 tile, 16 waves (32 x 32 each), BK 32, loads two k-tiles ahead
//   kSplit   x6   128 x 128 tile,  8 waves (32 x 64 each), BK 16, loads two k-tiles ahead, two blocks per CU
bool gemm_use_split() {
  static const bool v = [] { const char* e = getenv("SGRL_SET_GEMM"); return !(e && e[0] == 'f'); }();
  return v;
}
// Split form of the forward in flight (set by run_forward from its handle): SGRL_SET_FORM_F16X3 = two f16 pieces, three matrix
// instructions per product block (default; every operand row pre-scaled by a power of two: float32's range), SGRL_SET_FORM_BF16X6 =
// three bf16 pieces, six instructions, f32's exponent range (gemm_f32.h).  SGRL_SET_GEMM=bf16x6 makes the latter the default.
struct GemmCtx {
  int form = SGRL_SET_FORM_F16X3;
  const float* w_base = nullptr; const unsigned* w_words = nullptr;   // flat weight buffer and its row-scaled, pre-split twin (k_encode_rows)
  const float* wsc = nullptr;                                         // inverse row scales of the twin's matrices ...
  const std::vector<std::pair<int64_t, int>>* enc_index = nullptr;    // ... found by a matrix' flat offset
};
thread_local GemmCtx g_gemm;
int gemm_default_form() {
  static const int v = [] { const char* e = getenv("SGRL_SET_GEMM"); return (e && e[0] == 'b') ? SGRL_SET_FORM_BF16X6 : SGRL_SET_FORM_F16X3; }();
  return v;
}
// arguments of a two-piece launch: the event counter, and W taken from the pre-split twin of the weight buffer
// inverse row scales of the product matrix that starts at W (null: W is not a whole matrix of the table -> unscaled words do not exist)
const float* wsc_of(const float* W) {
  if (!g_gemm.enc_index) return nullptr;
  const int64_t off = W - g_gemm.w_base;
  for (const auto& e : *g_gemm.enc_index) if (e.first == off) return g_gemm.wsc + e.second;
  return nullptr;
}
GemmArgs with_words(const GemmArgs& a) {
  GemmArgs b = a;
  b.W = reinterpret_cast<const float*>(g_gemm.w_words + (a.W - g_gemm.w_base));
  b.wscale = wsc_of(a.W);
  return b;
}
template <int F> struct GemmKernels {
  static constexpr auto kNarrow = k_gemm2<F, 4, 1, 1, 2, 16, 1>;
  static constexpr auto kWide = k_gemm2<F, 4, 2, 1, 2, 32, 2>;
  static constexpr auto kSquare = k_gemm2<F, 4, 4, 1, 1, 32, 2>;
  static constexpr auto kSplit = sgrl_gemm::k_gemm3<F, 4, 2, 1, 2, 16, 2>;
  static constexpr auto kSplitH = sgrl_gemm::k_gemm3<F, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
  static constexpr int kSplitHLds = sgrl_gemm::TileCfg3<4, 2, 1, 2, 16, 2>::kLdsBytes;
  static constexpr int kNarrowLds = sgrl_gemm::TileCfg<4, 1, 1, 2, 16>::kLdsBytes;
  static constexpr int kWideLds = sgrl_gemm::TileCfg<4, 2, 1, 2, 32>::kLdsBytes;
  static constexpr int kSquareLds = sgrl_gemm::TileCfg<4, 4, 1, 1, 32>::kLdsBytes;
  static constexpr int kSplitLds = sgrl_gemm::TileCfg3<4, 2, 1, 2, 16>::kLdsBytes;
  static bool raise_lds_limits() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kNarrow), hipFuncAttributeMaxDynamicSharedMemorySize, kNarrowLds) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(kWide), hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(kSquare), hipFuncAttributeMaxDynamicSharedMemorySize, kSquareLds) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(kSplit), hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLds) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(kSplitH), hipFuncAttributeMaxDynamicSharedMemorySize, kSplitHLds) == hipSuccess;
  }
  static void launch(hipStream_t st, const GemmArgs& a) {
    const int tiles128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (a.N <= 64 || (a.K % 32) != 0) {
      hipLaunchKernelGGL(kNarrow, dim3(((a.M + 127) / 128) * ((a.N + 63) / 64)), dim3(256), kNarrowLds, st, a);
    } else if (gemm_use_split()) {
      if (g_gemm.form == SGRL_SET_FORM_F16X3) hipLaunchKernelGGL(kSplitH, dim3(tiles128), dim3(512), kSplitHLds, st, with_words(a));
      else hipLaunchKernelGGL(kSplit, dim3(tiles128), dim3(512), kSplitLds, st, a);
    } else if (tiles128 < 512) {
      hipLaunchKernelGGL(kNarrow, dim3(((a.M + 127) / 128) * ((a.N + 63) / 64)), dim3(256), kNarrowLds, st, a);
    } else if (a.N >= 512) {
      hipLaunchKernelGGL(kWide, dim3(tiles128), dim3(512), kWideLds, st, a);
    } else {
      hipLaunchKernelGGL(kSquare, dim3(tiles128), dim3(1024), kSquareLds, st, a);
    }
  }
};

// the stacked projections (N = 32 / 64) in the two-piece form
constexpr auto kProjH = sgrl_gemm::k_gemm3<EPI_ZSPLIT, 4, 1, 1, 2, 16, 2, false, false, false, 0, false, 2, false, 2>;
constexpr int kProjHLds = sgrl_gemm::TileCfg3<4, 1, 1, 2, 16, 2>::kLdsBytes;

// linear4 / linear2_m (N = 1024, columns ordered c * 32 + a) with the equivariant contraction in the epilogue:
// tout[m][s][c] = sum_a zq[m][s][a] * ((A . W^T + b)[m][c * 32 + a] / rowdiv[m]); always the split-precision kernel
constexpr auto kGemmEquiv = sgrl_gemm::k_gemm3<EPI_ROWDIV | EPI_EQUIV, 4, 2, 1, 2, 16, 2>;
constexpr auto kGemmEquivH = sgrl_gemm::k_gemm3<EPI_ROWDIV | EPI_EQUIV, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
// the equivariant epilogue stages the block's 128 z rows (pitch 100 floats) in the tile's LDS: more than the k-tile stages need
constexpr int kEquivLds = GemmKernels<0>::kSplitLds > 128 * 100 * 4 ? GemmKernels<0>::kSplitLds : 128 * 100 * 4;
int launch_gemm_equiv(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, int M, int K,
                      const float* rowdiv, const float* zq, float* tout) {
  if (K % 32 != 0 || (lda & 3) || (ldw & 3)) return sfail(SGRL_ERR_ARG, "gemm_equiv: K must be a multiple of 32 and rows 16-byte aligned");
  GemmArgs a{A, lda, W, ldw, bias, nullptr, 0, M, 1024, K, EPI_ROWDIV | EPI_EQUIV, rowdiv, nullptr, 0};
  a.zq = zq; a.tout = tout;
  if (g_gemm.form == SGRL_SET_FORM_F16X3) hipLaunchKernelGGL(kGemmEquivH, dim3(((M + 127) / 128) * 8), dim3(512), kEquivLds, st, with_words(a));
  else hipLaunchKernelGGL(kGemmEquiv, dim3(((M + 127) / 128) * 8), dim3(512), kEquivLds, st, a);
  return SGRL_OK;
}

// C[M,N] = relu(G(Z) . W^T + b): the Gram-operand GEMM (A generated from zc [M, 3, 32]; W [N, 576] folded); N = 128 or 256
constexpr auto kGemmGram = sgrl_gemm::k_gemm3<EPI_RELU, 4, 2, 1, 2, 16, 2, false, false, false, 0, true>;
constexpr auto kGemmGramH = sgrl_gemm::k_gemm3<EPI_RELU, 4, 2, 1, 2, 16, 2, false, false, false, 0, true, 2, true, 2>;
int launch_gemm_gram(hipStream_t st, const float* zc, const float* W, const float* bias, float* C, int ldc, int M, int N, float* fn) {
  if (N % 128 != 0) return sfail(SGRL_ERR_ARG, "gemm_gram: N must be a multiple of 128");
  GemmArgs a{zc, 96, W, GK, bias, C, ldc, M, N, GK, EPI_RELU, nullptr, nullptr, 0};
  a.rowdiv_out = fn;
  if (g_gemm.form == SGRL_SET_FORM_F16X3) hipLaunchKernelGGL(kGemmGramH, dim3(((M + 127) / 128) * (N / 128)), dim3(512), GemmKernels<0>::kSplitHLds, st, with_words(a));
  else hipLaunchKernelGGL(kGemmGram, dim3(((M + 127) / 128) * (N / 128)), dim3(512), GemmKernels<0>::kSplitLds, st, a);
  return SGRL_OK;
}

// ln_io[m][:] = LayerNorm(ln_io[m][:] + (A . W^T + b)[m][:] / rowdiv[m]) * ln_w + ln_b   (N = 128, residual stream in place)
constexpr auto kGemmLn = sgrl_gemm::k_gemm3<EPI_ROWDIV | EPI_LN, 4, 2, 1, 2, 16, 2>;
constexpr auto kGemmLnH = sgrl_gemm::k_gemm3<EPI_ROWDIV | EPI_LN, 4, 2, 1, 2, 16, 2, false, false, false, 0, false, 2, true, 2>;
int launch_gemm_ln(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, int M, int K,
                   const float* rowdiv, float* ln_io, int ln_ld, const float* ln_w, const float* ln_b) {
  if (K % 32 != 0 || (lda & 3) || (ldw & 3)) return sfail(SGRL_ERR_ARG, "gemm_ln: K must be a multiple of 32 and rows 16-byte aligned");
  GemmArgs a{A, lda, W, ldw, bias, nullptr, 0, M, 128, K, EPI_ROWDIV | EPI_LN, rowdiv, nullptr, 0};
  a.ln_io = ln_io; a.ln_ld = ln_ld; a.ln_w = ln_w; a.ln_b = ln_b;
  if (g_gemm.form == SGRL_SET_FORM_F16X3) hipLaunchKernelGGL(kGemmLnH, dim3((M + 127) / 128), dim3(512), GemmKernels<0>::kSplitHLds, st, with_words(a));
  else hipLaunchKernelGGL(kGemmLn, dim3((M + 127) / 128), dim3(512), GemmKernels<0>::kSplitLds, st, a);
  return SGRL_OK;
}

int launch_gemm(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                int M, int N, int K, int flags = 0, const float* rowdiv = nullptr, float* C2 = nullptr, int ldc2 = 0) {
  if (K % 16 != 0 || (lda & 3) || (ldw & 3)) return sfail(SGRL_ERR_ARG, "gemm: K must be a multiple of 16 and rows 16-byte aligned");
  GemmArgs a{A, lda, W, ldw, bias, C, ldc, M, N, K, flags, rowdiv, C2, ldc2};
  switch (flags) {
    case 0: GemmKernels<0>::launch(st, a); break;
    case EPI_RELU: GemmKernels<EPI_RELU>::launch(st, a); break;
    case EPI_ROWDIV: GemmKernels<EPI_ROWDIV>::launch(st, a); break;
    case EPI_ACC2: GemmKernels<EPI_ACC2>::launch(st, a); break;
    default: return sfail(SGRL_ERR_ARG, "gemm: unsupported epilogue combination");
  }
  return SGRL_OK;
}

// ---- back-to-back products in one kernel (chain_f16.h; two-piece form, weights pre-split by k_pack) -----------------------
// SGRL_SET_CHAIN=0 in the environment keeps every product a launch of its own (A/B comparisons).
bool chain_enabled() {
  static const bool v = [] { const char* e = getenv("SGRL_SET_CHAIN"); return !(e && e[0] == '0'); }();
  return v;
}
using sgrl_gemm::ChainArgs;
using sgrl_gemm::k_chain;
constexpr auto kSiteA = k_chain<1, 256, 0, 1>;                         // attention site: g -> Z -> Gram -> lg1 -> ReLU -> lg2
constexpr auto kSiteF = k_chain<1, 256, 0, 2>;                         // feed-forward site: g1 -> Z, Z2 -> ...
constexpr auto kSiteH1 = k_chain<1, 128, 0, 1>;                        // head (critic): outg -> Z -> Gram -> l1g -> ReLU -> l2g
constexpr auto kSiteH2 = k_chain<1, 128, 0, 2>;                        // head (actor): ... Z, Z2
constexpr auto kChainLn = k_chain<0, 256, EPI_ROWDIV | EPI_LN, 0>;     // linear1 -> ReLU -> linear2 / fn -> residual + norm2
constexpr auto kChainNg = k_chain<0, 128, 0, 0>;                       // linear1_ng -> ReLU -> linear2_ng
constexpr auto kChainPlain = k_chain<0, 256, 0, 0>;                    // (test hook: the plain pair at hidden width 256)
bool chain_raise_lds_limits() {
  const void* ks[] = {reinterpret_cast<const void*>(kSiteA), reinterpret_cast<const void*>(kSiteF), reinterpret_cast<const void*>(kSiteH1),
                      reinterpret_cast<const void*>(kSiteH2), reinterpret_cast<const void*>(kChainLn), reinterpret_cast<const void*>(kChainNg),
                      reinterpret_cast<const void*>(kChainPlain)};
  for (const void* k : ks)
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, sgrl_gemm::kChainLds) != hipSuccess) return false;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<0, 256, EPI_ROWDIV | EPI_EQUIV, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             sgrl_gemm::kChainEqLds) == hipSuccess;
}
const unsigned* words_of(const float* W) { return g_gemm.w_words + (W - g_gemm.w_base); }
// projection site + Gram pair: X [3 M, Kp] -> zc (and z2) -> fn -> relu(G(Z) . W1^T + b1) . W2^T + b2 -> C[:, 0:128]
int launch_site(hipStream_t st, const float* X, int ldx, int Kp, const float* Wp, float* zc, float* z2, float* fn, const float* W1,
                const float* b1, int hid, const float* W2, const float* b2, float* C, int ldc, int M) {
  if (Kp % 16 != 0 || (ldx & 3) || (hid != 256 && hid != 128)) return sfail(SGRL_ERR_ARG, "site: K must be a multiple of 16, rows 16-byte aligned, hidden width 128 or 256");
  ChainArgs a{};
  a.A = zc; a.W1 = words_of(W1); a.ldw1 = GK; a.b1 = b1; a.W2 = words_of(W2); a.ldw2 = hid; a.b2 = b2; a.C = C; a.ldc = ldc; a.M = M; a.K1 = GK;
  a.fn_out = fn; a.X = X; a.ldx = ldx; a.Kp = Kp; a.Wp = words_of(Wp); a.zc = zc; a.z2 = z2;
  a.ws1 = wsc_of(W1); a.ws2 = wsc_of(W2); a.wsp = wsc_of(Wp);
  if (!a.ws1 || !a.ws2 || !a.wsp) return sfail(SGRL_ERR_ARG, "site: a weight operand is not a matrix of the row-scale table");
  const dim3 grid((M + sgrl_gemm::kChainRows - 1) / sgrl_gemm::kChainRows);
  if (hid == 256) {
    if (z2) hipLaunchKernelGGL(kSiteF, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
    else hipLaunchKernelGGL(kSiteA, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
  } else {
    if (z2) hipLaunchKernelGGL(kSiteH2, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
    else hipLaunchKernelGGL(kSiteH1, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
  }
  return SGRL_OK;
}
// ln_io[m][:] = LayerNorm(ln_io[m][:] + (relu(A . W1^T + b1) . W2^T + b2)[m][:] / rowdiv[m])      (hidden width 256)
int launch_chain_ln(hipStream_t st, const float* A, int lda, int K1, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* rowdiv, float* ln_io, int ln_ld, const float* ln_w, const float* ln_b, int M, float* ln_out = nullptr) {
  if (K1 % 16 != 0 || (lda & 3)) return sfail(SGRL_ERR_ARG, "chain: K must be a multiple of 16 and rows 16-byte aligned");
  ChainArgs a{};
  a.A = A; a.lda = lda; a.W1 = words_of(W1); a.ldw1 = K1; a.b1 = b1; a.W2 = words_of(W2); a.ldw2 = 256; a.b2 = b2; a.M = M; a.K1 = K1;
  a.rowdiv = rowdiv; a.ln_io = ln_io; a.ln_ld = ln_ld; a.ln_w = ln_w; a.ln_b = ln_b; a.ln_out = ln_out;
  a.ws1 = wsc_of(W1); a.ws2 = wsc_of(W2);
  if (!a.ws1 || !a.ws2) return sfail(SGRL_ERR_ARG, "chain: a weight operand is not a matrix of the row-scale table");
  hipLaunchKernelGGL(kChainLn, dim3((M + sgrl_gemm::kChainRows - 1) / sgrl_gemm::kChainRows), dim3(512), sgrl_gemm::kChainLds, st, a);
  return SGRL_OK;
}
// tout[m][s][c] = sum_q zq[m][s][q] * ((relu(A . W1^T + b1) . W2^T + b2)[m][c * 32 + q] / rowdiv[m])      (hidden width 256, N = 1024)
constexpr auto kChainEq = k_chain<0, 256, EPI_ROWDIV | EPI_EQUIV, 0>;   // linear3 -> ReLU -> linear4 -> contraction (and the head's linear1_m -> linear2_m)
int launch_chain_equiv(hipStream_t st, const float* A, int lda, int K1, const float* W1, const float* b1, const float* W2, const float* b2,
                       const float* rowdiv, const float* zq, float* tout, int M, float* g = nullptr, const float* g1 = nullptr,
                       const float* W5 = nullptr, float* outg = nullptr) {
  if (K1 % 16 != 0 || (lda & 3)) return sfail(SGRL_ERR_ARG, "chain: K must be a multiple of 16 and rows 16-byte aligned");
  ChainArgs a{};
  a.A = A; a.lda = lda; a.W1 = words_of(W1); a.ldw1 = K1; a.b1 = b1; a.W2 = words_of(W2); a.ldw2 = 256; a.b2 = b2; a.M = M; a.K1 = K1;
  a.rowdiv = rowdiv; a.zq = zq; a.tout = tout; a.ws1 = wsc_of(W1); a.ws2 = wsc_of(W2);
  if (!a.ws1 || !a.ws2) return sfail(SGRL_ERR_ARG, "chain: a weight operand is not a matrix of the row-scale table");
  if (g) {               // the vector stream's update in the same kernel
    a.g = g; a.g1 = g1; a.W5 = words_of(W5); a.ws5 = wsc_of(W5); a.outg = outg; a.outg_ld = OGLD;
    if (!a.ws5) return sfail(SGRL_ERR_ARG, "chain: linear5 is not a matrix of the row-scale table");
  }
  hipLaunchKernelGGL(kChainEq, dim3((M + sgrl_gemm::kChainRows - 1) / sgrl_gemm::kChainRows), dim3(512), sgrl_gemm::kChainEqLds, st, a);
  return SGRL_OK;
}
// C[:, 0:128] = relu(A . W1^T + b1) . W2^T + b2      (hidden width 128)
int launch_chain_ng(hipStream_t st, const float* A, int lda, int K1, const float* W1, const float* b1, const float* W2, const float* b2,
                    float* C, int ldc, int M) {
  if (K1 % 16 != 0 || (lda & 3)) return sfail(SGRL_ERR_ARG, "chain: K must be a multiple of 16 and rows 16-byte aligned");
  ChainArgs a{};
  a.A = A; a.lda = lda; a.W1 = words_of(W1); a.ldw1 = K1; a.b1 = b1; a.W2 = words_of(W2); a.ldw2 = 128; a.b2 = b2; a.C = C; a.ldc = ldc; a.M = M;
  a.K1 = K1; a.ws1 = wsc_of(W1); a.ws2 = wsc_of(W2);
  if (!a.ws1 || !a.ws2) return sfail(SGRL_ERR_ARG, "chain: a weight operand is not a matrix of the row-scale table");
  hipLaunchKernelGGL(kChainNg, dim3((M + sgrl_gemm::kChainRows - 1) / sgrl_gemm::kChainRows), dim3(512), sgrl_gemm::kChainLds, st, a);
  return SGRL_OK;
}

// ---- small batches ------------------------------------------------------------------------------------------------------
// Below kSmallNodes nodes (the TD3 update's no-grad target networks: 100 transitions of one morphology = 700..1400 nodes;
// single-environment action selection) a 128 x 128 tile kernel has a handful of workgroups that each walk the whole
// contraction: the forward is a chain of ~50 latency-bound launches.  Those batches take the products through the 32 x 32 tile
// kernels of the training path instead (train_gemm.hip, include/sgrl_train.h: exact-float32 matrix instruction, four waves
// splitting every k-tile); the epilogue fusions of the big path become the small kernels below.  Scratch: the attention's
// qkv | vg block (contiguous, 1536 floats per node), idle whenever these run.
constexpr int kSmallNodesDefault = 2048;
int small_nodes() {
  return kSmallNodesDefault;       // (tests move the threshold per handle: sgrl_set_debug_small_nodes)
}

// A[n][0:576] = blocked lower triangle of Z'Z (order of the folded weights), fn[n] = ||Z'Z||_F + 1; one wave per node
__global__ __launch_bounds__(256) void k_gram576(const float* __restrict__ zc, const unsigned short* __restrict__ tri,
                                                 float* __restrict__ A, float* __restrict__ fn, int N) {
  __shared__ float zs[4][96];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, node = blockIdx.x * 4 + w;
  const bool live = node < N;
  if (live) {
    zs[w][lane] = zc[(size_t)node * 96 + lane];
    if (lane < 32) zs[w][64 + lane] = zc[(size_t)node * 96 + 64 + lane];
  }
  __syncthreads();
  if (!live) return;
  float ss = 0.f;
  for (int k = lane; k < GK; k += 64) {
    const unsigned t = tri[k];
    float v = 0.f;
    if (t != 0xFFFFu) {
      const int a = t >> 8, b = t & 255;
      v = zs[w][a] * zs[w][b] + zs[w][32 + a] * zs[w][32 + b] + zs[w][64 + a] * zs[w][64 + b];
      ss += (a == b ? 1.f : 2.f) * v * v;
    }
    A[(size_t)node * GK + k] = v;
  }
  ss = wave_sum_f32(ss);
  if (lane == 0) fn[node] = sqrtf(ss) + 1.f;
}

// tout[n][s][c] = sum_a zq[n][s][a] * matp[n][c * 32 + a]   (the 32 x 32 matrix in the packed c-major order); block = node
__global__ __launch_bounds__(128) void k_zmat_perm(const float* __restrict__ zq, const float* __restrict__ matp,
                                                   float* __restrict__ tout) {
  __shared__ float ms[32 * 33], zs[96];
  const int t = threadIdx.x;
  const size_t n = blockIdx.x;
  for (int o = t; o < 1024; o += 128) ms[(o >> 5) * 33 + (o & 31)] = matp[n * 1024 + o];
  if (t < 96) zs[t] = zq[n * 96 + t];
  __syncthreads();
  if (t >= 96) return;
  const int sI = t >> 5, c = t & 31;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 32; a++) acc += zs[sI * 32 + a] * ms[c * 33 + a];
  tout[n * 96 + t] = acc;
}

int small_fail() { return sfail(SGRL_ERR_HIP, std::string("small-batch product: ") + sgrl_train_last_error()); }

int small_gemm(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
               int N, int K, int flags = 0, const float* rowdiv = nullptr) {
  if (flags & ~(EPI_RELU | EPI_ROWDIV)) return sfail(SGRL_ERR_ARG, "small gemm: unsupported epilogue");
  const int rc = sgrl_linear_forward(A, lda, W, ldw, bias, (flags & EPI_ROWDIV) ? rowdiv : nullptr, C, ldc, M, N, K,
                                     (flags & EPI_RELU) ? 1 : 0, st);
  return rc == SGRL_OK ? rc : small_fail();
}

// critic = false: actions[e, 0:3L] = max_action * tanh(actor(obs)).  critic = true: `action` holds the per-limb action
// slots that complete the critic's input rows and `act` receives the per-limb Q values (row stride act_ld).
int run_forward(sgrl_set* s, const float* obs, int obs_ld, float* act, int act_ld, float max_action, hipStream_t st,
                bool critic = false, const float* action = nullptr, int action_ld = 0) {
  const int N = s->N, N3 = 3 * s->N;
  const int ngf = critic ? 20 : 17;
  g_gemm.form = s->gemm_form ? s->gemm_form : gemm_default_form();
  if (!s->live || !s->wwords) g_gemm.form = SGRL_SET_FORM_BF16X6;   // the two-piece form takes W pre-split by k_pack (bound parameters)
  // SGRL_SET_GEMM=f32: the plain products run on the exact-f32 kernels and nobody encodes the weights' words -- the generated-operand
  // products (Gram, equivariant, LayerNorm epilogue), which are always split kernels, must then take the three-piece form, which
  // splits the f32 weights itself.  (Rounds 4-5 left them on the two-piece form in that mode: they read words that had never been
  // written, and every forward of 2 048 nodes or more under SGRL_SET_GEMM=f32 returned garbage -- found in round 6 through the
  // "control" arm of the learning A/B, tools/diag/stale_pack_probe.py; tests/test_set_gpu.py now runs the mode.)
  if (!gemm_use_split()) g_gemm.form = SGRL_SET_FORM_BF16X6;
  g_gemm.w_base = s->w;
  g_gemm.w_words = s->wwords;
  g_gemm.wsc = s->wsc;
  g_gemm.enc_index = &s->enc_index;
  NodeTab nt{s->d_node_env, s->d_node_limb, s->d_node_mnode, s->d_trav, s->TM};
  EnvTab et{s->d_env_off, s->d_env_L, s->d_env_relb};
  const bool small = N <= (s->small_nodes >= 0 ? s->small_nodes : small_nodes());
  // Independent GEMM chains go to the side stream: fork() makes it wait for everything issued so far on `st`, join()
  // makes `st` wait for it.  The chains' store-heavy epilogues and partial last tile waves overlap each other.
  // While `st` is being captured into a hipGraph (the TD3 update graphs, td3.GraphedUpdates: batches of 100 environments,
  // where every kernel is far too small to gain from overlap) everything stays on ONE stream: a graph with cross-stream
  // forks costs ~7 us of hipGraphLaunch CPU time per node on this ROCm, a single-stream graph ~0.4 us.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  static const bool serial_env = [] { const char* e = getenv("SGRL_SET_ONE_STREAM"); return e && e[0] == '1'; }();   // diagnostics: per-kernel times without overlap
  const bool one_stream = small || cap != hipStreamCaptureStatusNone || serial_env;
  if (!one_stream && sgrl_streams::enabled() && s->side_picks < 6 &&
      std::find(s->side_ok_for.begin(), s->side_ok_for.end(), st) == s->side_ok_for.end()) {
    // first forward on this caller stream: make sure the side stream sits on another hardware queue (stream_pick.h)
    s->side_picks++;
    hipStream_t chosen = sgrl_streams::pick({st}, s->side);
    if (chosen != s->side) s->side_ok_for.clear();     // a new side stream: nothing is known about the other callers
    s->side = chosen;
    s->side_ok_for.push_back(st);
  }
  hipStream_t sd = one_stream ? st : s->side;
  auto fork = [&]() { if (!one_stream) { (void)hipEventRecord(s->ev_fork, st); (void)hipStreamWaitEvent(sd, s->ev_fork, 0); } };
  auto join = [&]() { if (!one_stream) { (void)hipEventRecord(s->ev_join, sd); (void)hipStreamWaitEvent(st, s->ev_join, 0); } };
  (void)hipMemsetAsync(act, 0, sizeof(float) * (size_t)s->n_env * act_ld, st);
  // live weights: flat buffer (and the stacked projection operands in it) rebuilt from the parameters -- by every forward, unless
  // the caller holds the weights (sgrl_set_hold_weights) and this handle has packed them in the same product form since
  const bool want_words = !small && g_gemm.form == SGRL_SET_FORM_F16X3 && gemm_use_split();
  const bool reuse = s->live && s->hold && s->packed_ok && (!want_words || s->packed_form == SGRL_SET_FORM_F16X3) && cap == hipStreamCaptureStatusNone;
  if (reuse) (void)hipStreamWaitEvent(st, s->ev_pack, 0);      // (a forward on another stream than the one that packed)
  if (s->live && !reuse)
    hipLaunchKernelGGL(k_pack, dim3(s->n_chunks), dim3(256), 0, st, s->d_segs, s->d_chunks, s->d_tri, s->wflat);
  const bool encode = s->live && !reuse && want_words;
  if (encode) {          // the product matrices as row-scaled words: beside the embedding, which reads the f32 buffer only
    fork();
    hipLaunchKernelGGL(sgrl_gemm::k_encode_rows, dim3((s->enc_rows + 3) / 4), dim3(256), 0, sd, s->wflat, s->wwords, s->wsc, s->d_enc, s->n_enc,
                       s->enc_rows);
  }
  hipLaunchKernelGGL(k_relbias, dim3(s->n_morph), dim3(256), 0, st, s->d_rel, s->W(SGRL_SET_REL_W), s->W(SGRL_SET_REL_B),
                     s->d_relb, s->d_m_off, s->d_m_L, s->n_morph);
  hipLaunchKernelGGL(k_embed, dim3((N + kEmbedNodes - 1) / kEmbedNodes), dim3(128), 0, st, obs, obs_ld, action, action_ld, ngf, nt,
                     s->W(SGRL_SET_GENC), s->W(SGRL_SET_ENC_W), s->W(SGRL_SET_ENC_B), s->W(SGRL_SET_EMB0), s->W(SGRL_SET_EMB1),
                     s->W(SGRL_SET_EMB2), s->g, s->cat, s->outg, s->outng, s->gdir, s->zc, s->z2, N);
  if (encode) join();
  if (s->live && !reuse && cap == hipStreamCaptureStatusNone) {
    s->packed_ok = s->hold;
    s->packed_form = want_words ? SGRL_SET_FORM_F16X3 : SGRL_SET_FORM_BF16X6;
    if (s->hold) (void)hipEventRecord(s->ev_pack, st);
  }
  // cat = [invariants | ng] is double-buffered in the fused form: norm2 writes the next ng into the OTHER buffer, so that the two
  // readers of [inv | ng] behind the feed-forward site (linear1 -> linear2 -> norm2 on the side stream, linear3 -> linear4 on the
  // main one) never wait for each other
  float* catc = s->cat;
  float* cato = s->cat2;
  s->cat_cur = catc;
  float* ng = catc + 128;
  int rc = SGRL_OK;
  // back-to-back products as one kernel each: the tile path in its two-piece form on bound (pre-split) weights
  const bool chain = !small && chain_enabled() && gemm_use_split() && g_gemm.form == SGRL_SET_FORM_F16X3;
  // linear3 -> linear4 -> contraction (-> vector-stream update) as ONE kernel exists (chain_f16.h, EPI_EQUIV; tests hold it against
  // float64) but LOSES to the two launches: a 64-row workgroup streams the 1 MB of linear4 words out of L2 twice as often as the
  // 128-row tiles do (587 MB per launch) and its three-instruction k-steps are barrier-bound -- 151 us against 135 us
  // (profiles/r4_chain_lab_ffn.txt).  SGRL_SET_CHAIN_EQ=1 selects it for A/B runs (SGRL_SET_FUSE_UPDATE=0: k_equiv stays a launch).
  // linear3 -> ReLU -> linear4 -> contraction as ONE kernel (launch_chain_equiv) is written, tested in tools/chain_lab.hip and 12 % slower
  // than its two launches (profiles/r4_chain_lab_ffn.txt): not part of the forward; the head folded through linear2_m and the fused
  // residual update are (the A/B switches of rounds 4-5 are gone, their numbers are in LAB_LOG)
  constexpr bool chain_eq = false, head_fold = true, fuse_update = true;
  float* const scratch = s->qkv;      // small path: [N, 576] Gram triangle / [N, 1024] per-node matrices (spans qkv | vg)
#define G(...) do { rc = small ? small_gemm(st, __VA_ARGS__) : launch_gemm(st, __VA_ARGS__); if (rc != SGRL_OK) return rc; } while (0)
  auto gram_gemm = [&](const float* W_, const float* b_, float* C_, int ldc_, int N_) -> int {
    if (!small) return launch_gemm_gram(st, s->zc, W_, b_, C_, ldc_, N, N_, s->fn);
    hipLaunchKernelGGL(k_gram576, dim3((N + 3) / 4), dim3(256), 0, st, s->zc, s->d_tri, scratch, s->fn, N);
    return small_gemm(st, scratch, GK, W_, GK, b_, C_, ldc_, N, N_, GK, EPI_RELU);
  };
  // linear4 / linear2_m + equivariant contraction; linear2 + residual + LayerNorm
  auto equiv_gemm = [&](const float* A_, const float* W_, const float* b_) -> int {
    if (!small) return launch_gemm_equiv(st, A_, 256, W_, 256, b_, N, 256, s->fn, s->z2, s->mat);
    const int r = small_gemm(st, A_, 256, W_, 256, b_, scratch, 1024, N, 1024, 256, EPI_ROWDIV, s->fn);
    if (r != SGRL_OK) return r;
    hipLaunchKernelGGL(k_zmat_perm, dim3(N), dim3(128), 0, st, s->z2, scratch, s->mat);
    return SGRL_OK;
  };
#define GG(W_, b_, C_, ldc_, N_) do { rc = gram_gemm(W_, b_, C_, ldc_, N_); if (rc != SGRL_OK) return rc; } while (0)
  // proj + gram site: Z (and Z2) = X . [Wp; Wq]^T on the matrix cores (stacked, zero-padded weights), then the packed
  // Gram triangle per node
  auto site_w = [&](int site) -> const float* {   // site 6 (the head, Cpad 144) is last
    return s->live ? s->site_ptr[site] : s->wstack + (size_t)site * 64 * 128;
  };
  if (!s->live && (s->stack_dirty || s->stack_critic != critic)) {
    s->stack_critic = critic;
    auto sw = [&](int site) { return s->wstack + (size_t)site * 64 * 128; };
    for (int l = 0; l < SGRL_SET_LAYERS; l++) {
      hipLaunchKernelGGL(k_stack_proj, dim3(32), dim3(256), 0, st, s->WL(l, SGRL_SET_A_GPROJ), (const float*)nullptr, D, D, sw(2 * l));
      hipLaunchKernelGGL(k_stack_proj, dim3(32), dim3(256), 0, st, s->WL(l, SGRL_SET_F_GPROJ2), s->WL(l, SGRL_SET_F_GPROJ3), D, D, sw(2 * l + 1));
    }
    hipLaunchKernelGGL(k_stack_proj, dim3(36), dim3(256), 0, st, s->W(SGRL_SET_GGPROJ), critic ? (const float*)nullptr : s->W(SGRL_SET_GPROJ), 136, OGLD, sw(6));
    s->stack_dirty = false;
  }
  // projection site: Z (and Z2) = X . [Wp; Wq]^T on the matrix cores (stacked, zero-padded weights), written straight into
  // the compact rows zc / z2 the Gram GEMM and the equivariant epilogues read
  auto pg = [&](const float* X, int ldx, int K, int site, float* z2) -> int {
    if (K % 16 != 0 || (ldx & 3)) return sfail(SGRL_ERR_ARG, "projection: K must be a multiple of 16 and rows 16-byte aligned");
    if (small) {          // rows 0..29 / 32..61 of the stacked operand: two narrow products (columns 30, 31 hold gdir)
      int r = small_gemm(st, X, ldx, site_w(site), K, nullptr, s->zc, ZD, N3, 30, K);
      if (r == SGRL_OK && z2) r = small_gemm(st, X, ldx, site_w(site) + (size_t)32 * K, K, nullptr, z2, ZD, N3, 30, K);
      return r;
    }
    GemmArgs a{X, ldx, site_w(site), K, nullptr, s->zc, ZD, N3, z2 ? 64 : 32, K, EPI_ZSPLIT, nullptr, z2, ZD};
    if (g_gemm.form == SGRL_SET_FORM_F16X3 && gemm_use_split())      // 128 x 64 tiles, four waves, W pre-split
      hipLaunchKernelGGL(kProjH, dim3(((a.M + 127) / 128) * ((a.N + 63) / 64)), dim3(256), kProjHLds, st, with_words(a));
    else
      GemmKernels<EPI_ZSPLIT>::launch(st, a);
    return SGRL_OK;
  };
#define PG(...) do { rc = pg(__VA_ARGS__); if (rc != SGRL_OK) return rc; } while (0)
  const int lnb = (N + 3) / 4;
#define GS(...) do { rc = small ? small_gemm(sd, __VA_ARGS__) : launch_gemm(sd, __VA_ARGS__); if (rc != SGRL_OK) return rc; } while (0)
  for (int l = 0; l < SGRL_SET_LAYERS; l++) {
    // --- attention ---
    fork();
    GS(s->g, D, s->WL(l, SGRL_SET_VG_W), D, nullptr, s->vg, 256, N3, 256, D);          // U = g . (Wgo_h Wvg_h)^T, both heads
    if (chain) {
      rc = launch_site(st, s->g, D, D, site_w(2 * l), s->zc, nullptr, s->fn, s->WL(l, SGRL_SET_A_LG1_W), s->WL(l, SGRL_SET_A_LG1_B), 256,
                       s->WL(l, SGRL_SET_A_LG2_W), s->WL(l, SGRL_SET_A_LG2_B), catc, 256, N);
      if (rc != SGRL_OK) return rc;
    } else {
      PG(s->g, D, D, 2 * l, nullptr);
      GG(s->WL(l, SGRL_SET_A_LG1_W), s->WL(l, SGRL_SET_A_LG1_B), s->h256, 256, 256);
      G(s->h256, 256, s->WL(l, SGRL_SET_A_LG2_W), 256, s->WL(l, SGRL_SET_A_LG2_B), catc, 256, N, 128, 256);
    }
    G(catc, 256, s->WL(l, SGRL_SET_QKV_W), 256, s->WL(l, SGRL_SET_QKV_B), s->qkv, 768, N, 768, 256, EPI_ROWDIV, s->fn);
    join();
    hipLaunchKernelGGL(k_attention, dim3(s->n_env), dim3(256), 0, st, s->qkv, s->vg, s->gdir, s->d_relb, et, l == 0 ? 1 : 0,
                       s->WL(l, SGRL_SET_NGOUT_B), s->WL(l, SGRL_SET_A_GD), s->stop_after == 2 * l ? s->delta : (float*)nullptr,
                       s->g1, ng, 256, s->WL(l, SGRL_SET_N1_W), s->WL(l, SGRL_SET_N1_B));
    if (s->stop_after == 2 * l) return SGRL_OK;      // probe: g1 = attention's vector output, delta = its scalar output
    // --- equivariant feed-forward ---
    if (chain) {
      rc = launch_site(st, s->g1, D, D, site_w(2 * l + 1), s->zc, s->z2, s->fn, s->WL(l, SGRL_SET_F_LG1_W), s->WL(l, SGRL_SET_F_LG1_B), 256,
                       s->WL(l, SGRL_SET_F_LG2_W), s->WL(l, SGRL_SET_F_LG2_B), catc, 256, N);
      if (rc != SGRL_OK) return rc;
    } else {
      PG(s->g1, D, D, 2 * l + 1, s->z2);
      GG(s->WL(l, SGRL_SET_F_LG1_W), s->WL(l, SGRL_SET_F_LG1_B), s->h256, 256, 256);
      G(s->h256, 256, s->WL(l, SGRL_SET_F_LG2_W), 256, s->WL(l, SGRL_SET_F_LG2_B), catc, 256, N, 128, 256);
    }
    fork();
    if (chain) {
      // side: linear1 -> ReLU -> linear2 -> / fn -> residual + norm2, the new ng into the other buffer; main: linear3 -> ReLU ->
      // linear4 -> contraction with z, then the update of the vector stream
      rc = launch_chain_ln(sd, catc, 256, 256, s->WL(l, SGRL_SET_L1_W), s->WL(l, SGRL_SET_L1_B), s->WL(l, SGRL_SET_L2_W), s->WL(l, SGRL_SET_L2_B),
                           s->fn, ng, 256, s->WL(l, SGRL_SET_N2_W), s->WL(l, SGRL_SET_N2_B), N, cato + 128);
      if (rc == SGRL_OK && chain_eq) {
        rc = launch_chain_equiv(st, catc, 256, 256, s->WL(l, SGRL_SET_L3_W), s->WL(l, SGRL_SET_L3_B), s->WL(l, SGRL_SET_L4_W), s->WL(l, SGRL_SET_L4_B),
                                s->fn, s->z2, s->mat, N, fuse_update ? s->g : (float*)nullptr, s->g1, s->WL(l, SGRL_SET_L5_W),
                                l == SGRL_SET_LAYERS - 1 ? s->outg : (float*)nullptr);
      } else if (rc == SGRL_OK) {
        rc = launch_gemm(st, catc, 256, s->WL(l, SGRL_SET_L3_W), 256, s->WL(l, SGRL_SET_L3_B), s->t256, 256, N, 256, 256, EPI_RELU);
        if (rc == SGRL_OK) rc = equiv_gemm(s->t256, s->WL(l, SGRL_SET_L4_W), s->WL(l, SGRL_SET_L4_B));
      }
      if (rc != SGRL_OK) return rc;
      if (!(chain_eq && fuse_update))
        hipLaunchKernelGGL(k_equiv, dim3((N + 7) / 8), dim3(128), 0, st, s->mat, s->WL(l, SGRL_SET_L5_W), s->g1, s->g,
                           l == SGRL_SET_LAYERS - 1 ? s->outg : (float*)nullptr, N);
      join();
      std::swap(catc, cato);
      ng = catc + 128;
      s->cat_cur = catc;
      if (s->stop_after == 2 * l + 1) return SGRL_OK;  // probe: g / ng (= the current cat[:, 128:]) are this layer's outputs
      continue;
    }
    GS(s->cat, 256, s->WL(l, SGRL_SET_L1_W), 256, s->WL(l, SGRL_SET_L1_B), s->t256b, 256, N, 256, 256, EPI_RELU);
    G(s->cat, 256, s->WL(l, SGRL_SET_L3_W), 256, s->WL(l, SGRL_SET_L3_B), s->t256, 256, N, 256, 256, EPI_RELU);
    // linear2 carries the scalar stream's second residual + norm2 in its epilogue (ng rewritten in place): it must not start
    // before linear3 -- the other reader of cat = [inv | ng] -- is done
    if (!one_stream) {
      (void)hipEventRecord(s->ev_l3, st);
      (void)hipStreamWaitEvent(sd, s->ev_l3, 0);
    }
    if (small) {
      rc = small_gemm(st, s->t256b, 256, s->WL(l, SGRL_SET_L2_W), 256, s->WL(l, SGRL_SET_L2_B), s->delta, 128, N, 128, 256, EPI_ROWDIV, s->fn);
      if (rc != SGRL_OK) return rc;
      hipLaunchKernelGGL(k_add_ln, dim3(lnb), dim3(256), 0, st, ng, 256, s->delta, 128, s->WL(l, SGRL_SET_N2_W),
                         s->WL(l, SGRL_SET_N2_B), (float*)nullptr, 0, ng, 256, N);
    } else {
      rc = launch_gemm_ln(sd, s->t256b, 256, s->WL(l, SGRL_SET_L2_W), 256, s->WL(l, SGRL_SET_L2_B), N, 256, s->fn, ng, 256,
                          s->WL(l, SGRL_SET_N2_W), s->WL(l, SGRL_SET_N2_B));
      if (rc != SGRL_OK) return rc;
    }
    rc = equiv_gemm(s->t256, s->WL(l, SGRL_SET_L4_W), s->WL(l, SGRL_SET_L4_B));
    if (rc != SGRL_OK) return rc;
    hipLaunchKernelGGL(k_equiv, dim3((N + 7) / 8), dim3(128), 0, st, s->mat, s->WL(l, SGRL_SET_L5_W), s->g1, s->g,
                       l == SGRL_SET_LAYERS - 1 ? s->outg : (float*)nullptr, N);
    join();
    if (s->stop_after == 2 * l + 1) return SGRL_OK;  // probe: g / ng (= cat[:, 128:]) are this layer's outputs
  }
  // final norm -> outng[:, 17:145]; head
  hipLaunchKernelGGL(k_add_ln, dim3(lnb), dim3(256), 0, st, ng, 256, (const float*)nullptr, 0, s->W(SGRL_SET_FNORM_W),
                     s->W(SGRL_SET_FNORM_B), (float*)nullptr, 0, s->outng + ngf, 160, N);
  float* const hd = chain ? cato : s->cat2;     // the head's [l2g | l2ng] rows: the buffer the last norm2 did not write
  fork();
  if (chain) {
    rc = launch_chain_ng(sd, s->outng, 160, 160, s->W(SGRL_SET_L1NG_W), s->W(SGRL_SET_L1NG_B), s->W(SGRL_SET_L2NG_W), s->W(SGRL_SET_L2NG_B),
                         hd + 128, 256, N);
    if (rc == SGRL_OK)
      rc = launch_site(st, s->outg, OGLD, OGLD, site_w(6), s->zc, critic ? (float*)nullptr : s->z2, s->fn, s->W(SGRL_SET_L1G_W),
                       s->W(SGRL_SET_L1G_B), 128, s->W(SGRL_SET_L2G_W), s->W(SGRL_SET_L2G_B), hd, 256, N);
    if (rc != SGRL_OK) return rc;
  } else {
    GS(s->outng, 160, s->W(SGRL_SET_L1NG_W), 160, s->W(SGRL_SET_L1NG_B), s->t128b, D, N, D, 160, EPI_RELU);
    GS(s->t128b, D, s->W(SGRL_SET_L2NG_W), D, s->W(SGRL_SET_L2NG_B), s->cat2 + 128, 256, N, D, D);
    PG(s->outg, OGLD, OGLD, 6, critic ? (float*)nullptr : s->z2);
    GG(s->W(SGRL_SET_L1G_W), s->W(SGRL_SET_L1G_B), s->t128a, D, D);
    G(s->t128a, D, s->W(SGRL_SET_L2G_W), D, s->W(SGRL_SET_L2G_B), s->cat2, 256, N, D, D);
  }
  join();
  if (critic) {
    // slots reused by the critic head: DECG = decoder_ng.weight [256], L1M_B = decoder_ng.bias [1]
    hipLaunchKernelGGL(k_q_head, dim3((N + 3) / 4), dim3(256), 0, st, hd, s->W(SGRL_SET_DECG), s->W(SGRL_SET_L1M_B), s->fn,
                       nt, act, act_ld, N);
  } else {
    const bool folded = !small && s->live && s->l2mf_w && head_fold;
    if (folded) {
      // decoder_g folded through linear2_m: the 1024-wide product and its contraction collapse into a 32-wide product
      G(hd, 256, s->W(SGRL_SET_L1M_W), 256, s->W(SGRL_SET_L1M_B), s->t256, 256, N, 256, 256, EPI_RELU);
      G(s->t256, 256, s->l2mf_w, 256, s->l2mf_b, s->mat, 32, N, 32, 256, EPI_ROWDIV, s->fn);
      hipLaunchKernelGGL(k_head_out2, dim3((N + 3) / 4), dim3(128), 0, st, s->mat, s->z2, obs, obs_ld, nt, act, act_ld, max_action, N);
    } else if (chain && chain_eq) {
      rc = launch_chain_equiv(st, hd, 256, 256, s->W(SGRL_SET_L1M_W), s->W(SGRL_SET_L1M_B), s->W(SGRL_SET_L2M_W), s->W(SGRL_SET_L2M_B), s->fn, s->z2,
                              s->mat, N);
    } else {
      G(hd, 256, s->W(SGRL_SET_L1M_W), 256, s->W(SGRL_SET_L1M_B), s->t256, 256, N, 256, 256, EPI_RELU);
      rc = equiv_gemm(s->t256, s->W(SGRL_SET_L2M_W), s->W(SGRL_SET_L2M_B));
    }
    if (rc != SGRL_OK) return rc;
    if (!folded)
      hipLaunchKernelGGL(k_head_out, dim3((N + 3) / 4), dim3(128), 0, st, s->mat, s->W(SGRL_SET_DECG), obs, obs_ld, nt,
                         act, act_ld, max_action, N);
  }
#undef GS
#undef GG
#undef G
#undef PG
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess)
      return sfail(SGRL_ERR_HIP, std::string("kernel launch failed in sgrl_set_forward (") + hipGetErrorName(le) + ": " + hipGetErrorString(le) + ")");
  }
  return SGRL_OK;
}

}  // namespace

extern "C" {

int sgrl_set_create(sgrl_set** out) {
  if (!out) return sfail(SGRL_ERR_ARG, "out is null");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    *out = nullptr;
    return sfail(SGRL_ERR_HIP, "no HIP device visible: the SET actor fast path needs an MI355X (there is no CPU fallback)");
  }
  const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kGemmEquiv), hipFuncAttributeMaxDynamicSharedMemorySize, kEquivLds) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(kGemmGram), hipFuncAttributeMaxDynamicSharedMemorySize, GemmKernels<0>::kSplitLds) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(kGemmLn), hipFuncAttributeMaxDynamicSharedMemorySize, GemmKernels<0>::kSplitLds) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(kGemmEquivH), hipFuncAttributeMaxDynamicSharedMemorySize, kEquivLds) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(kGemmGramH), hipFuncAttributeMaxDynamicSharedMemorySize, GemmKernels<0>::kSplitHLds) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(kGemmLnH), hipFuncAttributeMaxDynamicSharedMemorySize, GemmKernels<0>::kSplitHLds) == hipSuccess &&
                       GemmKernels<0>::raise_lds_limits() && GemmKernels<EPI_RELU>::raise_lds_limits() &&
                       GemmKernels<EPI_ROWDIV>::raise_lds_limits() && GemmKernels<EPI_ACC2>::raise_lds_limits() &&
                       GemmKernels<EPI_ZSPLIT>::raise_lds_limits() && chain_raise_lds_limits();
  if (!attr_ok) {
    *out = nullptr;
    return sfail(SGRL_ERR_HIP, "cannot raise the dynamic LDS limit of the GEMM kernel");
  }
  sgrl_set* s = new sgrl_set();
  // index table of the blocked Gram triangle (gemm_f32.h GRAM): k = 16 (A (A + 1) / 2 + B) + 4 i + j  ->  (a << 8 | b) with
  // a = 4 A + i, b = 4 B + j; the entries above the diagonal inside a diagonal block carry no weight: marked 0xFFFF
  std::vector<unsigned short> tri(GK, (unsigned short)0xFFFF);
  for (int A = 0, o = 0; A < ZD / 4; A++)
    for (int B = 0; B <= A; B++)
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++, o++)
          if (4 * A + i >= 4 * B + j) tri[o] = (unsigned short)(((4 * A + i) << 8) | (4 * B + j));
  const size_t wstack_floats = 6 * 64 * 128 + 64 * OGLD;
  if (hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_l3, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_pack, hipEventDisableTiming) != hipSuccess) {
    delete s;
    *out = nullptr;
    return sfail(SGRL_ERR_HIP, "cannot create the side stream of the SET actor");
  }
  if (hipMalloc(&s->wstack, sizeof(float) * wstack_floats) != hipSuccess || hipMalloc(&s->d_tri, sizeof(unsigned short) * GK) != hipSuccess ||
      hipMemcpy(s->d_tri, tri.data(), sizeof(unsigned short) * GK, hipMemcpyHostToDevice) != hipSuccess) {
    if (s->wstack) (void)hipFree(s->wstack);
    if (s->d_tri) (void)hipFree(s->d_tri);
    delete s;
    *out = nullptr;
    return sfail(SGRL_ERR_HIP, "device allocation failed in sgrl_set_create");
  }
  *out = s;
  return SGRL_OK;
}

void sgrl_set_destroy(sgrl_set* s) {
  if (!s) return;
  if (s->side) (void)hipStreamSynchronize(s->side);
  free_graphs(s);
  if (s->wstack) (void)hipFree(s->wstack);
  if (s->d_tri) (void)hipFree(s->d_tri);
  if (s->d_enc) (void)hipFree(s->d_enc);
  if (s->wsc) (void)hipFree(s->wsc);
  if (s->wflat) (void)hipFree(s->wflat);
  if (s->wwords) (void)hipFree(s->wwords);
  if (s->d_segs) (void)hipFree(s->d_segs);
  if (s->d_chunks) (void)hipFree(s->d_chunks);
  if (s->side) (void)hipStreamDestroy(s->side);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->ev_l3) (void)hipEventDestroy(s->ev_l3);
  if (s->ev_pack) (void)hipEventDestroy(s->ev_pack);
  delete s;
}

int sgrl_set_weights(sgrl_set* s, const float* w, const int64_t* offsets, int n_offsets) {
  if (!s || !w || !offsets || n_offsets != SGRL_SET_NW) return sfail(SGRL_ERR_ARG, "sgrl_set_weights: bad argument");
  s->w = w;
  std::memcpy(s->off, offsets, sizeof(int64_t) * SGRL_SET_NW);
  s->have_w = true;
  s->live = false;
  s->l2mf_w = nullptr; s->l2mf_b = nullptr;
  s->stack_dirty = true;   // the stacked projection operands are rebuilt by the next forward, on its stream
  return SGRL_OK;
}

int sgrl_set_bind_params(sgrl_set* s, const sgrl_pack_seg* segs, int n_segs, const int64_t* offsets, int n_offsets,
                         int64_t total_floats) {
  if (!s || !segs || n_segs <= 0 || !offsets || n_offsets != SGRL_SET_NW + SGRL_SET_NSITES + SGRL_SET_NEXTRA || total_floats <= 0)
    return sfail(SGRL_ERR_ARG, "sgrl_set_bind_params: bad argument");
  // the segments must tile [0, total_floats) exactly (sorted by dst): every float of the buffer is rewritten per forward
  std::vector<int> order(n_segs);
  for (int i = 0; i < n_segs; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return segs[x].dst < segs[y].dst; });
  int64_t pos = 0;
  std::vector<int2> chunks;
  for (int oi = 0; oi < n_segs; oi++) {
    const sgrl_pack_seg& g = segs[order[oi]];
    if (g.dst != pos || g.n <= 0 || !g.src0) return sfail(SGRL_ERR_ARG, "sgrl_set_bind_params: segments must tile the buffer without gaps");
    bool ok = false;
    switch (g.kind) {
      case SGRL_PACK_COPY: ok = g.a >= 0 && g.a <= g.n; break;
      case SGRL_PACK_PADCOL: ok = g.a > 0 && g.b >= g.a && g.n % g.b == 0; break;
      case SGRL_PACK_FOLD: ok = g.n % GK == 0; break;
      case SGRL_PACK_STACK: ok = g.a > 0 && g.b >= g.a && g.n == 64 * g.b; break;
      case SGRL_PACK_MATMUL: ok = g.src1 && g.a > 0 && g.b > 0 && g.n % g.b == 0 && g.lda >= g.a && g.ldb >= g.b; break;
      case SGRL_PACK_SUBMAT: ok = g.b > 0 && g.n % g.b == 0 && g.lda >= g.b; break;
      case SGRL_PACK_PERM32: ok = g.a > 0 && g.n == 1024 * g.a; break;
      default: break;
    }
    if (!ok) return sfail(SGRL_ERR_ARG, "sgrl_set_bind_params: inconsistent segment " + std::to_string(order[oi]));
    for (int st = 0; st < g.n; st += (g.kind == SGRL_PACK_MATMUL ? PACK_CHUNK_MM : PACK_CHUNK)) chunks.push_back(make_int2(order[oi], st));
    pos += g.n;
  }
  if (pos != total_floats) return sfail(SGRL_ERR_ARG, "sgrl_set_bind_params: segments do not add up to total_floats");
  for (int k = 0; k < n_offsets; k++)
    if (offsets[k] < 0 || offsets[k] >= total_floats || (offsets[k] & 3)) return sfail(SGRL_ERR_ARG, "sgrl_set_bind_params: bad offset");
  // replace the previous binding (a forward may still be reading it: hipFree waits for the device)
  if (s->wflat) s->generation++;
  if (s->wflat) (void)hipFree(s->wflat);
  if (s->wwords) (void)hipFree(s->wwords);
  if (s->d_segs) (void)hipFree(s->d_segs);
  if (s->d_chunks) (void)hipFree(s->d_chunks);
  s->wwords = nullptr;
  s->packed_ok = false;
  s->wflat = nullptr; s->d_segs = nullptr; s->d_chunks = nullptr; s->live = false; s->have_w = false;
  if (hipMalloc(&s->wflat, sizeof(float) * total_floats) != hipSuccess ||
      hipMalloc(&s->wwords, sizeof(unsigned) * total_floats) != hipSuccess ||
      hipMalloc(&s->d_segs, sizeof(sgrl_pack_seg) * n_segs) != hipSuccess ||
      hipMalloc(&s->d_chunks, sizeof(int2) * chunks.size()) != hipSuccess ||
      hipMemcpy(s->d_segs, segs, sizeof(sgrl_pack_seg) * n_segs, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->d_chunks, chunks.data(), sizeof(int2) * chunks.size(), hipMemcpyHostToDevice) != hipSuccess)
    return sfail(SGRL_ERR_HIP, "device allocation failed in sgrl_set_bind_params");
  s->wflat_floats = total_floats;
  {
    // the product matrices of the buffer (slot shapes: include/sgrl_set.h), for k_encode_rows.  A slot the bound network does not
    // fill (the critic has no linear1_m / linear2_m) is shorter than its matrix and is left out.
    std::vector<int64_t> bounds(offsets, offsets + n_offsets);
    bounds.push_back(total_floats);
    std::sort(bounds.begin(), bounds.end());
    std::vector<sgrl_gemm::EncMat> mats;
    s->enc_index.clear();
    int rows_total = 0;
    auto add = [&](int64_t off, int rows, int K, int group = 1) {
      const int64_t next = *std::upper_bound(bounds.begin(), bounds.end(), off);
      if (next - off < (int64_t)rows * K) return;
      mats.push_back(sgrl_gemm::EncMat{off, rows, K, rows_total, group});
      s->enc_index.emplace_back(off, rows_total);
      rows_total += rows;
    };
    add(offsets[SGRL_SET_L1G_W], 128, GK); add(offsets[SGRL_SET_L2G_W], 128, 128);
    add(offsets[SGRL_SET_L1NG_W], 128, 160); add(offsets[SGRL_SET_L2NG_W], 128, 128);
    add(offsets[SGRL_SET_L1M_W], 256, 256); add(offsets[SGRL_SET_L2M_W], 1024, 256, 32);
    for (int l = 0; l < SGRL_SET_LAYERS; l++) {
      const int64_t* o = offsets + SGRL_SET_NGLOBAL + l * SGRL_SET_NLAYER;
      add(o[SGRL_SET_A_LG1_W], 256, GK); add(o[SGRL_SET_A_LG2_W], 128, 256); add(o[SGRL_SET_QKV_W], 768, 256); add(o[SGRL_SET_VG_W], 256, 128);
      add(o[SGRL_SET_F_LG1_W], 256, GK); add(o[SGRL_SET_F_LG2_W], 128, 256); add(o[SGRL_SET_L3_W], 256, 256); add(o[SGRL_SET_L4_W], 1024, 256, 32);
      add(o[SGRL_SET_L1_W], 256, 256); add(o[SGRL_SET_L2_W], 128, 256); add(o[SGRL_SET_L5_W], 128, 32);
    }
    for (int k = 0; k < SGRL_SET_NSITES; k++) add(offsets[SGRL_SET_NW + k], 64, k == 6 ? OGLD : 128);
    if (s->d_enc) (void)hipFree(s->d_enc);
    if (s->wsc) (void)hipFree(s->wsc);
    s->d_enc = nullptr; s->wsc = nullptr;
    s->n_enc = (int)mats.size(); s->enc_rows = rows_total;
    if (hipMalloc(&s->d_enc, sizeof(sgrl_gemm::EncMat) * mats.size()) != hipSuccess || hipMalloc(&s->wsc, sizeof(float) * rows_total) != hipSuccess ||
        hipMemcpy(s->d_enc, mats.data(), sizeof(sgrl_gemm::EncMat) * mats.size(), hipMemcpyHostToDevice) != hipSuccess)
      return sfail(SGRL_ERR_HIP, "device allocation failed in sgrl_set_bind_params (row-scale table)");
  }
  s->n_chunks = (int)chunks.size();
  s->w = s->wflat;
  std::memcpy(s->off, offsets, sizeof(int64_t) * SGRL_SET_NW);
  for (int k = 0; k < SGRL_SET_NSITES; k++) s->site_ptr[k] = s->wflat + offsets[SGRL_SET_NW + k];
  {
    // the folded head exists when its slot is long enough for the [32, 256] matrix (an actor network; the critic binds a filler)
    std::vector<int64_t> b2(offsets, offsets + n_offsets);
    b2.push_back(total_floats);
    std::sort(b2.begin(), b2.end());
    const int64_t ow = offsets[SGRL_SET_NW + SGRL_SET_NSITES], ob = offsets[SGRL_SET_NW + SGRL_SET_NSITES + 1];
    const bool have = *std::upper_bound(b2.begin(), b2.end(), ow) - ow >= 32 * 256;
    s->l2mf_w = have ? s->wflat + ow : nullptr;
    s->l2mf_b = have ? s->wflat + ob : nullptr;
  }
  s->live = true;
  s->have_w = true;
  return SGRL_OK;
}

int sgrl_set_graph(sgrl_set* s, int n_morph, const int32_t* morph_L, const int32_t* morph_count, const int32_t* trav,
                   const float* rel) {
  if (!s || n_morph <= 0 || !morph_L || !morph_count || !trav || !rel) return sfail(SGRL_ERR_ARG, "sgrl_set_graph: bad argument");
  // content key of the request
  std::vector<int32_t> key_i;
  size_t ntrav = 0, nrel = 0;
  for (int k = 0; k < n_morph; k++) {
    if (morph_L[k] < 2 || morph_L[k] > 14) return sfail(SGRL_ERR_LIMIT, "limb count must be in [2, 14]");
    if (morph_count[k] < 0) return sfail(SGRL_ERR_ARG, "negative morph_count");
    ntrav += 3 * (size_t)morph_L[k];
    nrel += 3 * (size_t)morph_L[k] * morph_L[k];
  }
  key_i.push_back(n_morph);
  key_i.insert(key_i.end(), morph_L, morph_L + n_morph);
  key_i.insert(key_i.end(), morph_count, morph_count + n_morph);
  key_i.insert(key_i.end(), trav, trav + ntrav);
  for (GraphCfg* c : s->cfgs)
    if (c->key_i == key_i && c->key_f.size() == nrel && std::memcmp(c->key_f.data(), rel, sizeof(float) * nrel) == 0)
      return use_cfg(s, c);        // seen before: no allocation, upload or synchronisation
  // new batch structure: build and upload its tables once
  std::vector<int32_t> node_env, node_limb, node_mnode, env_off, env_L, env_relb, m_off, m_L, travT;
  int TM = 0, relb_off = 0, env = 0, node = 0, Lmax = 0;
  std::vector<int> m_node0;
  for (int k = 0; k < n_morph; k++) {
    m_node0.push_back(TM);
    m_off.push_back(relb_off);
    m_L.push_back(morph_L[k]);
    TM += morph_L[k];
    relb_off += 2 * morph_L[k] * morph_L[k];
    if (morph_L[k] > Lmax) Lmax = morph_L[k];
  }
  travT.assign(3 * (size_t)TM, 0);
  {
    size_t tp = 0;
    for (int k = 0; k < n_morph; k++) {
      const int L = morph_L[k];
      for (int q = 0; q < 3; q++)
        for (int i = 0; i < L; i++) {
          const int v = trav[tp + q * L + i];
          if (v < 0 || v >= 15) return sfail(SGRL_ERR_ARG, "traversal index out of range");
          travT[(size_t)q * TM + m_node0[k] + i] = v;
        }
      tp += 3 * (size_t)L;
    }
  }
  for (int k = 0; k < n_morph; k++) {
    for (int c = 0; c < morph_count[k]; c++) {
      env_off.push_back(node);
      env_L.push_back(morph_L[k]);
      env_relb.push_back(m_off[k]);
      for (int i = 0; i < morph_L[k]; i++) {
        node_env.push_back(env);
        node_limb.push_back(i);
        node_mnode.push_back(m_node0[k] + i);
        node++;
      }
      env++;
    }
  }
  if (node == 0) return sfail(SGRL_ERR_ARG, "no environments");
  if ((int)s->cfgs.size() >= SGRL_SET_GRAPH_CACHE) {      // evict the least recently used structure
    size_t lru = 0;
    for (size_t i = 1; i < s->cfgs.size(); i++) if (s->cfgs[i]->last_use < s->cfgs[lru]->last_use) lru = i;
    s->cfgs[lru]->release();                              // hipFree waits for the device
    s->generation++;
    delete s->cfgs[lru];
    s->cfgs.erase(s->cfgs.begin() + lru);
  }
  GraphCfg* c = new GraphCfg();
  c->key_i = std::move(key_i);
  c->key_f.assign(rel, rel + nrel);
  c->n_env = env; c->N = node; c->n_morph = n_morph; c->TM = TM; c->Lmax = Lmax;
  bool ok = upload(&c->d_node_env, node_env) == 0 && upload(&c->d_node_limb, node_limb) == 0 &&
            upload(&c->d_node_mnode, node_mnode) == 0 && upload(&c->d_trav, travT) == 0 &&
            upload(&c->d_env_off, env_off) == 0 && upload(&c->d_env_L, env_L) == 0 && upload(&c->d_env_relb, env_relb) == 0 &&
            upload(&c->d_m_off, m_off) == 0 && upload(&c->d_m_L, m_L) == 0 && upload(&c->d_rel, c->key_f) == 0;
  if (ok) ok = hipMalloc(&c->d_relb, sizeof(float) * relb_off) == hipSuccess;
  if (!ok) { c->release(); delete c; return sfail(SGRL_ERR_HIP, "device allocation failed in sgrl_set_graph"); }
  s->cfgs.push_back(c);
  return use_cfg(s, c);
}

int sgrl_set_forward(sgrl_set* s, const float* obs, int obs_ld, float* act, int act_ld, float max_action, void* stream) {
  if (!s || !obs || !act) return sfail(SGRL_ERR_ARG, "sgrl_set_forward: null argument");
  if (!s->have_w || !s->have_graph) return sfail(SGRL_ERR_ARG, "sgrl_set_forward: weights or graph not set");
  if (obs_ld < 41 * s->Lmax || act_ld < 3 * s->Lmax)
    return sfail(SGRL_ERR_ARG, "sgrl_set_forward: obs_ld < 41 * Lmax or act_ld < 3 * Lmax (rows too narrow for the largest morphology)");
  return run_forward(s, obs, obs_ld, act, act_ld, max_action, (hipStream_t)stream);
}

int sgrl_set_forward_q(sgrl_set* s, const float* obs, int obs_ld, const float* action, int action_ld, float* q, int q_ld,
                       void* stream) {
  if (!s || !obs || !action || !q) return sfail(SGRL_ERR_ARG, "sgrl_set_forward_q: null argument");
  if (!s->have_w || !s->have_graph) return sfail(SGRL_ERR_ARG, "sgrl_set_forward_q: weights or graph not set");
  if (obs_ld < 41 * s->Lmax || action_ld < 3 * s->Lmax || q_ld < s->Lmax)
    return sfail(SGRL_ERR_ARG, "sgrl_set_forward_q: obs_ld < 41 * Lmax, action_ld < 3 * Lmax or q_ld < Lmax");
  return run_forward(s, obs, obs_ld, q, q_ld, 0.f, (hipStream_t)stream, true, action, action_ld);
}

int sgrl_set_time_forward(sgrl_set* s, const float* obs, int obs_ld, float* act, int act_ld, float max_action,
                          int reps, void* stream, float* ms_out) {
  if (!s || !obs || !act || !ms_out || reps <= 0) return sfail(SGRL_ERR_ARG, "sgrl_set_time_forward: bad argument");
  if (!s->have_w || !s->have_graph) return sfail(SGRL_ERR_ARG, "weights or graph not set");
  if (obs_ld < 41 * s->Lmax || act_ld < 3 * s->Lmax) return sfail(SGRL_ERR_ARG, "sgrl_set_time_forward: rows too narrow for the largest morphology");
  hipEvent_t t0, t1;
  SHIP_TRY(hipEventCreate(&t0));
  SHIP_TRY(hipEventCreate(&t1));
  SHIP_TRY(hipEventRecord(t0, (hipStream_t)stream));
  for (int r = 0; r < reps; r++) {
    int rc = run_forward(s, obs, obs_ld, act, act_ld, max_action, (hipStream_t)stream);
    if (rc != SGRL_OK) return rc;
  }
  SHIP_TRY(hipEventRecord(t1, (hipStream_t)stream));
  SHIP_TRY(hipEventSynchronize(t1));
  float ms = 0;
  SHIP_TRY(hipEventElapsedTime(&ms, t0, t1));
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  *ms_out = ms / reps;
  return SGRL_OK;
}

int sgrl_set_hold_weights(sgrl_set* s, int hold) {
  if (!s) return sfail(SGRL_ERR_ARG, "sgrl_set_hold_weights: null handle");
  s->hold = hold != 0;
  s->packed_ok = false;          // every call is also "the weights may have changed just now": the next forward packs
  return SGRL_OK;
}

int sgrl_set_num_nodes(const sgrl_set* s) { return s ? s->N : SGRL_ERR_ARG; }
int64_t sgrl_set_workspace_bytes(const sgrl_set* s) { return s ? s->ws_floats * 4 : -1; }
int64_t sgrl_set_generation(const sgrl_set* s) { return s ? s->generation : -1; }

int sgrl_set_peek(sgrl_set* s, int which, float* host, int64_t n_floats) {
  if (!s || !host || !s->have_graph) return sfail(SGRL_ERR_ARG, "sgrl_set_peek: bad argument");
  const float* src[] = {s->g, s->cat_cur ? s->cat_cur : s->cat, s->zc, s->fn, s->qkv, nullptr, nullptr, s->mat, s->g1, s->delta, s->outng};
  const int64_t per[] = {384, 256, 96, 1, 768, 0, 0, 96, 384, 128, 160};
  if (which < 0 || which > 10 || !src[which] || n_floats > per[which] * s->N) return sfail(SGRL_ERR_ARG, "sgrl_set_peek: bad buffer or size");
  SHIP_TRY(hipDeviceSynchronize());
  SHIP_TRY(hipMemcpy(host, src[which], sizeof(float) * n_floats, hipMemcpyDeviceToHost));
  return SGRL_OK;
}

int sgrl_set_debug_stop_after(sgrl_set* s, int stage) {
  if (!s || stage < -1 || stage >= 2 * SGRL_SET_LAYERS) return sfail(SGRL_ERR_ARG, "sgrl_set_debug_stop_after: bad argument");
  s->stop_after = stage;
  return SGRL_OK;
}

int sgrl_set_debug_small_nodes(sgrl_set* s, int nodes) {
  if (!s || nodes < -1) return sfail(SGRL_ERR_ARG, "sgrl_set_debug_small_nodes: bad argument");
  s->small_nodes = nodes;
  return SGRL_OK;
}

int sgrl_set_gemm_form(sgrl_set* s, int form) {
  if (!s || (form != 0 && form != SGRL_SET_FORM_F16X3 && form != SGRL_SET_FORM_BF16X6)) return sfail(SGRL_ERR_ARG, "sgrl_set_gemm_form: bad argument");
  if (s->gemm_form != form) s->generation++;     // a captured graph holds the kernels of the previous form
  s->gemm_form = form;
  return SGRL_OK;
}

namespace {
// a caller-supplied weight matrix as k_encode_rows leaves the bound weights: row-scaled words + inverse scales (test hooks)
struct TempWords {
  unsigned* w = nullptr; float* sc = nullptr; sgrl_gemm::EncMat* dm = nullptr;
  ~TempWords() { if (w) (void)hipFree(w); if (sc) (void)hipFree(sc); if (dm) (void)hipFree(dm); }
  int make(hipStream_t st, const float* W, int rows, int K, int group = 1) {
    const sgrl_gemm::EncMat m{0, rows, K, 0, group};
    if (hipMalloc(&w, sizeof(unsigned) * (size_t)rows * K) != hipSuccess || hipMalloc(&sc, sizeof(float) * rows) != hipSuccess ||
        hipMalloc(&dm, sizeof(m)) != hipSuccess || hipMemcpy(dm, &m, sizeof(m), hipMemcpyHostToDevice) != hipSuccess)
      return sfail(SGRL_ERR_HIP, "debug product: allocation failed");
    hipLaunchKernelGGL(sgrl_gemm::k_encode_rows, dim3((rows + 3) / 4), dim3(256), 0, st, W, w, sc, dm, 1, rows);
    return SGRL_OK;
  }
};
}  // namespace

// Test hook (tests/test_split_products_gpu.py): ONE product through the production tile kernels on caller-supplied operands, so
// that every k_gemm3 instantiation the forward launches is held against float64 by `pytest -m gpu`, not only by a lab executable.
//   kind 0 plain | 1 ReLU | 2 row division (C = (A W' + b) / rowdiv)                 A [M, K], W [N, K], C [M, N]
//        3 Gram operand, ReLU (A = Z [M, 96]; W [N, 576] in the folded order; fn [M] out in aux_out)
//        4 equivariant epilogue (N = 1024: W [1024, K], rowdiv, zq [M, 96] in aux_in; tout [M, 96] out in C)
//        5 stacked projections (N = 64: C [M, 32] <- columns 0..29, aux_out [M, 32] <- columns 32..61)
//        6 residual + LayerNorm epilogue (N = 128: rowdiv; C [M, 128] is ln_io, read and rewritten; aux_in = ln_w | ln_b [256])
//   form SGRL_SET_FORM_F16X3 | SGRL_SET_FORM_BF16X6 | 1 = the exact-f32 matrix instruction (k_gemm2; kinds 0..2 only)
// Weights of the two-piece form are cut into row-scaled words here exactly as the forward does behind k_pack (k_encode_rows).
int sgrl_set_debug_product(sgrl_set* s, int kind, int form, const float* A, int lda, const float* W, int ldw, const float* bias,
                           float* C, int ldc, int M, int N, int K, const float* rowdiv, const float* aux_in, float* aux_out,
                           void* stream) {
  if (!s || !A || !W || !C || M <= 0 || N <= 0 || K <= 0) return sfail(SGRL_ERR_ARG, "sgrl_set_debug_product: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!GemmKernels<0>::raise_lds_limits() || !GemmKernels<EPI_RELU>::raise_lds_limits() || !GemmKernels<EPI_ROWDIV>::raise_lds_limits())
    return sfail(SGRL_ERR_HIP, "cannot raise the dynamic LDS limit of the tile kernels");
  TempWords tw;
  const std::vector<std::pair<int64_t, int>> index{{0, 0}};
  g_gemm.form = form == 1 ? SGRL_SET_FORM_BF16X6 : form;
  g_gemm.w_base = W;
  g_gemm.w_words = nullptr; g_gemm.wsc = nullptr; g_gemm.enc_index = nullptr;
  if (form == SGRL_SET_FORM_F16X3) {
    if (ldw != K) return sfail(SGRL_ERR_ARG, "debug product: the two-piece form takes W rows of K contiguous floats");
    if (tw.make(st, W, N, K, kind == 4 ? 32 : 1) != SGRL_OK) return SGRL_ERR_HIP;
    g_gemm.w_words = tw.w; g_gemm.wsc = tw.sc; g_gemm.enc_index = &index;
  }
  int rc = SGRL_OK;
  if (form == 1) {
    if (kind > 2 || (K % 16) != 0) rc = sfail(SGRL_ERR_ARG, "debug product: the exact-f32 kernel serves kinds 0..2");
    else {
      GemmArgs a{A, lda, W, ldw, bias, C, ldc, M, N, K, kind == 1 ? EPI_RELU : (kind == 2 ? EPI_ROWDIV : 0), rowdiv, nullptr, 0};
      const dim3 grid(((M + 127) / 128) * ((N + 63) / 64));
      if (kind == 0) hipLaunchKernelGGL(GemmKernels<0>::kNarrow, grid, dim3(256), GemmKernels<0>::kNarrowLds, st, a);
      else if (kind == 1) hipLaunchKernelGGL(GemmKernels<EPI_RELU>::kNarrow, grid, dim3(256), GemmKernels<0>::kNarrowLds, st, a);
      else hipLaunchKernelGGL(GemmKernels<EPI_ROWDIV>::kNarrow, grid, dim3(256), GemmKernels<0>::kNarrowLds, st, a);
    }
  } else if (kind <= 2) {
    if (N <= 64 || (K % 32) != 0) rc = sfail(SGRL_ERR_ARG, "debug product: the split kernels serve N > 64, K % 32 == 0");
    else rc = launch_gemm(st, A, lda, W, ldw, bias, C, ldc, M, N, K, kind == 1 ? EPI_RELU : (kind == 2 ? EPI_ROWDIV : 0), rowdiv);
  } else if (kind == 3) {
    if (K != GK || ldw != GK || !aux_out) rc = sfail(SGRL_ERR_ARG, "debug product: Gram operand needs K = ldw = 576 and aux_out");
    else rc = launch_gemm_gram(st, A, W, bias, C, ldc, M, N, aux_out);
  } else if (kind == 4) {
    if (N != 1024 || !rowdiv || !aux_in) rc = sfail(SGRL_ERR_ARG, "debug product: equivariant epilogue needs N = 1024, rowdiv, zq");
    else rc = launch_gemm_equiv(st, A, lda, W, ldw, bias, M, K, rowdiv, aux_in, C);
  } else if (kind == 5) {
    if (N != 64 || !aux_out || (K % 16) != 0) rc = sfail(SGRL_ERR_ARG, "debug product: stacked projections need N = 64 and aux_out");
    else {
      GemmArgs a{A, lda, W, ldw, nullptr, C, ZD, M, 64, K, EPI_ZSPLIT, nullptr, aux_out, ZD};
      if (form == SGRL_SET_FORM_F16X3) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kProjH), hipFuncAttributeMaxDynamicSharedMemorySize, kProjHLds) != hipSuccess)
          rc = sfail(SGRL_ERR_HIP, "debug product: LDS limit");
        else hipLaunchKernelGGL(kProjH, dim3(((M + 127) / 128) * 1), dim3(256), kProjHLds, st, with_words(a));
      } else if (GemmKernels<EPI_ZSPLIT>::raise_lds_limits()) GemmKernels<EPI_ZSPLIT>::launch(st, a);
      else rc = sfail(SGRL_ERR_HIP, "debug product: LDS limit");
    }
  } else if (kind == 6) {
    if (N != 128 || !rowdiv || !aux_in) rc = sfail(SGRL_ERR_ARG, "debug product: LayerNorm epilogue needs N = 128, rowdiv, ln_w | ln_b");
    else rc = launch_gemm_ln(st, A, lda, W, ldw, bias, M, K, rowdiv, C, ldc, aux_in, aux_in + 128);
  } else rc = sfail(SGRL_ERR_ARG, "debug product: unknown kind");
  const hipError_t le = hipGetLastError();
  if (rc == SGRL_OK && le != hipSuccess) rc = sfail(SGRL_ERR_HIP, std::string("debug product: launch failed: ") + hipGetErrorString(le));
  (void)hipStreamSynchronize(st);
  g_gemm.enc_index = nullptr;
  return rc;
}

// Test hook for the fused back-to-back products (chain_f16.h), same idea:
//   kind 0  C[:, 0:128] = relu(A W1' + b1) W2' + b2                               A [M, K], W1 [hid, K], W2 [128, hid]
//        1  ln_io = LayerNorm(ln_io + (relu(A W1' + b1) W2' + b2) / rowdiv)       hid = 256; C [M, ldc] is ln_io; ln = ln_w | ln_b [256]
//        2  projection site: X = A [3 M, K] -> Z (zc, z2 [3 M, 32]; z2 may be null), fn [M], C[:, 0:128] = relu(G(Z) W1' + b1) W2' + b2
//           with Wp [64, K] the stacked projections and W1 [hid, 576] in the folded Gram order
//        3  equivariant pair: C [M, 96] = tout[m][s][c] = sum_q zq[m][s][q] (relu(A W1' + b1) W2' + b2)[m][c * 32 + q] / rowdiv[m], hid = 256,
//           W2 [1024, 256], b2 [1024], zq [M, 96] passed in `ln`
int sgrl_set_debug_chain(sgrl_set* s, int kind, const float* A, int lda, int K, const float* Wp, const float* W1, const float* b1, int hid,
                         const float* W2, const float* b2, float* C, int ldc, int M, const float* rowdiv, const float* ln, float* zc,
                         float* z2, float* fn, void* stream) {
  if (!s || !A || !W1 || !W2 || !C || M <= 0 || K <= 0 || (K % 16) || (lda & 3) || (hid != 128 && hid != 256))
    return sfail(SGRL_ERR_ARG, "sgrl_set_debug_chain: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!chain_raise_lds_limits()) return sfail(SGRL_ERR_HIP, "cannot raise the dynamic LDS limit of the chain kernels");
  TempWords t1, t2, tp;
  const int K1 = kind == 2 ? GK : K;
  if (t1.make(st, W1, hid, K1) != SGRL_OK || t2.make(st, W2, kind == 3 ? 1024 : 128, hid, kind == 3 ? 32 : 1) != SGRL_OK) return SGRL_ERR_HIP;
  ChainArgs a{};
  a.W1 = t1.w; a.ldw1 = K1; a.b1 = b1; a.W2 = t2.w; a.ldw2 = hid; a.b2 = b2; a.M = M; a.K1 = K1; a.ws1 = t1.sc; a.ws2 = t2.sc;
  const dim3 grid((M + sgrl_gemm::kChainRows - 1) / sgrl_gemm::kChainRows);
  int rc = SGRL_OK;
  if (kind == 0) {
    a.A = A; a.lda = lda; a.C = C; a.ldc = ldc;
    if (hid == 256) hipLaunchKernelGGL(kChainPlain, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
    else hipLaunchKernelGGL(kChainNg, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
  } else if (kind == 1) {
    if (hid != 256 || !rowdiv || !ln) rc = sfail(SGRL_ERR_ARG, "debug chain: the LayerNorm pair needs hid = 256, rowdiv, ln_w | ln_b");
    else {
      a.A = A; a.lda = lda; a.rowdiv = rowdiv; a.ln_io = C; a.ln_ld = ldc; a.ln_w = ln; a.ln_b = ln + 128;
      hipLaunchKernelGGL(kChainLn, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
    }
  } else if (kind == 2) {
    if (!Wp || !zc || !fn) rc = sfail(SGRL_ERR_ARG, "debug chain: the site needs Wp, zc, fn");
    else if (tp.make(st, Wp, 64, K) != SGRL_OK) rc = SGRL_ERR_HIP;
    else {
      a.A = zc; a.C = C; a.ldc = ldc; a.fn_out = fn; a.X = A; a.ldx = lda; a.Kp = K; a.Wp = tp.w; a.wsp = tp.sc; a.zc = zc; a.z2 = z2;
      if (hid == 256) {
        if (z2) hipLaunchKernelGGL(kSiteF, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
        else hipLaunchKernelGGL(kSiteA, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
      } else {
        if (z2) hipLaunchKernelGGL(kSiteH2, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
        else hipLaunchKernelGGL(kSiteH1, grid, dim3(512), sgrl_gemm::kChainLds, st, a);
      }
    }
  } else if (kind == 3) {
    if (hid != 256 || !rowdiv || !ln) rc = sfail(SGRL_ERR_ARG, "debug chain: the equivariant pair needs hid = 256, rowdiv, zq (in `ln`)");
    else {
      a.A = A; a.lda = lda; a.rowdiv = rowdiv; a.zq = ln; a.tout = C;
      hipLaunchKernelGGL(kChainEq, grid, dim3(512), sgrl_gemm::kChainEqLds, st, a);
    }
  } else rc = sfail(SGRL_ERR_ARG, "debug chain: unknown kind");
  const hipError_t le = hipGetLastError();
  if (rc == SGRL_OK && le != hipSuccess) rc = sfail(SGRL_ERR_HIP, std::string("debug chain: launch failed: ") + hipGetErrorString(le));
  (void)hipStreamSynchronize(st);
  return rc;
}

long long sgrl_set_debug_redos(int reset) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(sgrl_gemm::g_scale_redos), sizeof(n)) != hipSuccess) return -1;
  const unsigned zero = 0;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(sgrl_gemm::g_scale_redos), &zero, sizeof(zero)) != hipSuccess) return -1;
  return (long long)n;
}

const char* sgrl_set_last_error(void) { return g_set_err.c_str(); }

}  // extern "C"
