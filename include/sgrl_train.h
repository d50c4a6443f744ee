/* sgrl_train.h -- C ABI of the dense products the TD3 update back-propagates through (gfx950).
 *
 * What it replaces: the `torch.nn.Linear` forward / backward calls inside the SET actor and critic when `Agent.update`
 * (reference src/agent.py:117-183) differentiates through them (reference src/SEActor.py:34-287, src/SECritic.py:8-124;
 * this repository's differentiable module is sgrl_amd/set_policy.py).  At the update's size -- batch 100 x 7..14 limbs = 700..1400
 * rows (x 3 for the vector channels) -- the vendor libraries spend most of the update inside a handful of single-workgroup
 * launches (a 256 x 256 macro tile looping over a 2 100-long contraction for a 30 x 128 weight gradient: 165 us); the kernels
 * behind this header are built for the latency of small products instead (32 x 32 output tiles, 128-deep k-tiles).
 *
 * Arithmetic: the float32 matrix instruction (exact float32 products, float32 accumulation); partial sums (the four waves of a
 * workgroup, the splits of a long weight-gradient contraction) are added in a fixed order: results are bit-reproducible.
 *
 * All pointers are DEVICE pointers owned by the caller; rows are `ld*` floats apart; no alignment is required (16-byte aligned
 * rows take the vector path).  `ws` (backward only; may be null: no split then) is scratch for split weight-gradient contractions:
 * sgrl_train_ws_floats() floats, ZERO filled before the first call; every call leaves its counter region zero again, so one
 * buffer serves any number of consecutive calls ON ONE STREAM.  Calls on different streams need different buffers.
 * Error codes as in sgrl.h; message via sgrl_train_last_error().  No CPU fallback.
 */
#ifndef SGRL_TRAIN_H
#define SGRL_TRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int64_t sgrl_train_ws_floats(void);

/* y[M, N] = act(x[M, K] . w[N, K]^T + bias[N]) / rowdiv[M]; bias and rowdiv may be null; relu != 0 applies max(., 0)
 * (torch.nn.functional.linear, the ReLU that follows it, and the `/ F_norm` of reference SEActor.py:101,117 in one launch) */
int sgrl_linear_forward(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowdiv, float* y,
                        int ldy, int M, int N, int K, int relu, void* stream);

/* sgrl_linear_forward with two optional fusions of what follows the product in the SET layers (one launch instead of two):
 *   addend [M, N] (row stride ldadd): y += addend after the epilogue -- the residual of the vector stream, g + linear5(.)
 *     (reference SEActor.py:110);
 *   tail [M, ntail]: y[m][N + j] = tail[m][j] -- columns appended to the product's N (y has N + ntail <= ldy columns): the
 *     gravity / direction pair behind the 30 projected channels, z = [proj(x) | gdir] (reference SEActor.py:93-94), or the scalar
 *     stream behind the invariant features, c = [inv | ng] (SEActor.py:98, 101); any ntail > 0.
 * The twin form does the same for the two critics' layers in one launch. */
int sgrl_linear_forward_fused(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowdiv,
                              const float* addend, int ldadd, const float* tail, int ntail, float* y, int ldy, int M, int N, int K,
                              int relu, void* stream);
int sgrl_linear_forward_twin_fused(const float* x0, const float* x1, int ldx, const float* w0, const float* w1, int ldw, const float* b0,
                                   const float* b1, const float* rd0, const float* rd1, const float* add0, const float* add1, int ldadd,
                                   const float* tail0, const float* tail1, int ntail, float* y0, float* y1, int ldy, int M, int N, int K,
                                   int relu, void* stream);

/* Backward of the call above (`y` = its output, needed when relu != 0 or drowdiv != null; relu and rowdiv exclude each other).
 * g = dy masked by (y > 0) if relu, divided row-wise by rowdiv if given.
 *   dx[M, K]    = g . w                                  (skipped when dx == null)
 *   dw[N, K]    = g^T . x                                (skipped when dw == null)
 *   db[N]       = column sums of g                       (skipped when db == null)
 *   drowdiv[M]  = -(sum_n dy[m][n] y[m][n]) / rowdiv[m]  (skipped when drowdiv == null) */
int sgrl_linear_backward(const float* dy, int lddy, const float* y, int ldyo, int relu, const float* rowdiv, const float* x,
                         int ldx, const float* w, int ldw, float* dx, int lddx, float* dw, int lddw, float* db,
                         float* drowdiv, int M, int N, int K, float* ws, void* stream);
/* The same with the ReLU of the layer BELOW folded into the input gradient's epilogue: x_relu != 0 says that this layer's input x is
 * the output of a ReLU layer (x = max(., 0): linear1 -> ReLU -> linear2 of the reference's feed-forward pairs, SEActor.py:101-121),
 * and dx comes out masked by x > 0 -- which is the gradient that layer's backward needs, so it is then called with relu = 0 and
 * reads no mask in its two products (mask applied once per element here instead of once per k-tile there, twice).  The twin form
 * takes the two inputs x0 / x1 (both null: no mask). */
int sgrl_linear_backward_xrelu(const float* dy, int lddy, const float* y, int ldyo, int relu, const float* rowdiv, const float* x,
                               int ldx, const float* w, int ldw, float* dx, int lddx, float* dw, int lddw, float* db,
                               float* drowdiv, int M, int N, int K, int x_relu, float* ws, void* stream);

/* The same with the input gradient ACCUMULATED: acc_dx != 0 adds g . w (masked as above) onto what dx already holds instead of
 * overwriting it -- a tensor that feeds several linear layers (the vector stream g into g_proj and vg_proj, the invariant features c
 * into linear3 and linear1: reference SEActor.py:93-121) collects their input gradients in one buffer, product by product, without
 * the element-wise additions autograd would launch (sgrl_amd/train_ops.py fan_out).  Same stream: the products run in order. */
int sgrl_linear_backward_acc(const float* dy, int lddy, const float* y, int ldyo, int relu, const float* rowdiv, const float* x,
                             int ldx, const float* w, int ldw, float* dx, int lddx, float* dw, int lddw, float* db,
                             float* drowdiv, int M, int N, int K, int x_relu, int acc_dx, float* ws, void* stream);

/* The weight (+ bias) gradients of several layers in one launch per 12 layers: dw = g^T x, db = column sums of g, g as in
 * sgrl_linear_backward (which then is called with dw = db = null).  They are not on the backward pass's critical path -- only the
 * optimizer needs them -- so a caller may collect the descriptors during the pass and issue them together at its end. */
typedef struct sgrl_wgrad_desc {
  const float* dy; const float* y; const float* rowdiv; const float* x;   /* y: the forward's output, read when relu != 0 */
  float* dw; float* db;                                                    /* db may be null */
  int32_t lddy, ldy, ldx, lddw, M, N, K, relu;
} sgrl_wgrad_desc;
int sgrl_linear_wgrad_group(int n, const sgrl_wgrad_desc* d, float* ws, void* stream);

/* Twin products: the same layer of the reference's two critics (SECritic.py:8-124: critic1 / critic2, two TransformerModels on one
 * batch; agent.py:150-160 updates both from one loss) in ONE launch -- two argument sets of identical shape, the workgroup's z index
 * picks one.  Forward: y_i = act(x_i w_i^T + b_i) / rd_i; input gradient (and the row divisor's): dx_i = g_i w_i with g_i = dy_i
 * masked by y_i > 0 (relu) or divided by rd_i, drd_i[m] = -(dy_i[m] . y_i[m]) / rd_i[m].  The weight gradients of both go
 * through sgrl_linear_wgrad_group.  b / rd / drd: both null or both given. */
int sgrl_linear_forward_twin(const float* x0, const float* x1, int ldx, const float* w0, const float* w1, int ldw, const float* b0,
                             const float* b1, const float* rd0, const float* rd1, float* y0, float* y1, int ldy, int M, int N, int K,
                             int relu, void* stream);
int sgrl_linear_dgrad_twin(const float* dy0, const float* dy1, int lddy, const float* y0, const float* y1, int ldyo, int relu,
                           const float* rd0, const float* rd1, const float* w0, const float* w1, int ldw, float* dx0, float* dx1,
                           int lddx, float* drd0, float* drd1, int M, int N, int K, void* stream);
int sgrl_linear_dgrad_twin_xrelu(const float* dy0, const float* dy1, int lddy, const float* y0, const float* y1, int ldyo, int relu,
                                 const float* rd0, const float* rd1, const float* w0, const float* w1, int ldw, float* dx0, float* dx1,
                                 int lddx, float* drd0, float* drd1, const float* x0, const float* x1, int ldx, int M, int N, int K,
                                 void* stream);
int sgrl_linear_dgrad_twin_acc(const float* dy0, const float* dy1, int lddy, const float* y0, const float* y1, int ldyo, int relu,
                               const float* rd0, const float* rd1, const float* w0, const float* w1, int ldw, float* dx0, float* dx1,
                               int lddx, float* drd0, float* drd1, const float* x0, const float* x1, int ldx, int M, int N, int K,
                               int acc_dx, void* stream);

/* Gram invariants of M nodes' three 32-vectors z[M, 3, 32] (reference SEActor.py:94-98): gram[M, 1024] = vec(Z'Z),
 * fn[M] = ||Z'Z||_F + 1; and their backward: dz = Z (D + D'), D = dgram + (dfn / ||Z'Z||_F) Z'Z (dgram or dfn may be null). */
int sgrl_gram_forward(const float* z, float* gram, float* fn, int M, void* stream);
int sgrl_gram_backward(const float* z, const float* dgram, const float* dfn, const float* fn, float* dz, int M, void* stream);
/* The same invariants on the lower triangle of the symmetric Z'Z only: tri[M, 528], tri[m][a (a + 1) / 2 + b] = (Z'Z)[a][b], a >= b
 * (fn as above, the norm of the FULL matrix).  A linear layer W on vec(Z'Z) equals W' on tri(Z'Z) with the mirror columns of W added
 * (sgrl_sym_fold), so its three products run over 528 columns instead of 1 024; dz = Z S with S[a][b] = S[b][a] = dtri[k] +
 * 2 (dfn / ||Z'Z||_F) (Z'Z)[a][b], the diagonal with 2 dtri[k].
 * sgrl_sym_fold: n <= 16 matrices per launch.  unfold == 0: wtri[i][rows[i], 528] = fold of w[i][rows[i], 1024] (W'[r][k] = W[r][a 32 + b] +
 * W[r][b 32 + a], the diagonal once); unfold != 0: w[i][r][a 32 + b] = w[i][r][b 32 + a] = wtri[i][r][k] -- the gradient of the fold
 * (w is then written, wtri read). */
int sgrl_gram_tri_forward(const float* z, float* tri, float* fn, int M, void* stream);
int sgrl_gram_tri_backward(const float* z, const float* dtri, const float* dfn, const float* fn, float* dz, int M, void* stream);
int sgrl_sym_fold(int n, const float* const* w, float* const* wtri, const int* rows, int unfold, void* stream);

/* Equivariant contraction of M nodes' three 32-vectors with their 32 x 32 matrices (reference SEActor.py:108-110, 262-264):
 * t[M, 3, 32] = z[M, 3, 32] . mat[M, 32, 32] per node; backward: dz = dt . mat', dmat = z' . dt. */
int sgrl_zmat_forward(const float* z, const float* mat, float* t, int M, void* stream);
int sgrl_zmat_backward(const float* z, const float* mat, const float* dt, float* dz, float* dmat, int M, void* stream);

/* Residual + LayerNorm over 128 columns (reference SEActor.py:90-91, 113-114: norm1(ng + attention update), norm2(ng + feed-forward
 * update); SEActor.py:164: the encoder's final norm, res = null):  y = LayerNorm(x + res) * w + b, eps inside the square root.
 * `nets` (1 or 2) networks stacked along the row axis, `rows` rows each ([nets * rows, 128]); network i uses w_i, b_i [128].
 * xhat [nets * rows, 128] (the normalised rows) and rstd [nets * rows] are what the backward needs (both null: not saved).
 * Backward: dx [nets * rows, 128] = the gradient of x AND of res; dw_i = sum_rows dy xhat, db_i = sum_rows dy (written, not
 * accumulated; null = not wanted; rows summed in a fixed order: bit-reproducible).  One launch each. */
int sgrl_add_ln_forward(const float* x, const float* res, const float* w0, const float* b0, const float* w1, const float* b1, float* y,
                        float* xhat, float* rstd, int rows, int nets, float eps, void* stream);
int sgrl_add_ln_backward(const float* dy, const float* xhat, const float* rstd, const float* w0, const float* w1, float* dx, float* dw0,
                         float* db0, float* dw1, float* db1, int rows, int nets, void* stream);

/* Three traversal-index embeddings concatenated (reference SEActor.py:18-31, ConcatPositionalEmbedding): idx [3, L] int64 (one index
 * vector per table, all < rows), w_t [rows, n_t], n0 + n1 + n2 <= 128:  out[l][:] = w0[idx[0][l]] | w1[idx[1][l]] | w2[idx[2][l]].
 * Backward: dw_t [rows, n_t] written densely (rows no limb points at receive zero; null = not wanted), limbs summed in order. */
int sgrl_embed3_forward(const long long* idx, const float* w0, const float* w1, const float* w2, int n0, int n1, int n2, float* out, int L,
                        void* stream);
int sgrl_embed3_backward(const long long* idx, const float* dout, float* dw0, float* dw1, float* dw2, int n0, int n1, int n2, int L, int rows,
                         void* stream);

/* Optimizer steps over a device TABLE of tensors instead of torch's multi-tensor launches (reference agent.py:161-177:
 * clip_grad_norm_ + Adam.step; common/functional.py:7-10: soft target update).  table: n_tensors rows of six 64-bit words
 * (param*, grad*, exp_avg*, exp_avg_sq*, step* (one float, the tensor's step count), numel); chunks: n_chunks pairs of int32
 * (row, first element), each covering at most sgrl_optim_chunk() elements of one tensor.
 *   sgrl_optim_clip_adam: max_norm > 0: total = sqrt(sum of grad^2 over every row), grads *= min(1, max_norm / (total + 1e-6)) in
 *     place (torch.nn.utils.clip_grad_norm_); every step += 1; Adam exactly as torch's fused kernel (no weight decay, no amsgrad):
 *     m = lerp(m, g, 1 - b1), v = b2 v + (1 - b2) g^2, p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
 *     scratch: 1 + n_chunks floats (total of squares, partials).  Three launches (two without clipping), fixed summation order.
 *   sgrl_optim_lerp: rows (dst*, src*, -, -, -, numel): dst = dst (1 - tau) + tau src.  One launch. */
int sgrl_optim_chunk(void);
int sgrl_optim_clip_adam(const void* table, int n_tensors, const void* chunks, int n_chunks, double lr, double beta1, double beta2, double eps,
                         float max_norm, float* scratch, void* stream);
int sgrl_optim_lerp(const void* table, const void* chunks, int n_chunks, float tau, void* stream);

/* Limb attention of B environments with L <= 14 limbs, 2 heads x 128 channels (reference subequivariant_attentions.py:90-151
 * between the projections).  qkv [B, L, 768] = q | k | v as the stacked projection leaves them (q is multiplied by `scale` inside);
 * the vector values are given in parts and never concatenated: vgp [B, L, 3, 252] (126 projected channels per head) and gdir
 * [B, L, 3, 2] (channels 126, 127 of both heads); bias [2, L, L] or null.
 *   w[b][h][i][:] = softmax_j(scale q_i . k_j + bias),  o[b][i][c] = sum_j w[h(c)] v[j][c],  og[b][i][s][c] = sum_j w[h(c)] vg[j][s][c]
 * w [B, 2, L, L] is returned for the backward, which yields dqkv [B, L, 768], dvgp [B, L, 3, 252], dgdh [B, L, 3, 2 heads, 2] (the
 * gradient of gdir per head: sum over the heads) and ds [B, 2, L, L] (the score gradient: its sum over the environments is the
 * gradient of `bias`). */
int sgrl_attention_forward(const float* qkv, const float* vgp, const float* gdir, const float* bias, float scale, float* w,
                           float* o, float* og, int B, int L, void* stream);
int sgrl_attention_backward(const float* qkv, const float* vgp, const float* gdir, float scale, const float* w, const float* d_o,
                            const float* d_og, float* dqkv, float* dvgp, float* dgdh, float* ds, int B, int L, void* stream);

const char* sgrl_train_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
