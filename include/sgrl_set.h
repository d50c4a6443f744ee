/* sgrl_set.h -- C ABI of the SET (subequivariant transformer) actor forward in libsgrl_hip.so.
 *
 * Replaces, for inference under torch.no_grad(), the chain
 *   Agent.select_action                 reference src/agent.py:189-198
 *   -> SEPolicy.forward                 reference src/SEActor.py:334-347
 *   -> TransformerModel.forward         reference src/SEActor.py:237-287
 *   -> RepeatTransformerEncoder.forward reference src/SEActor.py:138-167
 *   -> MyTransformerEncoderLayer.forward reference src/SEActor.py:82-125
 *   -> multi_head_attention_forward     reference src/subequivariant_attentions.py:4-154
 * for a whole batch of environments of mixed morphologies in one call: every per-node linear layer runs over the
 * nodes of ALL morphologies at once (weights are shared), attention runs per environment over its own limbs.
 *
 * Weights: one device float buffer + a host table of SGRL_SET_NW offsets (in floats) in the slot order below.
 * Layout conventions of the packed tensors (sgrl_amd/set_hip.py does the packing from an nn.Module state_dict):
 *   Linear weights are row-major [out, in] exactly as torch stores them, except
 *     QKV_W  = rows of q_proj (pre-multiplied by (2*head_dim)^-0.5), k_proj and -- folded through ng_out, see below --
 *              v' stacked -> [768, 256]; QKV_B alike
 *     VG_W   = the vector-value projection folded through g_out -> [256, 128] (see below)
 *   Attention output folds (exact algebra, reference subequivariant_attentions.py:138-151): the scalar output is
 *     ng_out(concat_h(sum_j w_h[i,j] v_h[j])) = sum_h sum_j w_h[i,j] (Wng_h v_h[j]) + b_ng, so the value projection and ng_out
 *     collapse into ONE 256 -> 128 map per head:  QKV_W rows 512 + 128 h + r = (Wng[:, 128h:128h+128] . Wv[128h:128h+128, :])[r],
 *     QKV_B likewise with b_v; the vector output g_out(concat_h(sum_j w_h[i,j] [vg_h[j] | gdir[j]])) collapses the same way:
 *     VG_W rows 128 h + r = (Wgo[:, 128h:128h+126] . Wvg[126h:126h+126, :])[r]  and  A_GD[h][r][0:2] = Wgo[r, 128h+126:128h+128]
 *     (the gravity / direction columns).  The attention kernel then produces the layer's two outputs directly; the slots
 *     NGOUT_W and GOUT_W are kept for reference but no longer read by the kernels.
 *     L1NG_W = linear1_ng.weight padded with 15 zero columns -> [128, 160]
 *     A_LG1_W, F_LG1_W, L1G_W (the layers fed by the symmetric 32x32 Gram matrix G = Z'Z) are folded onto the BLOCKED lower
 *              triangle the GEMM generates its operand in (the Gram matrix itself is never stored): the 36 4x4 blocks (A, B),
 *              B <= A, one per 16-wide k-tile: column k = 16 (A (A + 1) / 2 + B) + 4 i + j <-> a = 4 A + i, b = 4 B + j;
 *              W'[n][k] = W[n][32a+b] + W[n][32b+a] (b < a), W[n][33a] (b = a), 0 (b > a, diagonal blocks only); 576 columns
 */
#ifndef SGRL_SET_H
#define SGRL_SET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgrl_set sgrl_set;

/* global slots */
enum {
  SGRL_SET_EMB0 = 0, SGRL_SET_EMB1, SGRL_SET_EMB2, SGRL_SET_REL_W, SGRL_SET_REL_B, SGRL_SET_FNORM_W, SGRL_SET_FNORM_B,
  SGRL_SET_GENC, SGRL_SET_ENC_W, SGRL_SET_ENC_B, SGRL_SET_GGPROJ, SGRL_SET_L1G_W, SGRL_SET_L1G_B, SGRL_SET_L2G_W,
  SGRL_SET_L2G_B, SGRL_SET_L1NG_W, SGRL_SET_L1NG_B, SGRL_SET_L2NG_W, SGRL_SET_L2NG_B, SGRL_SET_DECG, SGRL_SET_L1M_W,
  SGRL_SET_L1M_B, SGRL_SET_L2M_W, SGRL_SET_L2M_B, SGRL_SET_GPROJ,
  SGRL_SET_NGLOBAL
};
/* per-layer slots (slot = SGRL_SET_NGLOBAL + layer * SGRL_SET_NLAYER + k) */
enum {
  SGRL_SET_A_GPROJ = 0, SGRL_SET_A_LG1_W, SGRL_SET_A_LG1_B, SGRL_SET_A_LG2_W, SGRL_SET_A_LG2_B, SGRL_SET_QKV_W,
  SGRL_SET_QKV_B, SGRL_SET_VG_W, SGRL_SET_NGOUT_W, SGRL_SET_NGOUT_B, SGRL_SET_GOUT_W, SGRL_SET_F_GPROJ2,
  SGRL_SET_F_GPROJ3, SGRL_SET_F_LG1_W, SGRL_SET_F_LG1_B, SGRL_SET_F_LG2_W, SGRL_SET_F_LG2_B, SGRL_SET_L3_W,
  SGRL_SET_L3_B, SGRL_SET_L4_W, SGRL_SET_L4_B, SGRL_SET_L5_W, SGRL_SET_L1_W, SGRL_SET_L1_B, SGRL_SET_L2_W,
  SGRL_SET_L2_B, SGRL_SET_N1_W, SGRL_SET_N1_B, SGRL_SET_N2_W, SGRL_SET_N2_B, SGRL_SET_A_GD,
  SGRL_SET_NLAYER
};
#define SGRL_SET_LAYERS 3
#define SGRL_SET_NW (SGRL_SET_NGLOBAL + SGRL_SET_LAYERS * SGRL_SET_NLAYER)

int sgrl_set_create(sgrl_set** out);
void sgrl_set_destroy(sgrl_set* s);

/* w: DEV float buffer (kept by reference: the caller keeps it alive); offsets: HOST int64[SGRL_SET_NW].
 * Static variant: the caller packs once and promises not to change the values behind `w`. */
int sgrl_set_weights(sgrl_set* s, const float* w, const int64_t* offsets, int n_offsets);

/* Live variant -- what an nn.Module binding should use (reference agent.py: the same modules are updated in place by
 * the optimizers, agent.py:155-176, and by functional.soft_update_network's `target_param.data.copy_(...)`,
 * common/functional.py:7-10, which no version counter reveals): the handle keeps the parameters' DEVICE addresses and
 * rebuilds its flat weight buffer from them at the top of EVERY forward, on the forward's stream (one kernel, ~46 MB of
 * traffic).  A segment describes one run of the flat buffer:
 *   COPY   dst[i] = i < a ? src0[i] * scale : 0                        (plain tensors, QKV stacking, zero row padding)
 *   PADCOL src0 [rows, a] -> dst [rows, b], zero columns appended      (L1NG_W)
 *   FOLD   src0 [rows, 1024] -> dst [rows, 576] Gram-triangle folding  (A_LG1_W, F_LG1_W, L1G_W; see above)
 *   STACK  dst [64, b]: rows 0..29 = src0 [30, a], rows 32..61 = src1 [30, a] or zero, columns a..b-1 zero
 *          (the two 30-row projections of a proj+Gram site as ONE zero-padded GEMM operand)
 *   MATMUL dst [n / b, b] = src0 [n / b, a] (row stride lda) . src1 [a, b] (row stride ldb), times scale   (weight folds)
 *   SUBMAT dst [n / b, b] = src0 [n / b, b] (row stride lda)                                                (column blocks)
 *   PERM32 dst [1024, a]: row c * 32 + q = src0 row q * 32 + c   (L4_W / L4_B / L2M_W / L2M_B: the 1024 outputs are a 32 x 32
 *          matrix mat[q][c] per node, reference SEActor.py:105-107; stored c-major so that one 32-column GEMM tile holds all q of
 *          one c and the epilogue can contract it with z[s][q] without ever writing the matrix: the GEMM emits z . mat [3, 32])
 * offsets: HOST int64[SGRL_SET_NW + SGRL_SET_NSITES + SGRL_SET_NEXTRA] -- the slot table followed by the offsets of the seven stacked
 * projection operands (sites 2l = attention g_proj of layer l [64,128]; 2l+1 = g_proj2 | g_proj3 [64,128];
 * 6 = gg_proj | g_proj (actor) [64,144]).  The segments must cover [0, total_floats) entirely.  Parameter storage must
 * stay allocated while the handle is bound; re-bind after anything that moves it (module.to(), new tensors). */
enum { SGRL_PACK_COPY = 0, SGRL_PACK_PADCOL = 1, SGRL_PACK_FOLD = 2, SGRL_PACK_STACK = 3, SGRL_PACK_MATMUL = 4, SGRL_PACK_SUBMAT = 5,
       SGRL_PACK_PERM32 = 6 };
typedef struct sgrl_pack_seg {
  int64_t dst;        /* first float of the run in the flat buffer */
  const void* src0;   /* DEV float* */
  const void* src1;   /* DEV float* or null (STACK) */
  int32_t n;          /* floats in the run */
  int32_t kind;
  int32_t a, b;
  float scale;        /* COPY, MATMUL */
  int32_t lda, ldb;   /* MATMUL, SUBMAT: row strides of src0 / src1 in floats */
  int32_t reserved;
} sgrl_pack_seg;
#define SGRL_SET_NSITES 7
/* ... followed by SGRL_SET_NEXTRA more offsets: the actor head with decoder_g FOLDED through linear2_m (exact algebra, reference
 * SEActor.py:272-279: decoder_g(z . mat) = z . (mat . w_dec), so only the 32 numbers m2[q] = sum_c mat[q][c] w_dec[c] per node are
 * needed):  [32, 256] rows q = sum_c w_dec[c] * linear2_m.weight[q * 32 + c]  and  [32] = sum_c w_dec[c] * linear2_m.bias[q * 32 + c]
 * (MATMUL segments over the live parameters).  A critic network binds 64-float fillers there. */
#define SGRL_SET_NEXTRA 2
int sgrl_set_bind_params(sgrl_set* s, const sgrl_pack_seg* segs, int n_segs, const int64_t* offsets, int n_offsets,
                         int64_t total_floats);

/* Weight hold.  By default every forward of a bound handle rebuilds its flat buffer from the live parameters (~55 us at the top of
 * the forward).  A rollout loop knows better: between two rounds of updates the actor's parameters do not change (reference
 * trainer.py:155-251: collect a round with `select_action`, then `agent.update`).  sgrl_set_hold_weights(s, 1) is the caller's promise
 * that the parameters stay as they are until the next call of this function: the first forward after it packs, the following ones
 * reuse that buffer.  Every call (hold = 1 again, or 0 = back to packing on every forward) also means "the parameters may just have
 * changed".  Forwards recorded into a hipGraph always pack.  `DeviceTrainer` holds across a collection round and calls again after
 * every round's updates / actor broadcast; bench.py's rollout (constant random-init weights) holds throughout. */
int sgrl_set_hold_weights(sgrl_set* s, int hold);

/* Batch structure (SEPolicy.change_morphology for every morphology at once, reference SEActor.py:349-355):
 *   n_morph, morph_L[n_morph] limbs, morph_count[n_morph] envs per morphology (env blocks in this order),
 *   trav: HOST int32, per morphology 3*L traversal indices (pre, inlcrs, postlcrs), concatenated,
 *   rel:  HOST float, per morphology L*L*3 relation tensor (graph_dict['relation']), concatenated. */
int sgrl_set_graph(sgrl_set* s, int n_morph, const int32_t* morph_L, const int32_t* morph_count, const int32_t* trav,
                   const float* rel);
/* The handle remembers the batch structures it has seen (keyed on the CONTENT of the arguments, up to
 * SGRL_SET_GRAPH_CACHE of them): switching back to one -- the reference changes morphology before every select_action
 * and every update, trainer.py:173-176,246-247 -- swaps a few pointers, with no device synchronisation, allocation or
 * upload.  The workspace is shared and only ever grows. */
#define SGRL_SET_GRAPH_CACHE 128

/* actions[e, 0:3*L_e] = max_action * tanh(actor(obs[e, 0:41*L_e])), rest of the row zero.
 * obs: DEV float [n_env, obs_ld]; act: DEV float [n_env, act_ld].  SGRL_ERR_ARG unless obs_ld >= 41 * Lmax and
 * act_ld >= 3 * Lmax (Lmax = most limbs of the configured morphologies); likewise action_ld / q_ld (>= Lmax) below. */
int sgrl_set_forward(sgrl_set* s, const float* obs, int obs_ld, float* act, int act_ld, float max_action, void* stream);

/* Critic variant of the same network (reference src/SECritic.py:8-124: TransformerModel with feature_size 44 = 41
 * state + 3 action values per limb, scalar head):  q[e, l] = critic(obs[e], action[e])[l] for the limbs of env e, rest
 * of the row zero.  Weights in the same slot table, packed from a critic's state_dict: ENC_W = encoder.weight [128,20],
 * L1NG_W = linear1_ng.weight [128,148] padded to 160 columns, DECG = decoder_ng.weight [256], L1M_B = decoder_ng.bias [1];
 * the slots L1M_W, L2M_W, L2M_B and GPROJ are not read.  One handle serves one network (actor OR one critic). */
int sgrl_set_forward_q(sgrl_set* s, const float* obs, int obs_ld, const float* action, int action_ld, float* q, int q_ld,
                       void* stream);

int sgrl_set_num_nodes(const sgrl_set* s);
int64_t sgrl_set_workspace_bytes(const sgrl_set* s);
/* Counter bumped whenever the handle FREES device memory that a forward recorded earlier may point into: a batch structure
 * evicted from the content cache of sgrl_set_graph, the flat weight buffers replaced by sgrl_set_bind_params, a regrown
 * workspace.  A hipGraph that captured forwards of this handle must be captured again once the value has changed (replaying
 * it would dereference freed memory: a GPU fault, not an error code).  No reference counterpart (PyTorch owns its tensors). */
int64_t sgrl_set_generation(const sgrl_set* s);
/* Time `reps` forwards with HIP events on `stream` (mean ms per forward). */
int sgrl_set_time_forward(sgrl_set* s, const float* obs, int obs_ld, float* act, int act_ld, float max_action,
                          int reps, void* stream, float* ms_out);
/* Debug/parity: copy an intermediate buffer of the LAST forward to the host.  which: 0 g[N,3,128], 1 cat[N,256]
 * (inv | ng), 2 zc[N,3,32] (the projected vectors Z whose Gram matrix Z'Z the lg1 GEMMs consume), 3 fn[N], 4 qkv[N,768] (q | k | v' folded), 5 / 6 unused, 7 T[N,3,32] (z . mat),
 * 8 g1[N,3,128] (attention's vector output), 9 delta[N,128] (attention's / FFN's scalar output before the residual norm),
 * 10 outng[N,160] (input features | final-norm ng | zero padding). */
int sgrl_set_peek(sgrl_set* s, int which, float* host, int64_t n_floats);
/* Parity probes (tests only): make the following forwards return after stage 2l (attention block of layer l done: g1 and
 * delta hold MyMultiheadAttention's two outputs, reference SEActor.py:48-66) or 2l+1 (layer l done: g and cat[:,128:] hold
 * MyTransformerEncoderLayer's outputs, SEActor.py:82-125); the action / Q output of such a forward is not written.
 * stage = -1 restores the full forward. */
int sgrl_set_debug_stop_after(sgrl_set* s, int stage);
/* Batches of at most `nodes` nodes run their dense products through the 32 x 32 tile kernels of sgrl_train.h (latency-bound
 * sizes: the TD3 update's target networks, single-environment action selection) instead of the 128 x 128 tile kernels; same
 * results to float32 rounding.  Default 2048 (SGRL_SET_SMALL_NODES in the environment); 0 = never; -1 restores the default.
 * Tests use it to run one input through both paths. */
int sgrl_set_debug_small_nodes(sgrl_set* s, int nodes);
/* Form of the 128 x 128 tile products (reference: plain f32 `F.linear`, subequivariant_attentions.py:90-151 / SEActor.py:82-125).
 * Both forms carry the f32 product on the 16-bit matrix cores and measure the same error against float64 as an f32 FMA chain
 * (tests/test_split_products_gpu.py):
 *   SGRL_SET_FORM_F16X3  (default) every operand = two f16 pieces, three matrix instructions per product block.  Every operand
 *                        ROW is first multiplied by a power of two that brings its largest magnitude into f16's range (exact; undone
 *                        in the epilogue), so the form has float32's exponent range as the reference's `F.linear` has: nothing is
 *                        clamped, there is no range contract to watch (csrc/gemm_f32.h, pow2_scale);
 *   SGRL_SET_FORM_BF16X6 three bf16 pieces, six instructions, ~25 % slower products (kept for A/B comparisons).
 * form 0 restores the default (SGRL_SET_GEMM=bf16x6 in the environment selects the second form).  The back-to-back products of a
 * forward (projection -> Gram -> linear_g1 -> linear_g2 sites, linear1 -> linear2 + norm2, the head's linear1_ng -> linear2_ng) run
 * as ONE kernel each in the default form (csrc/chain_f16.h; SGRL_SET_CHAIN=0 in the environment keeps them apart). */
#define SGRL_SET_FORM_F16X3 2
#define SGRL_SET_FORM_BF16X6 3
int sgrl_set_gemm_form(sgrl_set* s, int form);
/* Test hook: ONE product through the production tile kernels of the forward on caller-supplied device operands (kinds: plain,
 * ReLU, row division, Gram operand, equivariant epilogue, stacked projections, residual + LayerNorm; forms as above, 1 = the
 * exact-f32 matrix instruction for the first three) -- tests/test_split_products_gpu.py holds every instantiation the forward
 * launches against float64 with it.  Synchronises `stream`.  Argument conventions: sgrl_amd/csrc/set_actor.hip. */
int sgrl_set_debug_product(sgrl_set* s, int kind, int form, const float* A, int lda, const float* W, int ldw, const float* bias,
                           float* C, int ldc, int M, int N, int K, const float* rowdiv, const float* aux_in, float* aux_out,
                           void* stream);
/* Test hook for the fused back-to-back products (csrc/chain_f16.h), on caller-supplied device operands:
 *   kind 0  C[:, 0:128] = relu(A W1' + b1) W2' + b2                            A [M, K], W1 [hid, K], W2 [128, hid], hid = 128 | 256
 *        1  C = LayerNorm(C + (relu(A W1' + b1) W2' + b2) / rowdiv)            hid = 256; C [M, ldc] read and rewritten; ln = ln_w | ln_b [256]
 *        2  projection site: X = A [3 M, K] -> Z (zc, z2 [3 M, 32]; z2 may be null; columns 30, 31 untouched), fn [M] = ||Z'Z||_F + 1,
 *           C[:, 0:128] = relu(G(Z) W1' + b1) W2' + b2 with Wp [64, K] the stacked projections, W1 [hid, 576] in the folded Gram order
 *        3  equivariant pair (linear3 -> ReLU -> linear4 -> contraction): C [M, 96] = tout[m][s][c] = sum_q zq[m][s][q] *
 *           (relu(A W1' + b1) W2' + b2)[m][c * 32 + q] / rowdiv[m]; hid = 256, W2 [1024, 256], b2 [1024], zq [M, 96] passed in `ln`
 * Synchronises `stream`. */
int sgrl_set_debug_chain(sgrl_set* s, int kind, const float* A, int lda, int K, const float* Wp, const float* W1, const float* b1, int hid,
                         const float* W2, const float* b2, float* C, int ldc, int M, const float* rowdiv, const float* ln, float* zc,
                         float* z2, float* fn, void* stream);
/* Diagnostics of the row scaling (csrc/gemm_f32.h): workgroups of the two-piece products that had to repeat a tile because a
 * sampled estimate of a row's magnitude fell short, since the last reset (process-wide; synchronises the device; -1 on error).
 * Zero on every input a working policy produces -- the tests and the bench line say so. */
long long sgrl_set_debug_redos(int reset);
const char* sgrl_set_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SGRL_SET_H */
