/*
 * gluonnlp_amd.h -- C ABI of the MI355X-native (gfx950) Transformer-encoder forward path.
 *
 * Drop-in boundary for GluonNLP's encoder hot path.  The reference (dmlc/gluon-nlp) has NO
 * C/FFI boundary of its own: its boundary is the Python call signature of the functions cited
 * below (paths relative to the reference checkout, src/gluonnlp/...).  Each entry point here is
 * what a ctypes/cffi binding inside those Python functions would call.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless named *_host; the caller owns all buffers
 *    (DLPack zero-copy: the Python shim passes DLTensor.data + byte_offset straight through);
 *  - shapes / strides are in ELEMENTS, int64_t; the innermost dimension is always contiguous;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call is
 *    asynchronous on that stream and allocates nothing;
 *  - return value 0 = ok, <0 = error; gnlp_last_error() gives the thread-local message;
 *  - the library is reentrant per stream.  Its only process-global state: the developer knobs
 *    (GNLP_* environment variables, read ONCE on first use -- never on a launch path -- and
 *    re-read only by gnlp_reload_knobs()), per-device kernel attributes set once, and the
 *    measurement hook at the end of this file (off unless armed).
 */
#ifndef GLUONNLP_AMD_H_
#define GLUONNLP_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *gnlp_stream_t;

enum { GNLP_OK = 0, GNLP_ERR_ARG = -1, GNLP_ERR_UNSUPPORTED = -2, GNLP_ERR_LAUNCH = -3 };

/* tensor element types of the activation / weight buffers */
enum { GNLP_F32 = 0, GNLP_BF16 = 1 };
/* integer element type of valid_length vectors */
enum { GNLP_I32 = 0, GNLP_I64 = 1 };
/* attention_cell.py:153,168 */
enum { GNLP_ATTN_FULL = 0, GNLP_ATTN_CAUSAL = 1 };
/* layers.py:236-271 get_activation */
enum {
  GNLP_ACT_IDENTITY = 0, GNLP_ACT_RELU = 1, GNLP_ACT_GELU_ERF = 2, GNLP_ACT_GELU_TANH = 3,
  GNLP_ACT_GELU_SIGMOID = 4, GNLP_ACT_SIGMOID = 5, GNLP_ACT_TANH = 6, GNLP_ACT_LEAKY = 7,
  GNLP_ACT_ELU = 8, GNLP_ACT_SOFTRELU = 9, GNLP_ACT_SOFTSIGN = 10
};
/* layers.py:38-77 get_norm_layer */
enum { GNLP_NORM_LAYER = 0, GNLP_NORM_RMS = 1, GNLP_NORM_NONE = 2 };
/* how gnlp_dense combines its optional `other` operand with the activated product */
enum { GNLP_COMBINE_ADD = 0, GNLP_COMBINE_MUL = 1 };

int gnlp_version(void);
const char *gnlp_last_error(void);
/* Number of HIP devices visible / name of the current one (diagnostics only). */
int gnlp_device_count(void);
/* Re-read the GNLP_* developer knobs from the environment (no counterpart in the reference; for
 * tests and A/B scripts that flip a kernel choice mid-process).  Not for concurrent use with
 * launches on other threads. */
void gnlp_reload_knobs(void);

/* ---------------------------------------------------------------------------------------------
 * Masks.  Integer/byte work: results are bit-exact with the reference.
 * ------------------------------------------------------------------------------------------- */

/* gen_self_attn_mask(data, valid_length, dtype, attn_type, layout)   attention_cell.py:40-183
 * mask: (B, L, L) bytes, 1 = attend.  full: (j < vl[b]) & (i < vl[b]); causal: (j <= i) &
 * (i < vl[b]).  valid_length may be NULL (no length predicate). */
int gnlp_gen_self_attn_mask(uint8_t *mask, const void *valid_length, int vl_dtype, int64_t B,
                            int64_t L, int attn_type, gnlp_stream_t stream);

/* gen_mem_attn_mask(mem, mem_valid_length, data, data_valid_length)  attention_cell.py:186-274
 * mask: (B, Lq, Lm) bytes: (j < mem_vl[b]) & (i < data_vl[b]); data_valid_length may be NULL. */
int gnlp_gen_mem_attn_mask(uint8_t *mask, const void *mem_valid_length,
                           const void *data_valid_length, int vl_dtype, int64_t B, int64_t Lq,
                           int64_t Lm, gnlp_stream_t stream);

/* npx.sequence_mask(x, valid_length, use_sequence_length=True, axis=time)
 * as called at models/bert.py:301-303,311-312 and models/electra.py:292-295,303-304.
 * x/out: (B, L, C) (time_major = 0) or (L, B, C) (time_major = 1); rows with t >= vl[b] become 0.
 * out may alias x. */
int gnlp_sequence_mask(void *out, const void *x, const void *valid_length, int vl_dtype, int dtype,
                       int64_t B, int64_t L, int64_t C, int time_major, gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Softmax / small ops
 * ------------------------------------------------------------------------------------------- */

/* masked_softmax(att_score, mask, axis=-1, temperature)   attention_cell.py:277-302
 * masked_logsoftmax(att_score, mask, axis=-1)             attention_cell.py:305-329 (log_mode=1)
 * scores/out: contiguous, viewed as shape[0..3] with the softmax over shape[3].
 * mask: bytes, addressed with mask_stride[0..3] (0 = broadcast) or NULL (plain softmax).
 * out = softmax(where(mask, x, -inf) / temperature) * mask; a fully masked row gives zeros
 * (log mode: masked positions are -inf).  temperature <= 0 means "none" (1.0). */
int gnlp_masked_softmax(void *out, const void *scores, int dtype, const uint8_t *mask,
                        const int64_t mask_stride[4], const int64_t shape[4], float temperature,
                        int log_mode, gnlp_stream_t stream);

/* l2_normalize(data, axis=-1, eps)   op.py:289-307:  x / (||x||_2 + eps), rows of length d. */
int gnlp_l2_normalize(void *out, const void *x, int dtype, int64_t rows, int64_t d, float eps,
                      gnlp_stream_t stream);

/* get_activation(act)(x)   layers.py:236-338 (GELU erf/tanh/sigmoid, relu, ...), n elements. */
int gnlp_activation(void *out, const void *x, int dtype, int64_t n, int act, gnlp_stream_t stream);

/* The epilogue of Dense as an operation of its own, and its backward (training: mx.autograd
 * differentiates nn.Dense + get_activation + the residual add op by op, layers.py:633-647,
 * models/transformer.py:266-268; the fused GEMM epilogues of gnlp_dense keep no pre-activation,
 * so under autograd the Python layer saves z = x W^T + b and applies these).  n elements, all
 * tensors contiguous and of `dtype`.
 *   gnlp_dense_epilogue:           y = combine(act(z), other)            (other may be NULL)
 *   gnlp_dense_epilogue_backward:  combine add: dz = g act'(z)           (d_other = g: not written)
 *                                  combine mul: dz = g other act'(z), d_other = g act(z) (or NULL)
 * act' uses the exact formulas (erff / tanhf / expf). */
int gnlp_dense_epilogue(void *y, const void *z, const void *other, int dtype, int64_t n, int act,
                        int combine, gnlp_stream_t stream);
int gnlp_dense_epilogue_backward(void *dz, void *d_other, const void *g, const void *z,
                                 const void *other, int dtype, int64_t n, int act, int combine,
                                 gnlp_stream_t stream);

/* nn.LayerNorm / RMSNorm / NoNorm over the last axis   models/transformer.py:217-218,
 * layers.py:63,81-138,151-181.  y = norm(x [+ residual]) * gamma + beta, statistics in fp32.
 * residual, beta may be NULL.  If valid_length != NULL the rows are interpreted as (B, L)
 * (time_major = 0) or (L, B) (time_major = 1) tokens and rows with t >= vl[b] are written as 0
 * (the npx.sequence_mask of models/bert.py:311-312 fused into the last norm). */
int gnlp_norm(void *out, const void *x, const void *residual, const void *gamma, const void *beta,
              int dtype, int kind, float eps, int64_t rows, int64_t C, const void *valid_length,
              int vl_dtype, int64_t B, int64_t L, int time_major, gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Embedding front-end (the callers either side of the encoder stack):
 *   BertModel.get_initial_embedding      models/bert.py:458-504
 *   ElectraModel.get_initial_embedding   models/electra.py:466-510
 *   RobertaModel.get_initial_embedding   models/roberta.py:337-372
 *   out[row] = LayerNorm( word_w[ids[row]] + type_w[token_types[row]] + pos_w[t(row)] )
 * ids / token_types: (B, L) (time_major = 0) or (L, B) (time_major = 1), int32 or int64 (id_dtype);
 * token_types NULL = all zero (bert.py:490-491); type_w NULL = no token-type table (RoBERTa);
 * pos_w NULL = pos_embed_type None; positions are arange over the time axis, clipped to
 * max_length - 1 (LearnedPositionalEmbedding, layers.py:481-482, mode='clip'); ids are clipped to
 * the table.  ln_gamma NULL = no embedding LayerNorm (RoBERTa encoder_normalize_before=False).
 * The pooler (apply_pooling, bert.py:506-529: Dense + tanh on the CLS rows) is gnlp_dense with
 * ldx = the CLS row stride.
 * ------------------------------------------------------------------------------------------- */
int gnlp_embedding_forward(void *out, const void *ids, const void *token_types, int id_dtype,
                           const void *word_w, const void *type_w, const void *pos_w,
                           const void *ln_gamma, const void *ln_beta, int dtype, float eps,
                           int64_t B, int64_t L, int64_t units, int64_t vocab_size,
                           int64_t num_token_types, int64_t max_length, int time_major,
                           gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense (nn.Dense(flatten=False): models/transformer.py:193-206, layers.py:590-612)
 *   out[m, n] = combine( act( sum_k x[m, k] * w[n, k] + bias[n] ), other[m, n] )
 * x: (M, K) row stride ldx;  w: (N, K) contiguous;  out: (M, N) row stride ldo;
 * bias (N) and other (M, N, row stride ldother) may be NULL.  combine = ADD gives the fused
 * residual (models/transformer.py:268, layers.py:644); MUL gives the gated FFN product
 * (layers.py:636-638).  bf16 uses MFMA bf16 with fp32 accumulation, f32 the exact-fp32 MFMA.
 * ------------------------------------------------------------------------------------------- */
int gnlp_dense(void *out, const void *x, const void *w, const void *bias, const void *other,
               int combine, int act, int dtype, int64_t M, int64_t N, int64_t K, int64_t ldx,
               int64_t ldo, int64_t ldother, gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * multi_head_dot_attn(query, key, value, mask, edge_scores, dropout, scaled, normalized, eps,
 *                     query_head_units, layout, use_einsum)      attention_cell.py:332-524
 * The NKT / NTK / TNK layouts are stride descriptors (no transposes): element (b, h, t, c) of
 * q lives at q[b*q_stride[0] + h*q_stride[1] + t*q_stride[2] + c], likewise k, v and ctx.
 * mask: bytes (B, Lq, Lm) addressed with mask_stride[0..2], or NULL.
 * valid_length: if mask == NULL and valid_length != NULL the predicate (i < vl[b]) & (j < vl[b])
 *   is evaluated in-kernel (what gen_self_attn_mask 'full' would have produced); with d->causal
 *   the predicate is (j <= i) & (i < vl[b]) (gen_self_attn_mask 'causal').
 * edge_scores: addressed with edge_stride[0..3] over (b, h, i, j) (0 = broadcast) or NULL; it is
 *   added BEFORE the temperature (attention_cell.py:453-455).
 * temperature: sqrt(query_head_units) when scaled, 0 / 1 otherwise.
 * scores_out / weights_out: (B, H, Lq, Lm) contiguous or NULL; scores_out = QK^T + edge_scores
 *   (not divided by the temperature), weights_out = the softmax (attention_cell.py:524).
 * When both are NULL and the shape allows it (bf16, dk = dv = 64) the fused bf16 MFMA kernel runs;
 * every other call with head dims <= 128 (float32, dk = 8 / dv = 6, scores or weights wanted) runs
 * the general kernel on the f32 matrix cores.  Neither forms (B, H, Lq, Lm) beyond the outputs
 * asked for.  l2-normalisation (normalized=True) is done by the caller with gnlp_l2_normalize.
 * workspace: only needed for head dims > 128 (strided GEMM + softmax + GEMM) when scores_out or
 *   weights_out is NULL (gnlp_attn_workspace_bytes); may be NULL otherwise.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t B, H, Lq, Lm, dk, dv;
  int64_t q_stride[3], k_stride[3], v_stride[3], o_stride[3]; /* (b, h, t) element strides */
  int64_t causal; /* != 0 (mask must be NULL, head dims <= 128): the in-kernel predicate is
                     gen_self_attn_mask(attn_type='causal')'s (attention_cell.py:168-180),
                     (j <= i) & (i < valid_length[b]) -- valid_length may be NULL (lower triangle);
                     key tiles above the diagonal are skipped */
  int64_t mem_lengths; /* != 0 (mask NULL, not causal, valid_length given): valid_length holds
                     2 B entries, the key lengths and then the query lengths, and the in-kernel
                     predicate is gen_mem_attn_mask's (attention_cell.py:186-274),
                     (j < valid_length[b]) & (i < valid_length[B + b]), in the forward and
                     backward kernels (head dims <= 128; GNLP_ERR_UNSUPPORTED beyond: pass the
                     byte mask) */
} gnlp_attn_desc;

size_t gnlp_attn_workspace_bytes(const gnlp_attn_desc *d, int dtype, int want_scores,
                                 int want_weights);

/* Which kernel the last gnlp_multi_head_dot_attn / block call on THIS thread dispatched its
 * attention to (diagnostics and tests; no counterpart in the reference):
 * FUSED_BF16 = attn_bf16.hip (bf16, dk = dv = 64, nothing but ctx wanted), GENERAL_MFMA =
 * attn_gen.hip (float32 or bf16, head dims <= 128, optional scores / weights, on the f32 matrix
 * cores), UNFUSED = strided GEMM + softmax + GEMM on the vector ALU (head dims > 128 only). */
enum { GNLP_ATTN_KERNEL_NONE = 0, GNLP_ATTN_KERNEL_FUSED_BF16 = 1,
       GNLP_ATTN_KERNEL_GENERAL_MFMA = 2, GNLP_ATTN_KERNEL_UNFUSED = 3 };
int gnlp_attn_last_kernel(void);

/* Whether the last gnlp_encoder_stack_forward / gnlp_t5_encoder_forward call on THIS thread ran its
 * ragged batch PACKED to the valid tokens (diagnostics and tests; no counterpart in the reference,
 * which computes every dense op over the padding): with valid_length given, final_seq_mask set,
 * bf16, layer shapes the persistent GEMM takes (B * L a multiple of 256, ...) and no
 * all_encodings, the stack gathers the rows t < valid_length[b] into one (sum vl, units) matrix,
 * runs the layers on it -- GEMMs and norms read the row count from device memory, the attention
 * kernel the sequence offsets; nothing is synchronised with the host -- and scatters the result
 * back with exact zeros beyond valid_length.  Row by row the results are bit-identical to the
 * padded run.  GNLP_PACK_RAGGED=0 (developer knob) disables it. */
int gnlp_stack_last_packed(void);

int gnlp_multi_head_dot_attn(void *ctx, void *scores_out, void *weights_out, const void *q,
                             const void *k, const void *v, const gnlp_attn_desc *d, int dtype,
                             const uint8_t *mask, const int64_t mask_stride[3],
                             const void *valid_length, int vl_dtype, const void *edge_scores,
                             const int64_t edge_stride[4], float temperature, void *workspace,
                             size_t workspace_bytes, gnlp_stream_t stream);

/* gnlp_multi_head_dot_attn for a caller that will ask for gradients (mx.autograd.record() around
 * multi_head_dot_attn, tests/test_attention_cell.py:82-151): ctx as above (no scores / weights),
 * plus lse (B, H, Lq) float32 = log2 of the row's softmax denominator with the scores in base-2
 * units (s * log2(e) / temperature; 0 for a fully masked row) and ctx_lo (bf16, shaped and strided
 * like ctx) = the residual of ctx's rounding to bf16, to be handed to
 * gnlp_multi_head_dot_attn_backward through gnlp_attn_grad_desc.  Only the fused bf16 kernel
 * (bf16, dk = dv = 64, 16-byte aligned rows) produces it: GNLP_ERR_UNSUPPORTED otherwise, nothing
 * launched -- call gnlp_multi_head_dot_attn and leave the grad desc's pointers NULL. */
int gnlp_multi_head_dot_attn_lse(void *ctx, void *ctx_lo, float *lse, const void *q, const void *k,
                                 const void *v, const gnlp_attn_desc *d, int dtype,
                                 const uint8_t *mask, const int64_t mask_stride[3],
                                 const void *valid_length, int vl_dtype, const void *edge_scores,
                                 const int64_t edge_stride[4], float temperature,
                                 gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward pass.  The reference takes these gradients with mx.autograd through
 * the forward ops; its tests drive them through multi_head_dot_attn for query / key / value /
 * rel_scores in every layout (tests/test_attention_cell.py:82-151) and through whole models
 * (tests/test_models_bert.py).  Each call below is the vector-Jacobian product of one forward
 * entry point of this header; nothing of size (B, H, Lq, Lm) is stored between forward and
 * backward (the softmax is recomputed from q, k, edge_scores and the mask).
 * ------------------------------------------------------------------------------------------- */

/* VJP of gnlp_multi_head_dot_attn (attention_cell.py:332-524; normalized = False, no dropout)
 * for grad_ctx = the cotangent of ctx:
 *   P = masked_softmax((q k^T + edge) / t, mask),  dV = P^T dO,  dP = dO V^T,
 *   dS = P (dP - rowsum(P dP)) / t,  dQ = dS k,  dK = dS^T q,  d_edge = dS.
 * d, q, k, v, mask / valid_length, edge_scores, temperature: exactly as in the forward call.
 * gd: (b, h, t) element strides of grad_ctx and of the three outputs (inner stride 1).
 * d_edge: fp32, or NULL when the edge scores need no gradient.  It has the shape of the caller's
 *   edge_scores BEFORE broadcasting: d_edge_stride[0..3] over (b, h, i, j) with 0 along every axis
 *   the scores were broadcast along (the contributions are summed there); d_edge_elems = its
 *   element count.  The call zero-fills it and accumulates with fp32 atomics (the summation order
 *   over the broadcast axes is not fixed); cast with gnlp_cast_from_f32 if a bf16 copy is wanted.
 * workspace: gnlp_attn_backward_workspace_bytes(d) bytes (two fp32 per (b, h, query) row).
 * Head dims <= 128 (the general f32-MFMA kernels of attn_bwd.hip); larger: GNLP_ERR_UNSUPPORTED.
 * bf16 tensors with d_k = d_v = 64 and 16-byte aligned rows run on the bf16 matrix cores
 * (attn_bwd_bf16.hip: P and dS rounded to bf16 as operands, like P in the forward kernel);
 * GNLP_ATTN_BWD_BF16=0 (developer knob) keeps them on the f32-MFMA kernels. */
typedef struct {
  int64_t g_stride[3], dq_stride[3], dk_stride[3], dv_stride[3];
  /* Optional, all three or none: what gnlp_multi_head_dot_attn_lse left for this call -- its ctx
   * and ctx_lo (addressed with d->o_stride) and its lse.  The bf16 kernels then skip their
   * statistics pass (2 of 9 matrix products) and take D = <grad_ctx, ctx + ctx_lo> per row, as the
   * usual flash-attention backward does (but with the output to 16 bits: with the bf16 ctx alone
   * the rounding of ctx reaches the gradients of rows that see only a few keys); the general
   * kernels ignore them and recompute. */
  const void *fwd_ctx;
  const void *fwd_ctx_lo;
  const float *fwd_lse;
} gnlp_attn_grad_desc;

size_t gnlp_attn_backward_workspace_bytes(const gnlp_attn_desc *d);

int gnlp_multi_head_dot_attn_backward(void *dq, void *dk, void *dv, float *d_edge,
                                      int64_t d_edge_elems, const int64_t d_edge_stride[4],
                                      const void *grad_ctx, const void *q, const void *k,
                                      const void *v, const gnlp_attn_desc *d,
                                      const gnlp_attn_grad_desc *gd, int dtype,
                                      const uint8_t *mask, const int64_t mask_stride[3],
                                      const void *valid_length, int vl_dtype,
                                      const void *edge_scores, const int64_t edge_stride[4],
                                      float temperature, void *workspace, size_t workspace_bytes,
                                      gnlp_stream_t stream);

/* VJP of gnlp_norm (without the fused sequence mask): dx (rows, C) in the tensor dtype -- it is
 * also the gradient of `residual` when one was added --, dgamma / dbeta (C) in fp32 (zero-filled
 * by the call, accumulated with fp32 atomics; dbeta may not be NULL).  x, residual, gamma, kind,
 * eps: as in the forward call.  C <= 8192. */
int gnlp_norm_backward(void *dx, float *dgamma, float *dbeta, const void *grad_out, const void *x,
                       const void *residual, const void *gamma, int dtype, int kind, float eps,
                       int64_t rows, int64_t C, gnlp_stream_t stream);

/* VJP of gnlp_l2_normalize (op.py:289-307): dx = g / (n + eps) - x <g, x> / (n (n + eps)^2). */
int gnlp_l2_normalize_backward(void *dx, const void *grad_out, const void *x, int dtype,
                               int64_t rows, int64_t d, float eps, gnlp_stream_t stream);

/* VJP of gnlp_rel_attn_score (RelAttentionScoreCell 'transformer_xl' / 'shaw',
 * attention_cell.py:816-875; the gradients tests/test_attention_cell.py:313-391 takes): for
 * grad_out (B, H, Lq, Lm) contiguous,
 *   dq[b, h, i, :]            = sum_j grad_out[b, h, i, j] pos_embed[idx(i, j), h, :]
 *   d_pos_embed[idx, h, :]   += grad_out[b, h, i, j] query[b, h, i, :]
 * dq (tensor dtype, addressed with dq_stride like the query) or NULL; d_pos_embed (R, H, C) fp32 or
 * NULL, zero-filled by the call, accumulated with fp32 atomics.  Everything else as in the forward. */
int gnlp_rel_attn_score_backward(void *dq, const int64_t *dq_stride, float *d_pos_embed,
                                 const void *grad_out, const void *query, const int64_t *q_stride,
                                 const void *pos_embed, const int32_t *rel_positions, int dtype,
                                 int64_t B, int64_t H, int64_t Lq, int64_t Lm, int64_t C, int64_t R,
                                 int32_t clip_lo, int32_t clip_hi, int32_t base, int wrap,
                                 gnlp_stream_t stream);

/* VJP of gnlp_bucket_embedding (BucketPositionalEmbedding, layers.py:486-524):
 * d_weight[bucket(rel[i]), :] += grad_out[i, :]; d_weight (num_buckets, units) fp32, zero-filled
 * by the call. */
int gnlp_bucket_embedding_backward(float *d_weight, const void *grad_out, const int32_t *rel,
                                   int dtype, int64_t n, int64_t units, int bidirectional,
                                   int num_buckets, int max_distance, gnlp_stream_t stream);

/* out[c] = sum_r x[r, c], fp32 (zero-filled by the call): the bias gradient of a Dense layer.  Its
 * other two gradients are gnlp_dense calls on transposed operands (dx = g W, dW = g^T x). */
int gnlp_colsum(float *out, const void *x, int dtype, int64_t rows, int64_t C,
                gnlp_stream_t stream);

/* out[i] = (dtype) in[i]: the fp32 accumulators above -> a parameter's dtype.  No counterpart in
 * the reference (MXNet accumulates in the parameter dtype). */
int gnlp_cast_from_f32(void *out, int dtype, const float *in, int64_t n, gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * T5 relative-position pieces
 * ------------------------------------------------------------------------------------------- */

/* gen_rel_position(data, layout)   attention_cell.py:611-656: out[i, j] = j - i, int32 (L, L). */
int gnlp_gen_rel_position(int32_t *out, int64_t L, gnlp_stream_t stream);

/* relative_position_bucket(relative_position, bidirectional, num_buckets, max_distance)
 * op.py:229-286.  int32 -> int32, n elements, bit-exact with the float32-log formulation. */
int gnlp_relative_position_bucket(int32_t *out, const int32_t *rel, int64_t n, int bidirectional,
                                  int num_buckets, int max_distance, gnlp_stream_t stream);

/* BucketPositionalEmbedding.forward   layers.py:519-524: out[i, :] = weight[bucket(rel[i]), :],
 * weight (num_buckets, units). */
int gnlp_bucket_embedding(void *out, const void *weight, const int32_t *rel, int dtype, int64_t n,
                          int64_t units, int bidirectional, int num_buckets, int max_distance,
                          gnlp_stream_t stream);

/* RelAttentionScoreCell(method='t5').forward(gen_rel_position(...))   attention_cell.py:876-878
 * fused: out (H, Lq, Lm), out[h, i, j] = weight[bucket(j - i), h]. */
int gnlp_t5_rel_bias(void *out, const void *weight, int dtype, int64_t H, int64_t Lq, int64_t Lm,
                     int bidirectional, int num_buckets, int max_distance, gnlp_stream_t stream);

/* The same bias as one row per head: out (H, Lq + Lm - 1), out[h, t] = weight[bucket(t - (Lq-1)), h],
 * i.e. bias[h, i, j] = out[h, (Lq - 1) + j - i].  gen_rel_position gives j - i only
 * (attention_cell.py:653-655), so the (H, Lq, Lm) tensor of :876-878 is Toeplitz. */
int gnlp_t5_rel_bias_table(void *out, const void *weight, int dtype, int64_t H, int64_t Lq,
                           int64_t Lm, int bidirectional, int num_buckets, int max_distance,
                           gnlp_stream_t stream);

/* SinusoidalPositionalEmbedding.forward   layers.py:429-450: out (n, units) =
 * concat(sin(p * base_mult), cos(p * base_mult) [, 0 when units is odd]); positions float32 (n),
 * base_mult float32 (units / 2) = exp(arange(units/2) * -log(10000) / (units/2 - 1)) (:418-422). */
int gnlp_sinusoidal_embedding(void *out, const float *positions, const float *base_mult, int dtype,
                              int64_t n, int64_t units, gnlp_stream_t stream);

/* RelAttentionScoreCell(method='transformer_xl' | 'shaw').forward   attention_cell.py:816-875:
 * out (B, H, Lq, Lm), out[b,h,i,j] = <query[b,h,i,:], pos_embed[idx(rel[i,j]), h, :]> with
 * idx(r) = clip(r, clip_lo, clip_hi) - base (+ R when negative and `wrap`, np.take mode='wrap').
 * query: C contiguous, q_stride = element strides of (batch, head, time) -- NKT / NTK / TNK are
 * stride descriptors; pos_embed (R, H, C) contiguous = the (projected) embeddings of relative
 * positions base .. base + R - 1 ('transformer_xl') or the learned table itself ('shaw');
 * rel_positions (Lq, Lm) int32.  Replaces unique + matmul + gather_nd (:827-875) by one kernel. */
int gnlp_rel_attn_score(void *out, const void *query, const int64_t *q_stride,
                        const void *pos_embed, const int32_t *rel_positions, int dtype, int64_t B,
                        int64_t H, int64_t Lq, int64_t Lm, int64_t C, int64_t R, int32_t clip_lo,
                        int32_t clip_hi, int32_t base, int wrap, gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole blocks (one C call launches every kernel of the block on `stream`)
 * ------------------------------------------------------------------------------------------- */

/* PositionwiseFFN.forward   layers.py:619-647 */
typedef struct {
  const void *ffn1_w, *ffn1_b;     /* (F, U), (F) */
  const void *gate_w, *gate_b;     /* gated_ffn_1, NULL unless use_gated_activation */
  const void *ffn2_w, *ffn2_b;     /* (U, F), (U) */
  const void *ln_g, *ln_b;         /* layer_norm.gamma / beta (beta may be NULL) */
} gnlp_ffn_params;

typedef struct {
  int64_t units, hidden;
  int act, gated, norm_kind, pre_norm, dtype;
  float ln_eps;
} gnlp_ffn_cfg;

size_t gnlp_ffn_workspace_bytes(const gnlp_ffn_cfg *cfg, int64_t rows);
int gnlp_positionwise_ffn(void *out, const void *x, const gnlp_ffn_params *p,
                          const gnlp_ffn_cfg *cfg, int64_t rows, const void *valid_length,
                          int vl_dtype, int64_t B, int64_t L, int time_major, void *workspace,
                          size_t workspace_bytes, gnlp_stream_t stream);

/* TransformerEncoderLayer.forward(data, attn_mask)   models/transformer.py:234-272 */
typedef struct {
  const void *qkv_w, *qkv_b;       /* attn_qkv (3U, U), (3U) or NULL */
  const void *proj_w, *proj_b;     /* attention_proj (U, U), (U) */
  const void *ln_g, *ln_b;         /* layer_norm */
  gnlp_ffn_params ffn;             /* ffn.* */
} gnlp_encoder_layer_params;

typedef struct {
  int64_t units, hidden, heads;
  int act, pre_norm, dtype, time_major; /* time_major: 0 = 'NT' (B, L, U), 1 = 'TN' (L, B, U) */
  float ln_eps;
} gnlp_encoder_cfg;

size_t gnlp_encoder_workspace_bytes(const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                                    int want_weights);

/* mask (B, L, L) bytes or NULL; valid_length or NULL (see gnlp_multi_head_dot_attn).
 * attn_weights_out (B, H, L, L) or NULL.  final_seq_mask != 0 fuses npx.sequence_mask into the
 * last norm (needs valid_length).  out may not alias x. */
int gnlp_encoder_layer_forward(void *out, const void *x, const gnlp_encoder_layer_params *p,
                               const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                               const uint8_t *mask, const void *valid_length, int vl_dtype,
                               void *attn_weights_out, int final_seq_mask, void *workspace,
                               size_t workspace_bytes, gnlp_stream_t stream);

/* gnlp_encoder_layer_forward for callers that CANNOT allocate device memory or size a workspace:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference.  Same arguments and results (bit for bit: the same kernels) as
 * gnlp_encoder_layer_forward minus the workspace: the layer's scratch (packed QKV, context,
 * pre-norm sums, FFN hidden) lives in a static device array of GNLP_NOALLOC_SCRATCH_BYTES inside
 * the library -- every GPU the library runs on holds one copy until the process ends.  The call
 * performs no hipMalloc / hipFree and calls no allocating routine; on its first use on a device it
 * looks the array's address up once (hipGetSymbolAddress): make one warm-up call per device OUTSIDE
 * any graph capture.  The stream must belong to the CURRENT device (each GPU has its own copy of
 * the array; a mismatch is refused with GNLP_ERR_ARG).  A layer whose scratch
 * (gnlp_encoder_workspace_bytes minus the stack's two ping-pong activations) exceeds the array is
 * refused with GNLP_ERR_ARG and a message.  Every call on a GPU shares that GPU's array: calls that
 * overlap in time on one GPU (different streams or threads) RACE -- serialise them, or use the
 * workspace form.  Nothing in the array has to start at zero (each call writes every region
 * before reading it), so the previous call's contents are harmless. */
#define GNLP_NOALLOC_SCRATCH_BYTES ((size_t)64 << 20)
int gnlp_encoder_layer_forward_noalloc(void *out, const void *x,
                                       const gnlp_encoder_layer_params *p,
                                       const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                                       const uint8_t *mask, const void *valid_length,
                                       int vl_dtype, void *attn_weights_out, int final_seq_mask,
                                       gnlp_stream_t stream);

/* BertTransformer / RobertaEncoder / ElectraEncoder / TransformerEncoder (post-norm) .forward
 * models/bert.py:259-315, roberta.py:171-188, electra.py:249-307, transformer.py:346-385.
 * Runs n_layers layers back to back; mask is the valid_length predicate.
 * final_seq_mask: 1 for BERT/ELECTRA (bert.py:311-312), 0 for RoBERTa (roberta.py:184).
 * all_encodings: NULL, or n_layers pointers receiving every layer's output
 *   (output_all_encodings; with final_seq_mask each one is masked and fed on, bert.py:300-304). */
int gnlp_encoder_stack_forward(void *out, const void *x, const gnlp_encoder_layer_params *layers,
                               int n_layers, const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                               const void *valid_length, int vl_dtype, int final_seq_mask,
                               void *const *all_encodings, void *workspace, size_t workspace_bytes,
                               gnlp_stream_t stream);

/* The same stacks with their LayerNorms FOLDED into the GEMMs around them (bf16, post-norm; no
 * counterpart in the reference, whose layers run nn.LayerNorm as an operator of its own,
 * models/transformer.py:268-270, layers.py:642-646 -- same arithmetic regrouped).  For a row s with
 * mean mu and r = rsqrt(var + eps):
 *     LayerNorm(s) . W^T + b  =  r (s . W'^T) - (r mu) c + b',
 *     W' = bf16(gamma * W),  c[n] = sum_k W'[n, k],  b'[n] = sum_k beta[k] W[n, k] + b[n],
 * so attn_qkv and ffn_1 take the RAW sums as their A operand and apply the two per-row scalars in
 * the GEMM drain, and the residual operands of attention_proj / ffn_2 (= LayerNorm(s)) are rebuilt
 * in their drains from s, the statistics and gamma / beta.  The normalised tensors are never
 * written; only the last layer's norm (+ sequence mask) runs as a kernel.
 *
 * gnlp_fold_norm_dense computes (W', c, b') of one Dense on the device; call it once per
 * parameter version.  The first n_scaled output rows are additionally multiplied by `scale`
 * (pass 0 / 1.0f for none).  gnlp_row_stats writes (r, r * mu) per row, fp32, shape (rows, 2).
 * A stack whose shapes the folded kernels do not take (rows not a multiple of 256, units > 2048,
 * a GEMM that is not on the persistent kernel, ...) runs exactly as gnlp_encoder_stack_forward;
 * gnlp_stack_last_folded() tells which.  GNLP_LN_FOLD=0 switches the folded path off. */
typedef struct {
  const void *qkv_wf;              /* attn_qkv folded with the PREVIOUS layer's ffn.layer_norm */
  const float *qkv_c, *qkv_bf;     /*   (3U, U) bf16, (3U) fp32 x 2; ignored for the first layer */
  const void *ffn1_wf;             /* ffn.ffn_1 folded with this layer's layer_norm */
  const float *ffn1_c, *ffn1_bf;   /*   (F, U) bf16, (F) fp32 x 2 */
} gnlp_encoder_layer_folded;

int gnlp_fold_norm_dense(void *wf, float *c, float *bf, const void *w, const void *b,
                         const void *gamma, const void *beta, int dtype, int64_t N, int64_t K,
                         int64_t n_scaled, float scale, gnlp_stream_t stream);
int gnlp_row_stats(float *rstats, const void *x, int dtype, float eps, int64_t rows, int64_t C,
                   gnlp_stream_t stream);
int gnlp_encoder_stack_forward_folded(void *out, const void *x,
                                      const gnlp_encoder_layer_params *layers,
                                      const gnlp_encoder_layer_folded *folded, int n_layers,
                                      const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                                      const void *valid_length, int vl_dtype, int final_seq_mask,
                                      void *workspace, size_t workspace_bytes,
                                      gnlp_stream_t stream);
int gnlp_stack_last_folded(void);

/* T5Block.forward (encoder) / T5Encoder.forward   models/t5.py:492-572, 666-711 */
typedef struct {
  const void *attn_ln_g, *attn_ln_b;     /* self_attn_layer_norm.gamma / .beta (beta may be NULL) */
  const void *q_w, *k_w, *v_w, *proj_w;  /* (inner, d_model) x3, (d_model, inner); bias-free */
  gnlp_ffn_params ffn;                   /* ffn.* (pre-norm RMS, optional gate) */
  const void *qkv_w;                     /* optional: the three projection weights packed row-wise,
                                            (3 * inner, d_model) = [q_w; k_w; v_w]; when non-NULL
                                            ONE GEMM of N = 3 * inner replaces the three (same
                                            results bit for bit); NULL = use q_w / k_w / v_w */
} gnlp_t5_block_params;

typedef struct {
  int64_t d_model, d_kv, d_ff, heads;
  int gated, dtype, time_major;          /* gated: 0 = relu, 1 = gated-gelu(tanh) */
  float ln_eps;
  int bias_toeplitz;                     /* position_bias is the (H, 2L-1) table of gnlp_t5_rel_bias_table */
} gnlp_t5_cfg;

size_t gnlp_t5_workspace_bytes(const gnlp_t5_cfg *cfg, int64_t B, int64_t L);

/* position_bias: (H, L, L) from gnlp_t5_rel_bias, or with cfg->bias_toeplitz the (H, 2L-1) table
 * from gnlp_t5_rel_bias_table (the bias depends on j - i only; the kernels read the table through
 * a row stride of -1, so the (H, L, L) tensor is never formed).  final_ln_g / final_ln_b: final RMSNorm gamma /
 * beta (final_ln_g == NULL = run the blocks only; beta may be NULL).  The RMSNorm inside the FFN
 * uses eps 1e-6 whatever cfg->ln_eps says, as the reference does (layers.py:67-68 does not
 * forward epsilon to RMSNorm). */
int gnlp_t5_encoder_forward(void *out, const void *x, const gnlp_t5_block_params *layers,
                            int n_layers, const void *final_ln_g, const void *final_ln_b,
                            const gnlp_t5_cfg *cfg,
                            int64_t B, int64_t L, const void *valid_length, int vl_dtype,
                            const void *position_bias, void *workspace, size_t workspace_bytes,
                            gnlp_stream_t stream);

/* The same forward with every RMSNorm folded into the GEMM that consumes it (round 3; no
 * counterpart in the reference, whose blocks run RMSNorm as an operator of its own,
 * models/t5.py:537,571, layers.py:634-635 -- the same arithmetic regrouped):
 *     RMSNorm(h) . W^T = r (h . W'^T) + b',   W' = bf16(gamma * W),  b' = beta . W^T,
 *     r = rsqrt(mean(h^2) + eps)
 * (gnlp_fold_norm_dense builds W', c, b'; c goes unused: there is no mean term).  The packed
 * q | k | v weights are folded with the block's self_attn_layer_norm, ffn_1 with the FFN's
 * layer_norm.  The residual GEMMs (proj, ffn_2) leave partial sums of squares of their output rows,
 * from which the next norm's r comes without a pass over the tensor.  Conditions: bf16, relu FFN
 * (not gated), qkv_w packed, B * L a multiple of 256, d_model <= 2048, all four GEMM shapes taken
 * by the persistent kernel; otherwise -- and with GNLP_LN_FOLD=0 -- exactly
 * gnlp_t5_encoder_forward runs.  gnlp_stack_last_folded() tells which. */
typedef struct {
  const void *qkv_wf;              /* (3 inner, d_model) bf16 */
  const float *qkv_c, *qkv_bf;     /* (3 inner) fp32 x 2 */
  const void *ffn1_wf;             /* (d_ff, d_model) bf16 */
  const float *ffn1_c, *ffn1_bf;   /* (d_ff) fp32 x 2 */
} gnlp_t5_block_folded;
int gnlp_t5_encoder_forward_folded(void *out, const void *x, const gnlp_t5_block_params *layers,
                                   const gnlp_t5_block_folded *folded, int n_layers,
                                   const void *final_ln_g, const void *final_ln_b,
                                   const gnlp_t5_cfg *cfg, int64_t B, int64_t L,
                                   const void *valid_length, int vl_dtype,
                                   const void *position_bias, void *workspace,
                                   size_t workspace_bytes, gnlp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hook (no counterpart in the reference; for bench.py's roofline object).  While
 * armed, every dense launch of exactly this (M, N, K) -- from gnlp_dense or from inside the block
 * / stack entry points -- is bracketed by a pair of HIP events on its own stream, up to `capacity`
 * launches; gnlp_profile_dense_read waits for the recorded pairs and returns their count, filling
 * ms[i] with the durations.  Process-global, not for concurrent use; disarm frees the events.
 * ------------------------------------------------------------------------------------------- */
int gnlp_profile_dense_arm(int64_t M, int64_t N, int64_t K, int capacity);
int gnlp_profile_dense_read(float *ms, int capacity);
int gnlp_profile_dense_disarm(void);

#ifdef __cplusplus
}
#endif
#endif /* GLUONNLP_AMD_H_ */
