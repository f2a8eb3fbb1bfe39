// Internal launch interface between api.hip (C ABI, argument checking) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace gnlp {

// Function attributes (hipFuncSetAttribute) are per device: run `f` once per device that launches
// the kernel.  Racing threads may both run it, which is harmless; none launches before it ran.
struct PerDeviceFlag {
  std::atomic<uint64_t> mask{0};
  template <class F>
  void once(F f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(mask.load(std::memory_order_acquire) & bit)) {
      f();
      mask.fetch_or(bit, std::memory_order_release);
    }
  }
};


// Developer knobs (GNLP_* environment variables), read ONCE per process into this struct -- the
// launch paths never call getenv.  gnlp_reload_knobs() (C ABI) re-reads them: tests and A/B
// scripts that flip a knob mid-process call it.  Values are clamped where they index anything.
struct Knobs {
  int group_m;      // GNLP_GROUP_M: tile-walk band height, clamped to [1, 64]; default 8
  int gemm_tile;    // GNLP_GEMM_TILE: 0 = by shape, 128 / 256 forced
  bool gemm_w4;     // GNLP_GEMM_W4=0: one tile per workgroup (gemm_bf16_pp.hip) instead of the persistent kernel
  int gemm_nt;      // GNLP_GEMM_NT: 0 = by shape, 192 / 256 forced tile width
  bool norm_nt;     // GNLP_NORM_NT=0: plain instead of non-temporal loads in the norm kernel
  bool norm_full;   // GNLP_NORM_MAXV=f: always the full-width register image
  bool pack_ragged; // GNLP_PACK_RAGGED=0: the stack drivers never pack a ragged batch
  bool attn_bwd_bf16; // GNLP_ATTN_BWD_BF16=0: attention backward on the fp32 matrix cores for bf16 too
  bool ln_fold;     // GNLP_LN_FOLD=0: stacks never run with their LayerNorms folded into the GEMMs
  int skinny_max_m; // GNLP_SKINNY_MAX_M: bf16 Dense calls with at most this many rows take the
                    // K-split kernel (gemm_bf16.hip); 0 = never
};
const Knobs &knobs();
void reload_knobs();

// relative_position_bucket thresholds (see elementwise.hip); passed by value as a kernel argument
struct BucketTable {
  int bidirectional;
  int half;        // num_buckets (after the bidirectional halving)
  int max_exact;
  int n_thr;       // number of valid thresholds
  int thr[128];    // thr[q] = smallest n whose (unclipped) bucket is >= max_exact + q + 1
};

// relscore.hip
struct RelScoreArgs {
  int64_t B, H, Lq, Lm, C, R;
  int64_t qs[3];        // query strides in elements for (batch, head, time)
  int lo, hi;           // clip range of the relative positions (INT_MIN / INT_MAX: none)
  int base;             // table row of relative position `base` is row 0
  int wrap;             // negative indices wrap by R (np.take mode='wrap')
};
void launch_rel_score(void *out, const void *q, const void *emb, const int32_t *rel, int dtype,
                      const RelScoreArgs &a, hipStream_t s);
void launch_rel_score_bwd(void *dq, const int64_t dq_stride[3], float *demb, const void *g,
                          const void *q, const void *emb, const int32_t *rel, int dtype,
                          const RelScoreArgs &a, hipStream_t s);
void launch_bucket_embedding_bwd(float *dweight, const void *g, const int32_t *rel, int dtype,
                                 int64_t n, int64_t units, const BucketTable &tb, hipStream_t s);
void launch_colsum(float *out, const void *g, int dtype, int64_t rows, int64_t C, hipStream_t s);
void launch_sinusoid(void *out, const float *pos, const float *base, int dtype, int64_t n,
                     int64_t units, hipStream_t s);

// elementwise.hip
void launch_self_mask(uint8_t *mask, const void *vl, int vl_dtype, int64_t B, int64_t L,
                      int attn_type, hipStream_t s);
void launch_mem_mask(uint8_t *mask, const void *mem_vl, const void *data_vl, int vl_dtype,
                     int64_t B, int64_t Lq, int64_t Lm, hipStream_t s);
void launch_sequence_mask(void *out, const void *x, const void *vl, int vl_dtype, int dtype,
                          int64_t B, int64_t L, int64_t C, int time_major, hipStream_t s);
void launch_activation(void *out, const void *x, int dtype, int64_t n, int act, hipStream_t s);
void launch_dense_epilogue(void *y, const void *z, const void *other, int dtype, int64_t n, int act,
                           int mul, hipStream_t s);
void launch_dense_epilogue_bwd(void *dz, void *d_other, const void *g, const void *z,
                               const void *other, int dtype, int64_t n, int act, int mul,
                               hipStream_t s);
void launch_l2_normalize(void *out, const void *x, int dtype, int64_t rows, int64_t d, float eps,
                         hipStream_t s);
void launch_rel_position(int32_t *out, int64_t L, hipStream_t s);
void launch_bucket(int32_t *out, const int32_t *rel, int64_t n, const BucketTable &tb,
                   hipStream_t s);
void launch_bucket_embedding(void *out, const void *weight, const int32_t *rel, int dtype,
                             int64_t n, int64_t units, const BucketTable &tb, hipStream_t s);
void launch_t5_rel_bias(void *out, const void *weight, int dtype, int64_t H, int64_t Lq,
                        int64_t Lm, int rel0, const BucketTable &tb, hipStream_t s);

// embed.hip (false: shape / alignment not supported by the vector kernel)
bool launch_embedding(void *out, const void *ids, const void *types, int id_dtype,
                      const void *word_w, const void *type_w, const void *pos_w, const void *gamma,
                      const void *beta, int dtype, float eps, int64_t B, int64_t L, int64_t C,
                      int time_major, int64_t vocab, int64_t n_types, int64_t max_len,
                      hipStream_t s);

// norm.hip
void launch_norm(void *out, const void *x, const void *residual, const void *gamma,
                 const void *beta, int dtype, int kind, float eps, int64_t rows, int64_t C,
                 const void *vl, int vl_dtype, int64_t B, int64_t L, int time_major,
                 hipStream_t s);

// LayerNorm folded into the GEMMs around it (norm.hip): row statistics (r, r * mean) of bf16 rows,
// and the folded parameters (wf, c, bf) of the Dense that consumes the norm (see DenseFold)
bool row_stats_supported(int64_t C, const void *x);
void launch_row_stats(float *rstats, const void *x, float eps, int64_t rows, int64_t C,
                      hipStream_t s, bool rms = false);
void launch_stats_finish(float *rstats, const float *part, int64_t slots, float eps, int64_t rows,
                         int64_t C, hipStream_t s, bool rms = false);
void launch_fold_consts(void *ones, void *zeros, float *rs_one, int64_t C, int64_t rows,
                        hipStream_t s);
void launch_fold_norm_dense(void *wf, float *c, float *bfold, const void *w, const void *b,
                            const void *gamma, const void *beta, int64_t N, int64_t K,
                            int64_t n_scaled, float scale, hipStream_t s);

// softmax.hip
void launch_masked_softmax(void *out, const void *scores, int dtype, const uint8_t *mask,
                           const int64_t mask_stride[4], const int64_t shape[4],
                           const void *vl, int vl_dtype, float inv_temperature, int log_mode,
                           hipStream_t s);

// gemm_generic.hip : batched strided GEMM on the vector ALU (any shape; fp32 accumulate)
struct GenericGemm {
  const void *A;   // (batch, M, K): A[b0*sAb0 + b1*sAb1 + m*sAm + k]
  const void *Bm;  // trans_b: (N, K) element (n,k) at n*sBn + k ; else (K, N): k*sBn + n
  void *C;         // C[b0*sCb0 + b1*sCb1 + m*sCm + n]
  const void *bias;   // (N) or null
  const void *other;  // same addressing family as C with its own strides, or null
  int64_t M, N, K, nb0, nb1;
  int64_t sAb0, sAb1, sAm;
  int64_t sBb0, sBb1, sBn;
  int64_t sCb0, sCb1, sCm;
  int64_t sOb0, sOb1, sOm, sOn;  // `other` strides (sOn may be 0/1; 0-strides broadcast)
  int trans_b, act, combine, in_dtype, out_dtype, other_dtype;
  float alpha;   // C = combine(act(alpha * A.B + bias), other)
};
void launch_generic_gemm(const GenericGemm &g, hipStream_t s);

// gemm_bf16.hip / gemm_f32.hip : MFMA dense  out = combine(act(x.w^T + bias), other)
bool mfma_dense_act_fused(int act);   // activations the MFMA epilogues implement
bool mfma_bf16_dense_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                               int64_t ldother, const void *x, const void *w, const void *out,
                               const void *other);
void launch_mfma_bf16_dense(void *out, const void *x, const void *w, const void *bias,
                            const void *other, int combine, int act, int64_t M, int64_t N,
                            int64_t K, int64_t ldx, int64_t ldo, int64_t ldother, hipStream_t s);
// A LayerNorm folded into the bf16 GEMMs around it (persistent kernel only).
// For a row with statistics mu, r = rsqrt(var + eps):  LN(s) . W^T + b  =
//   r (s . W'^T) - (r mu) c + b'   with W' = bf16(gamma * W), c[n] = sum_k W'[n, k], b' = beta . W^T + b
struct DenseFold {
  int mode;              // 0 none
                         // 1 x holds the raw sums s and w the folded weights: out = act(r acc - (r mu) c + b')
                         // 2 `other` holds raw sums: out = acc + bias + gamma (r s - r mu) + beta
  const float *rstats;   // (rows, 2) fp32: r, r * mu of the rows of x (mode 1) / of `other` (mode 2)
  const float *c, *b;    // mode 1: (N) fp32
  const void *ln_g, *ln_b;   // mode 2: (N) bf16
  float *stats_part;     // mode 2, out: (rows, fold_stat_slots(N), 2) fp32 partial (sum, sum of
                         // squares) of the OUTPUT rows -- launch_stats_finish turns them into the
                         // next norm's rstats
};
int64_t fold_stat_slots(int64_t M, int64_t N);   // gemm_bf16_w4.hip: 4 per tile column
bool mfma_bf16_dense_w4_fold_supported(const DenseFold &f, int act, int combine, const void *other,
                                       int64_t N);
// One-wave-per-SIMD form (gemm_bf16_w4.hip): PERSISTENT, 4 waves of 128x128; M a multiple of 256,
// N a multiple of 256 or of 192 (256x192 tiles, 4 waves of 128x96).
bool mfma_bf16_dense_w4_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                                  int64_t ldother, const void *out, const void *bias,
                                  const void *other);
void launch_mfma_bf16_dense_w4(void *out, const void *x, const void *w, const void *bias,
                               const void *other, int combine, int act, int64_t M, int64_t N,
                               int64_t K, int64_t ldx, int64_t ldo, int64_t ldother,
                               hipStream_t s, const DenseFold *fold = nullptr);
bool mfma_bf16_dense_pp_supported(int64_t N, int64_t ldo, int64_t ldother, const void *out,
                                  const void *bias, const void *other);
void launch_mfma_bf16_dense_pp(void *out, const void *x, const void *w, const void *bias,
                               const void *other, int combine, int act, int64_t M, int64_t N,
                               int64_t K, int64_t ldx, int64_t ldo, int64_t ldother,
                               hipStream_t s);
bool mfma_f32_dense_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                              int64_t ldother, const void *x, const void *w, const void *out,
                              const void *other);
void launch_mfma_f32_dense(void *out, const void *x, const void *w, const void *bias,
                           const void *other, int combine, int act, int64_t M, int64_t N,
                           int64_t K, int64_t ldx, int64_t ldo, int64_t ldother, hipStream_t s);

// attn_bf16.hip : fused flash-style attention (bf16, dk = dv = 64)
struct FusedAttn {
  const void *q, *k, *v;
  void *o;
  int64_t B, H, Lq, Lm;
  int64_t qs[3], ks[3], vs[3], os[3];   // (b, h, t) element strides
  const uint8_t *mask;                  // or null
  int64_t ms[3];                        // (b, i, j)
  const void *vl;                       // or null (predicate i<vl && j<vl)
  int vl_dtype;
  int causal;                           // mask null: predicate (j <= i) && (i < vl) instead
  int vl_pair;                          // vl holds 2 B entries: key lengths, then query lengths
                                        // (gen_mem_attn_mask): predicate (j < vl[b]) && (i < vl[B + b])
  const int32_t *cu;                    // or null; packed ragged batch: sequence b holds token rows
                                        // cu[b] .. cu[b] + vl[b] - 1 (vl required, batch strides 0)
  const void *bias;                     // edge scores (bf16) or null
  int64_t bs[4];                        // (b, h, i, j)
  int bias_f32;                         // bias holds float32 (no mask, not causal): the stack's copy
  float inv_temperature;
  float *lse;                           // or null; (B, H, Lq) fp32: log2 of the softmax denominator of
                                        // the scores in base-2 units (s * log2(e) / temperature), what
                                        // the backward kernels rebuild P from; 0 for a fully masked row
  void *o_lo;                           // with lse: bf16, laid out like o: bf16(ctx_fp32 - bf16(ctx_fp32)),
                                        // so that the backward's D = <dO, O> sees O to 16 bits
};
void launch_widen_bf16(float *out, const void *in, int64_t n, hipStream_t s);
bool fused_attn_supported(const FusedAttn &a, int64_t dk, int64_t dv);
void launch_fused_attn_bf16(const FusedAttn &a, hipStream_t s);

// Ragged batches packed to their valid tokens (elementwise.hip): sequence offsets, gather, scatter.
void launch_seq_offsets(int32_t *cu, const void *vl, int vl_dtype, int64_t B, int64_t L,
                        hipStream_t s);
void launch_pack_rows(void *packed, const void *x, const int32_t *cu, int dtype, int64_t B,
                      int64_t L, int64_t C, int time_major, hipStream_t s);
void launch_unpack_rows(void *out, const void *packed, const int32_t *cu, int dtype, int64_t B,
                        int64_t L, int64_t C, int time_major, hipStream_t s);
void launch_zero_pad_rows(void *buf, const int32_t *cu, int dtype, int64_t B, int64_t C,
                          hipStream_t s);
// While a packed stack runs (api.hip sets it, per host thread), the persistent GEMM and the norm
// kernels read the number of rows to process from this device word instead of the host-side M.
const int32_t *dense_rows_dev();
// whether launch_mfma_bf16_dense would take the persistent kernel (the only GEMM that reads it)
bool mfma_bf16_dense_takes_w4(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                              int64_t ldother, const void *out, const void *bias,
                              const void *other);

// attn_gen.hip : general attention on the f32 matrix cores (float32 or bf16 tensors, head dims
// <= 128, optional `scores` / `attn_weights` outputs); no workspace
struct GenAttn {
  const void *q, *k, *v;
  void *o;
  void *scores_out, *weights_out;       // (B, H, Lq, Lm) contiguous, or null
  int64_t B, H, Lq, Lm, dk, dv;
  int64_t qs[3], ks[3], vs[3], os[3];   // (b, h, t) element strides
  const uint8_t *mask;                  // or null
  int64_t ms[3];                        // (b, i, j)
  const void *vl;                       // or null (predicate i<vl && j<vl, used when mask is null)
  int vl_dtype;
  int causal;                           // mask null: predicate (j <= i) && (i < vl) instead
  int vl_pair;                          // vl holds 2 B entries: key lengths, then query lengths
                                        // (gen_mem_attn_mask): predicate (j < vl[b]) && (i < vl[B + b])
  const void *bias;                     // edge scores (tensor dtype) or null
  int64_t bs[4];                        // (b, h, i, j)
  float inv_temperature;
};
bool gen_attn_supported(const GenAttn &a);
void launch_gen_attn(const GenAttn &a, int dtype, hipStream_t s);

// attn_bwd.hip: gradients of the attention above w.r.t. q / k / v / edge scores
struct GenAttnBwd {
  const void *q, *k, *v, *g;            // g = cotangent of ctx
  void *dq, *dk_out, *dv_out;           // tensor dtype
  float *stats;                         // (B, H, Lq, 2) fp32 workspace: lse, D
  float *de;                            // fp32, zero-filled, or null
  int64_t B, H, Lq, Lm, dk, dv;
  int64_t qs[3], ks[3], vs[3], gs[3], dqs[3], dks[3], dvs[3];   // (b, h, t) element strides
  int64_t des[4];                       // (b, h, i, j); 0 along the axes the bias was broadcast
  const uint8_t *mask;
  int64_t ms[3];
  const void *vl;
  int vl_dtype;
  int causal;
  int vl_pair;                          // as GenAttn::vl_pair
  const void *bias;
  int64_t bs[4];
  float inv_temperature;
  // saved by the forward (both or neither; only the bf16 kernels use them): its output and its
  // FusedAttn::lse -- the statistics pass is skipped, D = <dO, O>
  const float *fwd_lse;
  const void *fwd_o, *fwd_o_lo;
  int64_t os[3];
};
bool attn_bwd_supported(const GenAttnBwd &a);
void launch_attn_bwd(const GenAttnBwd &a, int dtype, hipStream_t s);
// attn_bwd_bf16.hip: bf16 tensors with 64-wide heads on the bf16 matrix cores
bool attn_bwd_bf16_supported(const GenAttnBwd &a);
void launch_attn_bwd_bf16(const GenAttnBwd &a, hipStream_t s);

// norm_bwd.hip
constexpr int64_t kNormBwdMaxC = 8192;   // 2 * C fp32 LDS accumulators per workgroup
bool norm_bwd_supported(int64_t C);
void launch_norm_bwd(void *dx, float *dgamma, float *dbeta, const void *g, const void *x,
                     const void *residual, const void *gamma, int dtype, int kind, float eps,
                     int64_t rows, int64_t C, hipStream_t s);
void launch_l2_normalize_bwd(void *dx, const void *g, const void *x, int dtype, int64_t rows,
                             int64_t d, float eps, hipStream_t s);
void launch_cast_f32(void *out, int dtype, const float *in, int64_t n, hipStream_t s);

}  // namespace gnlp
