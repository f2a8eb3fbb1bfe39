// C ABI of the MI355X-native encoder forward path (declared in include/gluonnlp_amd.h).
// Host side only: argument checking, kernel selection, the per-block launch sequences.
// No allocation, no synchronisation, no global state: everything is enqueued on the caller's
// HIP stream into caller-owned buffers.  (The one exception is the measurement hook at the end of
// the dense dispatcher, gnlp_profile_dense_*: off unless a benchmark arms it.)
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>

#include "common.h"
#include "kernels.h"

using namespace gnlp;

namespace gnlp {
namespace {
// Two instances and an atomic index: a reload fills the idle instance and flips the index, so a
// reader on a launch path never sees a half-written struct.  (Two reloads racing each other are
// serialised by the mutex; gnlp_reload_knobs is still not meant to run beside launches that must
// agree on a knob between two of their reads.)
Knobs g_knobs[2];
std::atomic<int> g_knobs_idx{-1};   // -1: not read yet
std::mutex g_knobs_mu;
int env_int(const char *name, int dflt, int lo, int hi) {
  const char *e = getenv(name);
  if (!e || !*e) return dflt;
  char *end = nullptr;
  const long v = strtol(e, &end, 10);
  if (end == e) return dflt;
  return v < lo ? lo : (v > hi ? hi : (int)v);
}
bool env_on(const char *name) {   // on unless set to something starting with '0'
  const char *e = getenv(name);
  return !(e && e[0] == '0');
}
}  // namespace
void reload_knobs() {
  std::lock_guard<std::mutex> lock(g_knobs_mu);
  Knobs k;
  k.group_m = env_int("GNLP_GROUP_M", 8, 1, 64);
  const int tile = env_int("GNLP_GEMM_TILE", 0, 0, 256);
  k.gemm_tile = (tile == 128 || tile == 256) ? tile : 0;
  k.gemm_w4 = env_on("GNLP_GEMM_W4");
  const int nt = env_int("GNLP_GEMM_NT", 0, 0, 256);
  k.gemm_nt = (nt == 192 || nt == 256) ? nt : 0;
  k.norm_nt = env_on("GNLP_NORM_NT");
  k.pack_ragged = env_on("GNLP_PACK_RAGGED");
  k.attn_bwd_bf16 = env_on("GNLP_ATTN_BWD_BF16");
  k.skinny_max_m = env_int("GNLP_SKINNY_MAX_M", 384, 0, 1 << 20);
  const char *mv = getenv("GNLP_NORM_MAXV");
  k.norm_full = mv && mv[0] == 'f';
  k.ln_fold = env_on("GNLP_LN_FOLD");
  const int cur = g_knobs_idx.load(std::memory_order_acquire);
  const int nxt = cur == 0 ? 1 : 0;
  g_knobs[nxt] = k;
  g_knobs_idx.store(nxt, std::memory_order_release);
}
const Knobs &knobs() {
  int i = g_knobs_idx.load(std::memory_order_acquire);
  if (i < 0) {
    reload_knobs();
    i = g_knobs_idx.load(std::memory_order_acquire);
  }
  return g_knobs[i];
}
}  // namespace gnlp

namespace {

thread_local std::string g_err;
thread_local int g_attn_kernel = 0;   // which kernel the last attention dispatch on this thread took
thread_local const int32_t *g_rows_dev = nullptr;   // see dense_rows_dev() in kernels.h
thread_local int g_stack_packed = 0;   // whether the last stack call on this thread ran packed
thread_local int g_stack_folded = 0;   // ... with its LayerNorms folded into the GEMMs
}  // namespace
namespace gnlp {
const int32_t *dense_rows_dev() { return g_rows_dev; }
}  // namespace gnlp
namespace {
struct RowsDevScope {   // set for the calls of one packed stack, cleared on every way out
  explicit RowsDevScope(const int32_t *p) { g_rows_dev = p; }
  ~RowsDevScope() { g_rows_dev = nullptr; }
};

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNLP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return GNLP_OK;
}

inline bool dtype_ok(int d) { return d == GNLP_F32 || d == GNLP_BF16; }
inline bool vl_ok(int d) { return d == GNLP_I32 || d == GNLP_I64; }
inline size_t esize(int d) { return d == GNLP_BF16 ? 2 : 4; }
inline hipStream_t S(gnlp_stream_t s) { return (hipStream_t)s; }
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

#define REQUIRE(cond, ...) \
  do { if (!(cond)) return fail(GNLP_ERR_ARG, __VA_ARGS__); } while (0)

// Host part of relative_position_bucket (op.py:261-286): find, with the reference's own float32
// formulation (libm logf), the integer distances at which the bucket index increments.
int bucket_value(int n, int max_exact, int nb, float denom) {
  float rr = logf((float)n / (float)max_exact) / denom * (float)(nb - max_exact);
  return max_exact + (int)rr;
}

int build_bucket_table(BucketTable *tb, int bidirectional, int num_buckets, int max_distance) {
  REQUIRE(num_buckets >= 2 && num_buckets <= 256, "num_buckets=%d out of range [2, 256]", num_buckets);
  REQUIRE(!bidirectional || num_buckets % 2 == 0,
          "When bidirectional is True, the number of buckets must be divisible by 2.");
  int nb = bidirectional ? num_buckets / 2 : num_buckets;
  const int max_exact = nb / 2;
  REQUIRE(max_exact >= 1, "num_buckets too small");
  REQUIRE(max_distance > max_exact, "max_distance=%d must exceed max_exact=%d", max_distance, max_exact);
  tb->bidirectional = bidirectional ? 1 : 0;
  tb->half = nb;
  tb->max_exact = max_exact;
  tb->n_thr = nb - 1 - max_exact;
  const float denom = (float)log((double)max_distance / (double)max_exact);
  const int HI = 1 << 30;
  for (int q = 0; q < tb->n_thr; ++q) {
    const int target = max_exact + q + 1;
    // smallest n >= max_exact with bucket_value(n) >= target (monotone in n)
    if (bucket_value(HI, max_exact, nb, denom) < target) { tb->thr[q] = 0x7fffffff; continue; }
    int lo = max_exact, hi = HI;  // invariant: value(lo) < target <= value(hi)
    if (bucket_value(lo, max_exact, nb, denom) >= target) { tb->thr[q] = lo; continue; }
    while (hi - lo > 1) {
      const int mid = lo + (hi - lo) / 2;
      if (bucket_value(mid, max_exact, nb, denom) >= target) hi = mid; else lo = mid;
    }
    tb->thr[q] = hi;
  }
  return GNLP_OK;
}

// ---- measurement hook: HIP events around the dense launches of one (M, N, K) shape -------------
struct DenseProfile {
  std::atomic<int> armed{0};
  int64_t M = 0, N = 0, K = 0;
  int cap = 0;
  std::atomic<int> n{0};
  hipEvent_t *start = nullptr, *stop = nullptr;
};
DenseProfile g_prof;

// ---- dense dispatcher --------------------------------------------------------------------------
int dense_launch(void *out, const void *x, const void *w, const void *bias, const void *other,
                 int combine, int act, int dtype, int64_t M, int64_t N, int64_t K, int64_t ldx,
                 int64_t ldo, int64_t ldother, hipStream_t s, const DenseFold *fold);

int dense_impl(void *out, const void *x, const void *w, const void *bias, const void *other,
               int combine, int act, int dtype, int64_t M, int64_t N, int64_t K, int64_t ldx,
               int64_t ldo, int64_t ldother, hipStream_t s, const DenseFold *fold = nullptr) {
  int slot = -1;
  if (g_prof.armed.load(std::memory_order_acquire) && M == g_prof.M && N == g_prof.N &&
      K == g_prof.K) {
    slot = g_prof.n.fetch_add(1);
    if (slot >= g_prof.cap) slot = -1;
  }
  if (slot >= 0) (void)hipEventRecord(g_prof.start[slot], s);
  const int rc = dense_launch(out, x, w, bias, other, combine, act, dtype, M, N, K, ldx, ldo,
                              ldother, s, fold);
  if (slot >= 0) (void)hipEventRecord(g_prof.stop[slot], s);
  return rc;
}

int dense_launch(void *out, const void *x, const void *w, const void *bias, const void *other,
               int combine, int act, int dtype, int64_t M, int64_t N, int64_t K, int64_t ldx,
               int64_t ldo, int64_t ldother, hipStream_t s, const DenseFold *fold) {
  if (M == 0 || N == 0) return GNLP_OK;
  if (fold && fold->mode) {   // a LayerNorm folded into this GEMM: the persistent kernel only
    REQUIRE(dtype == GNLP_BF16 && mfma_dense_act_fused(act) &&
                mfma_bf16_dense_takes_w4(M, N, K, ldx, ldo, ldother, out, bias, other) &&
                mfma_bf16_dense_w4_fold_supported(*fold, act, combine, other, N),
            "folded LayerNorm: unsupported Dense shape (M=%lld N=%lld K=%lld)", (long long)M,
            (long long)N, (long long)K);
    launch_mfma_bf16_dense_w4(out, x, w, bias, other, combine, act, M, N, K, ldx, ldo, ldother, s,
                              fold);
    return check_launch("gnlp_dense (folded LayerNorm)");
  }
  const bool bias_al = (!bias || ((uintptr_t)bias % 16 == 0)) && mfma_dense_act_fused(act);
  if (dtype == GNLP_BF16 && bias_al &&
      mfma_bf16_dense_supported(M, N, K, ldx, ldo, ldother, x, w, out, other)) {
    launch_mfma_bf16_dense(out, x, w, bias, other, combine, act, M, N, K, ldx, ldo, ldother, s);
  } else if (dtype == GNLP_F32 && bias_al &&
             mfma_f32_dense_supported(M, N, K, ldx, ldo, ldother, x, w, out, other)) {
    launch_mfma_f32_dense(out, x, w, bias, other, combine, act, M, N, K, ldx, ldo, ldother, s);
  } else {
    GenericGemm g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.Bm = w; g.C = out; g.bias = bias; g.other = other;
    g.M = M; g.N = N; g.K = K; g.nb0 = 1; g.nb1 = 1;
    g.sAm = ldx; g.sBn = K; g.sCm = ldo; g.sOm = ldother; g.sOn = 1;
    g.trans_b = 1; g.act = act; g.combine = combine;
    g.in_dtype = g.out_dtype = g.other_dtype = dtype;
    g.alpha = 1.0f;
    launch_generic_gemm(g, s);
  }
  return check_launch("gnlp_dense");
}

// ---- attention dispatcher ----------------------------------------------------------------------
size_t attn_ws_bytes(const gnlp_attn_desc *d, int dtype, bool want_scores, bool want_weights,
                     bool fused) {
  if (fused) return 0;
  {   // attn_gen.hip takes it (the SAME predicate the dispatcher asks): nothing is materialised
    GenAttn ga;
    memset(&ga, 0, sizeof(ga));
    ga.B = d->B; ga.H = d->H; ga.Lq = d->Lq; ga.Lm = d->Lm; ga.dk = d->dk; ga.dv = d->dv;
    if (gen_attn_supported(ga)) return 0;
  }
  if (want_scores && want_weights) return 0;
  return align256((size_t)d->B * d->H * d->Lq * d->Lm * esize(dtype));
}

bool can_fuse(const gnlp_attn_desc *d, int dtype, const void *q, const void *k, const void *v,
              void *ctx, const void *edge, const int64_t *es, FusedAttn *fa) {
  if (dtype != GNLP_BF16) return false;
  memset(fa, 0, sizeof(*fa));
  fa->q = q; fa->k = k; fa->v = v; fa->o = ctx;
  fa->B = d->B; fa->H = d->H; fa->Lq = d->Lq; fa->Lm = d->Lm;
  for (int i = 0; i < 3; ++i) {
    fa->qs[i] = d->q_stride[i]; fa->ks[i] = d->k_stride[i];
    fa->vs[i] = d->v_stride[i]; fa->os[i] = d->o_stride[i];
  }
  fa->bias = edge;
  if (edge) for (int i = 0; i < 4; ++i) fa->bs[i] = es[i];
  return fused_attn_supported(*fa, d->dk, d->dv);
}

int attn_impl(void *ctx, void *scores_out, void *weights_out, const void *q, const void *k,
              const void *v, const gnlp_attn_desc *d, int dtype, const uint8_t *mask,
              const int64_t *ms, const void *vl, int vl_dtype, const void *edge, const int64_t *es,
              float temperature, void *ws, size_t ws_bytes, hipStream_t s,
              const int32_t *cu = nullptr, const float *edge_f32 = nullptr, float *lse = nullptr,
              void *ctx_lo = nullptr) {
  // edge_f32: a float32 copy of `edge` (same strides), used by the fused kernel when it applies
  // lse, ctx_lo: the fused kernel's row statistics and the residual of ctx's bf16 rounding, for
  // the backward pass; no other kernel writes them
  if (d->B == 0 || d->H == 0 || d->Lq == 0) return GNLP_OK;
  const float inv_t = temperature > 0.0f ? 1.0f / temperature : 1.0f;
  FusedAttn fa;
  // (the fused kernel reads mask rows as 16-byte pieces along the key axis: unit key stride)
  if (!scores_out && !weights_out && (!mask || ms[2] == 1) &&
      can_fuse(d, dtype, q, k, v, ctx, edge, es, &fa)) {
    fa.cu = cu;
    fa.mask = mask;
    if (mask) { fa.ms[0] = ms[0]; fa.ms[1] = ms[1]; fa.ms[2] = ms[2]; }
    fa.vl = vl; fa.vl_dtype = vl_dtype;
    fa.causal = (!mask && d->causal) ? 1 : 0;
    fa.vl_pair = (!mask && vl && d->mem_lengths) ? 1 : 0;
    fa.bias = edge;
    if (edge) for (int i = 0; i < 4; ++i) fa.bs[i] = es[i];
    if (edge && edge_f32 && !mask && !fa.causal && (uintptr_t)edge_f32 % 4 == 0) {
      fa.bias = edge_f32;   // the loads become the accumulator init: no bf16 -> f32 converts
      fa.bias_f32 = 1;
    }
    fa.inv_temperature = inv_t;
    fa.lse = lse;
    fa.o_lo = ctx_lo;
    launch_fused_attn_bf16(fa, s);
    g_attn_kernel = GNLP_ATTN_KERNEL_FUSED_BF16;
    return check_launch("fused attention");
  }
  if (cu) return fail(GNLP_ERR_UNSUPPORTED, "packed batches need the fused bf16 attention kernel");
  if (lse)
    return fail(GNLP_ERR_UNSUPPORTED, "only the fused bf16 attention kernel (bf16, dk = dv = 64, "
                "16-byte aligned rows) writes the row statistics");
  {   // ---- general MFMA kernel: fp32, other head dims, scores / weights wanted; no workspace ----
    GenAttn ga;
    memset(&ga, 0, sizeof(ga));
    ga.q = q; ga.k = k; ga.v = v; ga.o = ctx;
    ga.scores_out = scores_out; ga.weights_out = weights_out;
    ga.B = d->B; ga.H = d->H; ga.Lq = d->Lq; ga.Lm = d->Lm; ga.dk = d->dk; ga.dv = d->dv;
    for (int i = 0; i < 3; ++i) {
      ga.qs[i] = d->q_stride[i]; ga.ks[i] = d->k_stride[i];
      ga.vs[i] = d->v_stride[i]; ga.os[i] = d->o_stride[i];
    }
    ga.mask = mask;
    if (mask) { ga.ms[0] = ms[0]; ga.ms[1] = ms[1]; ga.ms[2] = ms[2]; }
    ga.vl = vl; ga.vl_dtype = vl_dtype;
    ga.causal = (!mask && d->causal) ? 1 : 0;
    ga.vl_pair = (!mask && vl && d->mem_lengths) ? 1 : 0;
    ga.bias = edge;
    if (edge) for (int i = 0; i < 4; ++i) ga.bs[i] = es[i];
    ga.inv_temperature = inv_t;
    if (gen_attn_supported(ga)) {   // (Lm = 0 falls through: K = 0 GEMMs write the zeros)
      launch_gen_attn(ga, dtype, s);
      g_attn_kernel = GNLP_ATTN_KERNEL_GENERAL_MFMA;
      return check_launch("general attention");
    }
  }
  if (!mask && d->causal)
    return fail(GNLP_ERR_UNSUPPORTED, "the in-kernel causal predicate needs head dims <= 128 "
                "(dk=%lld dv=%lld): pass the byte mask", (long long)d->dk, (long long)d->dv);
  if (!mask && vl && d->mem_lengths)
    return fail(GNLP_ERR_UNSUPPORTED, "the in-kernel memory-mask predicate (two lengths per "
                "sequence) needs head dims <= 128 (dk=%lld dv=%lld): pass the byte mask",
                (long long)d->dk, (long long)d->dv);
  g_attn_kernel = GNLP_ATTN_KERNEL_UNFUSED;
  // ---- unfused path: scores -> masked softmax -> context (attention_cell.py:452-463) ----
  const size_t need = attn_ws_bytes(d, dtype, scores_out != nullptr, weights_out != nullptr, false);
  REQUIRE(ws_bytes >= need && (need == 0 || ws != nullptr),
          "attention workspace too small: need %zu bytes, got %zu", need, ws_bytes);
  // whichever of the two the caller does not want lives in the workspace; with neither wanted the
  // softmax runs in place over the scores
  void *scores = scores_out ? scores_out : ws;
  void *weights = weights_out ? weights_out : ws;
  const int64_t LL = d->Lq * d->Lm;
  GenericGemm g;
  memset(&g, 0, sizeof(g));
  g.A = q; g.Bm = k; g.C = scores; g.other = edge;
  g.M = d->Lq; g.N = d->Lm; g.K = d->dk; g.nb0 = d->B; g.nb1 = d->H;
  g.sAb0 = d->q_stride[0]; g.sAb1 = d->q_stride[1]; g.sAm = d->q_stride[2];
  g.sBb0 = d->k_stride[0]; g.sBb1 = d->k_stride[1]; g.sBn = d->k_stride[2];
  g.sCb0 = d->H * LL; g.sCb1 = LL; g.sCm = d->Lm;
  if (edge) { g.sOb0 = es[0]; g.sOb1 = es[1]; g.sOm = es[2]; g.sOn = es[3]; }
  g.trans_b = 1; g.act = GNLP_ACT_IDENTITY; g.combine = GNLP_COMBINE_ADD;
  g.in_dtype = g.out_dtype = g.other_dtype = dtype;
  g.alpha = 1.0f;
  launch_generic_gemm(g, s);
  const int64_t shape[4] = {d->B, d->H, d->Lq, d->Lm};
  int64_t mstr[4] = {0, 0, 0, 0};
  if (mask) { mstr[0] = ms[0]; mstr[1] = 0; mstr[2] = ms[1]; mstr[3] = ms[2]; }
  launch_masked_softmax(weights, scores, dtype, mask, mstr, shape, mask ? nullptr : vl, vl_dtype,
                        inv_t, 0, s);
  memset(&g, 0, sizeof(g));
  g.A = weights; g.Bm = v; g.C = ctx;
  g.M = d->Lq; g.N = d->dv; g.K = d->Lm; g.nb0 = d->B; g.nb1 = d->H;
  g.sAb0 = d->H * LL; g.sAb1 = LL; g.sAm = d->Lm;
  g.sBb0 = d->v_stride[0]; g.sBb1 = d->v_stride[1]; g.sBn = d->v_stride[2];
  g.sCb0 = d->o_stride[0]; g.sCb1 = d->o_stride[1]; g.sCm = d->o_stride[2];
  g.trans_b = 0; g.act = GNLP_ACT_IDENTITY; g.combine = GNLP_COMBINE_ADD;
  g.in_dtype = g.out_dtype = g.other_dtype = dtype;
  g.alpha = 1.0f;
  launch_generic_gemm(g, s);
  return check_launch("unfused attention");
}

// Whether the stack drivers' own attention call (packed QKV in the workspace, no scores / weights
// wanted, bias -- if any -- with unit key stride) takes the fused kernel: the SAME conditions
// fused_attn_supported() checks on the real descriptor, given that the drivers require a
// 256-byte aligned workspace and build every stride as a multiple of 8 elements.
bool stack_attn_fuses(int dtype, int64_t B, int64_t H, int64_t dk, int64_t qkv_row_elems,
                      int64_t ctx_row_elems) {
  return dtype == GNLP_BF16 && dk == 64 && B <= 65535 && H <= 65535 && qkv_row_elems % 8 == 0 &&
         ctx_row_elems % 8 == 0;
}

// ---- FFN ---------------------------------------------------------------------------------------
struct FfnWs { size_t normed, hidden, gate, total; };
FfnWs ffn_ws(const gnlp_ffn_cfg *c, int64_t rows) {
  FfnWs w;
  const size_t e = esize(c->dtype);
  size_t off = 0;
  w.normed = off; off += c->pre_norm ? align256((size_t)rows * c->units * e) : 0;
  w.hidden = off; off += align256((size_t)rows * c->hidden * e);
  w.gate = off; off += c->gated ? align256((size_t)rows * c->hidden * e) : 0;
  // post-norm needs the pre-LN sum; it is written over `normed`-less layouts into its own slot
  w.total = off + (c->pre_norm ? 0 : align256((size_t)rows * c->units * e));
  return w;
}

int ffn_impl(void *out, const void *x, const gnlp_ffn_params *p, const gnlp_ffn_cfg *c,
             int64_t rows, const void *vl, int vl_dtype, int64_t B, int64_t L, int time_major,
             void *ws, size_t ws_bytes, hipStream_t s) {
  const FfnWs w = ffn_ws(c, rows);
  REQUIRE(ws && ws_bytes >= w.total, "ffn workspace too small: need %zu bytes, got %zu", w.total,
          ws_bytes);
  char *base = (char *)ws;
  const int64_t U = c->units, F = c->hidden;
  const void *in = x;
  int rc;
  if (c->pre_norm) {  // layers.py:634-635
    launch_norm(base + w.normed, x, nullptr, p->ln_g, p->ln_b, c->dtype, c->norm_kind, c->ln_eps,
                rows, U, nullptr, 0, 0, 0, 0, s);
    in = base + w.normed;
  }
  void *hidden = base + w.hidden;
  if (c->gated) {     // layers.py:636-638: act(gated_ffn_1(x)) * ffn_1(x)
    void *gate = base + w.gate;
    rc = dense_impl(gate, in, p->gate_w, p->gate_b, nullptr, 0, c->act, c->dtype, rows, F, U, U, F,
                    0, s);
    if (rc) return rc;
    rc = dense_impl(hidden, in, p->ffn1_w, p->ffn1_b, gate, GNLP_COMBINE_MUL, GNLP_ACT_IDENTITY,
                    c->dtype, rows, F, U, U, F, F, s);
    if (rc) return rc;
  } else {            // layers.py:640
    rc = dense_impl(hidden, in, p->ffn1_w, p->ffn1_b, nullptr, 0, c->act, c->dtype, rows, F, U, U,
                    F, 0, s);
    if (rc) return rc;
  }
  if (c->pre_norm) {  // out = ffn_2(h) + residual(x)   layers.py:642-644
    rc = dense_impl(out, hidden, p->ffn2_w, p->ffn2_b, x, GNLP_COMBINE_ADD, GNLP_ACT_IDENTITY,
                    c->dtype, rows, U, F, F, U, U, s);
    if (rc) return rc;
    if (vl) launch_sequence_mask(out, out, vl, vl_dtype, c->dtype, B, L, U, time_major, s);
  } else {            // out = norm(ffn_2(h) + x)       layers.py:642-646
    void *sum = base + w.total - align256((size_t)rows * U * esize(c->dtype));
    rc = dense_impl(sum, hidden, p->ffn2_w, p->ffn2_b, x, GNLP_COMBINE_ADD, GNLP_ACT_IDENTITY,
                    c->dtype, rows, U, F, F, U, U, s);
    if (rc) return rc;
    launch_norm(out, sum, nullptr, p->ln_g, p->ln_b, c->dtype, c->norm_kind, c->ln_eps, rows, U,
                vl, vl_dtype, B, L, time_major, s);
  }
  return check_launch("positionwise_ffn");
}

gnlp_ffn_cfg ffn_cfg_of(const gnlp_encoder_cfg *c) {
  gnlp_ffn_cfg f;
  f.units = c->units; f.hidden = c->hidden; f.act = c->act; f.gated = 0;
  f.norm_kind = GNLP_NORM_LAYER; f.pre_norm = c->pre_norm; f.dtype = c->dtype; f.ln_eps = c->ln_eps;
  return f;
}

// ---- encoder layer -----------------------------------------------------------------------------
struct EncWs { size_t normed, qkv, ctx, sum, x1, attn, ffn, total; };
EncWs enc_ws(const gnlp_encoder_cfg *c, int64_t B, int64_t L, bool want_weights) {
  EncWs w;
  const size_t e = esize(c->dtype);
  const int64_t M = B * L, U = c->units;
  size_t off = 0;
  w.normed = off; off += c->pre_norm ? align256((size_t)M * U * e) : 0;
  w.qkv = off; off += align256((size_t)M * 3 * U * e);
  w.ctx = off; off += align256((size_t)M * U * e);
  w.sum = off; off += align256((size_t)M * U * e);
  w.x1 = off; off += c->pre_norm ? 0 : align256((size_t)M * U * e);
  w.attn = off;
  gnlp_attn_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.H = c->heads; d.Lq = L; d.Lm = L; d.dk = d.dv = U / (c->heads ? c->heads : 1);
  const bool fused = !want_weights && stack_attn_fuses(c->dtype, B, c->heads, d.dk, 3 * U, U);
  off += attn_ws_bytes(&d, c->dtype, false, want_weights, fused);
  w.ffn = off;
  gnlp_ffn_cfg f = ffn_cfg_of(c);
  off += ffn_ws(&f, M).total;
  w.total = off;
  return w;
}

int encoder_layer_impl(void *out, const void *x, const gnlp_encoder_layer_params *p,
                       const gnlp_encoder_cfg *c, int64_t B, int64_t L, const uint8_t *mask,
                       const void *vl, int vl_dtype, void *attn_w_out, int final_seq_mask,
                       void *ws, size_t ws_bytes, hipStream_t s, const int32_t *cu = nullptr) {
  // cu != null: x / out hold the batch PACKED to its valid tokens (row cu[b] + t); the dense
  // kernels read the row count from device memory (RowsDevScope of the caller), M stays the bound
  const int64_t U = c->units, H = c->heads, M = B * L, d = U / H;
  const EncWs w = enc_ws(c, B, L, attn_w_out != nullptr);
  REQUIRE(ws && ws_bytes >= w.total, "encoder workspace too small: need %zu bytes, got %zu",
          w.total, ws_bytes);
  REQUIRE((uintptr_t)ws % 256 == 0, "workspace must be 256-byte aligned");
  char *base = (char *)ws;
  const size_t e = esize(c->dtype);
  int rc;
  const void *in = x;
  if (c->pre_norm) {  // models/transformer.py:259-260 (data is REBOUND to the normalised tensor)
    launch_norm(base + w.normed, x, nullptr, p->ln_g, p->ln_b, c->dtype, GNLP_NORM_LAYER,
                c->ln_eps, M, U, nullptr, 0, 0, 0, 0, s);
    in = base + w.normed;
  }
  char *qkv = base + w.qkv;
  rc = dense_impl(qkv, in, p->qkv_w, p->qkv_b, nullptr, 0, GNLP_ACT_IDENTITY, c->dtype, M, 3 * U, U,
                  U, 3 * U, 0, s);  // :261
  if (rc) return rc;
  gnlp_attn_desc ad;
  memset(&ad, 0, sizeof(ad));
  ad.B = B; ad.H = H; ad.Lq = L; ad.Lm = L; ad.dk = d; ad.dv = d;
  // rows are tokens: NT -> row = b*L + t ; TN -> row = t*B + b   (:262-264 reshape, :207 layout)
  const int64_t sb = cu ? 0 : (c->time_major ? 1 : L), st = cu ? 1 : (c->time_major ? B : 1);
  for (int i = 0; i < 1; ++i) {
    ad.q_stride[0] = ad.k_stride[0] = ad.v_stride[0] = sb * 3 * U;
    ad.q_stride[1] = ad.k_stride[1] = ad.v_stride[1] = d;
    ad.q_stride[2] = ad.k_stride[2] = ad.v_stride[2] = st * 3 * U;
    ad.o_stride[0] = sb * U; ad.o_stride[1] = d; ad.o_stride[2] = st * U;
  }
  const int64_t mstr[3] = {L * L, L, 1};
  rc = attn_impl(base + w.ctx, nullptr, attn_w_out, qkv, qkv + (size_t)U * e,
                 qkv + (size_t)2 * U * e, &ad, c->dtype, mask, mstr, vl, vl_dtype, nullptr, nullptr,
                 sqrtf((float)d), base + w.attn, w.ffn - w.attn, s, cu);  // :265, scaled=True :213
  if (rc) return rc;
  // :266-268  out = attention_proj(ctx) + data   (data = normalised input when pre_norm)
  rc = dense_impl(base + w.sum, base + w.ctx, p->proj_w, p->proj_b, in, GNLP_COMBINE_ADD,
                  GNLP_ACT_IDENTITY, c->dtype, M, U, U, U, U, U, s);
  if (rc) return rc;
  const void *ffn_in = base + w.sum;
  if (!c->pre_norm) {  // :269-270
    launch_norm(base + w.x1, base + w.sum, nullptr, p->ln_g, p->ln_b, c->dtype, GNLP_NORM_LAYER,
                c->ln_eps, M, U, nullptr, 0, 0, 0, 0, s);
    ffn_in = base + w.x1;
  }
  gnlp_ffn_cfg f = ffn_cfg_of(c);
  return ffn_impl(out, ffn_in, &p->ffn, &f, M, final_seq_mask ? vl : nullptr, vl_dtype, B, L,
                  c->time_major, base + w.ffn, w.total - w.ffn, s);  // :271
}

// ---- static scratch of gnlp_encoder_layer_forward_noalloc ----------------------------------------
// A file-scope device array: every GPU the library runs on holds its own copy until the process
// ends.  Its NAME is a host-side stand-in, not a device address: the address of a device's copy is
// looked up once per device with hipGetSymbolAddress (that device current) and only that pointer
// is ever handed to kernels.
constexpr size_t kStaticScratchBytes = GNLP_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
std::atomic<void *> g_scratch_ptr[64];   // per device ordinal

int static_scratch(void **ptr, hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(GNLP_ERR_LAUNCH, "hipGetDevice failed");
  if (stream) {   // the kernels run on the stream's device: it must be the one whose copy we hand out
    hipDevice_t sdev = -1;
    if (hipStreamGetDevice(stream, &sdev) == hipSuccess && (int)sdev != dev)
      return fail(GNLP_ERR_ARG, "gnlp_encoder_layer_forward_noalloc: the stream belongs to device "
                  "%d but device %d is current; make the stream's device current", (int)sdev, dev);
  }
  if (dev < 0 || dev >= 64) return fail(GNLP_ERR_UNSUPPORTED, "device ordinal %d out of range", dev);
  void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    const hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p)
      return fail(GNLP_ERR_LAUNCH, "hipGetSymbolAddress(scratch) on device %d: %s", dev,
                  hipGetErrorString(e));
    g_scratch_ptr[dev].store(p, std::memory_order_release);   // (a racing thread stores the same)
  }
  *ptr = p;
  return GNLP_OK;
}

// Whether a ragged batch may run packed to its valid tokens: bf16, every GEMM of the layer takes the
// persistent kernel (the one that reads the row count from device memory) at the padded size, the
// attention call takes the fused kernel, rows move as 16-byte vectors, offsets fit int32.
bool stack_can_pack(const gnlp_encoder_cfg *c, int64_t B, int64_t L, const void *ws,
                    const gnlp_encoder_layer_params *layers, int n_layers) {
  const int64_t U = c->units, F = c->hidden, M = B * L;
  if (c->dtype != GNLP_BF16 || M % 256 != 0 || M >= ((int64_t)1 << 31) || U % 8 != 0) return false;
  if (c->pre_norm) return false;   // (no stack with a final sequence mask is pre-norm)
  if (!stack_attn_fuses(c->dtype, B, c->heads, U / c->heads, 3 * U, U)) return false;
  // activations live in the 256-byte aligned workspace (`ws` stands for all of them); the bias
  // vectors are the only operands whose alignment differs from layer to layer
  for (int i = 0; i < n_layers; ++i) {
    const gnlp_encoder_layer_params *p = &layers[i];
    if (p->ffn.gate_w) return false;
    if (!mfma_bf16_dense_takes_w4(M, 3 * U, U, U, 3 * U, 0, ws, p->qkv_b, nullptr) ||
        !mfma_bf16_dense_takes_w4(M, U, U, U, U, U, ws, p->proj_b, ws) ||
        !mfma_bf16_dense_takes_w4(M, F, U, U, F, 0, ws, p->ffn.ffn1_b, nullptr) ||
        !mfma_bf16_dense_takes_w4(M, U, F, F, U, U, ws, p->ffn.ffn2_b, ws))
      return false;
  }
  return true;
}

// Whether a post-norm bf16 stack may run with its LayerNorms folded into the GEMMs around them:
// every GEMM of a layer on the persistent kernel, rows of at most 2048 units, beta present.
bool stack_can_fold(const gnlp_encoder_cfg *c, int64_t B, int64_t L, const void *ws,
                    const gnlp_encoder_layer_params *layers,
                    const gnlp_encoder_layer_folded *folded, int n_layers) {
  const int64_t U = c->units, F = c->hidden, M = B * L;
  if (!folded || c->dtype != GNLP_BF16 || c->pre_norm || M % 256 != 0 || U > 2048) return false;
  if (!mfma_dense_act_fused(c->act) || !row_stats_supported(U, ws)) return false;
  for (int i = 0; i < n_layers; ++i) {
    const gnlp_encoder_layer_params *p = &layers[i];
    const gnlp_encoder_layer_folded *f = &folded[i];
    if (p->ffn.gate_w || !p->ln_b || !p->ffn.ln_b) return false;
    if (!f->ffn1_wf || !f->ffn1_c || !f->ffn1_bf) return false;
    if (i > 0 && (!f->qkv_wf || !f->qkv_c || !f->qkv_bf)) return false;
    if (((uintptr_t)f->ffn1_c | (uintptr_t)f->ffn1_bf | (uintptr_t)f->ffn1_wf |
         (uintptr_t)f->qkv_c | (uintptr_t)f->qkv_bf | (uintptr_t)f->qkv_wf |
         (uintptr_t)p->ln_g | (uintptr_t)p->ln_b | (uintptr_t)p->ffn.ln_g |
         (uintptr_t)p->ffn.ln_b) % 16)
      return false;
    if (!mfma_bf16_dense_takes_w4(M, 3 * U, U, U, 3 * U, 0, ws, p->qkv_b, nullptr) ||
        !mfma_bf16_dense_takes_w4(M, U, U, U, U, U, ws, p->proj_b, ws) ||
        !mfma_bf16_dense_takes_w4(M, F, U, U, F, 0, ws, p->ffn.ffn1_b, nullptr) ||
        !mfma_bf16_dense_takes_w4(M, U, F, F, U, U, ws, p->ffn.ffn2_b, ws))
      return false;
  }
  // what the folded instantiations of the persistent kernel take (e.g. no FOLD 1 with the tanh
  // GELU, N <= 4096 with an activation): asked with stand-in pointers of the right alignment
  const DenseFold f1{1, (const float *)ws, (const float *)ws, (const float *)ws, nullptr, nullptr,
                     nullptr};
  const DenseFold f2{2, (const float *)ws, nullptr, nullptr, ws, ws, (float *)ws};
  return mfma_bf16_dense_w4_fold_supported(f1, GNLP_ACT_IDENTITY, 0, nullptr, 3 * U) &&
         mfma_bf16_dense_w4_fold_supported(f1, c->act, 0, nullptr, F) &&
         mfma_bf16_dense_w4_fold_supported(f2, GNLP_ACT_IDENTITY, GNLP_COMBINE_ADD, ws, U);
}

// One post-norm layer of a folded stack (models/transformer.py:259-271, layers.py:640-646, the same
// arithmetic regrouped).  raw_in == null: x is a real tensor (the stack's input); otherwise the
// layer's input is LayerNorm(raw_in) with the statistics rs_in and the previous layer's
// ffn.layer_norm (prev), never materialised.  Leaves the raw sum of ffn_2 in raw_out and, unless
// `last`, its row statistics in rs_out; the last layer's LayerNorm (+ sequence mask) writes `out`.
int encoder_layer_folded(void *out, const void *x, const void *raw_in, const float *rs_in,
                         const gnlp_encoder_layer_params *prev, const gnlp_encoder_layer_params *p,
                         const gnlp_encoder_layer_folded *f, const gnlp_encoder_cfg *c, int64_t B,
                         int64_t L, const void *vl, int vl_dtype, int final_seq_mask, bool last,
                         void *raw_out, float *rs_mid, float *rs_out, float *part, void *ws,
                         size_t ws_bytes, hipStream_t s, const int32_t *cu) {
  const int64_t U = c->units, H = c->heads, M = B * L, d = U / H, F = c->hidden;
  const EncWs w = enc_ws(c, B, L, false);
  REQUIRE(ws && ws_bytes >= w.total, "encoder workspace too small: need %zu bytes, got %zu",
          w.total, ws_bytes);
  char *base = (char *)ws;
  const size_t e = esize(c->dtype);
  int rc;
  char *qkv = base + w.qkv;
  DenseFold fd;
  if (!raw_in) {
    rc = dense_impl(qkv, x, p->qkv_w, p->qkv_b, nullptr, 0, GNLP_ACT_IDENTITY, c->dtype, M, 3 * U,
                    U, U, 3 * U, 0, s);
  } else {   // attn_qkv(LN(raw_in))
    fd = DenseFold{1, rs_in, f->qkv_c, f->qkv_bf, nullptr, nullptr, nullptr};
    rc = dense_impl(qkv, raw_in, f->qkv_wf, nullptr, nullptr, 0, GNLP_ACT_IDENTITY, c->dtype, M,
                    3 * U, U, U, 3 * U, 0, s, &fd);
  }
  if (rc) return rc;
  gnlp_attn_desc ad;
  memset(&ad, 0, sizeof(ad));
  ad.B = B; ad.H = H; ad.Lq = L; ad.Lm = L; ad.dk = d; ad.dv = d;
  const int64_t sb = cu ? 0 : (c->time_major ? 1 : L), st = cu ? 1 : (c->time_major ? B : 1);
  ad.q_stride[0] = ad.k_stride[0] = ad.v_stride[0] = sb * 3 * U;
  ad.q_stride[1] = ad.k_stride[1] = ad.v_stride[1] = d;
  ad.q_stride[2] = ad.k_stride[2] = ad.v_stride[2] = st * 3 * U;
  ad.o_stride[0] = sb * U; ad.o_stride[1] = d; ad.o_stride[2] = st * U;
  const int64_t mstr[3] = {L * L, L, 1};
  rc = attn_impl(base + w.ctx, nullptr, nullptr, qkv, qkv + (size_t)U * e, qkv + (size_t)2 * U * e,
                 &ad, c->dtype, nullptr, mstr, vl, vl_dtype, nullptr, nullptr, sqrtf((float)d),
                 base + w.attn, w.ffn - w.attn, s, cu);
  if (rc) return rc;
  // sum = attention_proj(ctx) + data, data = x or LN(raw_in) rebuilt in the drain
  if (!raw_in) {
    rc = dense_impl(base + w.sum, base + w.ctx, p->proj_w, p->proj_b, x, GNLP_COMBINE_ADD,
                    GNLP_ACT_IDENTITY, c->dtype, M, U, U, U, U, U, s);
    if (rc) return rc;
    launch_row_stats(rs_mid, base + w.sum, c->ln_eps, M, U, s);   // (a plain residual: one read)
  } else {   // the drain leaves partial row sums of its output: no pass over the sum
    fd = DenseFold{2, rs_in, nullptr, nullptr, prev->ffn.ln_g, prev->ffn.ln_b, part};
    rc = dense_impl(base + w.sum, base + w.ctx, p->proj_w, p->proj_b, raw_in, GNLP_COMBINE_ADD,
                    GNLP_ACT_IDENTITY, c->dtype, M, U, U, U, U, U, s, &fd);
    if (rc) return rc;
    launch_stats_finish(rs_mid, part, fold_stat_slots(M, U), c->ln_eps, M, U, s);
  }
  // hidden = act(ffn_1(LN(sum)));  raw_out = ffn_2(hidden) + LN(sum)
  void *hidden = base + w.ffn;   // (ffn_ws of a post-norm, ungated FFN: hidden first)
  fd = DenseFold{1, rs_mid, f->ffn1_c, f->ffn1_bf, nullptr, nullptr, nullptr};
  rc = dense_impl(hidden, base + w.sum, f->ffn1_wf, nullptr, nullptr, 0, c->act, c->dtype, M, F, U,
                  U, F, 0, s, &fd);
  if (rc) return rc;
  fd = DenseFold{2, rs_mid, nullptr, nullptr, p->ln_g, p->ln_b, part};
  rc = dense_impl(raw_out, hidden, p->ffn.ffn2_w, p->ffn.ffn2_b, base + w.sum, GNLP_COMBINE_ADD,
                  GNLP_ACT_IDENTITY, c->dtype, M, U, F, F, U, U, s, &fd);
  if (rc) return rc;
  if (last) {
    launch_norm(out, raw_out, nullptr, p->ffn.ln_g, p->ffn.ln_b, c->dtype, GNLP_NORM_LAYER,
                c->ln_eps, M, U, final_seq_mask ? vl : nullptr, vl_dtype, B, L, c->time_major, s);
  } else {
    launch_stats_finish(rs_out, part, fold_stat_slots(M, U), c->ln_eps, M, U, s);
  }
  return check_launch("folded encoder layer");
}

// partial row sums a folded GEMM's drain leaves: 4 (sum, sum of squares) pairs per tile column
size_t fold_part_bytes(int64_t M, int64_t U) {
  return align256((size_t)M * 4 * ((U + 191) / 192) * 2 * sizeof(float));
}

int check_encoder_cfg(const gnlp_encoder_cfg *c, int64_t B, int64_t L) {
  REQUIRE(c != nullptr, "cfg is NULL");
  REQUIRE(dtype_ok(c->dtype), "unsupported dtype %d", c->dtype);
  REQUIRE(c->units > 0 && c->hidden > 0 && c->heads > 0, "bad units/hidden/heads");
  REQUIRE(c->units % c->heads == 0, "units must be divisive by the number of heads");
  REQUIRE(B >= 0 && L >= 0, "negative batch / length");
  return GNLP_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int gnlp_version(void) { return 200; }
int gnlp_attn_last_kernel(void) { return g_attn_kernel; }
int gnlp_stack_last_packed(void) { return g_stack_packed; }
void gnlp_reload_knobs(void) { reload_knobs(); }
const char *gnlp_last_error(void) { return g_err.c_str(); }
int gnlp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int gnlp_gen_self_attn_mask(uint8_t *mask, const void *valid_length, int vl_dtype, int64_t B,
                            int64_t L, int attn_type, gnlp_stream_t stream) {
  REQUIRE(B >= 0 && L >= 0, "negative shape");
  REQUIRE(attn_type == GNLP_ATTN_FULL || attn_type == GNLP_ATTN_CAUSAL, "unknown attn_type %d",
          attn_type);
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype %d", vl_dtype);
  REQUIRE(mask || B * L * L == 0, "mask is NULL");
  launch_self_mask(mask, valid_length, vl_dtype, B, L, attn_type, S(stream));
  return check_launch("gnlp_gen_self_attn_mask");
}

int gnlp_gen_mem_attn_mask(uint8_t *mask, const void *mem_valid_length,
                           const void *data_valid_length, int vl_dtype, int64_t B, int64_t Lq,
                           int64_t Lm, gnlp_stream_t stream) {
  REQUIRE(B >= 0 && Lq >= 0 && Lm >= 0, "negative shape");
  REQUIRE(mem_valid_length != nullptr, "mem_valid_length is NULL");
  REQUIRE(vl_ok(vl_dtype), "bad valid_length dtype %d", vl_dtype);
  REQUIRE(mask || B * Lq * Lm == 0, "mask is NULL");
  launch_mem_mask(mask, mem_valid_length, data_valid_length, vl_dtype, B, Lq, Lm, S(stream));
  return check_launch("gnlp_gen_mem_attn_mask");
}

int gnlp_sequence_mask(void *out, const void *x, const void *valid_length, int vl_dtype, int dtype,
                       int64_t B, int64_t L, int64_t C, int time_major, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(valid_length && vl_ok(vl_dtype), "valid_length missing or of bad dtype");
  REQUIRE(B >= 0 && L >= 0 && C >= 0, "negative shape");
  REQUIRE((out && x) || B * L * C == 0, "NULL tensor");
  launch_sequence_mask(out, x, valid_length, vl_dtype, dtype, B, L, C, time_major, S(stream));
  return check_launch("gnlp_sequence_mask");
}

int gnlp_masked_softmax(void *out, const void *scores, int dtype, const uint8_t *mask,
                        const int64_t mask_stride[4], const int64_t shape[4], float temperature,
                        int log_mode, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(shape != nullptr, "shape is NULL");
  for (int i = 0; i < 4; ++i) REQUIRE(shape[i] >= 0, "negative shape");
  REQUIRE(!mask || mask_stride, "mask given without strides");
  const int64_t n = shape[0] * shape[1] * shape[2] * shape[3];
  REQUIRE((out && scores) || n == 0, "NULL tensor");
  REQUIRE(shape[0] * shape[1] * shape[2] < ((int64_t)1 << 33), "too many rows");
  const float inv_t = temperature > 0.0f ? 1.0f / temperature : 1.0f;
  int64_t zero[4] = {0, 0, 0, 0};
  launch_masked_softmax(out, scores, dtype, mask, mask ? mask_stride : zero, shape, nullptr, 0,
                        inv_t, log_mode, S(stream));
  return check_launch("gnlp_masked_softmax");
}

int gnlp_l2_normalize(void *out, const void *x, int dtype, int64_t rows, int64_t d, float eps,
                      gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(rows >= 0 && d >= 0, "negative shape");
  REQUIRE((out && x) || rows * d == 0, "NULL tensor");
  launch_l2_normalize(out, x, dtype, rows, d, eps, S(stream));
  return check_launch("gnlp_l2_normalize");
}

int gnlp_activation(void *out, const void *x, int dtype, int64_t n, int act, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(act >= GNLP_ACT_IDENTITY && act <= GNLP_ACT_SOFTSIGN, "act=%d is not supported", act);
  REQUIRE(n >= 0 && ((out && x) || n == 0), "bad tensor");
  launch_activation(out, x, dtype, n, act, S(stream));
  return check_launch("gnlp_activation");
}

int gnlp_dense_epilogue(void *y, const void *z, const void *other, int dtype, int64_t n, int act,
                        int combine, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(act >= GNLP_ACT_IDENTITY && act <= GNLP_ACT_SOFTSIGN, "act=%d is not supported", act);
  REQUIRE(combine == GNLP_COMBINE_ADD || combine == GNLP_COMBINE_MUL, "bad combine %d", combine);
  REQUIRE(n >= 0 && ((y && z) || n == 0), "bad tensor");
  launch_dense_epilogue(y, z, other, dtype, n, act, combine == GNLP_COMBINE_MUL, S(stream));
  return check_launch("gnlp_dense_epilogue");
}

int gnlp_dense_epilogue_backward(void *dz, void *d_other, const void *g, const void *z,
                                 const void *other, int dtype, int64_t n, int act, int combine,
                                 gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(act >= GNLP_ACT_IDENTITY && act <= GNLP_ACT_SOFTSIGN, "act=%d is not supported", act);
  REQUIRE(combine == GNLP_COMBINE_ADD || combine == GNLP_COMBINE_MUL, "bad combine %d", combine);
  REQUIRE(n >= 0 && ((dz && g && z) || n == 0), "bad tensor");
  const bool mul = combine == GNLP_COMBINE_MUL;
  REQUIRE(!mul || other, "combine mul needs `other`");
  REQUIRE(!d_other || mul, "d_other is only written for combine mul (for add it equals g)");
  launch_dense_epilogue_bwd(dz, d_other, g, z, mul ? other : nullptr, dtype, n, act, mul, S(stream));
  return check_launch("gnlp_dense_epilogue_backward");
}

int gnlp_norm(void *out, const void *x, const void *residual, const void *gamma, const void *beta,
              int dtype, int kind, float eps, int64_t rows, int64_t C, const void *valid_length,
              int vl_dtype, int64_t B, int64_t L, int time_major, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(kind == GNLP_NORM_LAYER || kind == GNLP_NORM_RMS || kind == GNLP_NORM_NONE,
          "normalization kind %d is not supported", kind);
  REQUIRE(rows >= 0 && C >= 0, "negative shape");
  REQUIRE((out && x && gamma) || rows * C == 0, "NULL tensor");
  if (valid_length) {
    REQUIRE(vl_ok(vl_dtype), "bad valid_length dtype %d", vl_dtype);
    REQUIRE(B * L == rows, "valid_length given but B*L (%lld) != rows (%lld)", (long long)(B * L),
            (long long)rows);
  }
  launch_norm(out, x, residual, gamma, beta, dtype, kind, eps, rows, C, valid_length, vl_dtype, B,
              L, time_major, S(stream));
  return check_launch("gnlp_norm");
}

int gnlp_embedding_forward(void *out, const void *ids, const void *token_types, int id_dtype,
                           const void *word_w, const void *type_w, const void *pos_w,
                           const void *ln_gamma, const void *ln_beta, int dtype, float eps,
                           int64_t B, int64_t L, int64_t units, int64_t vocab_size,
                           int64_t num_token_types, int64_t max_length, int time_major,
                           gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(vl_ok(id_dtype), "token ids must be int32 or int64 (got %d)", id_dtype);
  REQUIRE(B >= 0 && L >= 0 && units >= 0, "negative shape");
  if (B * L * units == 0) return GNLP_OK;
  REQUIRE(out && ids && word_w, "NULL tensor");
  REQUIRE(vocab_size > 0, "vocab_size must be positive");
  REQUIRE(!type_w || num_token_types > 0, "num_token_types must be positive");
  REQUIRE(!pos_w || max_length > 0, "max_length must be positive");
  REQUIRE(!token_types || type_w, "token_types given without a token-type table");
  REQUIRE(!ln_beta || ln_gamma, "beta given without gamma");
  const bool ok = launch_embedding(out, ids, token_types, id_dtype, word_w, type_w, pos_w, ln_gamma,
                                   ln_beta, dtype, eps, B, L, units, time_major, vocab_size,
                                   num_token_types, max_length, S(stream));
  REQUIRE(ok, "embedding: units=%lld must be a multiple of %d, <= 2048 (%s), tables 16-byte aligned",
          (long long)units, dtype == GNLP_BF16 ? 8 : 4, dtype == GNLP_BF16 ? "bf16" : "f32");
  return check_launch("gnlp_embedding_forward");
}

int gnlp_dense(void *out, const void *x, const void *w, const void *bias, const void *other,
               int combine, int act, int dtype, int64_t M, int64_t N, int64_t K, int64_t ldx,
               int64_t ldo, int64_t ldother, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative shape");
  REQUIRE(act >= GNLP_ACT_IDENTITY && act <= GNLP_ACT_SOFTSIGN, "act=%d is not supported", act);
  REQUIRE(combine == GNLP_COMBINE_ADD || combine == GNLP_COMBINE_MUL, "bad combine %d", combine);
  REQUIRE(ldx >= K && ldo >= N && (!other || ldother >= N), "leading dimension too small");
  REQUIRE((out && x && w) || M * N == 0, "NULL tensor");
  return dense_impl(out, x, w, bias, other, combine, act, dtype, M, N, K, ldx, ldo, ldother,
                    S(stream));
}

size_t gnlp_attn_workspace_bytes(const gnlp_attn_desc *d, int dtype, int want_scores,
                                 int want_weights) {
  if (!d) return 0;
  // conservative: the size the unfused path needs (the fused kernel needs none)
  return attn_ws_bytes(d, dtype, want_scores != 0, want_weights != 0, false);
}

int gnlp_multi_head_dot_attn(void *ctx, void *scores_out, void *weights_out, const void *q,
                             const void *k, const void *v, const gnlp_attn_desc *d, int dtype,
                             const uint8_t *mask, const int64_t mask_stride[3],
                             const void *valid_length, int vl_dtype, const void *edge_scores,
                             const int64_t edge_stride[4], float temperature, void *workspace,
                             size_t workspace_bytes, gnlp_stream_t stream) {
  REQUIRE(d != nullptr, "descriptor is NULL");
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(d->B >= 0 && d->H >= 0 && d->Lq >= 0 && d->Lm >= 0 && d->dk >= 0 && d->dv >= 0,
          "negative shape");
  REQUIRE(!mask || mask_stride, "mask given without strides");
  REQUIRE(!edge_scores || edge_stride, "edge_scores given without strides");
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype %d", vl_dtype);
  REQUIRE(d->B * d->H <= 0x7fffffffLL, "batch*heads too large");
  REQUIRE(!(mask && d->causal), "d->causal selects the in-kernel predicate: mask must be NULL");
  REQUIRE(!d->mem_lengths || (!mask && !d->causal && valid_length),
          "d->mem_lengths selects the memory-mask predicate: mask NULL, not causal, valid_length "
          "with 2 B entries");
  if (d->B * d->H * d->Lq * d->dv == 0) return GNLP_OK;
  // d_k == 0: query and key hold no element (their pointers may be NULL), every score is 0
  REQUIRE(ctx && v && ((q && k) || d->dk == 0), "NULL tensor");
  return attn_impl(ctx, scores_out, weights_out, q, k, v, d, dtype, mask, mask_stride,
                   valid_length, vl_dtype, edge_scores, edge_stride, temperature, workspace,
                   workspace_bytes, S(stream));
}

int gnlp_multi_head_dot_attn_lse(void *ctx, void *ctx_lo, float *lse, const void *q, const void *k,
                                 const void *v, const gnlp_attn_desc *d, int dtype,
                                 const uint8_t *mask, const int64_t mask_stride[3],
                                 const void *valid_length, int vl_dtype, const void *edge_scores,
                                 const int64_t edge_stride[4], float temperature,
                                 gnlp_stream_t stream) {
  REQUIRE(lse != nullptr && ctx_lo != nullptr, "lse / ctx_lo is NULL");
  REQUIRE(((uintptr_t)lse & 3) == 0, "lse must be 4-byte aligned");
  REQUIRE(((uintptr_t)ctx_lo & 15) == 0, "ctx_lo must be 16-byte aligned");
  REQUIRE(d != nullptr, "descriptor is NULL");
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(d->B >= 0 && d->H >= 0 && d->Lq >= 0 && d->Lm >= 0 && d->dk >= 0 && d->dv >= 0,
          "negative shape");
  REQUIRE(!mask || mask_stride, "mask given without strides");
  REQUIRE(!edge_scores || edge_stride, "edge_scores given without strides");
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype %d", vl_dtype);
  REQUIRE(d->B * d->H <= 0x7fffffffLL, "batch*heads too large");
  REQUIRE(!(mask && d->causal), "d->causal selects the in-kernel predicate: mask must be NULL");
  REQUIRE(!d->mem_lengths || (!mask && !d->causal && valid_length),
          "d->mem_lengths selects the memory-mask predicate: mask NULL, not causal, valid_length "
          "with 2 B entries");
  if (d->B * d->H * d->Lq * d->dv == 0 || d->Lm == 0 || d->dk == 0)
    return fail(GNLP_ERR_UNSUPPORTED, "empty axis: use gnlp_multi_head_dot_attn");
  REQUIRE(ctx && q && k && v, "NULL tensor");
  return attn_impl(ctx, nullptr, nullptr, q, k, v, d, dtype, mask, mask_stride, valid_length,
                   vl_dtype, edge_scores, edge_stride, temperature, nullptr, 0, S(stream), nullptr,
                   nullptr, lse, ctx_lo);
}

size_t gnlp_attn_backward_workspace_bytes(const gnlp_attn_desc *d) {
  if (!d || d->B < 0 || d->H < 0 || d->Lq < 0) return 0;
  const size_t n = (size_t)d->B * (size_t)d->H * (size_t)d->Lq * 2 * sizeof(float);
  return (n + 255) / 256 * 256;
}

int gnlp_multi_head_dot_attn_backward(void *dq, void *dk, void *dv, float *d_edge,
                                      int64_t d_edge_elems, const int64_t d_edge_stride[4],
                                      const void *grad_ctx, const void *q, const void *k,
                                      const void *v, const gnlp_attn_desc *d,
                                      const gnlp_attn_grad_desc *gd, int dtype,
                                      const uint8_t *mask, const int64_t mask_stride[3],
                                      const void *valid_length, int vl_dtype,
                                      const void *edge_scores, const int64_t edge_stride[4],
                                      float temperature, void *workspace, size_t workspace_bytes,
                                      gnlp_stream_t stream) {
  REQUIRE(d != nullptr && gd != nullptr, "descriptor is NULL");
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(d->B >= 0 && d->H >= 0 && d->Lq >= 0 && d->Lm >= 0 && d->dk >= 0 && d->dv >= 0,
          "negative shape");
  REQUIRE(!mask || mask_stride, "mask given without strides");
  REQUIRE(!edge_scores || edge_stride, "edge_scores given without strides");
  REQUIRE(!d_edge || (edge_scores && d_edge_stride && d_edge_elems >= 0),
          "d_edge needs edge_scores, its strides and its element count");
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype %d", vl_dtype);
  REQUIRE(!(mask && d->causal), "d->causal selects the in-kernel predicate: mask must be NULL");
  REQUIRE(!d->mem_lengths || (!mask && !d->causal && valid_length),
          "d->mem_lengths selects the memory-mask predicate: mask NULL, not causal, valid_length "
          "= 2 B lengths");
  hipStream_t s = S(stream);
  if (d_edge && d_edge_elems > 0) {
    hipError_t e = hipMemsetAsync(d_edge, 0, (size_t)d_edge_elems * sizeof(float), s);
    if (e != hipSuccess) return fail(GNLP_ERR_LAUNCH, "hipMemsetAsync(d_edge): %s", hipGetErrorString(e));
  }
  if (d->B * d->H == 0) return GNLP_OK;
  if (d->Lq * d->Lm * d->dk * d->dv == 0)
    return fail(GNLP_ERR_UNSUPPORTED, "attention backward with an empty axis (Lq=%lld Lm=%lld "
                "dk=%lld dv=%lld): the gradients are all zero, fill them on the host side",
                (long long)d->Lq, (long long)d->Lm, (long long)d->dk, (long long)d->dv);
  REQUIRE(dq && dk && dv && grad_ctx && q && k && v, "NULL tensor");
  GenAttnBwd a;
  a.q = q; a.k = k; a.v = v; a.g = grad_ctx;
  a.dq = dq; a.dk_out = dk; a.dv_out = dv;
  a.de = d_edge;
  a.B = d->B; a.H = d->H; a.Lq = d->Lq; a.Lm = d->Lm; a.dk = d->dk; a.dv = d->dv;
  for (int i = 0; i < 3; ++i) {
    a.qs[i] = d->q_stride[i]; a.ks[i] = d->k_stride[i]; a.vs[i] = d->v_stride[i];
    a.gs[i] = gd->g_stride[i]; a.dqs[i] = gd->dq_stride[i]; a.dks[i] = gd->dk_stride[i];
    a.dvs[i] = gd->dv_stride[i];
    a.ms[i] = mask ? mask_stride[i] : 0;
  }
  for (int i = 0; i < 4; ++i) {
    a.bs[i] = edge_scores ? edge_stride[i] : 0;
    a.des[i] = d_edge ? d_edge_stride[i] : 0;
  }
  a.mask = mask;
  a.vl = mask ? nullptr : valid_length;
  a.vl_dtype = vl_dtype;
  a.causal = (!mask && d->causal) ? 1 : 0;
  a.vl_pair = (a.vl && d->mem_lengths) ? 1 : 0;
  a.bias = edge_scores;
  a.inv_temperature = temperature > 0.0f ? 1.0f / temperature : 1.0f;
  REQUIRE((gd->fwd_ctx == nullptr) == (gd->fwd_lse == nullptr) &&
          (gd->fwd_ctx_lo == nullptr) == (gd->fwd_lse == nullptr),
          "fwd_ctx, fwd_ctx_lo and fwd_lse come together (gnlp_multi_head_dot_attn_lse) or not at all");
  REQUIRE(((uintptr_t)gd->fwd_lse & 3) == 0, "fwd_lse must be 4-byte aligned");
  a.fwd_lse = gd->fwd_lse;
  a.fwd_o = gd->fwd_ctx;
  a.fwd_o_lo = gd->fwd_ctx_lo;
  for (int i = 0; i < 3; ++i) a.os[i] = d->o_stride[i];
  if (!attn_bwd_supported(a))
    return fail(GNLP_ERR_UNSUPPORTED, "attention backward: head dims %lld / %lld > 128 (or the "
                "grid limit) are not supported", (long long)d->dk, (long long)d->dv);
  const size_t need = gnlp_attn_backward_workspace_bytes(d);
  REQUIRE(workspace && workspace_bytes >= need, "workspace too small (%zu < %zu)", workspace_bytes,
          need);
  REQUIRE(((uintptr_t)workspace & 3) == 0, "workspace must be 4-byte aligned");
  a.stats = (float *)workspace;
  launch_attn_bwd(a, dtype, s);
  return check_launch("gnlp_multi_head_dot_attn_backward");
}

int gnlp_norm_backward(void *dx, float *dgamma, float *dbeta, const void *grad_out, const void *x,
                       const void *residual, const void *gamma, int dtype, int kind, float eps,
                       int64_t rows, int64_t C, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(kind == GNLP_NORM_LAYER || kind == GNLP_NORM_RMS || kind == GNLP_NORM_NONE,
          "normalization kind %d is not supported", kind);
  REQUIRE(rows >= 0 && C >= 0, "negative shape");
  REQUIRE((dgamma && dbeta) || C == 0, "NULL dgamma / dbeta");
  hipStream_t s = S(stream);
  if (C > 0) {
    hipError_t e = hipMemsetAsync(dgamma, 0, (size_t)C * sizeof(float), s);
    if (e == hipSuccess) e = hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), s);
    if (e != hipSuccess) return fail(GNLP_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
  if (rows * C == 0) return GNLP_OK;
  REQUIRE(dx && grad_out && x && gamma, "NULL tensor");
  if (!norm_bwd_supported(C))
    return fail(GNLP_ERR_UNSUPPORTED, "norm backward: C=%lld > %lld", (long long)C,
                (long long)kNormBwdMaxC);
  launch_norm_bwd(dx, dgamma, dbeta, grad_out, x, residual, gamma, dtype, kind, eps, rows, C, s);
  return check_launch("gnlp_norm_backward");
}

int gnlp_l2_normalize_backward(void *dx, const void *grad_out, const void *x, int dtype,
                               int64_t rows, int64_t d, float eps, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(rows >= 0 && d >= 0, "negative shape");
  REQUIRE((dx && grad_out && x) || rows * d == 0, "NULL tensor");
  launch_l2_normalize_bwd(dx, grad_out, x, dtype, rows, d, eps, S(stream));
  return check_launch("gnlp_l2_normalize_backward");
}

int gnlp_cast_from_f32(void *out, int dtype, const float *in, int64_t n, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(n >= 0 && ((out && in) || n == 0), "bad arguments");
  launch_cast_f32(out, dtype, in, n, S(stream));
  return check_launch("gnlp_cast_from_f32");
}

int gnlp_gen_rel_position(int32_t *out, int64_t L, gnlp_stream_t stream) {
  REQUIRE(L >= 0 && (out || L == 0), "bad arguments");
  REQUIRE(L < 46341, "L too large for an (L, L) int32 index grid");
  launch_rel_position(out, L, S(stream));
  return check_launch("gnlp_gen_rel_position");
}

int gnlp_relative_position_bucket(int32_t *out, const int32_t *rel, int64_t n, int bidirectional,
                                  int num_buckets, int max_distance, gnlp_stream_t stream) {
  BucketTable tb;
  int rc = build_bucket_table(&tb, bidirectional, num_buckets, max_distance);
  if (rc) return rc;
  REQUIRE(n >= 0 && ((out && rel) || n == 0), "bad arguments");
  launch_bucket(out, rel, n, tb, S(stream));
  return check_launch("gnlp_relative_position_bucket");
}

int gnlp_bucket_embedding(void *out, const void *weight, const int32_t *rel, int dtype, int64_t n,
                          int64_t units, int bidirectional, int num_buckets, int max_distance,
                          gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  BucketTable tb;
  int rc = build_bucket_table(&tb, bidirectional, num_buckets, max_distance);
  if (rc) return rc;
  REQUIRE(n >= 0 && units >= 0 && ((out && weight && rel) || n * units == 0), "bad arguments");
  launch_bucket_embedding(out, weight, rel, dtype, n, units, tb, S(stream));
  return check_launch("gnlp_bucket_embedding");
}

int gnlp_sinusoidal_embedding(void *out, const float *positions, const float *base_mult, int dtype,
                              int64_t n, int64_t units, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "dtype=%d", dtype);
  REQUIRE(n >= 0 && units >= 0, "n=%lld units=%lld", (long long)n, (long long)units);
  if (n * units == 0) return GNLP_OK;
  REQUIRE(out && positions, "null out / positions");
  REQUIRE(units < 2 || base_mult, "null base_mult");
  launch_sinusoid(out, positions, base_mult, dtype, n, units, S(stream));
  return check_launch("gnlp_sinusoidal_embedding");
}

int gnlp_rel_attn_score(void *out, const void *query, const int64_t *q_stride,
                        const void *pos_embed, const int32_t *rel_positions, int dtype, int64_t B,
                        int64_t H, int64_t Lq, int64_t Lm, int64_t C, int64_t R, int32_t clip_lo,
                        int32_t clip_hi, int32_t base, int wrap, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "dtype=%d", dtype);
  REQUIRE(B >= 0 && H >= 0 && Lq >= 0 && Lm >= 0 && C >= 0, "negative extent");
  if (B * H * Lq * Lm == 0) return GNLP_OK;
  REQUIRE(out && query && q_stride && pos_embed && rel_positions, "null argument");
  REQUIRE(R >= 1, "empty position table (R=%lld)", (long long)R);
  REQUIRE(clip_lo <= clip_hi, "clip range [%d, %d]", clip_lo, clip_hi);
  REQUIRE(C <= 8192, "head units %lld > 8192", (long long)C);
  REQUIRE(B * H * Lq <= 0x7fffffffLL, "B*H*Lq=%lld exceeds the grid limit", (long long)(B * H * Lq));
  RelScoreArgs a;
  a.B = B; a.H = H; a.Lq = Lq; a.Lm = Lm; a.C = C; a.R = R;
  for (int i = 0; i < 3; ++i) a.qs[i] = q_stride[i];
  a.lo = clip_lo; a.hi = clip_hi; a.base = base; a.wrap = wrap ? 1 : 0;
  launch_rel_score(out, query, pos_embed, rel_positions, dtype, a, S(stream));
  return check_launch("gnlp_rel_attn_score");
}

int gnlp_rel_attn_score_backward(void *dq, const int64_t *dq_stride, float *d_pos_embed,
                                 const void *grad_out, const void *query, const int64_t *q_stride,
                                 const void *pos_embed, const int32_t *rel_positions, int dtype,
                                 int64_t B, int64_t H, int64_t Lq, int64_t Lm, int64_t C, int64_t R,
                                 int32_t clip_lo, int32_t clip_hi, int32_t base, int wrap,
                                 gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "dtype=%d", dtype);
  REQUIRE(B >= 0 && H >= 0 && Lq >= 0 && Lm >= 0 && C >= 0 && R >= 0, "negative extent");
  hipStream_t s = S(stream);
  if (d_pos_embed && R * H * C > 0) {
    hipError_t e = hipMemsetAsync(d_pos_embed, 0, (size_t)(R * H * C) * sizeof(float), s);
    if (e != hipSuccess) return fail(GNLP_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
  if (B * H * Lq == 0) return GNLP_OK;
  REQUIRE(grad_out && query && q_stride && pos_embed && rel_positions, "null argument");
  REQUIRE(!dq || dq_stride, "dq given without strides");
  REQUIRE(R >= 1, "empty position table (R=%lld)", (long long)R);
  REQUIRE(clip_lo <= clip_hi, "clip range [%d, %d]", clip_lo, clip_hi);
  REQUIRE(C <= 8192, "head units %lld > 8192", (long long)C);
  REQUIRE(B * H * Lq <= 0x7fffffffLL, "B*H*Lq=%lld exceeds the grid limit", (long long)(B * H * Lq));
  RelScoreArgs a;
  a.B = B; a.H = H; a.Lq = Lq; a.Lm = Lm; a.C = C; a.R = R;
  for (int i = 0; i < 3; ++i) a.qs[i] = q_stride[i];
  a.lo = clip_lo; a.hi = clip_hi; a.base = base; a.wrap = wrap ? 1 : 0;
  launch_rel_score_bwd(dq, dq_stride, d_pos_embed, grad_out, query, pos_embed, rel_positions,
                       dtype, a, s);
  return check_launch("gnlp_rel_attn_score_backward");
}

int gnlp_bucket_embedding_backward(float *d_weight, const void *grad_out, const int32_t *rel,
                                   int dtype, int64_t n, int64_t units, int bidirectional,
                                   int num_buckets, int max_distance, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  BucketTable tb;
  int rc = build_bucket_table(&tb, bidirectional, num_buckets, max_distance);
  if (rc) return rc;
  REQUIRE(n >= 0 && units >= 0, "negative extent");
  hipStream_t s = S(stream);
  if ((int64_t)num_buckets * units > 0) {
    REQUIRE(d_weight, "NULL d_weight");
    hipError_t e = hipMemsetAsync(d_weight, 0, (size_t)num_buckets * units * sizeof(float), s);
    if (e != hipSuccess) return fail(GNLP_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
  REQUIRE((grad_out && rel) || n * units == 0, "bad arguments");
  launch_bucket_embedding_bwd(d_weight, grad_out, rel, dtype, n, units, tb, s);
  return check_launch("gnlp_bucket_embedding_backward");
}

int gnlp_colsum(float *out, const void *x, int dtype, int64_t rows, int64_t C,
                gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(rows >= 0 && C >= 0 && (out || C == 0) && (x || rows * C == 0), "bad arguments");
  hipStream_t s = S(stream);
  if (C > 0) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)C * sizeof(float), s);
    if (e != hipSuccess) return fail(GNLP_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
  launch_colsum(out, x, dtype, rows, C, s);
  return check_launch("gnlp_colsum");
}

int gnlp_t5_rel_bias(void *out, const void *weight, int dtype, int64_t H, int64_t Lq, int64_t Lm,
                     int bidirectional, int num_buckets, int max_distance, gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  BucketTable tb;
  int rc = build_bucket_table(&tb, bidirectional, num_buckets, max_distance);
  if (rc) return rc;
  REQUIRE(H >= 0 && Lq >= 0 && Lm >= 0 && ((out && weight) || H * Lq * Lm == 0), "bad arguments");
  launch_t5_rel_bias(out, weight, dtype, H, Lq, Lm, 0, tb, S(stream));
  return check_launch("gnlp_t5_rel_bias");
}

int gnlp_t5_rel_bias_table(void *out, const void *weight, int dtype, int64_t H, int64_t Lq,
                           int64_t Lm, int bidirectional, int num_buckets, int max_distance,
                           gnlp_stream_t stream) {
  REQUIRE(dtype_ok(dtype), "unsupported dtype %d", dtype);
  REQUIRE(H >= 0 && Lq >= 1 && Lm >= 1, "bad shape");
  REQUIRE(Lq + Lm < (1LL << 30), "lengths too large");
  if (H == 0) return GNLP_OK;
  REQUIRE(out && weight, "NULL tensor");
  BucketTable tb;
  int rc = build_bucket_table(&tb, bidirectional, num_buckets, max_distance);
  if (rc) return rc;
  // one row per head: entry t holds the bias of relative position t - (Lq - 1)
  launch_t5_rel_bias(out, weight, dtype, H, 1, Lq + Lm - 1, -(int)(Lq - 1), tb, S(stream));
  return check_launch("gnlp_t5_rel_bias_table");
}

size_t gnlp_ffn_workspace_bytes(const gnlp_ffn_cfg *cfg, int64_t rows) {
  if (!cfg || rows < 0) return 0;
  return ffn_ws(cfg, rows).total;
}

int gnlp_positionwise_ffn(void *out, const void *x, const gnlp_ffn_params *p,
                          const gnlp_ffn_cfg *cfg, int64_t rows, const void *valid_length,
                          int vl_dtype, int64_t B, int64_t L, int time_major, void *workspace,
                          size_t workspace_bytes, gnlp_stream_t stream) {
  REQUIRE(p && cfg, "NULL params / cfg");
  REQUIRE(dtype_ok(cfg->dtype), "unsupported dtype %d", cfg->dtype);
  REQUIRE(cfg->units > 0 && cfg->hidden > 0 && rows >= 0, "bad shape");
  REQUIRE(cfg->act >= GNLP_ACT_IDENTITY && cfg->act <= GNLP_ACT_SOFTSIGN,
          "act=%d is not supported", cfg->act);
  REQUIRE(cfg->norm_kind >= GNLP_NORM_LAYER && cfg->norm_kind <= GNLP_NORM_NONE,
          "normalization kind %d is not supported", cfg->norm_kind);
  REQUIRE(p->ffn1_w && p->ffn2_w && p->ln_g, "missing ffn weights");
  REQUIRE(!cfg->gated || p->gate_w, "gated FFN without gated_ffn_1 weight");
  REQUIRE(!valid_length || (vl_ok(vl_dtype) && B * L == rows), "bad valid_length arguments");
  if (rows == 0) return GNLP_OK;
  REQUIRE(out && x && out != x, "NULL or aliased tensors");
  return ffn_impl(out, x, p, cfg, rows, valid_length, vl_dtype, B, L, time_major, workspace,
                  workspace_bytes, S(stream));
}

size_t gnlp_encoder_workspace_bytes(const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                                    int want_weights) {
  if (!cfg || B < 0 || L < 0 || cfg->heads <= 0) return 0;
  // layer scratch + two ping-pong activations for the stack driver
  // (+ the sequence offsets of a packed ragged batch: B + 2 int32)
  // (+ two (rows, 2) fp32 row-statistics vectors of a stack run with folded LayerNorms)
  return enc_ws(cfg, B, L, want_weights != 0).total +
         2 * align256((size_t)B * L * cfg->units * esize(cfg->dtype)) +
         align256((size_t)(B + 2) * sizeof(int32_t)) + 2 * align256((size_t)B * L * 2 * sizeof(float)) +
         fold_part_bytes(B * L, cfg->units);
}

int gnlp_encoder_layer_forward(void *out, const void *x, const gnlp_encoder_layer_params *p,
                               const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                               const uint8_t *mask, const void *valid_length, int vl_dtype,
                               void *attn_weights_out, int final_seq_mask, void *workspace,
                               size_t workspace_bytes, gnlp_stream_t stream) {
  int rc = check_encoder_cfg(cfg, B, L);
  if (rc) return rc;
  REQUIRE(p && p->qkv_w && p->proj_w && p->ln_g && p->ffn.ffn1_w && p->ffn.ffn2_w && p->ffn.ln_g,
          "missing layer weights");
  if (B * L == 0) return GNLP_OK;
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype");
  REQUIRE(!final_seq_mask || valid_length, "final_seq_mask needs valid_length");
  REQUIRE(out && x && out != x, "NULL or aliased tensors");
  return encoder_layer_impl(out, x, p, cfg, B, L, mask, valid_length, vl_dtype, attn_weights_out,
                            final_seq_mask, workspace, workspace_bytes, S(stream));
}

int gnlp_encoder_layer_forward_noalloc(void *out, const void *x,
                                       const gnlp_encoder_layer_params *p,
                                       const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                                       const uint8_t *mask, const void *valid_length,
                                       int vl_dtype, void *attn_weights_out, int final_seq_mask,
                                       gnlp_stream_t stream) {
  int rc = check_encoder_cfg(cfg, B, L);
  if (rc) return rc;
  REQUIRE(p && p->qkv_w && p->proj_w && p->ln_g && p->ffn.ffn1_w && p->ffn.ffn2_w && p->ffn.ln_g,
          "missing layer weights");
  if (B * L == 0) return GNLP_OK;
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype");
  REQUIRE(!final_seq_mask || valid_length, "final_seq_mask needs valid_length");
  REQUIRE(out && x && out != x, "NULL or aliased tensors");
  const size_t need = enc_ws(cfg, B, L, attn_weights_out != nullptr).total;
  REQUIRE(need <= kStaticScratchBytes,
          "gnlp_encoder_layer_forward_noalloc: this layer needs %zu bytes of scratch, the static "
          "scratch holds %zu (GNLP_NOALLOC_SCRATCH_BYTES); use gnlp_encoder_layer_forward with a "
          "caller-owned workspace", need, kStaticScratchBytes);
  void *ws = nullptr;
  rc = static_scratch(&ws, S(stream));   // first call on this device: one hipGetSymbolAddress, no allocation
  if (rc) return rc;
  // every region of the scratch is written by this call before it is read: nothing to clear
  return encoder_layer_impl(out, x, p, cfg, B, L, mask, valid_length, vl_dtype, attn_weights_out,
                            final_seq_mask, ws, kStaticScratchBytes, S(stream));
}

static int encoder_stack_impl(void *out, const void *x, const gnlp_encoder_layer_params *layers,
                              const gnlp_encoder_layer_folded *folded, int n_layers,
                              const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                              const void *valid_length, int vl_dtype, int final_seq_mask,
                              void *const *all_encodings, void *workspace, size_t workspace_bytes,
                              gnlp_stream_t stream);

int gnlp_encoder_stack_forward(void *out, const void *x, const gnlp_encoder_layer_params *layers,
                               int n_layers, const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                               const void *valid_length, int vl_dtype, int final_seq_mask,
                               void *const *all_encodings, void *workspace, size_t workspace_bytes,
                               gnlp_stream_t stream) {
  return encoder_stack_impl(out, x, layers, nullptr, n_layers, cfg, B, L, valid_length, vl_dtype,
                            final_seq_mask, all_encodings, workspace, workspace_bytes, stream);
}

int gnlp_encoder_stack_forward_folded(void *out, const void *x,
                                      const gnlp_encoder_layer_params *layers,
                                      const gnlp_encoder_layer_folded *folded, int n_layers,
                                      const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                                      const void *valid_length, int vl_dtype, int final_seq_mask,
                                      void *workspace, size_t workspace_bytes,
                                      gnlp_stream_t stream) {
  REQUIRE(folded != nullptr, "folded parameters are NULL (use gnlp_encoder_stack_forward)");
  return encoder_stack_impl(out, x, layers, folded, n_layers, cfg, B, L, valid_length, vl_dtype,
                            final_seq_mask, nullptr, workspace, workspace_bytes, stream);
}

int gnlp_stack_last_folded(void) { return g_stack_folded; }

int gnlp_fold_norm_dense(void *wf, float *c, float *bf, const void *w, const void *b,
                         const void *gamma, const void *beta, int dtype, int64_t N, int64_t K,
                         int64_t n_scaled, float scale, gnlp_stream_t stream) {
  REQUIRE(dtype == GNLP_BF16, "gnlp_fold_norm_dense: bfloat16 parameters only");
  REQUIRE(N >= 0 && K >= 0 && K % 8 == 0, "bad shape N=%lld K=%lld (K a multiple of 8)",
          (long long)N, (long long)K);
  if (N == 0) return GNLP_OK;
  REQUIRE(wf && c && bf && w && gamma, "NULL tensors");
  REQUIRE(((uintptr_t)wf | (uintptr_t)w | (uintptr_t)gamma | (uintptr_t)(beta ? beta : gamma)) % 16 == 0,
          "gnlp_fold_norm_dense: 16-byte aligned tensors");
  launch_fold_norm_dense(wf, c, bf, w, b, gamma, beta, N, K, n_scaled, scale, S(stream));
  return check_launch("gnlp_fold_norm_dense");
}

int gnlp_row_stats(float *rstats, const void *x, int dtype, float eps, int64_t rows, int64_t C,
                   gnlp_stream_t stream) {
  REQUIRE(dtype == GNLP_BF16, "gnlp_row_stats: bfloat16 rows only");
  REQUIRE(rows >= 0 && C >= 1, "bad shape");
  if (rows == 0) return GNLP_OK;
  REQUIRE(rstats && x && row_stats_supported(C, x) && (uintptr_t)rstats % 8 == 0,
          "gnlp_row_stats: rows of at most 2048 elements, a multiple of 8, 16-byte aligned");
  launch_row_stats(rstats, x, eps, rows, C, S(stream));
  return check_launch("gnlp_row_stats");
}

static int encoder_stack_impl(void *out, const void *x, const gnlp_encoder_layer_params *layers,
                              const gnlp_encoder_layer_folded *folded, int n_layers,
                              const gnlp_encoder_cfg *cfg, int64_t B, int64_t L,
                              const void *valid_length, int vl_dtype, int final_seq_mask,
                              void *const *all_encodings, void *workspace, size_t workspace_bytes,
                              gnlp_stream_t stream) {
  int rc = check_encoder_cfg(cfg, B, L);
  if (rc) return rc;
  REQUIRE(layers && n_layers >= 1, "no layers");
  if (B * L == 0) return GNLP_OK;
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype");
  REQUIRE(!final_seq_mask || valid_length, "final_seq_mask needs valid_length");
  REQUIRE(x && (out || all_encodings), "NULL tensors");
  const size_t act_bytes = align256((size_t)B * L * cfg->units * esize(cfg->dtype));
  const size_t layer_bytes = enc_ws(cfg, B, L, false).total;
  REQUIRE(workspace && workspace_bytes >= layer_bytes + 2 * act_bytes,
          "encoder stack workspace too small: need %zu bytes, got %zu", layer_bytes + 2 * act_bytes,
          workspace_bytes);
  char *base = (char *)workspace;
  void *pp[2] = {base + layer_bytes, base + layer_bytes + act_bytes};
  const void *in = x;
  // Ragged batch whose padded rows come out as zeros anyway (final_seq_mask): run the layers on the
  // valid tokens only -- gather, n layers on (sum vl) rows, scatter + zero fill.  Row by row the
  // results are bit-identical to the padded run (no kernel mixes rows except attention, which sees
  // the same keys).  Needs the persistent GEMM for all four shapes and the fused attention kernel.
  const size_t cu_bytes = align256((size_t)(B + 2) * sizeof(int32_t));
  g_stack_packed = 0;
  g_stack_folded = 0;
  // LayerNorms folded into the GEMMs around them: the layers hand each
  // other RAW pre-norm sums plus two floats of statistics per row; only the last layer's norm runs
  // as a kernel.  Needs the caller's folded parameters and the persistent GEMM for all four shapes.
  const size_t rs_bytes = align256((size_t)B * L * 2 * sizeof(float));
  if (folded && !all_encodings && knobs().ln_fold &&
      workspace_bytes >= layer_bytes + 2 * act_bytes + cu_bytes + 2 * rs_bytes +
                             fold_part_bytes(B * L, cfg->units) &&
      stack_can_fold(cfg, B, L, base, layers, folded, n_layers)) {
    hipStream_t s = S(stream);
    float *rs[2] = {(float *)(base + layer_bytes + 2 * act_bytes + cu_bytes),
                    (float *)(base + layer_bytes + 2 * act_bytes + cu_bytes + rs_bytes)};
    float *part = (float *)(base + layer_bytes + 2 * act_bytes + cu_bytes + 2 * rs_bytes);
    void *raw = base + enc_ws(cfg, B, L, false).x1;   // (the materialised LN(sum) of the plain path)
    const bool pack = valid_length && final_seq_mask && knobs().pack_ragged &&
                      stack_can_pack(cfg, B, L, base, layers, n_layers);
    const int32_t *cu = nullptr;
    const void *x0 = x;
    void *dst = out;
    if (pack) {
      int32_t *cuw = (int32_t *)(base + layer_bytes + 2 * act_bytes);
      launch_seq_offsets(cuw, valid_length, vl_dtype, B, L, s);
      launch_pack_rows(pp[1], x, cuw, cfg->dtype, B, L, cfg->units, cfg->time_major, s);
      launch_zero_pad_rows(base + enc_ws(cfg, B, L, false).ctx, cuw, cfg->dtype, B, cfg->units, s);
      cu = cuw; x0 = pp[1]; dst = pp[0];
      g_stack_packed = 1;
    }
    g_stack_folded = 1;
    RowsDevScope scope(pack ? cu + B + 1 : nullptr);
    for (int i = 0; i < n_layers; ++i) {
      const gnlp_encoder_layer_params *p = &layers[i];
      REQUIRE(p->qkv_w && p->proj_w && p->ln_g && p->ffn.ffn1_w && p->ffn.ffn2_w && p->ffn.ln_g,
              "missing weights in layer %d", i);
      const bool last = i == n_layers - 1;
      rc = encoder_layer_folded(dst, x0, i ? raw : nullptr, rs[1], i ? &layers[i - 1] : nullptr, p,
                                &folded[i], cfg, B, L, valid_length, vl_dtype,
                                pack ? 0 : final_seq_mask, last, raw, rs[0], rs[1], part, base,
                                layer_bytes, s, cu);
      if (rc) return rc;
    }
    if (pack) launch_unpack_rows(out, dst, cu, cfg->dtype, B, L, cfg->units, cfg->time_major, s);
    return check_launch("gnlp_encoder_stack_forward (folded LayerNorm)");
  }
  if (valid_length && final_seq_mask && !all_encodings && knobs().pack_ragged &&
      workspace_bytes >= layer_bytes + 2 * act_bytes + cu_bytes &&
      stack_can_pack(cfg, B, L, base, layers, n_layers)) {
    hipStream_t s = S(stream);
    g_stack_packed = 1;
    int32_t *cu = (int32_t *)(base + layer_bytes + 2 * act_bytes);
    launch_seq_offsets(cu, valid_length, vl_dtype, B, L, s);
    launch_pack_rows(pp[1], x, cu, cfg->dtype, B, L, cfg->units, cfg->time_major, s);
    // the context rows between the last token and the dense row count belong to no sequence: the
    // attention kernel never writes them, the projection GEMM reads them (every layer: once is enough)
    launch_zero_pad_rows(base + enc_ws(cfg, B, L, false).ctx, cu, cfg->dtype, B, cfg->units, s);
    RowsDevScope scope(cu + B + 1);
    in = pp[1];
    for (int i = 0; i < n_layers; ++i) {
      void *dst = pp[i & 1];
      const gnlp_encoder_layer_params *p = &layers[i];
      REQUIRE(p->qkv_w && p->proj_w && p->ln_g && p->ffn.ffn1_w && p->ffn.ffn2_w && p->ffn.ln_g,
              "missing weights in layer %d", i);
      rc = encoder_layer_impl(dst, in, p, cfg, B, L, nullptr, valid_length, vl_dtype, nullptr, 0,
                              base, layer_bytes, s, cu);
      if (rc) return rc;
      in = dst;
    }
    launch_unpack_rows(out, in, cu, cfg->dtype, B, L, cfg->units, cfg->time_major, s);
    return check_launch("gnlp_encoder_stack_forward (packed)");
  }
  for (int i = 0; i < n_layers; ++i) {
    void *dst;
    if (all_encodings) dst = all_encodings[i];
    else dst = (i == n_layers - 1) ? out : pp[i & 1];
    REQUIRE(dst && dst != in, "NULL or aliased layer output");
    const gnlp_encoder_layer_params *p = &layers[i];
    REQUIRE(p->qkv_w && p->proj_w && p->ln_g && p->ffn.ffn1_w && p->ffn.ffn2_w && p->ffn.ln_g,
            "missing weights in layer %d", i);
    // bert.py:300-304: with output_all_encodings every layer output is masked and fed on
    const int seqm = final_seq_mask && (all_encodings || i == n_layers - 1);
    rc = encoder_layer_impl(dst, in, p, cfg, B, L, nullptr, valid_length, vl_dtype, nullptr, seqm,
                            base, layer_bytes, S(stream));
    if (rc) return rc;
    in = dst;
  }
  return GNLP_OK;
}

// ---- T5 ------------------------------------------------------------------------------------------
namespace {
struct T5Ws { size_t normed, qkv, ctx, h1, ffn, total; };
gnlp_ffn_cfg t5_ffn_cfg(const gnlp_t5_cfg *c) {
  gnlp_ffn_cfg f;
  f.units = c->d_model; f.hidden = c->d_ff;
  f.act = c->gated ? GNLP_ACT_GELU_TANH : GNLP_ACT_RELU;   // t5.py:341-342
  f.gated = c->gated; f.norm_kind = GNLP_NORM_RMS; f.pre_norm = 1; f.dtype = c->dtype;
  f.ln_eps = 1e-6f;  // layers.py:67-68: epsilon is not forwarded to RMSNorm (default 1e-6)
  return f;
}
T5Ws t5_ws(const gnlp_t5_cfg *c, int64_t B, int64_t L) {
  T5Ws w;
  const size_t e = esize(c->dtype);
  const int64_t M = B * L, inner = c->heads * c->d_kv;
  size_t off = 0;
  w.normed = off; off += align256((size_t)M * c->d_model * e);
  w.qkv = off; off += align256((size_t)M * 3 * inner * e);
  w.ctx = off; off += align256((size_t)M * inner * e);
  w.h1 = off; off += align256((size_t)M * c->d_model * e);
  w.ffn = off;
  gnlp_attn_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.H = c->heads; d.Lq = L; d.Lm = L; d.dk = d.dv = c->d_kv;
  const bool fused = stack_attn_fuses(c->dtype, B, c->heads, c->d_kv, 3 * inner, inner);
  size_t attn = attn_ws_bytes(&d, c->dtype, false, false, fused);
  gnlp_ffn_cfg f = t5_ffn_cfg(c);
  size_t ffn = ffn_ws(&f, M).total;
  off += attn > ffn ? attn : ffn;   // attention scratch and FFN scratch are never live together
  w.total = off;
  return w;
}
}  // namespace

size_t gnlp_t5_workspace_bytes(const gnlp_t5_cfg *cfg, int64_t B, int64_t L) {
  if (!cfg || B < 0 || L < 0) return 0;
  // (+ a float32 copy of the (H, 2L - 1) Toeplitz bias table for the bf16 attention kernel)
  return t5_ws(cfg, B, L).total + 2 * align256((size_t)B * L * cfg->d_model * esize(cfg->dtype)) +
         align256((size_t)cfg->heads * (2 * L) * sizeof(float));
}

namespace {
// RMSNorm folded into the GEMMs that consume it (bf16, relu FFN, packed q | k | v weights, rows a
// multiple of 256): RMSNorm(h) . W^T = r (h . W'^T), W' = bf16(gamma * W),
// r = rsqrt(mean(h^2) + eps) -- FOLD 1 with r * mu = 0.  The pre-norm residuals are plain tensors:
// proj and ffn_2 run FOLD 2 with gamma = 1, beta = 0, r = 1 (exactly acc + other) for the sake
// of its by-product, the partial sums of squares of the output rows, from which the next norm's
// r comes without a pass over the tensor.
bool t5_can_fold(const gnlp_t5_cfg *c, int64_t B, int64_t L, const void *ws,
                 const gnlp_t5_block_params *layers, const gnlp_t5_block_folded *folded,
                 int n_layers) {
  const int64_t M = B * L, D = c->d_model, inner = c->heads * c->d_kv, F = c->d_ff;
  if (!folded || !knobs().ln_fold || c->dtype != GNLP_BF16 || c->gated || M % 256 != 0 ||
      D > 2048 || !row_stats_supported(D, ws))
    return false;
  for (int i = 0; i < n_layers; ++i) {
    const gnlp_t5_block_params *p = &layers[i];
    const gnlp_t5_block_folded *f = &folded[i];
    if (!p->qkv_w || !f->qkv_wf || !f->qkv_c || !f->qkv_bf || !f->ffn1_wf || !f->ffn1_c ||
        !f->ffn1_bf || p->ffn.gate_w || p->ffn.ffn1_b || p->ffn.ffn2_b)
      return false;
    if (((uintptr_t)f->qkv_wf | (uintptr_t)f->qkv_c | (uintptr_t)f->qkv_bf | (uintptr_t)f->ffn1_wf |
         (uintptr_t)f->ffn1_c | (uintptr_t)f->ffn1_bf) % 16)
      return false;
    if (!mfma_bf16_dense_takes_w4(M, 3 * inner, D, D, 3 * inner, 0, ws, nullptr, nullptr) ||
        !mfma_bf16_dense_takes_w4(M, D, inner, inner, D, D, ws, nullptr, ws) ||
        !mfma_bf16_dense_takes_w4(M, F, D, D, F, 0, ws, nullptr, nullptr) ||
        !mfma_bf16_dense_takes_w4(M, D, F, F, D, D, ws, nullptr, ws))
      return false;
  }
  DenseFold f1{1, (const float *)ws, (const float *)ws, (const float *)ws, nullptr, nullptr, nullptr};
  DenseFold f2{2, (const float *)ws, nullptr, nullptr, ws, ws, (float *)ws};
  return mfma_bf16_dense_w4_fold_supported(f1, GNLP_ACT_IDENTITY, 0, nullptr, 3 * inner) &&
         mfma_bf16_dense_w4_fold_supported(f1, GNLP_ACT_RELU, 0, nullptr, F) &&
         mfma_bf16_dense_w4_fold_supported(f2, GNLP_ACT_IDENTITY, GNLP_COMBINE_ADD, ws, D);
}
}  // namespace

static int t5_encoder_impl(void *out, const void *x, const gnlp_t5_block_params *layers,
                           const gnlp_t5_block_folded *folded, int n_layers,
                           const void *final_ln_g, const void *final_ln_b, const gnlp_t5_cfg *cfg,
                           int64_t B, int64_t L, const void *valid_length, int vl_dtype,
                           const void *position_bias, void *workspace, size_t workspace_bytes,
                           gnlp_stream_t stream) {
  g_stack_folded = 0;
  REQUIRE(cfg && layers && n_layers >= 1, "NULL cfg / layers");
  REQUIRE(dtype_ok(cfg->dtype), "unsupported dtype %d", cfg->dtype);
  REQUIRE(cfg->d_model > 0 && cfg->d_kv > 0 && cfg->d_ff > 0 && cfg->heads > 0, "bad T5 sizes");
  REQUIRE(!valid_length || vl_ok(vl_dtype), "bad valid_length dtype");
  REQUIRE(B >= 0 && L >= 0, "negative shape");
  if (B * L == 0) return GNLP_OK;
  REQUIRE(out && x, "NULL tensors");
  const T5Ws w = t5_ws(cfg, B, L);
  const size_t e = esize(cfg->dtype);
  const int64_t M = B * L, D = cfg->d_model, H = cfg->heads, dkv = cfg->d_kv, inner = H * dkv;
  const size_t act_bytes = align256((size_t)M * D * e);
  REQUIRE(workspace && workspace_bytes >= w.total + 2 * act_bytes,
          "t5 workspace too small: need %zu bytes, got %zu", w.total + 2 * act_bytes,
          workspace_bytes);
  REQUIRE((uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
  REQUIRE(!position_bias || (uintptr_t)position_bias % esize(cfg->dtype) == 0,
          "position_bias is not element aligned");
  hipStream_t s = S(stream);
  char *base = (char *)workspace;
  void *pp[2] = {base + w.total, base + w.total + act_bytes};
  gnlp_ffn_cfg f = t5_ffn_cfg(cfg);
  const void *h = x;
  int rc;
  // bf16 Toeplitz table: widened once per call so that the attention kernel's bias loads are its
  // accumulator init (optional: only when the caller's workspace has the room)
  const float *bias_f32 = nullptr;
  const size_t tab_bytes = align256((size_t)H * (2 * L) * sizeof(float));
  if (position_bias && cfg->bias_toeplitz && cfg->dtype == GNLP_BF16 &&
      workspace_bytes >= w.total + 2 * act_bytes + tab_bytes) {
    float *tab = (float *)(base + w.total + 2 * act_bytes);
    launch_widen_bf16(tab, position_bias, H * (2 * L - 1), s);
    bias_f32 = tab + (L - 1);
  }
  // folded path: constants and statistics live where the plain path keeps the normalised tensor
  const bool fold = t5_can_fold(cfg, B, L, base, layers, folded, n_layers);
  float *rs_a = nullptr, *rs_b = nullptr, *rs_one = nullptr, *part = nullptr;
  void *ones = nullptr, *zeros = nullptr;
  if (fold) {
    const size_t rs_bytes = align256((size_t)M * 2 * sizeof(float));
    const size_t part_bytes = fold_part_bytes(M, D), vec_bytes = align256((size_t)D * e);
    if (3 * rs_bytes + part_bytes + 2 * vec_bytes <= align256((size_t)M * D * e)) {
      char *q = base + w.normed;
      rs_a = (float *)q; rs_b = (float *)(q + rs_bytes); rs_one = (float *)(q + 2 * rs_bytes);
      part = (float *)(q + 3 * rs_bytes);
      ones = q + 3 * rs_bytes + part_bytes; zeros = (char *)ones + vec_bytes;
      launch_fold_consts(ones, zeros, rs_one, D, M, s);
      launch_row_stats(rs_a, x, cfg->ln_eps, M, D, s, true);   // t5.py:537 of the first block
      g_stack_folded = 1;
    }
  }
  for (int i = 0; i < n_layers; ++i) {
    const gnlp_t5_block_params *p = &layers[i];
    REQUIRE(p->attn_ln_g && (p->qkv_w || (p->q_w && p->k_w && p->v_w)) && p->proj_w && p->ffn.ffn1_w &&
            p->ffn.ffn2_w && p->ffn.ln_g, "missing weights in T5 block %d", i);
    const bool last = (i == n_layers - 1);
    void *dst = (last && !final_ln_g) ? out : pp[i & 1];
    char *qkv = base + w.qkv;
    if (rs_a) {   // q | k | v = r (h . W'^T) [+ b']
      const DenseFold fd{1, rs_a, folded[i].qkv_c, folded[i].qkv_bf, nullptr, nullptr, nullptr};
      rc = dense_impl(qkv, h, folded[i].qkv_wf, nullptr, nullptr, 0, GNLP_ACT_IDENTITY, cfg->dtype,
                      M, 3 * inner, D, D, 3 * inner, 0, s, &fd);
      if (rc) return rc;
    } else {
    // t5.py:537  out = self_attn_layer_norm(hidden)
    launch_norm(base + w.normed, h, nullptr, p->attn_ln_g, p->attn_ln_b, cfg->dtype, GNLP_NORM_RMS,
                cfg->ln_eps, M, D, nullptr, 0, 0, 0, 0, s);
    }
    // t5.py:538-542  bias-free q / k / v projections written side by side: [q | k | v] rows
    if (rs_a) {
    } else if (p->qkv_w) {   // the three weights packed row-wise: one GEMM of N = 3 * inner
      rc = dense_impl(qkv, base + w.normed, p->qkv_w, nullptr, nullptr, 0, GNLP_ACT_IDENTITY,
                      cfg->dtype, M, 3 * inner, D, D, 3 * inner, 0, s);
      if (rc) return rc;
    } else {
      const void *wq[3] = {p->q_w, p->k_w, p->v_w};
      for (int t = 0; t < 3; ++t) {
        rc = dense_impl(qkv + (size_t)t * inner * e, base + w.normed, wq[t], nullptr, nullptr, 0,
                        GNLP_ACT_IDENTITY, cfg->dtype, M, inner, D, D, 3 * inner, 0, s);
        if (rc) return rc;
      }
    }
    gnlp_attn_desc ad;
    memset(&ad, 0, sizeof(ad));
    ad.B = B; ad.H = H; ad.Lq = L; ad.Lm = L; ad.dk = dkv; ad.dv = dkv;
    const int64_t sb = cfg->time_major ? 1 : L, st = cfg->time_major ? B : 1;
    ad.q_stride[0] = ad.k_stride[0] = ad.v_stride[0] = sb * 3 * inner;
    ad.q_stride[1] = ad.k_stride[1] = ad.v_stride[1] = dkv;
    ad.q_stride[2] = ad.k_stride[2] = ad.v_stride[2] = st * 3 * inner;
    ad.o_stride[0] = sb * inner; ad.o_stride[1] = dkv; ad.o_stride[2] = st * inner;
    // dense (H, L, L) bias, or the Toeplitz table (H, 2L - 1) viewed as bias[h, i, j] =
    // table[h, (L - 1) + j - i]: row stride -1, so the kernels read it like any strided bias and
    // the (H, L, L) tensor never exists (49 KB instead of 25 MB at T5-base, L = 1024)
    const int64_t estr_dense[4] = {0, L * L, L, 1};
    const int64_t estr_toep[4] = {0, 2 * L - 1, -1, 1};
    const int64_t *estr = cfg->bias_toeplitz ? estr_toep : estr_dense;
    const void *bias0 = (position_bias && cfg->bias_toeplitz)
                            ? (const void *)((const char *)position_bias + (size_t)(L - 1) * e)
                            : position_bias;
    // t5.py:543-549  scaled=False (ctor :261-270), edge_scores = position_embeddings
    rc = attn_impl(base + w.ctx, nullptr, nullptr, qkv, qkv + (size_t)inner * e,
                   qkv + (size_t)2 * inner * e, &ad, cfg->dtype, nullptr, nullptr, valid_length,
                   vl_dtype, bias0, bias0 ? estr : nullptr, 0.0f, base + w.ffn,
                   w.total - w.ffn, s, nullptr, bias_f32);
    if (rc) return rc;
    if (rs_a) {
      // t5.py:550-551  hidden = hidden + proj(out), leaving the partial sums of squares of its rows
      const DenseFold f2{2, rs_one, nullptr, nullptr, ones, zeros, part};
      rc = dense_impl(base + w.h1, base + w.ctx, p->proj_w, nullptr, h, GNLP_COMBINE_ADD,
                      GNLP_ACT_IDENTITY, cfg->dtype, M, D, inner, inner, D, D, s, &f2);
      if (rc) return rc;
      launch_stats_finish(rs_b, part, fold_stat_slots(M, D), f.ln_eps, M, D, s, true);
      // t5.py:571  ffn: hidden + ffn_2(relu(ffn_1(RMSNorm(hidden))))
      void *hid = base + w.ffn;
      const DenseFold f1{1, rs_b, folded[i].ffn1_c, folded[i].ffn1_bf, nullptr, nullptr, nullptr};
      rc = dense_impl(hid, base + w.h1, folded[i].ffn1_wf, nullptr, nullptr, 0, f.act, cfg->dtype, M,
                      cfg->d_ff, D, D, cfg->d_ff, 0, s, &f1);
      if (rc) return rc;
      rc = dense_impl(dst, hid, p->ffn.ffn2_w, nullptr, base + w.h1, GNLP_COMBINE_ADD,
                      GNLP_ACT_IDENTITY, cfg->dtype, M, D, cfg->d_ff, cfg->d_ff, D, D, s, &f2);
      if (rc) return rc;
      if (!last) launch_stats_finish(rs_a, part, fold_stat_slots(M, D), cfg->ln_eps, M, D, s, true);
      h = dst;
      continue;
    }
    // t5.py:550-551  hidden = hidden + proj(out)
    rc = dense_impl(base + w.h1, base + w.ctx, p->proj_w, nullptr, h, GNLP_COMBINE_ADD,
                    GNLP_ACT_IDENTITY, cfg->dtype, M, D, inner, inner, D, D, s);
    if (rc) return rc;
    // t5.py:571  ffn (pre-norm RMS, residual inside)
    rc = ffn_impl(dst, base + w.h1, &p->ffn, &f, M, nullptr, 0, 0, 0, 0, base + w.ffn,
                  w.total - w.ffn, s);
    if (rc) return rc;
    h = dst;
  }
  if (final_ln_g) {  // t5.py:709
    launch_norm(out, h, nullptr, final_ln_g, final_ln_b, cfg->dtype, GNLP_NORM_RMS, cfg->ln_eps, M,
                D, nullptr, 0, 0, 0, 0, s);
  }
  return check_launch("gnlp_t5_encoder_forward");
}

int gnlp_t5_encoder_forward(void *out, const void *x, const gnlp_t5_block_params *layers,
                            int n_layers, const void *final_ln_g, const void *final_ln_b,
                            const gnlp_t5_cfg *cfg, int64_t B, int64_t L, const void *valid_length,
                            int vl_dtype, const void *position_bias, void *workspace,
                            size_t workspace_bytes, gnlp_stream_t stream) {
  return t5_encoder_impl(out, x, layers, nullptr, n_layers, final_ln_g, final_ln_b, cfg, B, L,
                         valid_length, vl_dtype, position_bias, workspace, workspace_bytes, stream);
}

int gnlp_t5_encoder_forward_folded(void *out, const void *x, const gnlp_t5_block_params *layers,
                                   const gnlp_t5_block_folded *folded, int n_layers,
                                   const void *final_ln_g, const void *final_ln_b,
                                   const gnlp_t5_cfg *cfg, int64_t B, int64_t L,
                                   const void *valid_length, int vl_dtype,
                                   const void *position_bias, void *workspace,
                                   size_t workspace_bytes, gnlp_stream_t stream) {
  return t5_encoder_impl(out, x, layers, folded, n_layers, final_ln_g, final_ln_b, cfg, B, L,
                         valid_length, vl_dtype, position_bias, workspace, workspace_bytes, stream);
}

/* ---- measurement hook (see include/gluonnlp_amd.h) ---- */
int gnlp_profile_dense_disarm(void) {
  g_prof.armed.store(0, std::memory_order_release);
  for (int i = 0; i < g_prof.cap; ++i) {
    if (g_prof.start) (void)hipEventDestroy(g_prof.start[i]);
    if (g_prof.stop) (void)hipEventDestroy(g_prof.stop[i]);
  }
  delete[] g_prof.start;
  delete[] g_prof.stop;
  g_prof.start = g_prof.stop = nullptr;
  g_prof.cap = 0;
  g_prof.n.store(0);
  return GNLP_OK;
}

int gnlp_profile_dense_arm(int64_t M, int64_t N, int64_t K, int capacity) {
  REQUIRE(capacity >= 1 && capacity <= 65536, "capacity=%d out of range [1, 65536]", capacity);
  gnlp_profile_dense_disarm();
  g_prof.start = new hipEvent_t[capacity];
  g_prof.stop = new hipEvent_t[capacity];
  for (int i = 0; i < capacity; ++i) {
    if (hipEventCreate(&g_prof.start[i]) != hipSuccess || hipEventCreate(&g_prof.stop[i]) != hipSuccess) {
      g_prof.cap = i;
      gnlp_profile_dense_disarm();
      return fail(GNLP_ERR_LAUNCH, "hipEventCreate failed");
    }
  }
  g_prof.cap = capacity;
  g_prof.M = M; g_prof.N = N; g_prof.K = K;
  g_prof.n.store(0);
  g_prof.armed.store(1, std::memory_order_release);
  return GNLP_OK;
}

int gnlp_profile_dense_read(float *ms, int capacity) {
  int n = g_prof.n.load();
  if (n > g_prof.cap) n = g_prof.cap;
  if (n > capacity) n = capacity;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.stop[i]) != hipSuccess ||
        hipEventElapsedTime(&ms[i], g_prof.start[i], g_prof.stop[i]) != hipSuccess)
      return fail(GNLP_ERR_LAUNCH, "reading profile event %d failed", i);
  }
  return n;
}

}  // extern "C"
