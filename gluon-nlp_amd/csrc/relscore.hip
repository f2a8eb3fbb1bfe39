// Query-dependent relative-position scores and the sinusoidal position table.
// Reference: RelAttentionScoreCell.forward, methods 'transformer_xl' and 'shaw'
// (attention_cell.py:816-875), SinusoidalPositionalEmbedding (layers.py:405-450).
//
//   out[b, h, i, j] = < query[b, h, i, :], E[idx(rel[i, j]), h, :] >
//   idx(r) = clip(r, lo, hi) - base, negative indices wrapped by R when the table is read with
//   np.take(mode='wrap') (the bidirectional 'shaw' table, attention_cell.py:768-775).
//
// The reference forms scores against the unique relative positions and gathers (L_q, L_m) of them
// per query row (np.unique + gather_nd, :827-875); the value of an element depends only on
// rel[i, j], so the kernel evaluates the dot product at the gathered index directly: no
// (L_q, #unique, B, H) intermediate, no host round trip for np.unique.
#include "common.h"
#include "kernels.h"

namespace gnlp {

// One workgroup per (b, h, i): the query row is staged in LDS as fp32, each thread owns keys
// j = t, t + 256, ...; stores are coalesced along j; E (R x H x C) stays in L2.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
rel_score_kernel(T *__restrict__ out, const T *__restrict__ q, const T *__restrict__ emb,
                 const int32_t *__restrict__ rel, RelScoreArgs a) {
  extern __shared__ float qrow[];
  const int64_t row = blockIdx.x;           // (b * H + h) * Lq + i
  const int64_t i = row % a.Lq;
  const int64_t bh = row / a.Lq;
  const int64_t h = bh % a.H, b = bh / a.H;
  const T *qp = q + b * a.qs[0] + h * a.qs[1] + i * a.qs[2];
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) qrow[c] = to_f32(qp[c]);
  __syncthreads();
  const int32_t *rrow = rel + i * a.Lm;
  T *orow = out + row * a.Lm;
  constexpr int N = VecOf<T>::N;
  typedef typename VecOf<T>::type vec_t;
  for (int64_t j = threadIdx.x; j < a.Lm; j += blockDim.x) {
    int r = rrow[j];
    r = r < a.lo ? a.lo : (r > a.hi ? a.hi : r);
    int64_t idx = (int64_t)r - a.base;
    if (a.wrap && idx < 0) idx += a.R;
    idx = idx < 0 ? 0 : (idx >= a.R ? a.R - 1 : idx);   // never out of the table
    const T *e = emb + (idx * a.H + h) * a.C;
    float acc = 0.0f;
    if (VEC) {
      for (int c = 0; c < a.C; c += N) {
        const vec_t v = *reinterpret_cast<const vec_t *>(e + c);
#pragma unroll
        for (int t = 0; t < N; ++t) acc = fmaf(qrow[c + t], to_f32(v[t]), acc);
      }
    } else {
      for (int c = 0; c < a.C; ++c) acc = fmaf(qrow[c], to_f32(e[c]), acc);
    }
    orow[j] = from_f32<T>(acc);
  }
}

void launch_rel_score(void *out, const void *q, const void *emb, const int32_t *rel, int dtype,
                      const RelScoreArgs &a, hipStream_t s) {
  const int64_t rows = a.B * a.H * a.Lq;
  if (rows == 0 || a.Lm == 0) return;
  const size_t lds = (size_t)a.C * sizeof(float);
  dim3 grid((unsigned)rows);
  if (dtype == GNLP_BF16) {
    const bool vec = a.C % 8 == 0 && (uintptr_t)emb % 16 == 0;
    if (vec) rel_score_kernel<bf16, true><<<grid, 256, lds, s>>>((bf16 *)out, (const bf16 *)q,
                                                                 (const bf16 *)emb, rel, a);
    else rel_score_kernel<bf16, false><<<grid, 256, lds, s>>>((bf16 *)out, (const bf16 *)q,
                                                              (const bf16 *)emb, rel, a);
  } else {
    const bool vec = a.C % 4 == 0 && (uintptr_t)emb % 16 == 0;
    if (vec) rel_score_kernel<float, true><<<grid, 256, lds, s>>>((float *)out, (const float *)q,
                                                                  (const float *)emb, rel, a);
    else rel_score_kernel<float, false><<<grid, 256, lds, s>>>((float *)out, (const float *)q,
                                                               (const float *)emb, rel, a);
  }
}

// Backward of rel_score_kernel (the gradients tests/test_attention_cell.py:313-391 takes through
// RelAttentionScoreCell: query and the cell's parameters):
//   dq[b, h, i, :]       = sum_j g[b, h, i, j] E[idx(i, j), h, :]
//   dE[idx(i, j), h, :] += g[b, h, i, j] q[b, h, i, :]          (fp32 atomics, zero-filled buffer)
// One workgroup per (b, h, i); a thread owns a contiguous run of keys and merges neighbours that
// map to the same table row (clipped ranges collapse to one flush) before touching LDS / memory.
template <typename T>
__global__ void __launch_bounds__(256)
rel_score_bwd_kernel(T *__restrict__ dq, float *__restrict__ demb, const T *__restrict__ g,
                     const T *__restrict__ q, const T *__restrict__ emb,
                     const int32_t *__restrict__ rel, RelScoreArgs a, int64_t dqs0, int64_t dqs1,
                     int64_t dqs2) {
  extern __shared__ float sh_rs[];   // qrow[C], dqrow[C]
  float *qrow = sh_rs, *dqrow = sh_rs + a.C;
  const int64_t row = blockIdx.x;
  const int64_t i = row % a.Lq;
  const int64_t bh = row / a.Lq;
  const int64_t h = bh % a.H, b = bh / a.H;
  const T *qp = q + b * a.qs[0] + h * a.qs[1] + i * a.qs[2];
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    qrow[c] = to_f32(qp[c]);
    dqrow[c] = 0.0f;
  }
  __syncthreads();
  const int32_t *rrow = rel + i * a.Lm;
  const T *grow = g + row * a.Lm;
  const int64_t chunk = (a.Lm + blockDim.x - 1) / blockDim.x;
  const int64_t j0 = threadIdx.x * chunk, j1 = j0 + chunk < a.Lm ? j0 + chunk : a.Lm;
  int64_t cur = -1;
  float gs = 0.0f;
  auto flush = [&]() {
    if (cur < 0 || gs == 0.0f) return;
    const T *e = emb + (cur * a.H + h) * a.C;
    float *de = demb ? demb + (cur * a.H + h) * a.C : nullptr;
    for (int c = 0; c < a.C; ++c) {
      atomicAdd(&dqrow[c], gs * to_f32(e[c]));
      if (de) atomicAdd(&de[c], gs * qrow[c]);
    }
  };
  for (int64_t j = j0; j < j1; ++j) {
    int r = rrow[j];
    r = r < a.lo ? a.lo : (r > a.hi ? a.hi : r);
    int64_t idx = (int64_t)r - a.base;
    if (a.wrap && idx < 0) idx += a.R;
    idx = idx < 0 ? 0 : (idx >= a.R ? a.R - 1 : idx);
    if (idx != cur) {
      flush();
      cur = idx;
      gs = 0.0f;
    }
    gs += to_f32(grow[j]);
  }
  flush();
  __syncthreads();
  if (dq) {
    T *dp = dq + b * dqs0 + h * dqs1 + i * dqs2;
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) dp[c] = from_f32<T>(dqrow[c]);
  }
}

void launch_rel_score_bwd(void *dq, const int64_t dq_stride[3], float *demb, const void *g,
                          const void *q, const void *emb, const int32_t *rel, int dtype,
                          const RelScoreArgs &a, hipStream_t s) {
  const int64_t rows = a.B * a.H * a.Lq;
  if (rows == 0) return;
  const size_t lds = (size_t)2 * a.C * sizeof(float);
  dim3 grid((unsigned)rows);
  const int64_t d0 = dq ? dq_stride[0] : 0, d1 = dq ? dq_stride[1] : 0, d2 = dq ? dq_stride[2] : 0;
  if (dtype == GNLP_BF16)
    rel_score_bwd_kernel<bf16><<<grid, 256, lds, s>>>((bf16 *)dq, demb, (const bf16 *)g,
                                                      (const bf16 *)q, (const bf16 *)emb, rel, a,
                                                      d0, d1, d2);
  else
    rel_score_bwd_kernel<float><<<grid, 256, lds, s>>>((float *)dq, demb, (const float *)g,
                                                       (const float *)q, (const float *)emb, rel,
                                                       a, d0, d1, d2);
}

// layers.py:442-450: emb = positions[:, None] * base_mult; concat(sin(emb), cos(emb) [, 0]).
template <typename T>
__global__ void __launch_bounds__(256)
sinusoid_kernel(T *__restrict__ out, const float *__restrict__ pos, const float *__restrict__ base,
                int64_t n, int64_t units) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * units) return;
  const int64_t p = idx / units, u = idx % units;
  const int64_t half = units / 2;
  float v = 0.0f;
  if (u < 2 * half) {
    const float arg = pos[p] * base[u < half ? u : u - half];
    v = u < half ? sinf(arg) : cosf(arg);
  }
  out[idx] = from_f32<T>(v);
}

void launch_sinusoid(void *out, const float *pos, const float *base, int dtype, int64_t n,
                     int64_t units, hipStream_t s) {
  if (n * units == 0) return;
  dim3 grid((unsigned)((n * units + 255) / 256));
  if (dtype == GNLP_BF16)
    sinusoid_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)out, pos, base, n, units);
  else
    sinusoid_kernel<float><<<grid, 256, 0, s>>>((float *)out, pos, base, n, units);
}

}  // namespace gnlp
