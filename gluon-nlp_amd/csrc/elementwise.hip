// HBM-bound integer / elementwise kernels of the encoder path (gfx950):
// mask generation, sequence_mask, activations, l2_normalize, T5 relative-position buckets.
// All are coalesced 16-byte-per-lane streaming kernels; none uses LDS.
#include "common.h"
#include "kernels.h"

namespace gnlp {

// ------------------------------------------------------------------------------------------
// gen_self_attn_mask / gen_mem_attn_mask   (attention_cell.py:40-183, 186-274)
// One thread writes 16 consecutive mask bytes of the flat (B, Lq, Lm) array.
// ------------------------------------------------------------------------------------------
template <int MODE>  // 0 = self/full, 1 = self/causal, 2 = mem
__global__ void __launch_bounds__(256)
mask_kernel(uint8_t *__restrict__ mask, const void *__restrict__ vl_k, const void *__restrict__ vl_q,
            int vl_dtype, int64_t B, int64_t Lq, int64_t Lm) {
  const int64_t total = B * Lq * Lm;
  int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (base >= total) return;
  int64_t j = base % Lm;
  int64_t r = base / Lm;
  int64_t i = r % Lq;
  int64_t b = r / Lq;
  int64_t nk = vl_k ? load_len(vl_k, vl_dtype, b) : Lm;
  int64_t nq = vl_q ? load_len(vl_q, vl_dtype, b) : Lq;
  uint32_t w[4] = {0, 0, 0, 0};
  const int n = (int)((total - base) < 16 ? (total - base) : 16);
  for (int e = 0; e < n; ++e) {
    bool on;
    if (MODE == 1) on = (j <= i) && (i < nq);
    else on = (j < nk) && (i < nq);
    w[e >> 2] |= (on ? 1u : 0u) << (8 * (e & 3));
    if (++j == Lm) {
      j = 0;
      if (++i == Lq) {
        i = 0;
        ++b;
        if (b < B) {
          nk = vl_k ? load_len(vl_k, vl_dtype, b) : Lm;
          nq = vl_q ? load_len(vl_q, vl_dtype, b) : Lq;
        }
      }
    }
  }
  if (n == 16) {
    *reinterpret_cast<uint4 *>(mask + base) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    for (int e = 0; e < n; ++e) mask[base + e] = (uint8_t)((w[e >> 2] >> (8 * (e & 3))) & 0xff);
  }
}

void launch_self_mask(uint8_t *mask, const void *vl, int vl_dtype, int64_t B, int64_t L,
                      int attn_type, hipStream_t s) {
  const int64_t total = B * L * L;
  if (total == 0) return;
  const int64_t threads = (total + 15) / 16;
  dim3 grid((unsigned)((threads + 255) / 256));
  if (attn_type == GNLP_ATTN_CAUSAL)
    mask_kernel<1><<<grid, 256, 0, s>>>(mask, nullptr, vl, vl_dtype, B, L, L);
  else
    mask_kernel<0><<<grid, 256, 0, s>>>(mask, vl, vl, vl_dtype, B, L, L);
}

void launch_mem_mask(uint8_t *mask, const void *mem_vl, const void *data_vl, int vl_dtype,
                     int64_t B, int64_t Lq, int64_t Lm, hipStream_t s) {
  const int64_t total = B * Lq * Lm;
  if (total == 0) return;
  const int64_t threads = (total + 15) / 16;
  dim3 grid((unsigned)((threads + 255) / 256));
  mask_kernel<2><<<grid, 256, 0, s>>>(mask, mem_vl, data_vl, vl_dtype, B, Lq, Lm);
}

// ------------------------------------------------------------------------------------------
// sequence_mask   (npx.sequence_mask at models/bert.py:311-312)
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
seqmask_kernel(T *__restrict__ out, const T *__restrict__ x, const void *__restrict__ vl,
               int vl_dtype, int64_t B, int64_t L, int64_t C, int time_major) {
  constexpr int N = VEC ? VecOf<T>::N : 1;
  const int64_t total = B * L * C;
  int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * N;
  if (e0 >= total) return;
  const int64_t row = e0 / C;
  const int64_t b = time_major ? row % B : row / L;
  const int64_t t = time_major ? row / B : row % L;
  const bool keep = t < load_len(vl, vl_dtype, b);
  if (VEC) {
    typedef typename VecOf<T>::type V;
    V v = *reinterpret_cast<const V *>(x + e0);
    if (!keep) {
#pragma unroll
      for (int q = 0; q < N; ++q) v[q] = (T)0.0f;
    }
    *reinterpret_cast<V *>(out + e0) = v;
  } else {
    out[e0] = keep ? x[e0] : (T)0.0f;
  }
}

template <typename T>
static void seqmask_t(void *out, const void *x, const void *vl, int vl_dtype, int64_t B, int64_t L,
                      int64_t C, int time_major, hipStream_t s) {
  const int64_t total = B * L * C;
  if (total == 0) return;
  const bool vec = (C % VecOf<T>::N == 0) && (((uintptr_t)out | (uintptr_t)x) % 16 == 0);
  if (vec) {
    int64_t threads = total / VecOf<T>::N;
    seqmask_kernel<T, true><<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(
        (T *)out, (const T *)x, vl, vl_dtype, B, L, C, time_major);
  } else {
    seqmask_kernel<T, false><<<dim3((unsigned)((total + 255) / 256)), 256, 0, s>>>(
        (T *)out, (const T *)x, vl, vl_dtype, B, L, C, time_major);
  }
}

void launch_sequence_mask(void *out, const void *x, const void *vl, int vl_dtype, int dtype,
                          int64_t B, int64_t L, int64_t C, int time_major, hipStream_t s) {
  if (dtype == GNLP_BF16) seqmask_t<bf16>(out, x, vl, vl_dtype, B, L, C, time_major, s);
  else seqmask_t<float>(out, x, vl, vl_dtype, B, L, C, time_major, s);
}

// ------------------------------------------------------------------------------------------
// ragged batches packed to their valid tokens (no counterpart in the reference, which computes
// every dense op over the padding): the stack drivers gather the rows t < valid_length[b] of a
// (B, L, C) / (L, B, C) activation into one (sum vl, C) matrix, run the layers on it -- the GEMMs
// and norms read the row count from device memory, the attention kernel the sequence offsets --
// and scatter the result back with exact zeros beyond valid_length (the npx.sequence_mask of
// models/bert.py:311-312).  HBM-bound: one read + one write of the valid rows each way.
// ------------------------------------------------------------------------------------------
// cu[0..B] = exclusive prefix sum of clamp(vl, 0, L);  cu[B + 1] = rows the dense kernels process:
// cu[B] + 64 (rows a K / V tile of the last sequence may touch) rounded up to 256, at most B * L.
__global__ void __launch_bounds__(256)
seq_offsets_kernel(int32_t *__restrict__ cu, const void *__restrict__ vl, int vl_dtype, int64_t B,
                   int64_t L) {
  __shared__ int64_t part[257];
  const int tid = threadIdx.x;
  const int64_t per = (B + 255) / 256;
  const int64_t b0 = tid * per < B ? tid * per : B, b1 = (b0 + per < B) ? b0 + per : B;
  auto len = [&](int64_t b) {
    const int64_t n = load_len(vl, vl_dtype, b);
    return n < 0 ? 0 : (n > L ? L : n);
  };
  int64_t sum = 0;
  for (int64_t b = b0; b < b1; ++b) sum += len(b);
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int64_t run = 0;
    for (int i = 0; i < 256; ++i) {
      const int64_t v = part[i];
      part[i] = run;
      run += v;
    }
    part[256] = run;
  }
  __syncthreads();
  int64_t run = part[tid];
  for (int64_t b = b0; b < b1; ++b) {
    cu[b] = (int32_t)run;
    run += len(b);
  }
  if (tid == 0) {
    const int64_t tot = part[256];
    cu[B] = (int32_t)tot;
    const int64_t m = (tot + 64 + 255) / 256 * 256, cap = (B * L + 255) / 256 * 256;
    cu[B + 1] = (int32_t)(m < cap ? m : cap);
  }
}

// packed[r] = x[row(b, t)] for r = cu[b] + t < cu[B]; zeros for cu[B] <= r < cu[B + 1]
template <typename T>
__global__ void __launch_bounds__(256)
pack_rows_kernel(T *__restrict__ packed, const T *__restrict__ x, const int32_t *__restrict__ cu,
                 int64_t B, int64_t L, int C, int time_major) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= cu[B + 1]) return;
  T *dst = packed + r * C;
  if (r >= cu[B]) {
    V z;
#pragma unroll
    for (int q = 0; q < N; ++q) z[q] = (T)0.0f;
    for (int c = lane * N; c < C; c += 64 * N) *reinterpret_cast<V *>(dst + c) = z;
    return;
  }
  int64_t lo = 0, hi = B;   // largest b with cu[b] <= r (wave-uniform)
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (cu[mid] <= r) lo = mid; else hi = mid;
  }
  const int64_t t = r - cu[lo];
  const T *src = x + (time_major ? t * B + lo : lo * L + t) * C;
  for (int c = lane * N; c < C; c += 64 * N)
    *reinterpret_cast<V *>(dst + c) = *reinterpret_cast<const V *>(src + c);
}

// out[row(b, t)] = t < vl[b] ? packed[cu[b] + t] : 0
template <typename T>
__global__ void __launch_bounds__(256)
unpack_rows_kernel(T *__restrict__ out, const T *__restrict__ packed,
                   const int32_t *__restrict__ cu, int64_t B, int64_t L, int C, int time_major) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * L) return;
  const int64_t b = time_major ? row % B : row / L;
  const int64_t t = time_major ? row / B : row % L;
  T *dst = out + row * C;
  if (t < cu[b + 1] - cu[b]) {
    const T *src = packed + ((int64_t)cu[b] + t) * C;
    for (int c = lane * N; c < C; c += 64 * N)
      *reinterpret_cast<V *>(dst + c) = *reinterpret_cast<const V *>(src + c);
  } else {
    V z;
#pragma unroll
    for (int q = 0; q < N; ++q) z[q] = (T)0.0f;
    for (int c = lane * N; c < C; c += 64 * N) *reinterpret_cast<V *>(dst + c) = z;
  }
}

// zero the rows cu[B] .. cu[B + 1] - 1 (fewer than 320) of a packed (rows, C) buffer that no kernel
// writes -- the attention context: no sequence owns them, the GEMM after it reads them
template <typename T>
__global__ void __launch_bounds__(256)
zero_pad_rows_kernel(T *__restrict__ buf, const int32_t *__restrict__ cu, int64_t B, int C) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)cu[B] + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= cu[B + 1]) return;
  V z;
#pragma unroll
  for (int q = 0; q < N; ++q) z[q] = (T)0.0f;
  for (int c = lane * N; c < C; c += 64 * N) *reinterpret_cast<V *>(buf + r * C + c) = z;
}

void launch_zero_pad_rows(void *buf, const int32_t *cu, int dtype, int64_t B, int64_t C,
                          hipStream_t s) {
  if (dtype == GNLP_BF16)
    zero_pad_rows_kernel<bf16><<<80, 256, 0, s>>>((bf16 *)buf, cu, B, (int)C);
  else
    zero_pad_rows_kernel<float><<<80, 256, 0, s>>>((float *)buf, cu, B, (int)C);
}

void launch_seq_offsets(int32_t *cu, const void *vl, int vl_dtype, int64_t B, int64_t L,
                        hipStream_t s) {
  seq_offsets_kernel<<<1, 256, 0, s>>>(cu, vl, vl_dtype, B, L);
}

// C a multiple of the 16-byte vector, tensors 16-byte aligned (checked by the caller)
void launch_pack_rows(void *packed, const void *x, const int32_t *cu, int dtype, int64_t B,
                      int64_t L, int64_t C, int time_major, hipStream_t s) {
  const unsigned grid = (unsigned)((B * L + 255) / 256 * 256 / 4);
  if (dtype == GNLP_BF16)
    pack_rows_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)packed, (const bf16 *)x, cu, B, L, (int)C,
                                                time_major);
  else
    pack_rows_kernel<float><<<grid, 256, 0, s>>>((float *)packed, (const float *)x, cu, B, L,
                                                 (int)C, time_major);
}

void launch_unpack_rows(void *out, const void *packed, const int32_t *cu, int dtype, int64_t B,
                        int64_t L, int64_t C, int time_major, hipStream_t s) {
  const unsigned grid = (unsigned)((B * L + 3) / 4);
  if (dtype == GNLP_BF16)
    unpack_rows_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)out, (const bf16 *)packed, cu, B, L,
                                                  (int)C, time_major);
  else
    unpack_rows_kernel<float><<<grid, 256, 0, s>>>((float *)out, (const float *)packed, cu, B, L,
                                                   (int)C, time_major);
}

// ------------------------------------------------------------------------------------------
// activations   (layers.py:236-338)
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
act_kernel(T *__restrict__ out, const T *__restrict__ x, int64_t n, int act) {
  constexpr int N = VEC ? VecOf<T>::N : 1;
  int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * N;
  if (e0 >= n) return;
  if (VEC) {
    typedef typename VecOf<T>::type V;
    V v = *reinterpret_cast<const V *>(x + e0);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = from_f32<T>(apply_act<true>(to_f32(v[q]), act));
    *reinterpret_cast<V *>(out + e0) = v;
  } else {
    out[e0] = from_f32<T>(apply_act<true>(to_f32(x[e0]), act));
  }
}

template <typename T>
static void act_t(void *out, const void *x, int64_t n, int act, hipStream_t s) {
  if (n == 0) return;
  const bool vec = (n % VecOf<T>::N == 0) && (((uintptr_t)out | (uintptr_t)x) % 16 == 0);
  if (vec) {
    int64_t threads = n / VecOf<T>::N;
    act_kernel<T, true><<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>((T *)out,
                                                                                 (const T *)x, n, act);
  } else {
    act_kernel<T, false><<<dim3((unsigned)((n + 255) / 256)), 256, 0, s>>>((T *)out, (const T *)x,
                                                                            n, act);
  }
}

// The epilogue of Dense as an operation of its own, and its backward (training: the fused GEMM
// epilogues keep no pre-activation; under autograd Dense saves z = x W^T + b and applies these).
//   forward:   y = combine(act(z), other)
//   backward:  dz = g act'(z) [* other],   d_other = g act(z)   (combine mul; for add d_other = g)
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
dense_epilogue_kernel(T *__restrict__ o0, T *__restrict__ o1, const T *__restrict__ g,
                      const T *__restrict__ z, const T *__restrict__ other, int64_t n, int act,
                      int mul) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float zv = to_f32(z[i]);
  const float ov = other ? to_f32(other[i]) : 0.0f;
  if (!BWD) {
    const float a = apply_act<true>(zv, act);
    o0[i] = from_f32<T>(other ? (mul ? a * ov : a + ov) : a);
  } else {
    const float gv = to_f32(g[i]);
    const float d = apply_dact(zv, act);
    o0[i] = from_f32<T>(mul ? gv * ov * d : gv * d);
    if (o1) o1[i] = from_f32<T>(gv * apply_act<true>(zv, act));
  }
}

void launch_dense_epilogue(void *y, const void *z, const void *other, int dtype, int64_t n, int act,
                           int mul, hipStream_t s) {
  if (n == 0) return;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == GNLP_BF16)
    dense_epilogue_kernel<bf16, false><<<grid, 256, 0, s>>>((bf16 *)y, nullptr, nullptr,
                                                            (const bf16 *)z, (const bf16 *)other, n,
                                                            act, mul);
  else
    dense_epilogue_kernel<float, false><<<grid, 256, 0, s>>>((float *)y, nullptr, nullptr,
                                                             (const float *)z, (const float *)other,
                                                             n, act, mul);
}

void launch_dense_epilogue_bwd(void *dz, void *d_other, const void *g, const void *z,
                               const void *other, int dtype, int64_t n, int act, int mul,
                               hipStream_t s) {
  if (n == 0) return;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == GNLP_BF16)
    dense_epilogue_kernel<bf16, true><<<grid, 256, 0, s>>>((bf16 *)dz, (bf16 *)d_other,
                                                           (const bf16 *)g, (const bf16 *)z,
                                                           (const bf16 *)other, n, act, mul);
  else
    dense_epilogue_kernel<float, true><<<grid, 256, 0, s>>>((float *)dz, (float *)d_other,
                                                            (const float *)g, (const float *)z,
                                                            (const float *)other, n, act, mul);
}

void launch_activation(void *out, const void *x, int dtype, int64_t n, int act, hipStream_t s) {
  if (dtype == GNLP_BF16) act_t<bf16>(out, x, n, act, s);
  else act_t<float>(out, x, n, act, s);
}

// ------------------------------------------------------------------------------------------
// l2_normalize   (op.py:289-307):  x / (||x||_2 + eps), one row per wave
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
l2norm_kernel(T *__restrict__ out, const T *__restrict__ x, int64_t rows, int64_t d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T *xr = x + row * d;
  float ss = 0.0f;
  for (int64_t c = lane; c < d; c += 64) {
    float v = to_f32(xr[c]);
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / (sqrtf(ss) + eps);
  T *orow = out + row * d;
  for (int64_t c = lane; c < d; c += 64) orow[c] = from_f32<T>(to_f32(xr[c]) * inv);
}

void launch_l2_normalize(void *out, const void *x, int dtype, int64_t rows, int64_t d, float eps,
                         hipStream_t s) {
  if (rows == 0 || d == 0) return;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == GNLP_BF16)
    l2norm_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)out, (const bf16 *)x, rows, d, eps);
  else
    l2norm_kernel<float><<<grid, 256, 0, s>>>((float *)out, (const float *)x, rows, d, eps);
}

// ------------------------------------------------------------------------------------------
// T5 relative positions   (attention_cell.py:611-656, op.py:229-286, layers.py:519-524)
//
// The bucket of a distance n >= max_exact is max_exact + trunc(logf(n/max_exact)/log(md/me)*(nb-me))
// in float32.  That map is monotone in n, so the host evaluates it (same libm logf as the
// reference's CPU path) only to find the integer thresholds where the bucket increments, and the
// device does pure integer compares: bit-exact by construction, no device transcendental.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int bucket_of(int rel, const BucketTable &tb) {
  int n = -rel;  // op.py:262
  int ret = 0;
  if (tb.bidirectional) {
    if (n < 0) ret = tb.half;  // op.py:267
    n = n < 0 ? -n : n;        // op.py:268
  } else {
    n = n < 0 ? 0 : n;         // op.py:271
  }
  int v;
  if (n < tb.max_exact) {
    v = n;
  } else {
    v = tb.max_exact;
    for (int q = 0; q < tb.n_thr; ++q) v += (n >= tb.thr[q]) ? 1 : 0;
  }
  return ret + v;
}

__global__ void __launch_bounds__(256)
rel_position_kernel(int32_t *__restrict__ out, int64_t L) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L * L) return;
  out[idx] = (int32_t)(idx % L) - (int32_t)(idx / L);  // mem_j - query_i
}

void launch_rel_position(int32_t *out, int64_t L, hipStream_t s) {
  if (L == 0) return;
  rel_position_kernel<<<dim3((unsigned)((L * L + 255) / 256)), 256, 0, s>>>(out, L);
}

__global__ void __launch_bounds__(256)
bucket_kernel(int32_t *__restrict__ out, const int32_t *__restrict__ rel, int64_t n,
              BucketTable tb) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  out[idx] = bucket_of(rel[idx], tb);
}

void launch_bucket(int32_t *out, const int32_t *rel, int64_t n, const BucketTable &tb,
                   hipStream_t s) {
  if (n == 0) return;
  bucket_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, s>>>(out, rel, n, tb);
}

template <typename T>
__global__ void __launch_bounds__(256)
bucket_embed_kernel(T *__restrict__ out, const T *__restrict__ weight,
                    const int32_t *__restrict__ rel, int64_t n, int64_t units, BucketTable tb) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * units) return;
  const int64_t i = idx / units, u = idx % units;
  out[idx] = weight[(int64_t)bucket_of(rel[i], tb) * units + u];
}

void launch_bucket_embedding(void *out, const void *weight, const int32_t *rel, int dtype,
                             int64_t n, int64_t units, const BucketTable &tb, hipStream_t s) {
  if (n * units == 0) return;
  dim3 grid((unsigned)((n * units + 255) / 256));
  if (dtype == GNLP_BF16)
    bucket_embed_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)out, (const bf16 *)weight, rel, n, units,
                                                   tb);
  else
    bucket_embed_kernel<float><<<grid, 256, 0, s>>>((float *)out, (const float *)weight, rel, n,
                                                    units, tb);
}

// backward of bucket_embed_kernel: dweight[bucket(rel[i]), u] += g[i, u]  (fp32 atomics)
template <typename T>
__global__ void __launch_bounds__(256)
bucket_embed_bwd_kernel(float *__restrict__ dweight, const T *__restrict__ g,
                        const int32_t *__restrict__ rel, int64_t n, int64_t units,
                        BucketTable tb) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * units) return;
  const int64_t i = idx / units, u = idx % units;
  atomicAdd(&dweight[(int64_t)bucket_of(rel[i], tb) * units + u], to_f32(g[idx]));
}

void launch_bucket_embedding_bwd(float *dweight, const void *g, const int32_t *rel, int dtype,
                                 int64_t n, int64_t units, const BucketTable &tb, hipStream_t s) {
  if (n * units == 0) return;
  dim3 grid((unsigned)((n * units + 255) / 256));
  if (dtype == GNLP_BF16)
    bucket_embed_bwd_kernel<bf16><<<grid, 256, 0, s>>>(dweight, (const bf16 *)g, rel, n, units, tb);
  else
    bucket_embed_bwd_kernel<float><<<grid, 256, 0, s>>>(dweight, (const float *)g, rel, n, units,
                                                        tb);
}

// out[h, i, j] = weight[bucket(j - i + rel0), h]; thread = one (h, i, j); j fastest => coalesced
// stores, the (num_buckets, H) table stays in L1/L2.  (rel0 != 0: the one-row Toeplitz table.)
template <typename T>
__global__ void __launch_bounds__(256)
t5_bias_kernel(T *__restrict__ out, const T *__restrict__ weight, int64_t H, int64_t Lq,
               int64_t Lm, int rel0, BucketTable tb) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * Lq * Lm) return;
  const int64_t j = idx % Lm;
  const int64_t r = idx / Lm;
  const int64_t i = r % Lq;
  const int64_t h = r / Lq;
  out[idx] = weight[(int64_t)bucket_of((int)(j - i) + rel0, tb) * H + h];
}

void launch_t5_rel_bias(void *out, const void *weight, int dtype, int64_t H, int64_t Lq,
                        int64_t Lm, int rel0, const BucketTable &tb, hipStream_t s) {
  const int64_t total = H * Lq * Lm;
  if (total == 0) return;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == GNLP_BF16)
    t5_bias_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)out, (const bf16 *)weight, H, Lq, Lm, rel0,
                                              tb);
  else
    t5_bias_kernel<float><<<grid, 256, 0, s>>>((float *)out, (const float *)weight, H, Lq, Lm, rel0,
                                               tb);
}

}  // namespace gnlp
