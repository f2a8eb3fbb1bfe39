// Fused (residual +) LayerNorm / RMSNorm / NoNorm (+ sequence_mask) over the last axis.
// Reference: mxnet nn.LayerNorm as used at models/transformer.py:217-218,268-270 and
// layers.py:63,642-646; RMSNorm layers.py:151-181; NoNorm layers.py:81-138;
// npx.sequence_mask models/bert.py:311-312.
//
// HBM-bound: one wave64 owns one row, 16-byte loads per lane, the row stays in registers between
// the statistics pass and the normalisation pass (read once, written once), statistics in fp32
// with the two-pass (mean, then centred variance) formulation.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace gnlp {

template <typename T, int MAXV, bool NT>
__global__ void __launch_bounds__(256)
norm_vec_kernel(T *__restrict__ out, const T *__restrict__ x, const T *__restrict__ residual,
                const T *__restrict__ gamma, const T *__restrict__ beta, int kind, float eps,
                int64_t rows, int C, const void *__restrict__ vl, int vl_dtype, int64_t B,
                int64_t L, int time_major, const int32_t *__restrict__ rows_dev) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (rows_dev && row >= *rows_dev) return;   // packed ragged batch: rows past the valid tokens
  const int nvec = C / N;  // vectors per row
  T *orow = out + row * (int64_t)C;
  if (vl) {
    const int64_t b = time_major ? row % B : row / L;
    const int64_t t = time_major ? row / B : row % L;
    if (t >= load_len(vl, vl_dtype, b)) {  // wave-uniform
      V z;
#pragma unroll
      for (int q = 0; q < N; ++q) z[q] = (T)0.0f;
      for (int v = lane; v < nvec; v += 64) *reinterpret_cast<V *>(orow + v * N) = z;
      return;
    }
  }
  const T *xr = x + row * (int64_t)C;
  const T *rr = residual ? residual + row * (int64_t)C : nullptr;
  float vals[MAXV][N];
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      V a = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(xr + v * N))
                 : *reinterpret_cast<const V *>(xr + v * N);
      if (rr) {
        V r = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(rr + v * N))
                   : *reinterpret_cast<const V *>(rr + v * N);
#pragma unroll
        for (int q = 0; q < N; ++q) vals[i][q] = to_f32(a[q]) + to_f32(r[q]);
      } else {
#pragma unroll
        for (int q = 0; q < N; ++q) vals[i][q] = to_f32(a[q]);
      }
#pragma unroll
      for (int q = 0; q < N; ++q) sum += vals[i][q];
    } else {
#pragma unroll
      for (int q = 0; q < N; ++q) vals[i][q] = 0.0f;
    }
  }
  float mean = 0.0f, rstd = 1.0f;
  if (kind == GNLP_NORM_LAYER) {
    mean = wave_sum(sum) / (float)C;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + i * 64 < nvec) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const float d = vals[i][q] - mean;
          ss += d * d;
        }
      }
    }
    rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
  } else if (kind == GNLP_NORM_RMS) {
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
#pragma unroll
      for (int q = 0; q < N; ++q) ss += vals[i][q] * vals[i][q];
    }
    rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      V g = *reinterpret_cast<const V *>(gamma + v * N);
      V o;
      if (beta) {
        V bt = *reinterpret_cast<const V *>(beta + v * N);
#pragma unroll
        for (int q = 0; q < N; ++q)
          o[q] = from_f32<T>((vals[i][q] - mean) * rstd * to_f32(g[q]) + to_f32(bt[q]));
      } else {
#pragma unroll
        for (int q = 0; q < N; ++q) o[q] = from_f32<T>((vals[i][q] - mean) * rstd * to_f32(g[q]));
      }
      *reinterpret_cast<V *>(orow + v * N) = o;
    }
  }
}

// Any C / alignment: scalar loads, three passes over the (L2-resident) row.
template <typename T>
__global__ void __launch_bounds__(256)
norm_scalar_kernel(T *__restrict__ out, const T *__restrict__ x, const T *__restrict__ residual,
                   const T *__restrict__ gamma, const T *__restrict__ beta, int kind, float eps,
                   int64_t rows, int64_t C, const void *__restrict__ vl, int vl_dtype, int64_t B,
                   int64_t L, int time_major, const int32_t *__restrict__ rows_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (rows_dev && row >= *rows_dev) return;
  T *orow = out + row * C;
  if (vl) {
    const int64_t b = time_major ? row % B : row / L;
    const int64_t t = time_major ? row / B : row % L;
    if (t >= load_len(vl, vl_dtype, b)) {
      for (int64_t c = lane; c < C; c += 64) orow[c] = (T)0.0f;
      return;
    }
  }
  const T *xr = x + row * C;
  const T *rr = residual ? residual + row * C : nullptr;
  float mean = 0.0f, rstd = 1.0f;
  if (kind == GNLP_NORM_LAYER) {
    float s = 0.0f;
    for (int64_t c = lane; c < C; c += 64) s += to_f32(xr[c]) + (rr ? to_f32(rr[c]) : 0.0f);
    mean = wave_sum(s) / (float)C;
    float ss = 0.0f;
    for (int64_t c = lane; c < C; c += 64) {
      const float d = to_f32(xr[c]) + (rr ? to_f32(rr[c]) : 0.0f) - mean;
      ss += d * d;
    }
    rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
  } else if (kind == GNLP_NORM_RMS) {
    float ss = 0.0f;
    for (int64_t c = lane; c < C; c += 64) {
      const float d = to_f32(xr[c]) + (rr ? to_f32(rr[c]) : 0.0f);
      ss += d * d;
    }
    rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
  }
  for (int64_t c = lane; c < C; c += 64) {
    const float v = to_f32(xr[c]) + (rr ? to_f32(rr[c]) : 0.0f);
    orow[c] = from_f32<T>((v - mean) * rstd * to_f32(gamma[c]) + (beta ? to_f32(beta[c]) : 0.0f));
  }
}

template <typename T>
static void norm_t(void *out, const void *x, const void *residual, const void *gamma,
                   const void *beta, int kind, float eps, int64_t rows, int64_t C, const void *vl,
                   int vl_dtype, int64_t B, int64_t L, int time_major, hipStream_t s) {
  if (rows == 0 || C == 0) return;
  constexpr int N = VecOf<T>::N;
  constexpr int MAXV = 32 / N;  // 32 fp32 values per lane => C <= 2048
  dim3 grid((unsigned)((rows + 3) / 4));
  uintptr_t al = (uintptr_t)out | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)(beta ? beta : gamma) |
                 (uintptr_t)(residual ? residual : x);
  const bool vec = (C % N == 0) && (C <= 64 * N * MAXV) && (al % 16 == 0);
  if (vec) {
    // x / residual are read once here and never again: non-temporal loads keep them from
    // displacing what the next GEMM re-reads (+0.45 % on the BERT-large forward, 4 of 4 A / B runs)
    const bool nt = knobs().norm_nt;
    const bool small = !knobs().norm_full && C <= 64 * N * (MAXV / 2);
#define GNLP_NORM_LAUNCH(MV, NTV)                                                                \
  norm_vec_kernel<T, MV, NTV><<<grid, 256, 0, s>>>((T *)out, (const T *)x, (const T *)residual,  \
                                                   (const T *)gamma, (const T *)beta, kind, eps, \
                                                   rows, (int)C, vl, vl_dtype, B, L, time_major, \
                                                   dense_rows_dev())
    if (small && nt) GNLP_NORM_LAUNCH(MAXV / 2, true);
    else if (small) GNLP_NORM_LAUNCH(MAXV / 2, false);
    else if (nt) GNLP_NORM_LAUNCH(MAXV, true);
    else GNLP_NORM_LAUNCH(MAXV, false);
#undef GNLP_NORM_LAUNCH
  } else {
    norm_scalar_kernel<T><<<grid, 256, 0, s>>>((T *)out, (const T *)x, (const T *)residual,
                                               (const T *)gamma, (const T *)beta, kind, eps, rows,
                                               C, vl, vl_dtype, B, L, time_major, dense_rows_dev());
  }
}

void launch_norm(void *out, const void *x, const void *residual, const void *gamma,
                 const void *beta, int dtype, int kind, float eps, int64_t rows, int64_t C,
                 const void *vl, int vl_dtype, int64_t B, int64_t L, int time_major,
                 hipStream_t s) {
  if (dtype == GNLP_BF16)
    norm_t<bf16>(out, x, residual, gamma, beta, kind, eps, rows, C, vl, vl_dtype, B, L,
                 time_major, s);
  else
    norm_t<float>(out, x, residual, gamma, beta, kind, eps, rows, C, vl, vl_dtype, B, L,
                  time_major, s);
}

// ---- LayerNorm folded into the GEMMs around it (bf16 post-norm stacks) ----
// Row statistics of the raw pre-norm sums: r = rsqrt(var + eps) and r * mean per row, fp32, with
// the same two-pass formulation as norm_vec_kernel (the row stays in registers).  Read-only over
// the rows: 2 bytes per element in, 8 bytes per row out.
template <int MAXV>
__global__ void __launch_bounds__(256)
row_stats_kernel(float *__restrict__ rstats, const bf16 *__restrict__ x, float eps, int64_t rows,
                 int C, const int32_t *__restrict__ rows_dev, int rms) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (rows_dev && row >= *rows_dev) return;
  const int nvec = C / 8;
  const bf16 *xr = x + row * (int64_t)C;
  float vals[MAXV][8];
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(xr + v * 8);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        vals[i][q] = (float)a[q];
        sum += vals[i][q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) vals[i][q] = 0.0f;
    }
  }
  // (rms: RMSNorm has no mean term -- r = rsqrt(mean(x^2) + eps), r * mean := 0)
  const float mean = rms ? 0.0f : wave_sum(sum) / (float)C;
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (lane + i * 64 < nvec) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float d = vals[i][q] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
  if (lane == 0) *reinterpret_cast<f32x2 *>(rstats + row * 2) = (f32x2){rstd, rstd * mean};
}

// The same statistics from the partial sums the producing GEMM's drain left (FOLD 2 of
// gemm_bf16_w4.hip): (sum, sum of squares) per row and 64-column group, added in slot order.
// var = E[x^2] - mean^2 in fp32 (relative error ~1e-7 E[x^2] / var: 1e-5 at |mean| = 10 sigma).
__global__ void __launch_bounds__(256)
stats_finish_kernel(float *__restrict__ rstats, const float *__restrict__ part, int slots,
                    float eps, int64_t rows, float inv_c, const int32_t *__restrict__ rows_dev,
                    int rms) {
  // 8 lanes per row, each adding the slots 2 j, 2 j + 1 (+ 16, ...) it loads as one 16-byte piece
  // (a row's 16 slots are one 128-byte line: coalesced), then the 8 lanes in a fixed order
  const int lane = threadIdx.x & 63, j = lane & 7;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
  const bool live = row < rows && !(rows_dev && row >= *rows_dev);
  float s = 0.0f, q = 0.0f;
  if (live) {
    const float *p = part + row * slots * 2;
    for (int i = 2 * j; i < slots; i += 16) {
      if (i + 1 < slots) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + i * 2);
        s += v[0]; q += v[1];
        s += v[2]; q += v[3];
      } else {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(p + i * 2);
        s += v[0]; q += v[1];
      }
    }
  }
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) {   // (xor butterfly: every lane ends with the same sum)
    s += __shfl_xor(s, m, 64);
    q += __shfl_xor(q, m, 64);
  }
  if (!live || j != 0) return;
  const float mean = rms ? 0.0f : s * inv_c;
  const float var = fmaxf(q * inv_c - mean * mean, 0.0f);
  const float rstd = 1.0f / sqrtf(var + eps);
  *reinterpret_cast<f32x2 *>(rstats + row * 2) = (f32x2){rstd, rstd * mean};
}

void launch_stats_finish(float *rstats, const float *part, int64_t slots, float eps, int64_t rows,
                         int64_t C, hipStream_t s, bool rms) {
  if (rows == 0) return;
  stats_finish_kernel<<<(unsigned)((rows + 31) / 32), 256, 0, s>>>(
      rstats, part, (int)slots, eps, rows, 1.0f / (float)C, dense_rows_dev(), rms ? 1 : 0);
}

// Constants of a folded pre-norm (RMSNorm) stack: its residuals are plain tensors, which the FOLD 2
// drain adds exactly when told gamma = 1, beta = 0, r = 1, r * mu = 0 (acc + 1 * (1 * o - 0)).
__global__ void __launch_bounds__(256)
fold_consts_kernel(bf16 *__restrict__ ones, bf16 *__restrict__ zeros, float *__restrict__ rs_one,
                   int64_t C, int64_t rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < C) { ones[i] = (bf16)1.0f; zeros[i] = (bf16)0.0f; }
  if (i < rows) *reinterpret_cast<f32x2 *>(rs_one + i * 2) = (f32x2){1.0f, 0.0f};
}
void launch_fold_consts(void *ones, void *zeros, float *rs_one, int64_t C, int64_t rows,
                        hipStream_t s) {
  const int64_t n = C > rows ? C : rows;
  if (n == 0) return;
  fold_consts_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((bf16 *)ones, (bf16 *)zeros, rs_one,
                                                                 C, rows);
}

bool row_stats_supported(int64_t C, const void *x) {
  return C % 8 == 0 && C <= 2048 && (uintptr_t)x % 16 == 0;
}

void launch_row_stats(float *rstats, const void *x, float eps, int64_t rows, int64_t C,
                      hipStream_t s, bool rms) {
  if (rows == 0) return;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (C <= 1024)
    row_stats_kernel<2><<<grid, 256, 0, s>>>(rstats, (const bf16 *)x, eps, rows, (int)C,
                                             dense_rows_dev(), rms ? 1 : 0);
  else
    row_stats_kernel<4><<<grid, 256, 0, s>>>(rstats, (const bf16 *)x, eps, rows, (int)C,
                                             dense_rows_dev(), rms ? 1 : 0);
}

// The folded parameters of a Dense that consumes LayerNorm(s) (run once per parameter version):
//   wf[n, k] = bf16(gamma[k] w[n, k] sc_n),  c[n] = sum_k float(wf[n, k]),
//   bf[n] = (sum_k beta[k] w[n, k] + b[n]) sc_n,        sc_n = scale for n < n_scaled, else 1
// c sums the ROUNDED folded weights, so that sum_k (s_k - mu) wf[n, k] = s . wf[n, :] - mu c[n]
// holds exactly in the arithmetic of the GEMM.  One wave per output row n.
__global__ void __launch_bounds__(256)
fold_norm_dense_kernel(bf16 *__restrict__ wf, float *__restrict__ c, float *__restrict__ bfold,
                       const bf16 *__restrict__ w, const bf16 *__restrict__ b,
                       const bf16 *__restrict__ gamma, const bf16 *__restrict__ beta, int64_t N,
                       int64_t K, int64_t n_scaled, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float sc = n < n_scaled ? scale : 1.0f;
  const bf16 *wr = w + n * K;
  bf16 *wo = wf + n * K;
  float csum = 0.0f, bsum = 0.0f;
  for (int64_t k = (int64_t)lane * 8; k < K; k += 64 * 8) {
    const bf16x8 w8 = *reinterpret_cast<const bf16x8 *>(wr + k);
    const bf16x8 g8 = *reinterpret_cast<const bf16x8 *>(gamma + k);
    bf16x8 t8, o8;
    if (beta) t8 = *reinterpret_cast<const bf16x8 *>(beta + k);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float wv = (float)w8[e];
      o8[e] = (bf16)((float)g8[e] * wv * sc);
      csum += (float)o8[e];
      if (beta) bsum += (float)t8[e] * wv;
    }
    *reinterpret_cast<bf16x8 *>(wo + k) = o8;
  }
  csum = wave_sum(csum);
  bsum = wave_sum(bsum);
  if (lane == 0) {
    c[n] = csum;
    bfold[n] = (bsum + (b ? (float)b[n] : 0.0f)) * sc;
  }
}

void launch_fold_norm_dense(void *wf, float *c, float *bfold, const void *w, const void *b,
                            const void *gamma, const void *beta, int64_t N, int64_t K,
                            int64_t n_scaled, float scale, hipStream_t s) {
  if (N == 0) return;
  fold_norm_dense_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(
      (bf16 *)wf, c, bfold, (const bf16 *)w, (const bf16 *)b, (const bf16 *)gamma,
      (const bf16 *)beta, N, K, n_scaled, scale);
}

}  // namespace gnlp
