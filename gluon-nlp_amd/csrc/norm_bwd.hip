// Backward pass of (residual +) LayerNorm / RMSNorm / NoNorm over the last axis.
// Reference: the gradients mx.autograd takes through mxnet nn.LayerNorm
// (models/transformer.py:217-218,268-270), RMSNorm (layers.py:151-181) and NoNorm
// (layers.py:81-138).  With v = x (+ residual), g = cotangent of the output:
//   layer_norm: xh = (v - mean) rstd,  gg = g gamma,
//               dx = rstd (gg - mean(gg) - xh mean(gg xh))
//   rms_norm:   xh = v rstd,           dx = rstd (gg - xh mean(gg xh))
//   no_norm:    xh = v,                dx = gg
//   dgamma = sum_rows g xh,   dbeta = sum_rows g            (fp32, caller casts)
// d(residual) = dx (the sum's gradient goes to both addends).
//
// HBM-bound: a wave owns a row at a time (rows strided over a grid of at most 1024 workgroups);
// rows of up to 2048 elements stay in registers between the statistics and the dx pass (x, g read
// once, dx written once); the wave's dgamma / dbeta partial sums live in registers across its
// rows, are combined per workgroup in LDS and leave as one fp32 atomic per column per workgroup.
#include "common.h"
#include "kernels.h"

namespace gnlp {

namespace {

template <typename T, int MAXE>
__global__ void __launch_bounds__(256)
norm_bwd_kernel(T *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                const T *__restrict__ g, const T *__restrict__ x, const T *__restrict__ residual,
                const T *__restrict__ gamma, int kind, float eps, int64_t rows, int C) {
  extern __shared__ float acc_lds[];   // [2][C]: dgamma, dbeta of this workgroup
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) acc_lds[c] = 0.0f;
  __syncthreads();
  const float invC = 1.0f / (float)C;

  if constexpr (MAXE > 0) {
    float pg[MAXE], pb[MAXE], gam[MAXE];
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int c = lane + 64 * i;
      pg[i] = pb[i] = 0.0f;
      gam[i] = c < C ? to_f32(gamma[c]) : 0.0f;
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
      const T *xr = x + row * C, *gr = g + row * C;
      const T *rr = residual ? residual + row * C : nullptr;
      float v[MAXE], go[MAXE];
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < MAXE; ++i) {
        const int c = lane + 64 * i;
        const bool in = c < C;
        v[i] = in ? to_f32(xr[c]) + (rr ? to_f32(rr[c]) : 0.0f) : 0.0f;
        go[i] = in ? to_f32(gr[c]) : 0.0f;
        s += v[i];
      }
      float mean = 0.0f, rstd = 1.0f;
      if (kind == GNLP_NORM_LAYER) {
        mean = wave_sum(s) * invC;
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXE; ++i)
          if (lane + 64 * i < C) ss += (v[i] - mean) * (v[i] - mean);
        rstd = 1.0f / sqrtf(wave_sum(ss) * invC + eps);
      } else if (kind == GNLP_NORM_RMS) {
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) ss += v[i] * v[i];
        rstd = 1.0f / sqrtf(wave_sum(ss) * invC + eps);
      }
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int i = 0; i < MAXE; ++i) {
        const bool in = lane + 64 * i < C;
        v[i] = in ? (v[i] - mean) * rstd : 0.0f;   // xh
        pg[i] += go[i] * v[i];
        pb[i] += go[i];
        go[i] *= gam[i];                            // gg
        s1 += go[i];
        s2 += go[i] * v[i];
      }
      float m1 = 0.0f, m2 = 0.0f;
      if (kind == GNLP_NORM_LAYER) m1 = wave_sum(s1) * invC;
      if (kind != GNLP_NORM_NONE) m2 = wave_sum(s2) * invC;
      T *dr = dx + row * C;
#pragma unroll
      for (int i = 0; i < MAXE; ++i) {
        const int c = lane + 64 * i;
        if (c < C) dr[c] = from_f32<T>(rstd * (go[i] - m1 - v[i] * m2));
      }
    }
    // the four waves add their partial sums in turn, with plain LDS accesses: as LDS float
    // atomics this step cost ~8 us per workgroup (tools/bwd_bench.py: 117 -> 60 us at 32768 x 1024)
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
          const int c = lane + 64 * i;
          if (c < C) {
            acc_lds[c] += pg[i];
            acc_lds[C + c] += pb[i];
          }
        }
      }
      __syncthreads();
    }
  } else {
    // any C that fits the LDS accumulators: three passes over the (L2-resident) row
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
      const T *xr = x + row * C, *gr = g + row * C;
      const T *rr = residual ? residual + row * C : nullptr;
      auto val = [&](int c) { return to_f32(xr[c]) + (rr ? to_f32(rr[c]) : 0.0f); };
      float mean = 0.0f, rstd = 1.0f;
      if (kind == GNLP_NORM_LAYER) {
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s += val(c);
        mean = wave_sum(s) * invC;
        float ss = 0.0f;
        for (int c = lane; c < C; c += 64) ss += (val(c) - mean) * (val(c) - mean);
        rstd = 1.0f / sqrtf(wave_sum(ss) * invC + eps);
      } else if (kind == GNLP_NORM_RMS) {
        float ss = 0.0f;
        for (int c = lane; c < C; c += 64) ss += val(c) * val(c);
        rstd = 1.0f / sqrtf(wave_sum(ss) * invC + eps);
      }
      float s1 = 0.0f, s2 = 0.0f;
      for (int c = lane; c < C; c += 64) {
        const float xh = (val(c) - mean) * rstd, go = to_f32(gr[c]);
        atomicAdd(&acc_lds[c], go * xh);
        atomicAdd(&acc_lds[C + c], go);
        const float gg = go * to_f32(gamma[c]);
        s1 += gg;
        s2 += gg * xh;
      }
      float m1 = 0.0f, m2 = 0.0f;
      if (kind == GNLP_NORM_LAYER) m1 = wave_sum(s1) * invC;
      if (kind != GNLP_NORM_NONE) m2 = wave_sum(s2) * invC;
      T *dr = dx + row * C;
      for (int c = lane; c < C; c += 64) {
        const float xh = (val(c) - mean) * rstd;
        dr[c] = from_f32<T>(rstd * (to_f32(gr[c]) * to_f32(gamma[c]) - m1 - xh * m2));
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomicAdd(&dgamma[c], acc_lds[c]);
    atomicAdd(&dbeta[c], acc_lds[C + c]);
  }
}

// The same with 16-byte accesses (C a multiple of the vector, rows 16-byte aligned, C <= 64 * N *
// MAXV): a lane owns MAXV vectors of N consecutive columns.  x, g are read once, dx written once;
// non-temporal loads (nothing here is read again).
template <typename T, int MAXV>
__global__ void __launch_bounds__(256)
norm_bwd_vec_kernel(T *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                    const T *__restrict__ g, const T *__restrict__ x,
                    const T *__restrict__ residual, const T *__restrict__ gamma, int kind,
                    float eps, int64_t rows, int C) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  extern __shared__ float acc_lds[];   // [2][C]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) acc_lds[c] = 0.0f;
  __syncthreads();
  const float invC = 1.0f / (float)C;
  const int nvec = C / N;
  float pg[MAXV][N], pb[MAXV][N], gam[MAXV][N];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + 64 * i;
#pragma unroll
    for (int q = 0; q < N; ++q) {
      pg[i][q] = pb[i][q] = 0.0f;
      gam[i][q] = vi < nvec ? to_f32(gamma[vi * N + q]) : 0.0f;
    }
  }
  // the loads of the wave's NEXT row are issued before the current one is reduced: a row is four
  // dependent wave reductions long, and with the loads inside that chain the kernel ran at a
  // third of the memory rate
  const int64_t stride = (int64_t)gridDim.x * 4;
  V na[MAXV], nb[MAXV], nr[MAXV];
  auto fetch = [&](int64_t row) {
    const T *xr = x + row * C, *gr = g + row * C;
    const T *rr = residual ? residual + row * C : nullptr;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        na[i] = __builtin_nontemporal_load(reinterpret_cast<const V *>(xr + vi * N));
        nb[i] = __builtin_nontemporal_load(reinterpret_cast<const V *>(gr + vi * N));
        if (rr) nr[i] = __builtin_nontemporal_load(reinterpret_cast<const V *>(rr + vi * N));
      }
    }
  };
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row < rows) fetch(row);
  for (; row < rows; row += stride) {
    float v[MAXV][N], go[MAXV][N];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
          v[i][q] = to_f32(na[i][q]);
          go[i][q] = to_f32(nb[i][q]);
        }
        if (residual) {
#pragma unroll
          for (int q = 0; q < N; ++q) v[i][q] += to_f32(nr[i][q]);
        }
#pragma unroll
        for (int q = 0; q < N; ++q) s += v[i][q];
      } else {
#pragma unroll
        for (int q = 0; q < N; ++q) v[i][q] = go[i][q] = 0.0f;
      }
    }
    if (row + stride < rows) fetch(row + stride);
    float mean = 0.0f, rstd = 1.0f;
    if (kind == GNLP_NORM_LAYER) {
      mean = wave_sum(s) * invC;
      float ss = 0.0f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nvec) {
#pragma unroll
          for (int q = 0; q < N; ++q) ss += (v[i][q] - mean) * (v[i][q] - mean);
        }
      rstd = 1.0f / sqrtf(wave_sum(ss) * invC + eps);
    } else if (kind == GNLP_NORM_RMS) {
      float ss = 0.0f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int q = 0; q < N; ++q) ss += v[i][q] * v[i][q];
      rstd = 1.0f / sqrtf(wave_sum(ss) * invC + eps);
    }
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const bool in = lane + 64 * i < nvec;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        v[i][q] = in ? (v[i][q] - mean) * rstd : 0.0f;   // xh
        pg[i][q] += go[i][q] * v[i][q];
        pb[i][q] += go[i][q];
        go[i][q] *= gam[i][q];                            // gg
        s1 += go[i][q];
        s2 += go[i][q] * v[i][q];
      }
    }
    float m1 = 0.0f, m2 = 0.0f;
    if (kind == GNLP_NORM_LAYER) m1 = wave_sum(s1) * invC;
    if (kind != GNLP_NORM_NONE) m2 = wave_sum(s2) * invC;
    T *dr = dx + row * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        V o;
#pragma unroll
        for (int q = 0; q < N; ++q) o[q] = from_f32<T>(rstd * (go[i][q] - m1 - v[i][q] * m2));
        *reinterpret_cast<V *>(dr + vi * N) = o;
      }
    }
  }
  // the four waves add their partial sums in turn, with plain 16-byte LDS accesses (see above)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
#pragma unroll
          for (int q = 0; q < N; q += 4) {
            f32x4 *ag = reinterpret_cast<f32x4 *>(&acc_lds[vi * N + q]);
            f32x4 *ab = reinterpret_cast<f32x4 *>(&acc_lds[C + vi * N + q]);
            *ag += (f32x4){pg[i][q], pg[i][q + 1], pg[i][q + 2], pg[i][q + 3]};
            *ab += (f32x4){pb[i][q], pb[i][q + 1], pb[i][q + 2], pb[i][q + 3]};
          }
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomicAdd(&dgamma[c], acc_lds[c]);
    atomicAdd(&dbeta[c], acc_lds[C + c]);
  }
}

template <typename T>
void norm_bwd_t(void *dx, float *dgamma, float *dbeta, const void *g, const void *x,
                const void *residual, const void *gamma, int kind, float eps, int64_t rows,
                int64_t C, hipStream_t s) {
  const int64_t want = (rows + 3) / 4;
  const unsigned grid = (unsigned)(want < 1024 ? want : 1024);
  const size_t lds = (size_t)2 * C * sizeof(float);
  constexpr int N = VecOf<T>::N;
  const uintptr_t al = (uintptr_t)dx | (uintptr_t)g | (uintptr_t)x | (uintptr_t)(residual ? residual : x);
  if (C % N == 0 && al % 16 == 0 && C / N <= 256) {   // at most 4 vectors per lane
#define GNLP_NBV(MV)                                                                            \
  norm_bwd_vec_kernel<T, MV><<<grid, 256, lds, s>>>((T *)dx, dgamma, dbeta, (const T *)g,       \
                                                    (const T *)x, (const T *)residual,          \
                                                    (const T *)gamma, kind, eps, rows, (int)C)
    const int nvec = (int)(C / N);
    if (nvec <= 64) GNLP_NBV(1);
    else if (nvec <= 128) GNLP_NBV(2);
    else GNLP_NBV(4);
#undef GNLP_NBV
    return;
  }
#define GNLP_NB(MAXE)                                                                          \
  norm_bwd_kernel<T, MAXE><<<grid, 256, lds, s>>>((T *)dx, dgamma, dbeta, (const T *)g,         \
                                                  (const T *)x, (const T *)residual,            \
                                                  (const T *)gamma, kind, eps, rows, (int)C)
  if (C <= 64) GNLP_NB(1);
  else if (C <= 128) GNLP_NB(2);
  else if (C <= 256) GNLP_NB(4);
  else if (C <= 512) GNLP_NB(8);
  else if (C <= 1024) GNLP_NB(16);
  else if (C <= 2048) GNLP_NB(32);
  else GNLP_NB(0);
#undef GNLP_NB
}

// l2_normalize (op.py:289-307): y = x / (n + eps), n = ||x||_2
//   dx = g / (n + eps) - x <g, x> / (n (n + eps)^2)     (n = 0: the norm's subgradient 0)
template <typename T>
__global__ void __launch_bounds__(256)
l2norm_bwd_kernel(T *__restrict__ dx, const T *__restrict__ g, const T *__restrict__ x,
                  int64_t rows, int64_t d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T *xr = x + row * d, *gr = g + row * d;
  float ss = 0.0f, gx = 0.0f;
  for (int64_t c = lane; c < d; c += 64) {
    const float xv = to_f32(xr[c]);
    ss += xv * xv;
    gx += to_f32(gr[c]) * xv;
  }
  const float nrm = sqrtf(wave_sum(ss));
  gx = wave_sum(gx);
  const float inv = 1.0f / (nrm + eps);
  const float k2 = nrm > 0.0f ? gx * inv * inv / nrm : 0.0f;
  for (int64_t c = lane; c < d; c += 64)
    dx[row * d + c] = from_f32<T>(to_f32(gr[c]) * inv - to_f32(xr[c]) * k2);
}

// column sums of a (rows, C) matrix: the bias gradient of a dense layer (fp32 atomics into a
// zero-filled vector, one per column per workgroup)
template <typename T>
__global__ void __launch_bounds__(256)
colsum_kernel(float *__restrict__ out, const T *__restrict__ g, int64_t rows, int64_t C) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float acc = 0.0f;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) acc += to_f32(g[r * C + c]);
  atomicAdd(&out[c], acc);
}

template <typename T>
__global__ void __launch_bounds__(256)
cast_f32_kernel(T *__restrict__ out, const float *__restrict__ in, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = from_f32<T>(in[i]);
}

}  // namespace

bool norm_bwd_supported(int64_t C) { return C >= 1 && C <= kNormBwdMaxC; }

void launch_norm_bwd(void *dx, float *dgamma, float *dbeta, const void *g, const void *x,
                     const void *residual, const void *gamma, int dtype, int kind, float eps,
                     int64_t rows, int64_t C, hipStream_t s) {
  if (rows == 0 || C == 0) return;
  if (dtype == GNLP_BF16)
    norm_bwd_t<bf16>(dx, dgamma, dbeta, g, x, residual, gamma, kind, eps, rows, C, s);
  else
    norm_bwd_t<float>(dx, dgamma, dbeta, g, x, residual, gamma, kind, eps, rows, C, s);
}

void launch_l2_normalize_bwd(void *dx, const void *g, const void *x, int dtype, int64_t rows,
                             int64_t d, float eps, hipStream_t s) {
  if (rows == 0 || d == 0) return;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (dtype == GNLP_BF16)
    l2norm_bwd_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)dx, (const bf16 *)g, (const bf16 *)x,
                                                 rows, d, eps);
  else
    l2norm_bwd_kernel<float><<<grid, 256, 0, s>>>((float *)dx, (const float *)g,
                                                  (const float *)x, rows, d, eps);
}

void launch_colsum(float *out, const void *g, int dtype, int64_t rows, int64_t C, hipStream_t s) {
  if (rows == 0 || C == 0) return;
  const int64_t gy = rows < 256 ? rows : 256;
  dim3 grid((unsigned)((C + 255) / 256), (unsigned)gy);
  if (dtype == GNLP_BF16) colsum_kernel<bf16><<<grid, 256, 0, s>>>(out, (const bf16 *)g, rows, C);
  else colsum_kernel<float><<<grid, 256, 0, s>>>(out, (const float *)g, rows, C);
}

void launch_cast_f32(void *out, int dtype, const float *in, int64_t n, hipStream_t s) {
  if (n == 0) return;
  const int64_t want = (n + 255) / 256;
  const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
  if (dtype == GNLP_BF16) cast_f32_kernel<bf16><<<grid, 256, 0, s>>>((bf16 *)out, in, n);
  else cast_f32_kernel<float><<<grid, 256, 0, s>>>((float *)out, in, n);
}

}  // namespace gnlp
