// Shared device helpers for the gfx950 kernels (wave64, bf16 <-> fp32, reductions, activations).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gluonnlp_amd.h"

namespace gnlp {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kWave = 64;

template <typename T> struct VecOf;   // 16-byte vector of T
template <> struct VecOf<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct VecOf<bf16> { typedef bf16x8 type; static constexpr int N = 8; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Workgroups are placed on the 8 XCDs round-robin by dispatch order.  Linear workgroup id ->
// (head = batch * H + h, block) such that the nblk blocks of one head share an XCD (one L2) and are
// dispatched together: what they all read (K / V, or Q / dO) is fetched from HBM once instead of
// once per block.  Bijective when the number of heads is a multiple of 8; plain order otherwise.
__device__ __forceinline__ void xcd_head_map(int64_t lin, int64_t nblk, int64_t nheads,
                                             int64_t &head, int64_t &blk) {
  if ((nheads & 7) == 0) {
    const int64_t grp = lin / (8 * nblk), i = lin - grp * 8 * nblk;
    head = grp * 8 + (i & 7);
    blk = i >> 3;
  } else {
    blk = lin % nblk;
    head = lin / nblk;
  }
}

__device__ __forceinline__ int64_t load_len(const void *p, int vl_dtype, int64_t i) {
  return vl_dtype == GNLP_I64 ? ((const int64_t *)p)[i] : (int64_t)((const int32_t *)p)[i];
}

// The lengths behind the in-kernel predicates of sequence b when no dense mask is given:
// klen = valid keys, qlen = valid queries (-1: no lengths, everything is on).  gen_self_attn_mask
// products have one length per sequence; gen_mem_attn_mask products (pair != 0) hold B memory
// lengths followed by B query lengths (attention_cell.py:153-161, :262-272).
__device__ __forceinline__ void load_len_pair(const void *p, int vl_dtype, int pair, int64_t B,
                                              int64_t b, int64_t &klen, int64_t &qlen) {
  klen = load_len(p, vl_dtype, b);
  klen = klen < 0 ? 0 : klen;
  qlen = klen;
  if (pair) {
    qlen = load_len(p, vl_dtype, B + b);
    qlen = qlen < 0 ? 0 : qlen;
  }
}

// layers.py:236-338.  EXACT selects erff (fp32 tolerance path); the bf16 path may use the
// cheaper rational erf (|err| < 1.5e-7, far below bf16 resolution).
__device__ __forceinline__ float erf_fast(float x) {
  // Abramowitz & Stegun 7.1.26
  float ax = fabsf(x);
  float t = __frcp_rn(1.0f + 0.3275911f * ax);
  float p = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t +
             0.254829592f) * t;
  float r = 1.0f - p * __expf(-ax * ax);
  return copysignf(r, x);
}

// erf-GELU for the bf16 epilogues without a transcendental instruction:  x * Phi(x)  with
//   Phi(x) ~= sat(0.5 + u Q(y)),   u = x / 4,   y = sat(u^2),   sat = the clamp modifier (to [0, 1])
// Q of degree 6: minimax fit by tools/fit/gelu_poly_fit.py against the bound tests/test_gpu_ops.py
// asserts; |gelu error| < 1.9e-4 absolute (bf16 resolution near 1 is 3.9e-3).  Beyond |x| = 4 the
// polynomial sees y = 1 and Q(1) >= 0.5: Phi is exactly 1 / 0, gelu(x) = x for large x and -0 for
// very negative x.  10 instructions per pair against 11 for round 2's sigmoid form (x^2 (a1 + a3 x^2
// + a5 x^4) through v_exp + v_rcp); what a drain pays is the instruction COUNT, ~9 cycles each,
// transcendental or not (same-box A / B) -- hence degree 6, not 7.
#define GELU_POLY_RC 0.25f
#define GELU_POLY_DEG 6
#define GELU_POLY_Q {1.5914379358291626f, -4.130927562713623f, 8.972149848937988f, \
                     -13.306079864501953f, 12.52656078338623f, -6.672885894775391f, \
                     1.5197588205337524f}

__device__ __forceinline__ float gelu_erf_fast(float x) {
  constexpr float q[GELU_POLY_DEG + 1] = GELU_POLY_Q;
  const float u = x * GELU_POLY_RC;
  float y, phi;
  asm("v_mul_f32_e64 %0, %1, %1 clamp" : "=v"(y) : "v"(u));
  float t = q[GELU_POLY_DEG];
#pragma unroll
  for (int k = GELU_POLY_DEG - 1; k >= 0; --k) t = fmaf(t, y, q[k]);
  asm("v_fma_f32 %0, %1, %2, 0.5 clamp" : "=v"(phi) : "v"(u), "v"(t));
  return x * phi;
}

// N pairs in lockstep (v_pk_* math): the Horner chain of one pair is serial and a drain has one
// wave per SIMD -- chain by chain the compiler pads every step with an s_nop
template <int N>
__device__ __forceinline__ void gelu_erf_fast2_n(f32x2 (&x)[N]) {
  constexpr float q[GELU_POLY_DEG + 1] = GELU_POLY_Q;
  const f32x2 rc = {GELU_POLY_RC, GELU_POLY_RC};
  const f32x2 half = {0.5f, 0.5f};
  f32x2 u[N], y[N], t[N];
#pragma unroll
  for (int c = 0; c < N; ++c) u[c] = x[c] * rc;
#pragma unroll
  for (int c = 0; c < N; ++c) asm("v_pk_mul_f32 %0, %1, %1 clamp" : "=v"(y[c]) : "v"(u[c]));
#pragma unroll
  for (int c = 0; c < N; ++c) t[c] = y[c] * (f32x2){q[GELU_POLY_DEG], q[GELU_POLY_DEG]} +
                                     (f32x2){q[GELU_POLY_DEG - 1], q[GELU_POLY_DEG - 1]};
#pragma unroll
  for (int k = GELU_POLY_DEG - 2; k >= 0; --k)
#pragma unroll
    for (int c = 0; c < N; ++c) t[c] = t[c] * y[c] + (f32x2){q[k], q[k]};
#pragma unroll
  for (int c = 0; c < N; ++c)
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t[c]) : "v"(u[c]), "v"(t[c]), "v"(half));
#pragma unroll
  for (int c = 0; c < N; ++c) x[c] = x[c] * t[c];
}

template <bool EXACT>
__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case GNLP_ACT_IDENTITY: return x;
    case GNLP_ACT_RELU: return fmaxf(x, 0.0f);
    case GNLP_ACT_GELU_ERF: {
      if (!EXACT) return gelu_erf_fast(x);
      return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    }
    case GNLP_ACT_GELU_TANH: {
      float u = 0.79788456080286535588f * (x + 0.044715f * x * x * x);
      return 0.5f * x * (1.0f + tanhf(u));
    }
    case GNLP_ACT_GELU_SIGMOID: return x / (1.0f + expf(-1.702f * x));
    case GNLP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case GNLP_ACT_TANH: return tanhf(x);
    case GNLP_ACT_LEAKY: return x > 0.0f ? x : 0.1f * x;
    case GNLP_ACT_ELU: return x > 0.0f ? x : expm1f(x);
    case GNLP_ACT_SOFTRELU: return log1pf(expf(-fabsf(x))) + fmaxf(x, 0.0f);
    case GNLP_ACT_SOFTSIGN: return x / (1.0f + fabsf(x));
    default: return x;
  }
}

// d act(x) / dx of the activations above, with the exact formulas (the backward of the fused Dense
// epilogues and of get_activation; layers.py:236-338 under autograd)
__device__ __forceinline__ float apply_dact(float x, int act) {
  switch (act) {
    case GNLP_ACT_IDENTITY: return 1.0f;
    case GNLP_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
    case GNLP_ACT_GELU_ERF: {
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
    }
    case GNLP_ACT_GELU_TANH: {
      const float k = 0.79788456080286535588f;
      const float t = tanhf(k * (x + 0.044715f * x * x * x));
      return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k * (1.0f + 3.0f * 0.044715f * x * x);
    }
    case GNLP_ACT_GELU_SIGMOID: {
      const float s = 1.0f / (1.0f + expf(-1.702f * x));
      return s + 1.702f * x * s * (1.0f - s);
    }
    case GNLP_ACT_SIGMOID: {
      const float s = 1.0f / (1.0f + expf(-x));
      return s * (1.0f - s);
    }
    case GNLP_ACT_TANH: {
      const float t = tanhf(x);
      return 1.0f - t * t;
    }
    case GNLP_ACT_LEAKY: return x > 0.0f ? 1.0f : 0.1f;
    case GNLP_ACT_ELU: return x > 0.0f ? 1.0f : expf(x);
    case GNLP_ACT_SOFTRELU: return 1.0f / (1.0f + expf(-x));
    case GNLP_ACT_SOFTSIGN: {
      const float d = 1.0f + fabsf(x);
      return 1.0f / (d * d);
    }
    default: return 1.0f;
  }
}

}  // namespace gnlp
