// Stand-alone masked softmax / masked log-softmax over the last axis.
// Reference: attention_cell.py:277-329 (npx.masked_softmax semantics restated in
// utils/tvm_utils.py:59-75): out = softmax(where(mask, x, -inf) / T) * mask, fully masked row -> 0.
//
// HBM-bound: one wave64 per row; rows up to 1024 elements are held in registers between the
// max / sum / write passes (one read, one write); longer rows re-read from L2.
// The fused attention kernel (attn_bf16.hip) does not use this; it exists for API parity and for
// the unfused path that returns `scores` / `attn_weights`.
#include "common.h"
#include "kernels.h"

namespace gnlp {

struct RowMask {
  const uint8_t *mask;
  int64_t ms3;
  const void *vl;
  int vl_dtype;
  int64_t nq;   // valid length of this row's sample when vl != null (0 if the query row is padded)
  __device__ __forceinline__ bool on(int64_t j) const {
    if (mask) return mask[j * ms3] != 0;
    if (vl) return j < nq;
    return true;
  }
};

template <typename T, int MAXE>
__global__ void __launch_bounds__(256)
softmax_kernel(T *__restrict__ out, const T *__restrict__ x, const uint8_t *__restrict__ mask,
               int64_t ms0, int64_t ms1, int64_t ms2, int64_t ms3, int64_t s1, int64_t s2,
               int64_t Lm, int64_t rows, const void *__restrict__ vl, int vl_dtype,
               float inv_t, int log_mode) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t i2 = row % s2;
  const int64_t i1 = (row / s2) % s1;
  const int64_t i0 = row / (s2 * s1);
  RowMask rm;
  rm.mask = mask ? mask + i0 * ms0 + i1 * ms1 + i2 * ms2 : nullptr;
  rm.ms3 = ms3;
  rm.vl = vl;
  rm.vl_dtype = vl_dtype;
  rm.nq = 0;
  if (!mask && vl) {
    const int64_t n = load_len(vl, vl_dtype, i0);
    rm.nq = (i2 < n) ? n : 0;
  }
  const T *xr = x + row * Lm;
  T *orow = out + row * Lm;
  const float NEG = -INFINITY;

  if constexpr (MAXE > 0) {
    float v[MAXE];
    bool m[MAXE];
    float mx = NEG;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int64_t j = lane + 64 * i;
      m[i] = (j < Lm) && rm.on(j);
      v[i] = m[i] ? to_f32(xr[j]) * inv_t : NEG;
      mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    if (mx == NEG) mx = 0.0f;  // fully masked row
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      if (!log_mode) v[i] = m[i] ? __expf(v[i] - mx) : 0.0f;
      sum += log_mode ? (m[i] ? __expf(v[i] - mx) : 0.0f) : v[i];
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
    const float lse = sum > 0.0f ? mx + __logf(sum) : 0.0f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int64_t j = lane + 64 * i;
      if (j < Lm) orow[j] = from_f32<T>(log_mode ? (m[i] ? v[i] - lse : NEG) : v[i] * inv);
    }
  } else {
    float mx = NEG;
    for (int64_t j = lane; j < Lm; j += 64)
      if (rm.on(j)) mx = fmaxf(mx, to_f32(xr[j]) * inv_t);
    mx = wave_max(mx);
    if (mx == NEG) mx = 0.0f;
    float sum = 0.0f;
    for (int64_t j = lane; j < Lm; j += 64)
      if (rm.on(j)) sum += __expf(to_f32(xr[j]) * inv_t - mx);
    sum = wave_sum(sum);
    const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
    const float lse = sum > 0.0f ? mx + __logf(sum) : 0.0f;
    for (int64_t j = lane; j < Lm; j += 64) {
      const bool on = rm.on(j);
      const float z = to_f32(xr[j]) * inv_t;
      orow[j] = from_f32<T>(log_mode ? (on ? z - lse : NEG) : (on ? __expf(z - mx) * inv : 0.0f));
    }
  }
}

template <typename T>
static void softmax_t(void *out, const void *x, const uint8_t *mask, const int64_t ms[4],
                      const int64_t shape[4], const void *vl, int vl_dtype, float inv_t,
                      int log_mode, hipStream_t s) {
  const int64_t rows = shape[0] * shape[1] * shape[2];
  const int64_t Lm = shape[3];
  if (rows == 0 || Lm == 0) return;
  dim3 grid((unsigned)((rows + 3) / 4));
  const int64_t m0 = mask ? ms[0] : 0, m1 = mask ? ms[1] : 0, m2 = mask ? ms[2] : 0,
                m3 = mask ? ms[3] : 0;
#define GNLP_SM(MAXE)                                                                           \
  softmax_kernel<T, MAXE><<<grid, 256, 0, s>>>((T *)out, (const T *)x, mask, m0, m1, m2, m3,    \
                                               shape[1], shape[2], Lm, rows, vl, vl_dtype,     \
                                               inv_t, log_mode)
  if (Lm <= 64) GNLP_SM(1);
  else if (Lm <= 128) GNLP_SM(2);
  else if (Lm <= 256) GNLP_SM(4);
  else if (Lm <= 512) GNLP_SM(8);
  else if (Lm <= 1024) GNLP_SM(16);
  else GNLP_SM(0);
#undef GNLP_SM
}

void launch_masked_softmax(void *out, const void *scores, int dtype, const uint8_t *mask,
                           const int64_t mask_stride[4], const int64_t shape[4], const void *vl,
                           int vl_dtype, float inv_temperature, int log_mode, hipStream_t s) {
  if (dtype == GNLP_BF16)
    softmax_t<bf16>(out, scores, mask, mask_stride, shape, vl, vl_dtype, inv_temperature, log_mode,
                    s);
  else
    softmax_t<float>(out, scores, mask, mask_stride, shape, vl, vl_dtype, inv_temperature,
                     log_mode, s);
}

}  // namespace gnlp
