// fp32 MFMA dense layer for gfx950 (the 1e-5-tolerance path):
//   out = combine(act(x . w^T + bias), other), exact fp32 (v_mfma_f32_32x32x2_f32 is bit-for-bit
// an fp32 fmaf chain; gfx950 has no xf32/TF32 shortcut).
// Reference op: mxnet nn.Dense(flatten=False), models/transformer.py:193-206, layers.py:590-612.
//
// 128x128x32 tiles (128-byte LDS rows of 32 floats, same direct-to-LDS staging and source-side
// XOR swizzle as gemm_bf16.hip).  K is consumed in blocks of 8: lane-half h reads the 16-byte
// chunk holding k = 8u + 4h .. +3 of its row with one ds_read_b128 and feeds element e to the
// e-th of 4 MFMAs; A and B use the same k permutation, so the sum is unchanged.  The MFMA is
// issued as D^T = W . X^T, which leaves 4 consecutive output columns per register group
// (16-byte stores).
#include "common.h"
#include "kernels.h"

namespace gnlp {

struct DenseArgsF32 {
  const float *x, *w, *bias, *other;
  float *out;
  int64_t M, N, K, ldx, ldo, ldother;
  int act, combine;
  int tiles_m, tiles_n;
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int ACT>
__global__ void __launch_bounds__(256)
dense_f32_kernel(DenseArgsF32 a) {
  constexpr int BM = 128, BN = 128, NW = 4;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = tile / a.tiles_n, tn = tile % a.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int srow = lane >> 3, cpos = lane & 7;
  const float *ga[A_INSTR];
  const float *gb[B_INSTR];
#pragma unroll
  for (int t = 0; t < A_INSTR; ++t) {
    const int row = (wave * A_INSTR + t) * 8 + srow;
    const int c = cpos ^ ((row >> 1) & 7);
    int64_t m = m0 + row;
    if (m > a.M - 1) m = a.M - 1;
    ga[t] = a.x + m * a.ldx + c * 4;
  }
#pragma unroll
  for (int t = 0; t < B_INSTR; ++t) {
    const int row = (wave * B_INSTR + t) * 8 + srow;
    const int c = cpos ^ ((row >> 1) & 7);
    int64_t n = n0 + row;
    if (n > a.N - 1) n = a.N - 1;
    gb[t] = a.w + n * a.K + c * 4;
  }

  auto stage = [&](int buf, int64_t k0) {
    char *base = smem + buf * STAGE;
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t)
      __builtin_amdgcn_global_load_lds((gptr_t)(ga[t] + k0),
                                       (lptr_t)(base + (wave * A_INSTR + t) * 1024), 16, 0, 0);
#pragma unroll
    for (int t = 0; t < B_INSTR; ++t)
      __builtin_amdgcn_global_load_lds((gptr_t)(gb[t] + k0),
                                       (lptr_t)(base + A_BYTES + (wave * B_INSTR + t) * 1024), 16,
                                       0, 0);
  };

  const int fr = lane & 31, fh = lane >> 5;
  const int sw = (fr >> 1) & 7;
  const int a_row_off = (wm * 64 + fr) * 128;
  const int b_row_off = A_BYTES + (wn * 64 + fr) * 128;

  f32x16 acc[2][2];  // [n-frag][m-frag]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

  auto compute = [&](int buf) {
    const char *base = smem + buf * STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int coff = (((2 * u + fh) ^ sw) << 4);
      f32x4 xf[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        xf[i] = *reinterpret_cast<const f32x4 *>(base + a_row_off + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[j] = *reinterpret_cast<const f32x4 *>(base + b_row_off + j * 32 * 128 + coff);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][e], xf[i][e], acc[j][i], 0, 0, 0);
    }
  };

  const int nk = (int)(a.K >> 5);
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk - 1; ++kt) {
    stage(cur ^ 1, (int64_t)(kt + 1) << 5);
    compute(cur);
    __syncthreads();
    cur ^= 1;
  }
  compute(cur);

  // D[row = n_local][col = m_local]: lane -> m = fr, n_local = 8g + 4*fh + e, g = reg >> 2
  const bool has_bias = a.bias != nullptr, has_other = a.other != nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t n = n0 + wn * 64 + j * 32 + g * 8 + fh * 4;
      const bool n_ok = n < a.N;
      f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (has_bias && n_ok) bv = *reinterpret_cast<const f32x4 *>(a.bias + n);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wm * 64 + i * 32 + fr;
        const bool ok = n_ok && (m < a.M);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<true>(acc[j][i][g * 4 + e] + bv[e], ACT);
        if (has_other && ok) {
          f32x4 o4 = *reinterpret_cast<const f32x4 *>(a.other + m * a.ldother + n);
          if (a.combine == GNLP_COMBINE_MUL) v *= o4;
          else v += o4;
        }
        if (ok) *reinterpret_cast<f32x4 *>(a.out + m * a.ldo + n) = v;
      }
    }
  }
}

bool mfma_f32_dense_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                              int64_t ldother, const void *x, const void *w, const void *out,
                              const void *other) {
  if (M < 1 || N < 4 || K < 32) return false;
  if (K % 32 != 0 || N % 4 != 0) return false;
  if (ldx % 4 != 0 || ldo % 4 != 0) return false;
  if (other && ldother % 4 != 0) return false;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)(other ? other : out)) % 16 != 0)
    return false;
  return true;
}

template <int ACT>
static void launch_f32_act(const DenseArgsF32 &a, hipStream_t s) {
  constexpr int LDS = 2 * 256 * 128;
  static PerDeviceFlag attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void *)dense_f32_kernel<ACT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  dense_f32_kernel<ACT><<<dim3((unsigned)(a.tiles_m * a.tiles_n)), 256, LDS, s>>>(a);
}

void launch_mfma_f32_dense(void *out, const void *x, const void *w, const void *bias,
                           const void *other, int combine, int act, int64_t M, int64_t N,
                           int64_t K, int64_t ldx, int64_t ldo, int64_t ldother, hipStream_t s) {
  DenseArgsF32 a;
  a.x = (const float *)x; a.w = (const float *)w; a.bias = (const float *)bias;
  a.other = (const float *)other; a.out = (float *)out;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.ldother = ldother;
  a.act = act; a.combine = combine;
  a.tiles_m = (int)((M + 127) / 128);
  a.tiles_n = (int)((N + 127) / 128);
  switch (act) {
    case GNLP_ACT_RELU: launch_f32_act<GNLP_ACT_RELU>(a, s); break;
    case GNLP_ACT_GELU_ERF: launch_f32_act<GNLP_ACT_GELU_ERF>(a, s); break;
    case GNLP_ACT_GELU_TANH: launch_f32_act<GNLP_ACT_GELU_TANH>(a, s); break;
    default: launch_f32_act<GNLP_ACT_IDENTITY>(a, s); break;
  }
}

}  // namespace gnlp
