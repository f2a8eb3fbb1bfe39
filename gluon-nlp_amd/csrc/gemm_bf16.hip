// bf16 MFMA dense layer for gfx950:  out = combine(act(x . w^T + bias), other)
// Reference op: mxnet nn.Dense(flatten=False) at models/transformer.py:193-206,261,266 and
// layers.py:590-612,636-642 with the bias / GELU / residual / gate epilogues fused.
//
// Both operands are K-contiguous (x is (M, K), the Dense weight is (N, K)), which is exactly the
// A / B^T fragment order of v_mfma_f32_16x16x32_bf16: every fragment is one 16-byte LDS read.
// Structure: BMxBNx64 tiles, double-buffered LDS filled with 16-byte direct-to-LDS loads
// (global_load_lds_dwordx4).  The LDS image is lane-linear, so the bank-conflict swizzle
// (16-byte chunk index XOR ((row >> 1) & 7)) is applied to the per-lane *source* address and to
// the ds_read address (both sides or neither).  The MFMA is issued as D^T = W_frag . X_frag^T so
// that each lane ends up with 4 consecutive output columns (8-byte stores, bias as a vector).
// Blocks are remapped so that each XCD (own L2) walks a contiguous range of tiles.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace gnlp {

struct DenseArgs {
  const bf16 *x, *w, *bias, *other;
  bf16 *out;
  int64_t M, N, K, ldx, ldo, ldother;
  int act, combine;
  int tiles_m, tiles_n, group_m;
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int BM, int BN, int WM, int WN, int ACT>
__global__ void __launch_bounds__(WM *WN * 64)
dense_bf16_kernel(DenseArgs a) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave output tile
  constexpr int FM = TM / 16, FN = TN / 16;  // 16x16 fragments per wave
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // 1-KiB glds per wave per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- XCD-aware tile mapping (bijective for any grid size) ----
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // grouped order: within a band of GROUP_M row-panels walk m fastest, so the ~32 tiles an XCD
  // has in flight form a compact patch (few distinct x / weight panels per K-step in its L2)
  const int GROUP_M = a.group_m;
  const int band = tile / (GROUP_M * a.tiles_n);
  const int band_m0 = band * GROUP_M;
  const int band_h = (a.tiles_m - band_m0) < GROUP_M ? (a.tiles_m - band_m0) : GROUP_M;
  const int in_band = tile - band * GROUP_M * a.tiles_n;
  const int tm = band_m0 + in_band % band_h, tn = in_band / band_h;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- staging addresses: lane -> (row within an 8-row piece, 16-byte chunk position) ----
  const int srow = lane >> 3, cpos = lane & 7;
  const bf16 *ga[A_INSTR];
  const bf16 *gb[B_INSTR];
#pragma unroll
  for (int t = 0; t < A_INSTR; ++t) {
    const int row = (wave * A_INSTR + t) * 8 + srow;
    const int c = cpos ^ ((row >> 1) & 7);
    int64_t m = m0 + row;
    if (m > a.M - 1) m = a.M - 1;
    ga[t] = a.x + m * a.ldx + c * 8;
  }
#pragma unroll
  for (int t = 0; t < B_INSTR; ++t) {
    const int row = (wave * B_INSTR + t) * 8 + srow;
    const int c = cpos ^ ((row >> 1) & 7);
    int64_t n = n0 + row;
    if (n > a.N - 1) n = a.N - 1;
    gb[t] = a.w + n * a.K + c * 8;
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

  // ---- fragment read offsets (bytes), k-step 0; k-step 1 is the same XOR 64 ----
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int frag_off = fr * 128 + ((fq ^ sw) << 4);
  const int a_off = (wm * TM) * 128 + frag_off;
  const int b_off = A_BYTES + (wn * TN) * 128 + frag_off;

  f32x4 acc[FN][FM];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char *base = smem + buf * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 xf[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        xf[i] = *reinterpret_cast<const bf16x8 *>(base + ((a_off + i * 2048) ^ (s << 6)));
#pragma unroll
      for (int j = 0; j < FN; ++j)
        wf[j] = *reinterpret_cast<const bf16x8 *>(base + ((b_off + j * 2048) ^ (s << 6)));
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
  };

  const int nk = (int)(a.K >> 6);
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk - 1; ++kt) {
    stage(cur ^ 1, (int64_t)(kt + 1) << 6);
    compute(cur);
    __syncthreads();
    cur ^= 1;
  }
  compute(cur);

  // ---- epilogue: lane holds out[m][n .. n+3] per fragment ----
  const bool has_bias = a.bias != nullptr, has_other = a.other != nullptr;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int64_t n = n0 + wn * TN + j * 16 + fq * 4;
    const bool n_ok = n < a.N;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_bias && n_ok) {
      bf16x4 b4 = *reinterpret_cast<const bf16x4 *>(a.bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (float)b4[e];
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int64_t m = m0 + wm * TM + i * 16 + fr;
      const bool ok = n_ok && (m < a.M);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act<false>(acc[j][i][e] + bv[e], ACT);
      if (has_other && ok) {
        bf16x4 o4 = *reinterpret_cast<const bf16x4 *>(a.other + m * a.ldother + n);
        if (a.combine == GNLP_COMBINE_MUL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= (float)o4[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)o4[e];
        }
      }
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
      if (ok) *reinterpret_cast<bf16x4 *>(a.out + m * a.ldo + n) = o;
    }
  }
}

// ---- a few rows (one decoding step of a batch: M = batch size) ----
// The tiled kernel above gives such a call N / 128 workgroups, each walking all of K through
// barriers two 64-wide tiles at a time: 6 workgroups and 19 us for N = K = 768 on a 256-CU chip,
// for 1.2 MB of weights.  Here a workgroup owns 16 output columns for all rows and its 8 waves
// split K: every wave streams its share of the weight rows straight from global memory into
// MFMA fragments (no LDS staging, no barrier in the loop, loads of several k-steps in flight), the
// partial sums meet once in LDS, and waves 0 .. FM - 1 run the epilogue.  Same MFMA, fp32
// accumulation and epilogue as above; the order of the K summation differs.
constexpr int SKINNY_WAVES = 8;

template <int FM>
__global__ void __launch_bounds__(SKINNY_WAVES * 64)
dense_bf16_skinny_kernel(DenseArgs a) {
  __shared__ __attribute__((aligned(16))) float red[SKINNY_WAVES][FM][64][4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * 16;
  const int64_t mb = (int64_t)blockIdx.y * (16 * FM);   // first row of this workgroup
  int64_t nrow = n0 + fr;
  if (nrow > a.N - 1) nrow = a.N - 1;
  const bf16 *wp = a.w + nrow * a.K + fq * 8;
  const bf16 *xp[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int64_t m = mb + i * 16 + fr;
    if (m > a.M - 1) m = a.M - 1;
    xp[i] = a.x + m * a.ldx + fq * 8;
  }
  f32x4 acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // k-steps of 32, dealt out round robin: neighbouring waves read the two halves of a 128-byte line
  const int nks = (int)(a.K >> 5);
  const int cnt = nks > wave ? (nks - wave + SKINNY_WAVES - 1) / SKINNY_WAVES : 0;
  // the loads of up to four k-steps are in flight before the first MFMA (the guards are uniform
  // per wave; K = 768 leaves a wave three k-steps: without them it would run them one by one)
  for (int t = 0; t < cnt; t += 4) {
    bf16x8 wf[4], xf[4][FM];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (t + u < cnt) {
        const int64_t off = (int64_t)(wave + SKINNY_WAVES * (t + u)) * 32;
        wf[u] = *reinterpret_cast<const bf16x8 *>(wp + off);
#pragma unroll
        for (int i = 0; i < FM; ++i) xf[u][i] = *reinterpret_cast<const bf16x8 *>(xp[i] + off);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (t + u < cnt) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u][i], acc[i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) *reinterpret_cast<f32x4 *>(&red[wave][i][lane][0]) = acc[i];
  __syncthreads();
  if (wave >= FM) return;
  const int i = wave;                       // this wave finishes rows 16 i .. 16 i + 15
  f32x4 sum = *reinterpret_cast<const f32x4 *>(&red[0][i][lane][0]);
#pragma unroll
  for (int w = 1; w < SKINNY_WAVES; ++w) {
    const f32x4 p = *reinterpret_cast<const f32x4 *>(&red[w][i][lane][0]);
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] += p[e];
  }
  // lane holds out[m][n .. n + 3]
  const int64_t n = n0 + fq * 4, m = mb + i * 16 + fr;
  if (n >= a.N || m >= a.M) return;
  float v[4];
  if (a.bias) {
    const bf16x4 b4 = *reinterpret_cast<const bf16x4 *>(a.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] += (float)b4[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = apply_act<false>(sum[e], a.act);
  if (a.other) {
    const bf16x4 o4 = *reinterpret_cast<const bf16x4 *>(a.other + m * a.ldother + n);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      v[e] = a.combine == GNLP_COMBINE_MUL ? v[e] * (float)o4[e] : v[e] + (float)o4[e];
  }
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
  *reinterpret_cast<bf16x4 *>(a.out + m * a.ldo + n) = o;
}

static void launch_skinny(const DenseArgs &a, hipStream_t s) {
  const dim3 block(SKINNY_WAVES * 64);
  const unsigned gx = (unsigned)((a.N + 15) / 16);
  if (a.M <= 16) dense_bf16_skinny_kernel<1><<<dim3(gx, 1), block, 0, s>>>(a);
  else if (a.M <= 32) dense_bf16_skinny_kernel<2><<<dim3(gx, 1), block, 0, s>>>(a);
  // more rows: 64-row groups along grid y (the weights then come from L2 once per group), 32-row
  // groups while the 64-row ones would leave CUs without a workgroup (measured, BERT-base forward
  // at M = 128: 0.70 -> 0.60 ms; at M = 256 the narrower groups lose)
  else if ((int64_t)gx * ((a.M + 63) / 64) < 128)
    dense_bf16_skinny_kernel<2><<<dim3(gx, (unsigned)((a.M + 31) / 32)), block, 0, s>>>(a);
  else
    dense_bf16_skinny_kernel<4><<<dim3(gx, (unsigned)((a.M + 63) / 64)), block, 0, s>>>(a);
}

bool mfma_bf16_dense_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                               int64_t ldother, const void *x, const void *w, const void *out,
                               const void *other) {
  if (M < 1 || N < 4 || K < 64) return false;
  if (K % 64 != 0 || N % 4 != 0) return false;
  if (ldx % 8 != 0 || ldo % 4 != 0) return false;
  if (other && ldother % 4 != 0) return false;
  if (((uintptr_t)x | (uintptr_t)w) % 16 != 0) return false;
  if (((uintptr_t)out | (uintptr_t)(other ? other : out)) % 8 != 0) return false;
  return true;
}

template <int BM, int BN, int WM, int WN, int ACT>
static void launch_act(const DenseArgs &a0, hipStream_t s) {
  DenseArgs a = a0;
  a.tiles_m = (int)((a.M + BM - 1) / BM);
  a.tiles_n = (int)((a.N + BN - 1) / BN);
  a.group_m = knobs().group_m;
  constexpr int LDS = 2 * (BM + BN) * 128;
  static PerDeviceFlag attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void *)dense_bf16_kernel<BM, BN, WM, WN, ACT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  dense_bf16_kernel<BM, BN, WM, WN, ACT>
      <<<dim3((unsigned)(a.tiles_m * a.tiles_n)), WM * WN * 64, LDS, s>>>(a);
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const DenseArgs &a, hipStream_t s) {
  switch (a.act) {
    case GNLP_ACT_RELU: launch_act<BM, BN, WM, WN, GNLP_ACT_RELU>(a, s); break;
    case GNLP_ACT_GELU_ERF: launch_act<BM, BN, WM, WN, GNLP_ACT_GELU_ERF>(a, s); break;
    case GNLP_ACT_GELU_TANH: launch_act<BM, BN, WM, WN, GNLP_ACT_GELU_TANH>(a, s); break;
    default: launch_act<BM, BN, WM, WN, GNLP_ACT_IDENTITY>(a, s); break;
  }
}

bool mfma_dense_act_fused(int act) {
  return act == GNLP_ACT_IDENTITY || act == GNLP_ACT_RELU || act == GNLP_ACT_GELU_ERF ||
         act == GNLP_ACT_GELU_TANH;
}

static bool big_tiles(int64_t M, int64_t N) {
  const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
  const Knobs &kn = knobs();   // developer knobs, read once per process (kernels.h)
  // 256x256 tiles once they fill at least half of the 256 CUs (measured on T5-base, 192 tiles per
  // GEMM: +25 % end to end over 768 tiles of 128x128)
  return kn.gemm_tile ? (kn.gemm_tile == 256) : (t256 >= 128 && M >= 256 && N >= 256);
}

bool mfma_bf16_dense_takes_w4(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                              int64_t ldother, const void *out, const void *bias,
                              const void *other) {
  return big_tiles(M, N) && knobs().gemm_w4 &&
         mfma_bf16_dense_w4_supported(M, N, K, ldx, ldo, ldother, out, bias, other);
}

void launch_mfma_bf16_dense(void *out, const void *x, const void *w, const void *bias,
                            const void *other, int combine, int act, int64_t M, int64_t N,
                            int64_t K, int64_t ldx, int64_t ldo, int64_t ldother, hipStream_t s) {
  DenseArgs a;
  a.x = (const bf16 *)x; a.w = (const bf16 *)w; a.bias = (const bf16 *)bias;
  a.other = (const bf16 *)other; a.out = (bf16 *)out;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.ldother = ldother;
  a.act = act; a.combine = combine; a.tiles_m = a.tiles_n = 0;
  const bool big = big_tiles(M, N);
  // persistent, one wave per SIMD (gemm_bf16_w4.hip) when the shape allows; otherwise (ragged M / N,
  // N > 8192, or the developer knob GNLP_GEMM_W4=0) one 256x256 tile per workgroup with clamped
  // edges (gemm_bf16_pp.hip) -- same accumulation order and epilogue, bit-identical results
  if (mfma_bf16_dense_takes_w4(M, N, K, ldx, ldo, ldother, out, bias, other))
    launch_mfma_bf16_dense_w4(out, x, w, bias, other, combine, act, M, N, K, ldx, ldo, ldother, s);
  else if (big && mfma_bf16_dense_pp_supported(N, ldo, ldother, out, bias, other))
    launch_mfma_bf16_dense_pp(out, x, w, bias, other, combine, act, M, N, K, ldx, ldo, ldother, s);
  else if (M <= knobs().skinny_max_m && M <= 64 * 65535LL && knobs().gemm_tile == 0)
    launch_skinny(a, s);     // a few rows: decoding steps
  else
    launch_cfg<128, 128, 2, 2>(a, s);
}

}  // namespace gnlp
