// bf16 MFMA dense layer, 256x256x64 "ping-pong" schedule for gfx950 (the large-GEMM hot path:
// QKV / attention_proj / FFN-1 / FFN-2 of models/transformer.py:193-206,261,266 and
// layers.py:590-612 at BERT shapes).
//
// 8 waves = 2 (M) x 4 (N), each owning a 128x64 output block (128 fp32 accumulators per lane).
// Waves 0-3 (upper row half) and waves 4-7 (lower half) share the 4 SIMDs pairwise and run the
// same program shifted by ONE barrier: while one group issues its ds_read_b128 fragment loads and
// its direct-to-LDS prefetches, the other group's 16-MFMA cluster owns the matrix pipe.
// A K-tile (64) is consumed in 2 phases of 32 MFMAs (two 64x32 output quadrants each):
//   p0: read B-sub0, B-sub1 (8 x b128) + A-sub0 (8) -> Q(0,0), Q(0,1)   stage A0, A1 (t+2)
//   p1: read A-sub1 (8)                              -> Q(1,1), Q(1,0)   stage B0, B1 (t+2),
//                                                                        s_waitcnt vmcnt(8)
// (a 4-phase split with 16-MFMA clusters measured 4 % slower: twice the barriers)
// LDS = 3 A buffers + 2 B buffers of one K-tile each = all 160 KiB; a buffer is two half-tiles of
// 16 KiB (128 rows x 128 B, 16-byte chunk index XOR ((row>>1)&7), applied on the global SOURCE
// address and on the ds_read address).  The third A buffer lets BOTH operands of K-tile t+2 be
// requested during K-tile t, so every direct-to-LDS load has >= 4 barrier intervals (one K-tile of
// MFMA time) to land before its counted wait; with two A buffers the A halves had 1-2 intervals.
// Each half-tile is filled by 2 global_load_lds_dwordx4 per lane; a slot is re-staged only after
// the last ds_read of it has retired behind a barrier (B halves after p0, A halves after p1 of
// both groups), and a staged slot is read only after the counted vmcnt of EVERY wave followed by a
// barrier -- prefetches stay in flight across barriers (never vmcnt(0) in the steady state).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace gnlp {

struct DensePPArgs {
  const bf16 *x, *w, *bias, *other;
  bf16 *out;
  int64_t M, N, K, ldx, ldo, ldother;
  int combine;
  int tiles_m, tiles_n, group_m;
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

#define GNLP_BAR()                                  \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    __builtin_amdgcn_s_barrier();                   \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
#define GNLP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int ACT>
__global__ void __launch_bounds__(512)
dense_bf16_pp_kernel(DensePPArgs a) {
  constexpr int HALF = 128 * 128;        // bytes of one half-tile (128 rows x 64 bf16)
  constexpr int ABUF = 2 * HALF;         // A0 A1 of one K-tile; 3 of them at offset 0
  constexpr int BBUF = 2 * HALF;         // B0 B1 of one K-tile; 2 of them behind the A buffers
  constexpr int BBASE = 3 * ABUF;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- XCD-aware, grouped tile mapping (see gemm_bf16.hip) ----
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int GROUP_M = a.group_m;
  const int band = tile / (GROUP_M * a.tiles_n);
  const int band_m0 = band * GROUP_M;
  const int band_h = (a.tiles_m - band_m0) < GROUP_M ? (a.tiles_m - band_m0) : GROUP_M;
  const int in_band = tile - band * GROUP_M * a.tiles_n;
  const int tm = band_m0 + in_band % band_h, tn = in_band / band_h;
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // group = wm

  // ---- staging: wave stages pieces 2w, 2w+1 (8 rows each) of every half-tile ----
  const int srow = lane >> 3, cpos = lane & 7;
  const bf16 *gA[2][2];   // [half][piece]
  const bf16 *gB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = (wave * 2 + t) * 8 + srow;          // row inside the half-tile
      const int c = cpos ^ ((row >> 1) & 7);
      int64_t m = m0 + h * 128 + row;
      if (m > a.M - 1) m = a.M - 1;
      gA[h][t] = a.x + m * a.ldx + c * 8;
      int64_t n = n0 + h * 128 + row;
      if (n > a.N - 1) n = a.N - 1;
      gB[h][t] = a.w + n * a.K + c * 8;
    }
  const int piece_off = wave * 2048;   // two consecutive 1-KiB pieces
  auto stage_a = [&](char *abuf, int h, int64_t k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_global_load_lds((gptr_t)(gA[h][t] + k0),
                                       (lptr_t)(abuf + h * HALF + piece_off + t * 1024), 16, 0, 0);
  };
  auto stage_b = [&](char *bbuf, int h, int64_t k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_global_load_lds((gptr_t)(gB[h][t] + k0),
                                       (lptr_t)(bbuf + h * HALF + piece_off + t * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets ----
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int frag_off = fr * 128 + ((fq ^ sw) << 4);
  const int a_off = wm * HALF + frag_off;                                   // + sub*64 rows
  const int b_off = (wn >> 1) * HALF + (wn & 1) * 64 * 128 + frag_off;      // + sub*32 rows

  f32x4 acc[4][8];   // [n-frag (2 subs x 2)][m-frag (2 subs x 4)]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 xf[4][2];      // current A-sub: [m-frag][k-step]
  bf16x8 wf[2][2][2];   // B-sub0 / B-sub1: [sub][n-frag][k-step]

  auto read_a = [&](const char *buf, int sub) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        xf[i][s] = *reinterpret_cast<const bf16x8 *>(
            buf + ((a_off + (sub * 64 + i * 16) * 128) ^ (s << 6)));
  };
  auto read_b = [&](const char *buf, int sub) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        wf[sub][j][s] = *reinterpret_cast<const bf16x8 *>(
            buf + ((b_off + (sub * 32 + j * 16) * 128) ^ (s << 6)));
  };
  auto mma = [&](int msub, int nsub) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[nsub * 2 + j][msub * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              wf[nsub][j][s], xf[i][s], acc[nsub * 2 + j][msub * 4 + i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nk = (int)(a.K >> 6);
  // ---- prologue: K-tiles 0 and 1 ----
  stage_a(smem, 0, 0);
  stage_a(smem, 1, 0);
  stage_b(smem + BBASE, 0, 0);
  stage_b(smem + BBASE, 1, 0);
  if (nk > 1) {
    stage_a(smem + ABUF, 0, 64);
    stage_a(smem + ABUF, 1, 64);
    stage_b(smem + BBASE + BBUF, 0, 64);
    stage_b(smem + BBASE + BBUF, 1, 64);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  GNLP_BAR();
  // Waves w and w + 4 share a SIMD (measured: staggering by wave parity instead is 25 % slower,
  // no stagger 15 % slower), so the late group is the lower row half, waves 4-7.
  if (wm == 1) GNLP_BAR();   // stagger the late group by one barrier

  int ai = 0;   // t % 3
  for (int t = 0; t < nk; ++t) {
    char *ca = smem + ai * ABUF;                                   // A of K-tile t
    char *na = smem + (ai == 0 ? 2 : ai - 1) * ABUF;               // A of K-tile t+2 (= t-1's)
    char *cbb = smem + BBASE + (t & 1) * BBUF;                     // B of K-tile t and of t+2
    const bool has2 = (t + 2 < nk);
    const int64_t k2 = (int64_t)(t + 2) << 6;
    read_b(cbb, 0);
    read_b(cbb, 1);
    read_a(ca, 0);
    if (has2) { stage_a(na, 0, k2); stage_a(na, 1, k2); }
    GNLP_LGKM0();
    GNLP_BAR();
    mma(0, 0);
    mma(0, 1);
    GNLP_BAR();
    read_a(ca, 1);
    if (has2) {
      stage_b(cbb, 0, k2);
      stage_b(cbb, 1, k2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // K-tile t+1 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    GNLP_LGKM0();
    GNLP_BAR();
    mma(1, 1);
    mma(1, 0);
    GNLP_BAR();
    ai = (ai == 2) ? 0 : ai + 1;
  }
  if (wm == 0) GNLP_BAR();   // balance the stagger barrier

  // ---- epilogue ----
  // The accumulator fragments (lane = one row x 4 columns) go through the now idle LDS so that
  // every global access is a full 128-byte row segment: each wave owns a private 16 KiB fp32
  // image [64 rows][64 cols] (16-byte chunk index XOR (row & 15): conflict-free ds_write_b128),
  // filled and drained twice (rows 0-63, 64-127 of the wave's block).  On the way out a lane
  // holds 8 consecutive columns of one row: bias is loaded once, residual / gate and the output
  // move as 16-byte accesses, 8 lanes per 128-byte line.
  const bool has_bias = a.bias != nullptr, has_other = a.other != nullptr;
  char *cw = smem + wave * 16384;
  const int c8 = lane & 7, rr8 = lane >> 3;
  const int64_t n = n0 + wn * 64 + c8 * 8;
  const bool n_ok = n < a.N;            // N % 8 == 0 is guaranteed by the dispatcher
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (has_bias && n_ok) {
    bf16x8 b8 = *reinterpret_cast<const bf16x8 *>(a.bias + n);
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (float)b8[e];
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // residual / gate rows of this pass: all 8 loads in flight before the LDS round trip
    bf16x8 o8[8];
    if (has_other) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        int64_t m = m0 + wm * 128 + p * 64 + it * 8 + rr8;
        m = m < a.M ? m : a.M - 1;
        o8[it] = *reinterpret_cast<const bf16x8 *>(a.other + m * a.ldother + (n_ok ? n : 0));
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4 *>(cw + (i * 16 + fr) * 256 + (((j * 4 + fq) ^ fr) << 4)) =
            acc[j][p * 4 + i];
    GNLP_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + rr8;
      const char *rp = cw + row * 256;
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(rp + (((2 * c8) ^ (row & 15)) << 4));
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(rp + (((2 * c8 + 1) ^ (row & 15)) << 4));
      float v[8];
      if (ACT == GNLP_ACT_GELU_ERF) {   // packed-math form: 2 elements per VALU instruction
        f32x2 g[4];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          g[e >> 1] = (f32x2){lo[e] + bv[e], lo[e + 1] + bv[e + 1]};
          g[2 + (e >> 1)] = (f32x2){hi[e] + bv[4 + e], hi[e + 1] + bv[5 + e]};
        }
        gelu_erf_fast2_n<4>(g);
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[2 * c] = g[c][0]; v[2 * c + 1] = g[c][1]; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = apply_act<false>(lo[e] + bv[e], ACT);
          v[4 + e] = apply_act<false>(hi[e] + bv[4 + e], ACT);
        }
      }
      const int64_t m = m0 + wm * 128 + p * 64 + row;
      const bool ok = n_ok && (m < a.M);
      if (has_other) {
        if (a.combine == GNLP_COMBINE_MUL) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= (float)o8[it][e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)o8[it][e];
        }
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
      if (ok) *reinterpret_cast<bf16x8 *>(a.out + m * a.ldo + n) = o;
    }
    GNLP_LGKM0();   // the wave's reads are done before the image is overwritten
    __builtin_amdgcn_sched_barrier(0);
  }
}

bool mfma_bf16_dense_pp_supported(int64_t N, int64_t ldo, int64_t ldother, const void *out,
                                  const void *bias, const void *other) {
  if (N % 8 != 0 || ldo % 8 != 0) return false;
  if (other && ldother % 8 != 0) return false;
  if (((uintptr_t)out | (uintptr_t)(other ? other : out) | (uintptr_t)(bias ? bias : out)) % 16)
    return false;
  return true;
}

template <int ACT>
static void launch_pp_act(const DensePPArgs &a, hipStream_t s) {
  constexpr int LDS = 5 * 2 * 128 * 128;   // 3 A + 2 B K-tile buffers = 160 KiB
  static PerDeviceFlag attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void *)dense_bf16_pp_kernel<ACT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  dense_bf16_pp_kernel<ACT><<<dim3((unsigned)(a.tiles_m * a.tiles_n)), 512, LDS, s>>>(a);
}

void launch_mfma_bf16_dense_pp(void *out, const void *x, const void *w, const void *bias,
                               const void *other, int combine, int act, int64_t M, int64_t N,
                               int64_t K, int64_t ldx, int64_t ldo, int64_t ldother,
                               hipStream_t s) {
  DensePPArgs a;
  a.x = (const bf16 *)x; a.w = (const bf16 *)w; a.bias = (const bf16 *)bias;
  a.other = (const bf16 *)other; a.out = (bf16 *)out;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.ldother = ldother;
  a.combine = combine;
  a.tiles_m = (int)((M + 255) / 256);
  a.tiles_n = (int)((N + 255) / 256);
  a.group_m = knobs().group_m;
  switch (act) {
    case GNLP_ACT_RELU: launch_pp_act<GNLP_ACT_RELU>(a, s); break;
    case GNLP_ACT_GELU_ERF: launch_pp_act<GNLP_ACT_GELU_ERF>(a, s); break;
    case GNLP_ACT_GELU_TANH: launch_pp_act<GNLP_ACT_GELU_TANH>(a, s); break;
    default: launch_pp_act<GNLP_ACT_IDENTITY>(a, s); break;
  }
}

}  // namespace gnlp
