// bf16 MFMA dense layer, PERSISTENT 256x256x64 ping-pong kernel for gfx950 (the large-GEMM hot
// path: QKV / attention_proj / FFN-1 / FFN-2 of models/transformer.py:193-206,261,266 and
// layers.py:590-612 at BERT shapes, where M, N are multiples of 256).
//
// Why persistent: with one 512-thread workgroup per CU (the tile needs all of LDS and 220 VGPRs)
// a one-tile-per-workgroup launch leaves the matrix pipes idle while a tile is written out, the
// next workgroup is dispatched and its first K-tiles arrive from L2/HBM -- measured 30-35 % of the
// GEMM time at K = 1024 (16 K-tiles per tile).  Here one workgroup per CU walks its tiles as ONE
// stream of K-tiles: the direct-to-LDS prefetch runs across tile boundaries, and a finished tile
// is written out from inside the memory phases of the stream, i.e. while the partner wave of the
// SIMD owns the matrix pipe.
//
// Schedule (same ping-pong as gemm_bf16_pp.hip): 8 waves = 2 (M) x 4 (N), each a 128x64 block.
// Waves w and w+4 share a SIMD and run the same program shifted by ONE barrier.  Per K-tile g:
//   P0 mem: ds_read B-sub0/1 + A-sub0, stage A(g+1) rows 64-127 of each half (2 LDS-DMA / wave)
//                                                        [drain rows 64-127 of the previous tile]
//   P0 mma: Q(0,0) Q(0,1)                                 (32 MFMA 16x16x32)
//   P1 mem: ds_read A-sub1, stage A(g+2) rows 0-63 + B(g+2) (6 LDS-DMA / wave), counted vmcnt
//                                                        [drain rows 0-63 if g ends the tile]
//   P1 mma: Q(1,1) Q(1,0)
// The accumulators of rows 0-63 get their last MFMA in P0 of the tile's last K-tile and are first
// needed again in P0 of the next tile, those of rows 64-127 in P1 / P1: each half has exactly one
// memory phase in which it is final and not yet reused, and that is where it is drained.
//
// Drain: per 16-row slab the fragments (lane = one row x 4 columns) go through a 4 KiB LDS image
// (fp32 [16][64], 16-byte chunk index XOR row) and come back as 8 consecutive columns per lane:
// bias, activation, residual / gate and the bf16 store then move whole 128-byte lines.  Waves w
// and w+4 drain in different barrier intervals and share one image.  The bias vector is copied
// to LDS once per workgroup, so a drain issues no global load except the residual rows.
// LDS: 2 K-tile buffers x {A0 A1 B0 B1} x 16 KiB + 4 x 4 KiB images + 16 KiB bias = 160 KiB.
//
// Staging addresses are a uniform 64-bit tile base plus a per-lane 32-bit offset that does not
// depend on the tile (M, N multiples of 256: no clamping), so crossing a tile costs no VGPRs.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace gnlp {

struct DensePSArgs {
  const bf16 *x, *w, *bias, *other;
  bf16 *out;
  int64_t ldx, ldo, ldother;
  int K, nk, combine, tiles_m, tiles_n, group_m;
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

template <int ACT, bool HAS_OTHER>
__global__ void __launch_bounds__(512)
dense_bf16_ps_kernel(DensePSArgs a) {
  constexpr int HALF = 128 * 128;        // bytes of one half-tile (128 rows x 64 bf16)
  constexpr int BUF = 4 * HALF;          // A0 A1 B0 B1
  constexpr int SCRATCH = 2 * BUF;       // 4 x 4 KiB transpose images behind the K-tile buffers
  constexpr int BIAS = SCRATCH + 16384;  // the whole bias vector (N <= 8192), filled once
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // group = wm

  // ---- the tiles of this workgroup: XCD x owns a contiguous range of the grouped tile order
  // (see gemm_bf16.hip), its workgroups take every S-th tile of it, so the 32 tiles an XCD has in
  // flight are neighbours (same L2) exactly as in the one-tile-per-workgroup launch ----
  const int nt = a.tiles_m * a.tiles_n;
  const int S = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q8 = nt >> 3, r8 = nt & 7;
  const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xcnt = q8 + (xcd < r8 ? 1 : 0);
  const int n_my = slot < xcnt ? (xcnt - slot + S - 1) / S : 0;
  if (n_my == 0) return;   // uniform over the workgroup
  auto coords = [&](int i, int &m0, int &n0) {
    const int tile = xbase + slot + i * S;
    const int GROUP_M = a.group_m;
    const int band = tile / (GROUP_M * a.tiles_n);
    const int band_m0 = band * GROUP_M;
    const int band_h = (a.tiles_m - band_m0) < GROUP_M ? (a.tiles_m - band_m0) : GROUP_M;
    const int in_band = tile - band * GROUP_M * a.tiles_n;
    m0 = (band_m0 + in_band % band_h) * 256;
    n0 = (in_band / band_h) * 256;
  };

  // ---- staging ----
  // A K-tile is staged in two steps so that both memory phases issue LDS-DMA (each instruction
  // costs its wave 55-60 cycles, and P0 -- 16 fragment reads -- is the longer phase): rows 0-63 of
  // each 128-row half ("sub 0", free once both groups have done their P0 reads) in P1 of the
  // K-tile TWO before, rows 64-127 ("sub 1", free after the P1 reads) in P0 of the K-tile before.
  // Wave w moves rows 8w..8w+7 of a sub; a B K-tile is moved in one step, pieces 2w, 2w+1.
  const int srow = lane >> 3, cpos = lane & 7;
  unsigned offA, offB[2];   // per-lane byte offsets inside a panel, tile-invariant
  {
    const int row = wave * 8 + srow;   // + 64 * sub: same swizzle (64 = 0 mod 16)
    offA = (unsigned)(row * (int)a.ldx + (cpos ^ ((row >> 1) & 7)) * 8) * 2u;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = (wave * 2 + t) * 8 + srow;
    const int c = cpos ^ ((row >> 1) & 7);
    offB[t] = (unsigned)(row * a.K + c * 8) * 2u;
  }
  const int piece_off = wave * 2048;   // two consecutive 1-KiB pieces
  const int64_t a_half = 128 * a.ldx, b_half = (int64_t)128 * a.K;   // elements
  const int64_t a_sub = 64 * a.ldx;
  // (the offsets pass through an empty asm so that "uniform base + lane offset" is formed at the
  // load -- SGPR-base addressing -- instead of hoisted 64-bit per-lane addresses per tile)
  auto stage_a = [&](char *buf, const bf16 *base, int sub) {   // base: uniform, &x[m0][k0]
    unsigned o0 = offA;
    asm volatile("" : "+v"(o0));
#pragma unroll
    for (int h = 0; h < 2; ++h)
      __builtin_amdgcn_global_load_lds(
          (gptr_t)((const char *)(base + h * a_half + sub * a_sub) + o0),
          (lptr_t)(buf + h * HALF + sub * 8192 + wave * 1024), 16, 0, 0);
  };
  auto stage_b = [&](char *buf, const bf16 *base) {   // base: uniform, &w[n0][k0]
    unsigned o0 = offB[0], o1 = offB[1];
    asm volatile("" : "+v"(o0), "+v"(o1));
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        __builtin_amdgcn_global_load_lds(
            (gptr_t)((const char *)(base + h * b_half) + (t ? o1 : o0)),
            (lptr_t)(buf + (2 + h) * HALF + piece_off + t * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets ----
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int frag_off = fr * 128 + ((fq ^ sw) << 4);
  const int a_off = wm * HALF + frag_off;                                     // + sub*64 rows
  const int b_off = (2 + (wn >> 1)) * HALF + (wn & 1) * 64 * 128 + frag_off;  // + sub*32 rows

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

  // ---- drain of one 64-row half of a finished tile (P = 0: rows 0-63, P = 1: rows 64-127) ----
  char *scr = smem + SCRATCH + (wave & 3) * 4096;
  const bool has_bias = a.bias != nullptr;
  // Global loads of the drains are inline asm with hand-counted vmcnt waits: behind a pending
  // direct-to-LDS load the compiler's own wait for a global load is always vmcnt(0), which would
  // drain the prefetch at every tile boundary.
  //  * bias: the tile's 8 values per lane are requested in the tile's first phase and have landed
  //    (any later counted wait covers them) long before the first drain of the tile;
  //  * residual / gate rows (HAS_OTHER): requested BEFORE the G staging loads of the drain's
  //    phase, into the fragment registers of the cluster that has just retired, and consumed
  //    after them.  Issue order: 8 rows, G staging loads, then 4 stores per slab pair, so both
  //    waits of a drain allow G + 4 younger operations.
  auto gload = [&](bf16x8 &dst, const bf16 *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
  };
  struct DrainIn {
    bf16x8 o8[4][2];
  };
  auto drain_loads = [&](auto Pc, int m0, int n0, DrainIn &d) {
    constexpr int P = decltype(Pc)::value;
    if (HAS_OTHER) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int c8 = ln & 7, rr8 = ln >> 3;
      const int64_t n = n0 + wn * 64 + c8 * 8;
      const int64_t mrow = m0 + wm * 128 + P * 64 + rr8;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int it = 0; it < 2; ++it)
          gload(d.o8[s][it], a.other + (mrow + s * 16 + it * 8) * a.ldother + n);
    }
  };
  auto drain = [&](auto Pc, auto Gc, int m0, int n0, DrainIn &d) {
    constexpr int P = decltype(Pc)::value;
    constexpr int G = decltype(Gc)::value;   // staging loads issued behind drain_loads: 6, 2, 0
    static_assert(G == 0 || G == 2 || G == 6, "wait counts are written out for these");
    // The lane-dependent addresses of the drain are recomputed here from an opaque copy of the
    // lane id: hoisted out of the K loop they would hold ~30 VGPRs for the whole kernel.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int c8 = ln & 7, rr8 = ln >> 3, fr = ln & 15, fq = ln >> 4;
    const int64_t n = n0 + wn * 64 + c8 * 8;
    const int64_t mrow = m0 + wm * 128 + P * 64 + rr8;
    const bf16x8 b8 =
        *reinterpret_cast<const bf16x8 *>(smem + BIAS + (n0 + wn * 64 + c8 * 8) * 2);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (float)b8[e];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {   // two slabs per LDS round trip
      if (HAS_OTHER) {   // allow the G staging loads + 4 younger rows / stores
        if (G == 6)
          asm volatile("s_waitcnt vmcnt(10)"
                       : "+v"(d.o8[2 * sp][0]), "+v"(d.o8[2 * sp][1]), "+v"(d.o8[2 * sp + 1][0]),
                         "+v"(d.o8[2 * sp + 1][1])::"memory");
        else if (G == 2)
          asm volatile("s_waitcnt vmcnt(6)"
                       : "+v"(d.o8[2 * sp][0]), "+v"(d.o8[2 * sp][1]), "+v"(d.o8[2 * sp + 1][0]),
                         "+v"(d.o8[2 * sp + 1][1])::"memory");
        else
          asm volatile("s_waitcnt vmcnt(4)"
                       : "+v"(d.o8[2 * sp][0]), "+v"(d.o8[2 * sp][1]), "+v"(d.o8[2 * sp + 1][0]),
                         "+v"(d.o8[2 * sp + 1][1])::"memory");
      }
      // One 4 KiB image per wave pair: LDS serves a wave's operations in issue order, so slab
      // q+1 may overwrite the image as soon as the reads of slab q have been ISSUED.
      f32x4 lo[2][2], hi[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int s = 2 * sp + q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          *reinterpret_cast<f32x4 *>(scr + fr * 256 + (((j * 4 + fq) ^ fr) << 4)) =
              acc[j][P * 4 + s];
          acc[j][P * 4 + s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + rr8;
          const char *rp = scr + row * 256;
          lo[q][it] = *reinterpret_cast<const f32x4 *>(rp + (((2 * c8) ^ row) << 4));
          hi[q][it] = *reinterpret_cast<const f32x4 *>(rp + (((2 * c8 + 1) ^ row) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);   // keep write -> read -> write -> read as written
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int s = 2 * sp + q;
          float v[8];
          if (ACT == GNLP_ACT_GELU_ERF) {   // packed-math form: 2 elements per VALU instruction
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const f32x2 g0 =
                  gelu_erf_fast2((f32x2){lo[q][it][e] + bv[e], lo[q][it][e + 1] + bv[e + 1]});
              const f32x2 g1 = gelu_erf_fast2(
                  (f32x2){hi[q][it][e] + bv[4 + e], hi[q][it][e + 1] + bv[5 + e]});
              v[e] = g0[0]; v[e + 1] = g0[1]; v[4 + e] = g1[0]; v[5 + e] = g1[1];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = apply_act<false>(lo[q][it][e] + bv[e], ACT);
              v[4 + e] = apply_act<false>(hi[q][it][e] + bv[4 + e], ACT);
            }
          }
          if (HAS_OTHER) {
            if (a.combine == GNLP_COMBINE_MUL) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= (float)d.o8[s][it][e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)d.o8[s][it][e];
            }
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          *reinterpret_cast<bf16x8 *>(a.out + (mrow + s * 16 + it * 8) * a.ldo + n) = o;
        }
    }
  };
  const std::integral_constant<int, 0> P0c;
  const std::integral_constant<int, 1> P1c;
  const std::integral_constant<int, 0> G0c;
  const std::integral_constant<int, 2> G2c;
  const std::integral_constant<int, 6> G6c;

  const int nk = a.nk;
  int m0c, n0c, m0n = 0, n0n = 0, m0p = 0, n0p = 0;
  coords(0, m0c, n0c);
  bool has_next = n_my > 1;
  if (has_next) coords(1, m0n, n0n);
  const bf16 *Ac = a.x + (int64_t)m0c * a.ldx, *Bc = a.w + (int64_t)n0c * a.K;
  const bf16 *An = a.x + (int64_t)m0n * a.ldx, *Bn = a.w + (int64_t)n0n * a.K;

  // ---- the bias vector goes to LDS once (zeros without a bias): no global load in the drains ----
  for (int i = threadIdx.x * 8; i < a.tiles_n * 256; i += 512 * 8) {
    bf16x8 b8;
    if (has_bias) {
      b8 = *reinterpret_cast<const bf16x8 *>(a.bias + i);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[e] = (bf16)0.f;
    }
    *reinterpret_cast<bf16x8 *>(smem + BIAS + i * 2) = b8;
  }
  // ---- prologue: K-tile 0 -> buffer 0; B and A sub 0 of K-tile 1 -> buffer 1 (nk >= 2) ----
  stage_b(smem, Bc);
  stage_a(smem, Ac, 0);
  stage_a(smem, Ac, 1);
  stage_b(smem + BUF, Bc + 64);
  stage_a(smem + BUF, Ac + 64, 0);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  GNLP_LGKM0();   // the bias image
  GNLP_BAR();
  // Waves w and w + 4 share a SIMD, so the late group is the lower row half, waves 4-7.
  if (wm == 1) GNLP_BAR();   // stagger the late group by one barrier

  int par = 0;
  bool pend = false;   // rows 64-127 of the previous tile still sit in the accumulators
  for (int ti = 0; ti < n_my; ++ti) {
    for (int t = 0; t < nk; ++t) {
      char *cb = smem + par * BUF;
      char *nb = smem + (par ^ 1) * BUF;
      // ---- P0: fragment reads B-sub0/1 + A-sub0; A(g+1) sub 1 -> other buffer ----
      if (t == 0 && pend) {   // (nk >= 2: K-tile 1 of this tile exists)
        DrainIn din;
        drain_loads(P1c, m0p, n0p, din);
        __builtin_amdgcn_sched_barrier(0);
        stage_a(nb, Ac + 64, 1);
        __builtin_amdgcn_sched_barrier(0);
        drain(P1c, G2c, m0p, n0p, din);
        pend = false;
        __builtin_amdgcn_sched_barrier(0);
        read_b(cb, 0);
        read_b(cb, 1);
        read_a(cb, 0);
      } else {
        read_b(cb, 0);
        read_b(cb, 1);
        read_a(cb, 0);
        if (t + 1 < nk) {
          stage_a(nb, Ac + (int64_t)(t + 1) * 64, 1);
        } else if (has_next) {
          stage_a(nb, An, 1);
        }
      }
      GNLP_LGKM0();
      GNLP_BAR();
      mma(0, 0);
      mma(0, 1);
      GNLP_BAR();
      // ---- P1: fragment reads A-sub1; A(g+2) sub 0 and B(g+2) -> this buffer ----
      // Counted wait: K-tile g+1 (B from P1 of g-1, A sub 0 likewise, A sub 1 from P0 above) must
      // have landed, the 6 loads of K-tile g+2 may stay in flight.  vmcnt retires in issue order,
      // so it is enough to allow as many operations as were issued after the A(g+1) loads; where
      // the count is not exact it errs towards waiting longer.
      if (t == nk - 1) {
        DrainIn din;
        drain_loads(P0c, m0c, n0c, din);
        __builtin_amdgcn_sched_barrier(0);
        // K-tile 1 of the next tile; after the last tile a harmless reload keeps the count
        stage_a(cb, has_next ? An + 64 : Ac, 0);
        stage_b(cb, has_next ? Bn + 64 : Bc);
        __builtin_amdgcn_sched_barrier(0);
        drain(P0c, G6c, m0c, n0c, din);   // 8 stores behind the 6 staging loads
        __builtin_amdgcn_sched_barrier(0);
        read_a(cb, 1);
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      } else {
        read_a(cb, 1);
        if (t + 2 < nk) {
          stage_a(cb, Ac + (int64_t)(t + 2) * 64, 0);
          stage_b(cb, Bc + (int64_t)(t + 2) * 64);
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (has_next) {
          stage_a(cb, An, 0);
          stage_b(cb, Bn);
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      GNLP_LGKM0();
      GNLP_BAR();
      mma(1, 1);
      mma(1, 0);
      GNLP_BAR();
      par ^= 1;
    }
    // ---- next tile ----
    m0p = m0c; n0p = n0c; pend = true;
    m0c = m0n; n0c = n0n; Ac = An; Bc = Bn;
    has_next = ti + 2 < n_my;
    if (has_next) {
      coords(ti + 2, m0n, n0n);
      An = a.x + (int64_t)m0n * a.ldx;
      Bn = a.w + (int64_t)n0n * a.K;
    }
  }
  // Rows 64-127 of the last tile.  The early group drains between its last barrier and the late
  // group's last barrier, the late group after it: the shared scratch is never used by both.
  {
    DrainIn din;
    drain_loads(P1c, m0p, n0p, din);
    drain(P1c, G0c, m0p, n0p, din);
  }
  if (wm == 0) {
    GNLP_LGKM0();
    GNLP_BAR();   // balance the stagger barrier
  }
}

bool mfma_bf16_dense_ps_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                                  int64_t ldother, const void *out, const void *bias,
                                  const void *other) {
  if (M % 256 != 0 || N % 256 != 0 || N > 8192 || K % 64 != 0 || K < 128) return false;
  if (ldo % 8 != 0 || (other && ldother % 8 != 0)) return false;
  if (((uintptr_t)out | (uintptr_t)(other ? other : out) | (uintptr_t)(bias ? bias : out)) % 16)
    return false;
  // per-lane staging offsets are 32-bit byte offsets inside a 256-row panel
  if (256 * ldx * 2 >= ((int64_t)1 << 31) || 256 * K * 2 >= ((int64_t)1 << 31)) return false;
  if ((M / 256) * (N / 256) >= ((int64_t)1 << 28)) return false;
  return true;
}

static int cu_count() {   // of the current device
  static std::atomic<int> cache[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = cache[dev & 63].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t p;
    n = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 0;
    if (n < 8) n = 256;
    cache[dev & 63].store(n, std::memory_order_relaxed);
  }
  return n;
}

template <int ACT, bool HAS_OTHER>
static void launch_ps_act(const DensePSArgs &a, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 128 + 4 * 4096 + 16384;   // 160 KiB
  static PerDeviceFlag attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void *)dense_bf16_ps_kernel<ACT, HAS_OTHER>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  const int nt = a.tiles_m * a.tiles_n;
  int grid = cu_count() & ~7;            // one workgroup per CU, a multiple of the 8 XCDs
  const int need = ((nt + 7) / 8) * 8;   // fewer tiles than CUs: one tile each
  if (grid > need) grid = need;
  dense_bf16_ps_kernel<ACT, HAS_OTHER><<<dim3((unsigned)grid), 512, LDS, s>>>(a);
}

template <int ACT>
static void launch_ps_other(const DensePSArgs &a, hipStream_t s) {
  if (a.other) launch_ps_act<ACT, true>(a, s);
  else launch_ps_act<ACT, false>(a, s);
}

void launch_mfma_bf16_dense_ps(void *out, const void *x, const void *w, const void *bias,
                               const void *other, int combine, int act, int64_t M, int64_t N,
                               int64_t K, int64_t ldx, int64_t ldo, int64_t ldother,
                               hipStream_t s) {
  DensePSArgs a;
  a.x = (const bf16 *)x; a.w = (const bf16 *)w; a.bias = (const bf16 *)bias;
  a.other = (const bf16 *)other; a.out = (bf16 *)out;
  a.ldx = ldx; a.ldo = ldo; a.ldother = ldother;
  a.K = (int)K; a.nk = (int)(K >> 6); a.combine = combine;
  a.tiles_m = (int)(M / 256);
  a.tiles_n = (int)(N / 256);
  a.group_m = knobs().group_m;
  switch (act) {
    case GNLP_ACT_RELU: launch_ps_other<GNLP_ACT_RELU>(a, s); break;
    case GNLP_ACT_GELU_ERF: launch_ps_other<GNLP_ACT_GELU_ERF>(a, s); break;
    case GNLP_ACT_GELU_TANH: launch_ps_other<GNLP_ACT_GELU_TANH>(a, s); break;
    default: launch_ps_other<GNLP_ACT_IDENTITY>(a, s); break;
  }
}

}  // namespace gnlp
