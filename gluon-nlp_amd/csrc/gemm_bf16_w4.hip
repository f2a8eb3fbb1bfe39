// bf16 MFMA dense layer, PERSISTENT 256x256x64 kernel with ONE wave per SIMD (4 waves, each a
// 128x128 block of the tile) for gfx950 -- the big GEMMs of the encoder layer (QKV /
// attention_proj / FFN-1 / FFN-2 of models/transformer.py:193-206,261,266 and layers.py:590-612).
//
// Why this shape (it is the shape of the vendor library's hand-written 256x256x64 kernel): with 8
// waves (128x64 each: round 1's persistent kernel, retired in round 2) a
// K-tile costs 192 KiB of LDS fragment reads and the two waves of a SIMD trade the matrix pipe at
// barriers.  With 4 waves of 128x128 the fragment reads
// drop to 128 KiB, each wave streams its 128 MFMAs per K-tile without handing the pipe over, and
// the memory instructions (32 ds_read_b128 + 16 LDS-DMA per wave and K-tile) ride in the issue
// slots between MFMAs.  256 accumulator registers + 128 fragment registers per lane: one wave per
// SIMD owns the whole 512-entry register file.  Measured (FFN-2 shape, GRBM_GUI_ACTIVE): this
// kernel spends MORE cycles per K-tile than the 8-wave one (3400 vs 3030; 128 bare MFMAs take
// 2640) but the chip holds 2.05 GHz under it against 1.71 GHz: less data moved per flop, and the
// GEMMs of the forward are power-limited.  End to end: 3050 -> 3230 seq/s (BERT-large, L = 512).
//
// The loads are buffer_load ... lds: one resource descriptor per operand (base = the K-tile's first
// element), the lane's offset in a VGPR that never changes, the piece's offset in an SGPR that
// never changes -- no vector-ALU instruction per piece.
//
// K-tile g lives in LDS buffer X = g & 1, g + 1 in Y.  Per K-tile (MFMA index t = 0..127, k-step
// s = t / 64 uses the fragment registers fa[s], fb[s]; slots: W4Slots below):
//   t  0..14    ds_read A fragments of k-step 1 from X         (k-step 0 was read during g - 1)
//   t  20, 21   lgkmcnt(0), barrier 1: every wave has read all of X's A panel
//   t  22..42   LDS-DMA A(g+2) -> X (5 of 8 pieces of 1 KiB) between ds_read B fragments of
//               k-step 1 from X
//   t  50, 51   lgkmcnt(0), barrier 2: every wave has read all of X's B panel
//   t  52..64   LDS-DMA: the last 3 A pieces, the first 2 B pieces; t 65..84 nothing but MFMAs
//   t  85       one more B piece
//   t  91, 92   vmcnt(11) (all of g + 1 has landed for this wave; the 11 requests of THIS K-tile
//               may fly on), barrier 3: Y is complete
//   t  93..123  ds_read A and B fragments of k-step 0 of g + 1 from Y, the last 5 B pieces
// so a K-tile is requested almost two K-tiles before its first use out of two 64 KiB buffers.
// (256x192 tiles, NJ = 6: 96 MFMAs, k-step s = t / 48, 6 weight pieces / fragments, the same
// shape of schedule: W4Slots<6>.)
// The stream of K-tiles runs across the tiles of the workgroup; between the last K-tile of a tile
// and the first of the next the 256 accumulators are drained (bias, activation, residual, bf16
// store in full 128-byte lines through a 4 KiB LDS transpose image per wave) while the next
// tile's first two K-tiles are already in flight.
//
// Same arithmetic, in the same order, as the one-tile-per-workgroup kernel gemm_bf16_pp.hip:
// results are bit-identical to its (tests/test_gpu_ops.py).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace gnlp {

#ifdef W4_STAMP   // diagnostic build only (tools/w4_stamp.py): where a K-tile's cycles go
__device__ unsigned w4_dbg[256 * 4 * 16];
#define W4_STAMP_AT(k)                                                      \
  do {                                                                      \
    W4_SB();                                                                \
    const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();           \
    if ((k) == 0) { if (st_v) { st_d[7] += now_ - st_last; ++st_n7; } st_v = 1; ++st_n; } \
    else st_d[(k) - 1] += now_ - st_last;                                   \
    st_last = now_;                                                         \
    W4_SB();                                                                \
  } while (0)
#else
#define W4_STAMP_AT(k) do {} while (0)
#endif

struct DenseW4Args {
  const bf16 *x, *w, *bias, *other;
  bf16 *out;
  int64_t ldx, ldo, ldother;
  int K, nk, combine, tiles_m, tiles_n, group_m;
  const int32_t *rows_dev;   // or null: the number of rows to process (a multiple of 256, at most
                             // 256 * tiles_m) is read from device memory (packed ragged batches)
  // LayerNorm folded into this GEMM (FOLD != 0; kernels.h DenseFold):
  const float *rstats;       // (rows, 2): r = rsqrt(var + eps) and r * mean of the row's raw sum
  const float *fold_c, *fold_b;   // FOLD == 1: c[n] = sum_k w'[n, k], b'[n] = beta . w[n, :] + b[n]
  const bf16 *ln_g, *ln_b;        // FOLD == 2: gamma / beta of the norm `other` goes through
  float *stats_part;              // FOLD == 2: (rows, 4 * tiles_n, 2) partial (sum, sum of squares)
                                  // of the OUTPUT rows, one slot per 64-column group of a tile
                                  // (plain stores, no atomics: a fixed summation order downstream)
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <class F, int... I>
__device__ __forceinline__ void w4_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void w4_static_for(F &&f) {
  w4_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Where the memory instructions of a K-tile sit between its MFMAs (index of the MFMA they follow).
// Shape of the schedule (read off the vendor library's hand-written kernel of the same tile, then
// varied on the forward: +2.6 % over one piece every 4 MFMAs with everything requested before
// barrier 3): requests come in short bursts with a quiet stretch of ~20 MFMAs in the middle, and
// the last weight pieces are requested AFTER barrier 3, so that the wait in front of it covers
// only requests that are at least a K-tile old.
template <int NJ> struct W4Slots;
template <> struct W4Slots<8> {
  static constexpr int RA1[8] = {0, 2, 4, 6, 8, 10, 12, 14};        // A fragments, k-step 1, of X
  static constexpr int BAR1 = 21;                                   // X's A panel is free
  static constexpr int SA[8] = {22, 25, 28, 31, 34, 52, 55, 58};    // A(g+2) -> X
  static constexpr int RB1[8] = {24, 27, 30, 33, 36, 38, 40, 42};   // B fragments, k-step 1, of X
  static constexpr int BAR2 = 51;                                   // X's B panel is free
  static constexpr int SB[8] = {61, 64, 85, 96, 99, 107, 110, 124}; // B(g+2) -> X, 5 after barrier 3
  static constexpr int BAR3 = 92;                                   // Y (K-tile g+1) is complete
  static constexpr int R0[16] = {93, 94, 95, 97, 98, 102, 103, 104,           // A, k-step 0, of Y
                                 105, 106, 109, 112, 114, 117, 120, 123};     // B, k-step 0, of Y
};
template <> struct W4Slots<6> {   // 96 MFMAs, k-step 1 from t = 48, 6 weight pieces / fragments
  static constexpr int RA1[8] = {0, 2, 4, 6, 8, 10, 12, 14};
  static constexpr int BAR1 = 17;
  static constexpr int SA[8] = {18, 21, 24, 27, 30, 40, 43, 46};
  static constexpr int RB1[6] = {20, 23, 26, 29, 32, 34};
  static constexpr int BAR2 = 39;
  static constexpr int SB[6] = {49, 52, 65, 73, 76, 93};            // 3 after barrier 3
  static constexpr int BAR3 = 70;
  static constexpr int R0[14] = {71, 72, 74, 75, 77, 78, 79, 80, 81, 82, 84, 86, 88, 90};
};
constexpr int w4_count_le(const int *tab, int n, int t) {
  int c = 0;
  for (int q = 0; q < n; ++q) c += tab[q] <= t ? 1 : 0;
  return c;
}
constexpr int w4_find(const int *tab, int n, int t) {
  for (int q = 0; q < n; ++q)
    if (tab[q] == t) return q;
  return -1;
}

// sum over the 8 lanes that share (lane >> 3): xor 1, xor 2 inside the quad, then the mirrored quad
__device__ __forceinline__ float w4_oct_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
  return x;
}

// weight row (inside a block of 32) that lands in LDS row x of the block in the DIRECT drain's
// instantiations: bits (b4 b3 b2 b1 b0) -> (b3 b2 b4 b1 b0), so that MFMA output row
// 16 (j & 1) + 4 g + e of weight fragment j is column 8 g + 4 (j & 1) + e of the block
__device__ __forceinline__ int w4_perm5(int x) {
  return (x & 3) | ((x & 16) >> 2) | ((x & 12) << 1);
}

#define W4_SB() __builtin_amdgcn_sched_barrier(0)
#define W4_BAR()                       \
  do {                                 \
    W4_SB();                           \
    __builtin_amdgcn_s_barrier();      \
    W4_SB();                           \
  } while (0)
#define W4_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// NJ = 16-column fragments per wave: 8 = 256x256 tiles, 6 = 256x192 tiles (each wave 128x96; the
// B half-panels keep their 16 KiB slots and use 12 KiB of them).  The narrower tile exists for
// N = 768 / 2304 (BERT-base, T5-base): 384 tiles of 256x256 are two rounds over 256 CUs with half
// of them idle in the second, 512 tiles of 256x192 are two full rounds of 3/4 the work.
// FOLD (LayerNorm folded into the GEMMs around it, bf16 post-norm stacks):
//   1: x holds the RAW pre-norm sums s and w = bf16(gamma * W); the drain applies the row's
//      statistics: out = act(r * acc - (r mu) * c[n] + b'[n])  ==  act(LN(s) . W^T + b)
//   2: `other` holds raw sums s too and the residual is LN(s), rebuilt in the drain in fp32:
//      out = acc + bias + gamma * (r s - r mu) + beta
template <int ACT, bool HAS_OTHER, int NJ, int FOLD>
__global__ void __launch_bounds__(256)
dense_bf16_w4_kernel(DenseW4Args a) {
#if __HIP_DEVICE_COMPILE__   // (the buffer-resource builtins exist in the device pass only)
  static_assert(NJ == 8 || NJ == 6, "256- or 192-wide tiles");
  static_assert(FOLD == 0 || (FOLD == 1 && !HAS_OTHER) || (FOLD == 2 && HAS_OTHER), "fold modes");
  // DIRECT: the drain stores straight from the accumulator layout (no LDS transpose): the
  // instantiations with an activation, whose drains are bound by their vector-ALU work.  The
  // identity drains are bound by the memory system (every CU stores at once) and keep the
  // transpose with its whole-line stores: measured, DESIGN.md section 6c.
  constexpr bool DIRECT = ACT != GNLP_ACT_IDENTITY;
  static_assert(!(DIRECT && FOLD == 2), "FOLD 2 is an identity drain");
  constexpr int NT = 32 * NJ;            // tile width
  constexpr int NQ = NJ / 2;             // 32-row pieces of a B half-panel per wave
  constexpr int HALF = 128 * 128;        // bytes of one half-panel slot (128 rows x 64 bf16)
  constexpr int BUF = 4 * HALF;          // A0 A1 B0 B1 of one K-tile
  constexpr int SCRATCH = 2 * BUF;       // 4 x 4 KiB transpose images
  constexpr int BIAS = SCRATCH + 16384;  // the whole bias vector (N <= 8192)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- the tiles of this workgroup (XCD-aware band walk) ----
  const int tiles_m = a.rows_dev ? (*a.rows_dev >> 8) : a.tiles_m;
  const int nt = tiles_m * a.tiles_n;
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
    const int band_h = (tiles_m - band_m0) < GROUP_M ? (tiles_m - band_m0) : GROUP_M;
    const int in_band = tile - band * GROUP_M * a.tiles_n;
    m0 = (band_m0 + in_band % band_h) * 256;
    n0 = (in_band / band_h) * NT;
  };

  // ---- staging: wave w moves pieces w, w+4, w+8, w+12 (8 rows x 128 B each) of every half-panel,
  // so that the source-side XOR swizzle of a lane is the same for all of its pieces ----
  const int srow = lane >> 3, cpos = lane & 7;
  unsigned offA, offB;   // per-lane byte offsets inside a panel, tile- and piece-invariant
  {
    const int row = wave * 8 + srow;
    const int c = cpos ^ ((row >> 1) & 7);
    offA = (unsigned)(row * (int)a.ldx + c * 8) * 2u;
    // (DIRECT: permuted weight rows, see the drain; the swizzle stays keyed by the LDS row)
    offB = (unsigned)((DIRECT ? w4_perm5(row) : row) * a.K + c * 8) * 2u;
  }
  int soffA[8], soffB[NJ];   // byte offsets of the pieces inside a K-tile panel (uniform)
#pragma unroll
  for (int p = 0; p < 8; ++p) soffA[p] = (int)(((p >> 2) * 128 + (p & 3) * 32) * a.ldx * 2);
#pragma unroll
  for (int p = 0; p < NJ; ++p)
    soffB[p] = (int)(((p / NQ) * (NT / 2) + (p % NQ) * 32) * (int64_t)a.K * 2);
  auto rsrc = [](const bf16 *base) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
  };
  auto stage_a = [&](char *buf, const bf16 *base, int p) {   // p = 0..7: half p>>2, quarter p&3
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rsrc(base), (lptr_t)(buf + (p >> 2) * HALF + (wave + 4 * (p & 3)) * 1024), 16, offA,
        soffA[p], 0, 0);
  };
  auto stage_b = [&](char *buf, const bf16 *base, int p) {   // p = 0..NJ-1: half p/NQ, piece p%NQ
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rsrc(base), (lptr_t)(buf + (2 + p / NQ) * HALF + (wave + 4 * (p % NQ)) * 1024), 16, offB,
        soffB[p], 0, 0);
  };

  // ---- fragment reads ----
  const int fr = lane & 15, fq = lane >> 4;
  const int frag_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
  const int a_off = wm * HALF + frag_off;
  const int b_off = (2 + wn) * HALF + frag_off;

  f32x4 acc[NJ][8];   // [n-frag][m-frag]; first written by the tile's first K-tile
  bf16x8 fa[2][8], fb[2][NJ];   // [k-step][frag]

  auto rd_a = [&](const char *buf, int s, int i) {
    fa[s][i] = *reinterpret_cast<const bf16x8 *>(buf + ((a_off + i * 2048) ^ (s << 6)));
  };
  auto rd_b = [&](const char *buf, int s, int j) {
    fb[s][j] = *reinterpret_cast<const bf16x8 *>(buf + ((b_off + j * 2048) ^ (s << 6)));
  };

  // ---- one K-tile: 128 MFMAs with the memory instructions in between (see the header) ----
  // FIRST: the tile's first K-tile starts the accumulators from the inline constant 0 (no
  // zeroing pass over 256 registers per tile).
#ifdef W4_STAMP
  unsigned st_d[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0, st_n = 0, st_n7 = 0, st_v = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  auto ktile = [&](auto firstc, char *X, char *Y, const bf16 *pA,
                   const bf16 *pB) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;
    if constexpr (!FIRST) W4_STAMP_AT(0);
#ifdef W4_STAMP
    if constexpr (FIRST) st_v = 0;   // the K-tile after a drain is not a steady-state one
#endif
    w4_static_for<16 * NJ>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      constexpr int s = t / (8 * NJ), j = (t % (8 * NJ)) >> 3, i = t & 7;
      // (inline asm: the accumulators are pinned to the AGPR half of the register file; left to
      // the register allocator they wander between the halves, hundreds of copies per K-tile)
      if constexpr (FIRST && s == 0)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0"
                     : "=a"(acc[j][i])
                     : "v"(fb[s][j]), "v"(fa[s][i]));
      else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                     : "+a"(acc[j][i])
                     : "v"(fb[s][j]), "v"(fa[s][i]));
      // slot tables: the memory instruction issued after MFMA t (see the header)
      using S = W4Slots<NJ>;
      constexpr int ia = w4_find(S::SA, 8, t), ib = w4_find(S::SB, NJ, t),
                    ra1 = w4_find(S::RA1, 8, t), rb1 = w4_find(S::RB1, NJ, t),
                    r0 = w4_find(S::R0, 8 + NJ, t);
      if constexpr (ra1 >= 0) rd_a(X, 1, ra1);
      if constexpr (t == S::BAR1 - 1) { W4_SB(); W4_LGKM0(); }
      if constexpr (t == S::BAR1 && !FIRST) W4_STAMP_AT(1);
      if constexpr (t == S::BAR1) W4_BAR();
      if constexpr (t == S::BAR1 && !FIRST) W4_STAMP_AT(2);
      if constexpr (ia >= 0) stage_a(X, pA, ia);
      if constexpr (rb1 >= 0) rd_b(X, 1, rb1);
      if constexpr (t == S::BAR2 - 1) { W4_SB(); W4_LGKM0(); }
      if constexpr (t == S::BAR2 && !FIRST) W4_STAMP_AT(3);
      if constexpr (t == S::BAR2) W4_BAR();
      if constexpr (t == S::BAR2 && !FIRST) W4_STAMP_AT(4);
      if constexpr (ib >= 0) stage_b(X, pB, ib);
      if constexpr (t == S::BAR3 - 1) {   // the requests of THIS K-tile issued so far may fly on
        // (a larger allowance in a tile's first K-tile, for the drain's stores that are younger
        // than the K-tile waited for, measured +-0: rounds 1 and 3)
        constexpr int NFLY = w4_count_le(S::SA, 8, t) + w4_count_le(S::SB, NJ, t);
        if constexpr (!FIRST) W4_STAMP_AT(5);
        W4_SB();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFLY) : "memory");
        if constexpr (!FIRST) W4_STAMP_AT(6);
      }
      if constexpr (t == S::BAR3 && S::BAR3 != S::BAR2) W4_BAR();
      if constexpr (t == S::BAR3 && !FIRST) W4_STAMP_AT(7);
      if constexpr (r0 >= 0 && r0 < 8) rd_a(Y, 0, r0);
      if constexpr (r0 >= 8) rd_b(Y, 0, r0 - 8);
      W4_SB();
    });
  };

  // ---- drain of the whole 128x128 block of this wave ----
  // Residual / gate rows (HAS_OTHER): the drain reads them two slabs ahead through a ring of three
  // register slabs (+0.5 % on the forward; touching the tile's lines during its last K-tile so
  // that they sit in L2 when the drain asks measured 1.4 % SLOWER and was dropped).  The loads
  // are inline asm with hand-counted vmcnt waits: the compiler does not wait for them at all, and
  // its own wait behind a pending LDS-DMA would be vmcnt(0).  Order of vector-memory operations
  // of a drain: L0 L1 | L2 [wait L0] S0 | L3 [wait L1] S1 | ... | L7 [wait L5] S5 | [wait L6] S6 |
  // [wait L7] S7, four operations per L (slab loads) and per S (slab stores).
  char *scr = smem + SCRATCH + wave * 4096;
  auto gload = [&](bf16x8 &dst, const bf16 *p) __attribute__((always_inline)) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
  };
  auto gload_f4 = [&](f32x4 &dst, const float *p) __attribute__((always_inline)) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
  };
  constexpr bool RING = HAS_OTHER;
  constexpr int NL = 4;   // operations of one L group
  constexpr int RD = (FOLD == 2 || ACT == GNLP_ACT_GELU_TANH) ? 2 : 3;   // ring depth (registers)
  constexpr int NST = FOLD == 2 ? 8 : 4;  // stores of one S group (FOLD 2: + the rows' statistics)
  constexpr int WCOLS = NT / 2;
  bf16x8 ring[RD][2][2];     // residual / gate rows of RD slabs (HAS_OTHER)
  f32x2 rst[16];             // FOLD: (r, r mu) of the lane's 16 rows, m0 + 128 wm + (lane >> 3) + 8 j
  f32x4 fdc[2][2], fdb[2][2];  // FOLD == 1: c, b' of the lane's 8 columns in each 64-column group
                             // (FOLD == 2: bias + beta and gamma sit in LDS)
  auto slab_n = [&](int n0, int hn, int c8) -> int64_t {
    const int nj = NJ - hn * 4 < 4 ? NJ - hn * 4 : 4;
    const bool on = c8 * 8 < nj * 16;
    return n0 + wn * WCOLS + hn * 64 + (on ? c8 : 0) * 8;
  };
  auto slab_loads = [&](int m0, int n0, int ln, int k) __attribute__((always_inline)) {   // k = hn * 4 + sp
    const int c8 = ln & 7, rr8 = ln >> 3;
    const int hn = k >> 2, sp = k & 3;
    const int64_t n = slab_n(n0, hn, c8);
    const int64_t mrow = m0 + wm * 128 + sp * 32 + rr8;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        gload(ring[k % RD][q][it], a.other + (mrow + q * 16 + it * 8) * a.ldother + n);
      }
  };
  // The statistics of the lane's 16 rows and (FOLD 1) the per-column vectors c, b': loaded BEFORE
  // the tile's last K-tile, whose own vmcnt wait -- for requests that are younger -- sees them
  // land: the drain starts on data that is there, and its counted waits see only the residual ring.
  auto fold_head = [&](int m0, int n0) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if constexpr (DIRECT) {   // 8 rows per lane, m0 + 128 wm + (lane & 15) + 16 s; c, b' sit in LDS
      const float *rp = a.rstats + ((int64_t)m0 + wm * 128 + (ln & 15)) * 2;
#define W4_RST_LD(j)                                            \
  asm volatile("global_load_dwordx2 %0, %1, off offset:%2"      \
               : "=&v"(rst[j])                                  \
               : "v"(rp), "n"((j) * 128)                        \
               : "memory")
      W4_RST_LD(0); W4_RST_LD(1); W4_RST_LD(2); W4_RST_LD(3);
      W4_RST_LD(4); W4_RST_LD(5); W4_RST_LD(6); W4_RST_LD(7);
#undef W4_RST_LD
      return;
    }
    if constexpr (FOLD == 1) {
#pragma unroll
      for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          gload_f4(fdc[hn][h], a.fold_c + slab_n(n0, hn, ln & 7) + h * 4);
          gload_f4(fdb[hn][h], a.fold_b + slab_n(n0, hn, ln & 7) + h * 4);
        }
    }
    const float *rp = a.rstats + ((int64_t)m0 + wm * 128 + (ln >> 3)) * 2;
    // (one address, 16 immediate offsets; written out: asm operands cannot name captures of a
    // nested lambda)
#define W4_RST_LD(j)                                            \
  asm volatile("global_load_dwordx2 %0, %1, off offset:%2"      \
               : "=&v"(rst[j])                                  \
               : "v"(rp), "n"((j) * 64)                         \
               : "memory")
    W4_RST_LD(0); W4_RST_LD(1); W4_RST_LD(2); W4_RST_LD(3); W4_RST_LD(4); W4_RST_LD(5);
    W4_RST_LD(6); W4_RST_LD(7); W4_RST_LD(8); W4_RST_LD(9); W4_RST_LD(10); W4_RST_LD(11);
    W4_RST_LD(12); W4_RST_LD(13); W4_RST_LD(14); W4_RST_LD(15);
#undef W4_RST_LD
  };
  auto drain_head = [&](int m0, int n0) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if (RING) {
      slab_loads(m0, n0, ln, 0);
      if (RD == 3) slab_loads(m0, n0, ln, 1);
    }
  };
  auto drain = [&](int m0, int n0) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // wait for the loads of slab k: the operations younger than L(k) may stay in flight
    auto slab_wait = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      // RD 3: L0 L1 | L2 [wait L0] S0 | L3 [wait L1] S1 | ... | [wait L6] S6 | [wait L7] S7
      // RD 2: L0 | L1 [wait L0] S0 | L2 [wait L1] S1 | ... | [wait L7] S7    (S groups: NST stores)
      constexpr int FLY = RD == 3 ? (k == 0 ? 2 * NL : k == 1 ? 2 * NL + NST
                                     : k <= 5 ? 2 * NL + 2 * NST : k == 6 ? NL + 2 * NST : 2 * NST)
                                  : (k + 1 < 8 ? NL : 0) + (k >= 1 ? NST : 0);
      bf16x8(&r)[2][2] = ring[k % RD];
      asm volatile("s_waitcnt vmcnt(%4)"
                   : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1])
                   : "n"(FLY)
                   : "memory");
    };
    drain_head(m0, n0);
    w4_static_for<2>([&](auto hnc) __attribute__((always_inline)) {   // 64-column groups of the wave's 128 (96: 64 + 32) columns
      constexpr int hn = decltype(hnc)::value;
      if constexpr (FOLD == 2) {   // (lane-derived addresses are recomputed per group instead of
        asm volatile("" : "+v"(ln));   // living across the drain: this instantiation spilled them)
      }
      const int c8 = ln & 7, rr8 = ln >> 3, dfr = ln & 15, dfq = ln >> 4;
      const int nj = NJ - hn * 4 < 4 ? NJ - hn * 4 : 4;      // fragments of this group: 4 or 2
      const bool on = c8 * 8 < nj * 16;                      // lanes that own columns of it
      const int64_t n = n0 + wn * WCOLS + hn * 64 + (on ? c8 : 0) * 8;
      float bv[8];
      if constexpr (FOLD == 2) {   // fp32 image: bias + beta (gamma: 8 KiB further on)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(smem + BIAS + (n + h * 4) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[h * 4 + e] = b4[e];
        }
      } else {
        const bf16x8 b8 = *reinterpret_cast<const bf16x8 *>(smem + BIAS + n * 2);
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (float)b8[e];
      }
      w4_static_for<4>([&](auto spc) __attribute__((always_inline)) {   // two 16-row slabs per LDS round trip
        constexpr int sp = decltype(spc)::value;
        constexpr int k = hn * 4 + sp;
        const int64_t mrow = m0 + wm * 128 + sp * 32 + rr8;
        if constexpr (RING) {   // (lanes that are not `on` re-read column group 0: in bounds, unused)
          if constexpr (k + RD - 1 < 8) slab_loads(m0, n0, ln, k + RD - 1);
          slab_wait(std::integral_constant<int, k>{});
        }
        f32x4 lo[2][2], hi[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int s = 2 * sp + q;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < nj)
              *reinterpret_cast<f32x4 *>(scr + dfr * 256 + (((j * 4 + dfq) ^ dfr) << 4)) =
                  acc[hn * 4 + j < NJ ? hn * 4 + j : 0][s];
          }
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int row = it * 8 + rr8;
            const char *rp = scr + row * 256;
            lo[q][it] = *reinterpret_cast<const f32x4 *>(rp + (((2 * c8) ^ row) << 4));
            hi[q][it] = *reinterpret_cast<const f32x4 *>(rp + (((2 * c8 + 1) ^ row) << 4));
          }
          W4_SB();   // keep write -> read -> write -> read as written (LDS serves a wave in order)
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            float v[8];
            // pairs throughout (v_pk_add_f32 / v_cvt_pk_bf16_f32): left to itself the compiler
            // shuffles the identity epilogue through v_perm / v_alignbit (600 instead of 256
            // vector-ALU instructions per tile)
            f32x2 g[4];   // columns 0 1 | 2 3 | 4 5 | 6 7
            if constexpr (FOLD == 1) {   // r * acc + (b' - (r mu) c)
              const f32x2 st = rst[sp * 4 + q * 2 + it];
              const f32x2 r2 = {st[0], st[0]}, nm2 = {-st[1], -st[1]};
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                const f32x2 t0 = (f32x2){fdc[hn][0][e], fdc[hn][0][e + 1]} * nm2 +
                                 (f32x2){fdb[hn][0][e], fdb[hn][0][e + 1]};
                const f32x2 t1 = (f32x2){fdc[hn][1][e], fdc[hn][1][e + 1]} * nm2 +
                                 (f32x2){fdb[hn][1][e], fdb[hn][1][e + 1]};
                g[e >> 1] = (f32x2){lo[q][it][e], lo[q][it][e + 1]} * r2 + t0;
                g[2 + (e >> 1)] = (f32x2){hi[q][it][e], hi[q][it][e + 1]} * r2 + t1;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                g[e >> 1] = (f32x2){lo[q][it][e], lo[q][it][e + 1]} + (f32x2){bv[e], bv[e + 1]};
                g[2 + (e >> 1)] =
                    (f32x2){hi[q][it][e], hi[q][it][e + 1]} + (f32x2){bv[4 + e], bv[5 + e]};
              }
            }
            if (ACT == GNLP_ACT_GELU_ERF) {
              gelu_erf_fast2_n<4>(g);
            } else if (ACT != GNLP_ACT_IDENTITY) {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                g[c] = (f32x2){apply_act<false>(g[c][0], ACT), apply_act<false>(g[c][1], ACT)};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[2 * c] = g[c][0]; v[2 * c + 1] = g[c][1]; }
            if constexpr (HAS_OTHER) {
              const bf16x8 o8 = ring[k % RD][q][it];
              if constexpr (FOLD == 2) {   // + gamma * (r s - r mu)   (beta sits in the bias image)
                // (gamma is re-read from LDS row by row: held in 8 registers this instantiation
                // spills, and spill traffic would sit in the counted vmcnt queue)
                const f32x2 st = rst[sp * 4 + q * 2 + it];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const f32x4 g4 =
                      *reinterpret_cast<const f32x4 *>(smem + BIAS + 8192 + (n + h * 4) * 4);
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    v[h * 4 + e] = __builtin_fmaf(
                        g4[e], __builtin_fmaf(st[0], (float)o8[h * 4 + e], -st[1]), v[h * 4 + e]);
                }
              } else if (a.combine == GNLP_COMBINE_MUL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (float)o8[e];
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)o8[e];
              }
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
            if (NJ == 8 || on)
              *reinterpret_cast<bf16x8 *>(a.out + (mrow + q * 16 + it * 8) * a.ldo + n) = o;
            if constexpr (FOLD == 2) {
              // statistics of the output row for the LayerNorm the NEXT GEMMs fold in: this
              // lane's 8 columns, then the 8 lanes of the row; one (sum, sum of squares) per row
              // and 64-column group, summed in a fixed order by stats_finish_kernel
              float ps = 0.0f, pq = 0.0f;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                ps += v[e];
                pq = __builtin_fmaf(v[e], v[e], pq);
              }
              if (NJ != 8 && !on) { ps = 0.0f; pq = 0.0f; }
              ps = w4_oct_sum(ps);
              pq = w4_oct_sum(pq);
              const int64_t slot = (int64_t)(n0 / NT) * 4 + wn * 2 + hn;
              float *sp_ = a.stats_part + ((mrow + q * 16 + it * 8) * (4 * a.tiles_n) + slot) * 2;
              if (c8 == 0) *reinterpret_cast<f32x2 *>(sp_) = (f32x2){ps, pq};
              W4_SB();   // one row at a time: registers (no spills)
            }
          }
        if (RING) W4_SB();   // the slab's four stores are issued before the next slab's loads
      });
    });
  };
  // ---- DIRECT drain (activations): no transpose ----
  // The MFMA's A operand is the WEIGHT fragment, so lane (g = lane >> 4, r = lane & 15) holds, in
  // acc[j][s], row 16 s + r of the wave's block and the 4 columns 4 g .. 4 g + 3 of weight fragment
  // j; the staging permutes the weight rows inside every block of 32 (w4_perm5) so that fragments
  // 2 p and 2 p + 1 together give the lane the 8 CONSECUTIVE columns 32 p + 8 g .. + 7: one
  // 16-byte store per (s, p), 16 rows x 64 contiguous bytes per instruction.  The fp32 round trip
  // of the accumulators through LDS (512 KiB per tile and CU) is gone; the per-column vectors
  // (bias; FOLD 1: b' and c as fp32) come from LDS images written in the prologue.
  // Slab k = 4 hn + sp: rows 32 sp .. + 31 of the wave's block (row fragments s = 2 sp, 2 sp + 1) x
  // the 64-column group hn (pairs p = 2 hn, 2 hn + 1; NJ = 6: group 1 is pair 2 alone).
  constexpr int NP = NJ / 2;
  auto d_pairs = [](int k) constexpr { return NP - (k >> 2) * 2 < 2 ? NP - (k >> 2) * 2 : 2; };
  auto slab_loads_direct = [&](int m0, int n0, int ln, int k) __attribute__((always_inline)) {
    const int hn = k >> 2, sp = k & 3;
    const bf16 *op = a.other + ((int64_t)m0 + wm * 128 + sp * 32 + (ln & 15)) * a.ldother + n0 +
                     wn * WCOLS + hn * 64 + (ln >> 4) * 8;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
        if (pp < d_pairs(k)) gload(ring[k % RD][q][pp], op + q * 16 * a.ldother + pp * 32);
  };
  auto drain_direct = [&](int m0, int n0) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // wait for the loads of slab k: L(k + 1) .. L(k + RD - 1) and S(k - RD + 1) .. S(k - 1) may
    // stay in flight (2 d_pairs operations per L and per S)
    auto slab_wait = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      constexpr int FLY = [=]() constexpr {
        int n = 0;
        for (int j = k + 1; j <= k + RD - 1 && j < 8; ++j) n += 2 * d_pairs(j);
        for (int j = k - RD + 1 > 0 ? k - RD + 1 : 0; j < k; ++j) n += 2 * d_pairs(j);
        return n;
      }();
      bf16x8(&r)[2][2] = ring[k % RD];
      asm volatile("s_waitcnt vmcnt(%4)"
                   : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1])
                   : "n"(FLY)
                   : "memory");
    };
    if (RING) {
      slab_loads_direct(m0, n0, ln, 0);
      if (RD == 3) slab_loads_direct(m0, n0, ln, 1);
    }
    w4_static_for<2>([&](auto hnc) __attribute__((always_inline)) {   // 64-column groups
      constexpr int hn = decltype(hnc)::value;
      constexpr int PPN = NP - hn * 2 < 2 ? NP - hn * 2 : 2;          // pairs of this group: 2 or 1
      asm volatile("" : "+v"(ln));   // (lane-derived addresses are recomputed per group: registers)
      const int g = ln >> 4, r16 = ln & 15;
      const int nloc = wn * WCOLS + hn * 64 + g * 8;   // the lane's 8 columns of pair pp: + 32 pp
      f32x2 bv[2][4], cv[2][4];   // FOLD 0: bv = bias;  FOLD 1: bv = b', cv = c
#pragma unroll
      for (int pp = 0; pp < PPN; ++pp) {
        const int n = n0 + nloc + pp * 32;
        if constexpr (FOLD == 0) {
          const bf16x8 b8 = *reinterpret_cast<const bf16x8 *>(smem + BIAS + n * 2);
#pragma unroll
          for (int c = 0; c < 4; ++c) bv[pp][c] = (f32x2){(float)b8[2 * c], (float)b8[2 * c + 1]};
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(smem + BIAS + (n + h * 4) * 4);
            const f32x4 c4 = *reinterpret_cast<const f32x4 *>(smem + SCRATCH + (n + h * 4) * 4);
            bv[pp][2 * h] = (f32x2){b4[0], b4[1]}; bv[pp][2 * h + 1] = (f32x2){b4[2], b4[3]};
            cv[pp][2 * h] = (f32x2){c4[0], c4[1]}; cv[pp][2 * h + 1] = (f32x2){c4[2], c4[3]};
          }
        }
      }
      w4_static_for<4>([&](auto spc) __attribute__((always_inline)) {   // 32-row slabs
        constexpr int sp = decltype(spc)::value;
        constexpr int k = hn * 4 + sp;
        if constexpr (RING) {
          if constexpr (k + RD - 1 < 8) slab_loads_direct(m0, n0, ln, k + RD - 1);
          slab_wait(std::integral_constant<int, k>{});
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int s = 2 * sp + q;
          const int64_t mrow = (int64_t)m0 + wm * 128 + s * 16 + r16;
          [[maybe_unused]] f32x2 r2, nm2;
          if constexpr (FOLD == 1) {
            r2 = (f32x2){rst[s][0], rst[s][0]};
            nm2 = (f32x2){-rst[s][1], -rst[s][1]};
          }
          bf16 *orow = a.out + mrow * a.ldo + n0 + nloc;
#pragma unroll
          for (int pp = 0; pp < PPN; ++pp) {
            const int p = hn * 2 + pp;
            const f32x4 a0 = acc[2 * p][s], a1 = acc[2 * p + 1][s];
            // pairs throughout (v_pk_fma_f32 / v_cvt_pk_bf16_f32)
            f32x2 gg[4] = {(f32x2){a0[0], a0[1]}, (f32x2){a0[2], a0[3]}, (f32x2){a1[0], a1[1]},
                           (f32x2){a1[2], a1[3]}};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if constexpr (FOLD == 1)   // r acc + (b' - (r mu) c)
                gg[c] = gg[c] * r2 + (cv[pp][c] * nm2 + bv[pp][c]);
              else
                gg[c] = gg[c] + bv[pp][c];
            }
            if (ACT == GNLP_ACT_GELU_ERF) {
              gelu_erf_fast2_n<4>(gg);
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                gg[c] = (f32x2){apply_act<false>(gg[c][0], ACT), apply_act<false>(gg[c][1], ACT)};
                if constexpr (ACT == GNLP_ACT_GELU_TANH) W4_SB();   // (registers)
              }
            }
            float v[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[2 * c] = gg[c][0]; v[2 * c + 1] = gg[c][1]; }
            if constexpr (HAS_OTHER) {
              const bf16x8 o8 = ring[k % RD][q][pp];
              if (a.combine == GNLP_COMBINE_MUL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (float)o8[e];
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)o8[e];
              }
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
            *reinterpret_cast<bf16x8 *>(orow + pp * 32) = o;
            W4_SB();   // 8 columns of a row at a time: registers (no spills)
          }
        }
        W4_SB();   // a slab at a time; its stores are issued before the next slab's loads
      });
    });
  };
  // ---- producer cursor: the next K-tile to request, running ahead of the MFMAs by two ----
  const int nk = a.nk;
  int p_tile = 0, p_kt = 0;
  int pm0, pn0;
  coords(0, pm0, pn0);
  const bf16 *pA = a.x + (int64_t)pm0 * a.ldx, *pB = a.w + (int64_t)pn0 * a.K;
  const bf16 *const dA = pA, *const dB = pB;   // a valid source for the two requests past the end
  auto advance = [&]() {
    if (++p_kt < nk) {
      pA += 64; pB += 64;
    } else {
      p_kt = 0;
      if (++p_tile < n_my) {
        coords(p_tile, pm0, pn0);
        pA = a.x + (int64_t)pm0 * a.ldx;
        pB = a.w + (int64_t)pn0 * a.K;
      } else {
        p_tile = n_my;   // stay past the end
        p_kt = nk;       // (re-enters this branch)
        pA = dA; pB = dB;
      }
    }
  };

  // ---- the bias vector goes to LDS once (zeros without a bias) ----
  const bool has_bias = a.bias != nullptr;
  if constexpr (DIRECT && FOLD == 1) {   // b' (fp32) at BIAS, c (fp32) at SCRATCH: N <= 4096
    for (int i = threadIdx.x * 4; i < a.tiles_n * NT; i += 256 * 4) {
      *reinterpret_cast<f32x4 *>(smem + BIAS + i * 4) = *reinterpret_cast<const f32x4 *>(a.fold_b + i);
      *reinterpret_cast<f32x4 *>(smem + SCRATCH + i * 4) = *reinterpret_cast<const f32x4 *>(a.fold_c + i);
    }
  } else if constexpr (FOLD == 2) {   // fp32 image (N <= 2048): bias + beta at BIAS, gamma at BIAS + 8 KiB
    for (int i = threadIdx.x * 8; i < a.tiles_n * NT; i += 256 * 8) {
      const bf16x8 g8 = *reinterpret_cast<const bf16x8 *>(a.ln_g + i);
      const bf16x8 t8 = *reinterpret_cast<const bf16x8 *>(a.ln_b + i);
      // (no branch around the bias load: without a bias beta is read twice and counted once)
      const bf16x8 b8 = *reinterpret_cast<const bf16x8 *>((has_bias ? a.bias : a.ln_b) + i);
      const float bsel = has_bias ? 1.0f : 0.0f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 bb, gg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bb[e] = (float)t8[h * 4 + e] + bsel * (float)b8[h * 4 + e];
          gg[e] = (float)g8[h * 4 + e];
        }
        *reinterpret_cast<f32x4 *>(smem + BIAS + (i + h * 4) * 4) = bb;
        *reinterpret_cast<f32x4 *>(smem + BIAS + 8192 + (i + h * 4) * 4) = gg;
      }
    }
  } else {
    for (int i = threadIdx.x * 8; i < a.tiles_n * NT; i += 256 * 8) {
      bf16x8 b8;
      if (has_bias) {
        b8 = *reinterpret_cast<const bf16x8 *>(a.bias + i);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = (bf16)0.f;
      }
      *reinterpret_cast<bf16x8 *>(smem + BIAS + i * 2) = b8;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- prologue: K-tile 0 -> buffer 0, K-tile 1 -> buffer 1, fragments of k-step 0 ----
#pragma unroll
  for (int p = 0; p < 8; ++p) stage_a(smem, pA, p);
#pragma unroll
  for (int p = 0; p < NJ; ++p) stage_b(smem, pB, p);
  advance();
#pragma unroll
  for (int p = 0; p < 8; ++p) stage_a(smem + BUF, pA, p);
#pragma unroll
  for (int p = 0; p < NJ; ++p) stage_b(smem + BUF, pB, p);
  advance();
  W4_SB();
  if constexpr (NJ == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  W4_LGKM0();   // the bias image
  W4_BAR();
#pragma unroll
  for (int i = 0; i < 8; ++i) rd_a(smem, 0, i);
#pragma unroll
  for (int j = 0; j < NJ; ++j) rd_b(smem, 0, j);

  int par = 0;
  for (int ti = 0; ti < n_my; ++ti) {
    int m0, n0;
    coords(ti, m0, n0);
    // (the first K-tile is peeled: one loop body, so that the fragment registers read at the end
    // of a K-tile are the ones the next K-tile starts from, without copies)
    ktile(std::true_type{}, smem + par * BUF, smem + (par ^ 1) * BUF, pA, pB);
    advance();
    par ^= 1;
    for (int t = 1; t < nk; ++t) {
      if constexpr (FOLD != 0) {
        if (t == nk - 1) fold_head(m0, n0);
      }
      ktile(std::false_type{}, smem + par * BUF, smem + (par ^ 1) * BUF, pA, pB);
      advance();
      par ^= 1;
    }
    if constexpr (DIRECT) drain_direct(m0, n0);
    else drain(m0, n0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two requests past the end
#ifdef W4_STAMP
  if (lane == 0 && blockIdx.x < 256) {
    unsigned *d = w4_dbg + (blockIdx.x * 4 + wave) * 16;
    for (int k = 0; k < 8; ++k) d[k] = st_d[k];
    d[8] = st_n; d[9] = st_n7;
    d[10] = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0);       // shader cycles
    d[11] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);   // 100 MHz ticks
  }
#endif
#endif
}

bool mfma_bf16_dense_w4_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldo,
                                  int64_t ldother, const void *out, const void *bias,
                                  const void *other) {
  // M a multiple of 256; N a multiple of 256 or of 192; K a multiple of 64, at least two K-tiles
  if (M % 256 != 0 || (N % 256 != 0 && N % 192 != 0) || N > 8192 || K % 64 != 0 || K < 128)
    return false;
  if (ldo % 8 != 0 || (other && ldother % 8 != 0)) return false;
  if (((uintptr_t)out | (uintptr_t)(other ? other : out) | (uintptr_t)(bias ? bias : out)) % 16)
    return false;
  // per-lane staging offsets are 32-bit byte offsets inside a 256-row panel
  if (256 * ldx * 2 >= ((int64_t)1 << 31) || 256 * K * 2 >= ((int64_t)1 << 31)) return false;
  if ((M / 256) * ((N + 191) / 192) >= ((int64_t)1 << 28)) return false;
  return true;
}

// Tile width for (M, N): 256 unless N is a multiple of 192 only, or both widths are a single
// round over the CUs and the narrower tile keeps more of them busy (T5-base, M = 16384, N = 768:
// 192 tiles of 256 against 256 tiles of 192, +1.8 % end to end).  Balancing ROUNDS with the
// narrower tile does not pay: at N = 768, M = 32768, 512 tiles of 192 (two full rounds of 3/4 the
// work) ran 1.3 % slower end to end than 384 tiles of 256 (a half-empty second round) -- these
// GEMMs are power-limited, idle CUs lift the clock of the busy ones, and the narrower tile moves
// 17 % more bytes per flop.  It does pay when the 256-wide tiles leave a last round less than
// half full and the 192-wide ones fill theirs: T5-base QKV, M = 16384, N = 2304: 576 tiles = 2.25
// rounds against 768 = 3 rounds of 3/4 the work, 65 -> 55 us, +1.2 % end to end with FFN-1
// (N = 3072: 3 rounds against 4) left at 256; not with a row count read from the device (packed
// ragged batches: M is only an upper bound).  GNLP_GEMM_NT=256 / 192 is the developer knob.
static int w4_tile_width(int64_t M, int64_t N, int cus, bool m_is_bound) {
  if (N % 192 != 0) return 256;
  if (N % 256 != 0) return 192;
  if (knobs().gemm_nt) return knobs().gemm_nt;
  const int64_t tm = M / 256;
  const int64_t t256 = tm * (N / 256), t192 = tm * (N / 192);
  if (t256 < cus && t192 <= cus) return 192;
  if (!m_is_bound && t256 > cus && t192 % cus == 0 && t256 % cus != 0 && 2 * (t256 % cus) < cus &&
      (t192 / cus) * 3 < ((t256 + cus - 1) / cus) * 4)
    return 192;
  return 256;
}

static int w4_cu_count() {   // of the current device
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

template <int ACT, bool HAS_OTHER, int NJ, int FOLD>
static void launch_w4_act(const DenseW4Args &a, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * 128 + 4 * 4096 + 16384;   // 160 KiB
  static PerDeviceFlag attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void *)dense_bf16_w4_kernel<ACT, HAS_OTHER, NJ, FOLD>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  const int nt = a.tiles_m * a.tiles_n;
  int grid = w4_cu_count() & ~7;         // one workgroup per CU, a multiple of the 8 XCDs
  const int need = ((nt + 7) / 8) * 8;   // fewer tiles than CUs: one tile each
  if (grid > need) grid = need;
  dense_bf16_w4_kernel<ACT, HAS_OTHER, NJ, FOLD><<<dim3((unsigned)grid), 256, LDS, s>>>(a);
}

template <int ACT>
static void launch_w4_other(const DenseW4Args &a, int nt_width, int fold, hipStream_t s) {
  if constexpr (ACT != GNLP_ACT_GELU_TANH) {
    if (fold == 1) {   // (mfma_bf16_dense_w4_fold_supported: no `other`, not the tanh GELU)
      if (nt_width == 192) launch_w4_act<ACT, false, 6, 1>(a, s);
      else launch_w4_act<ACT, false, 8, 1>(a, s);
      return;
    }
  }
  if (nt_width == 192) {
    if (a.other) launch_w4_act<ACT, true, 6, 0>(a, s);
    else launch_w4_act<ACT, false, 6, 0>(a, s);
  } else {
    if (a.other) launch_w4_act<ACT, true, 8, 0>(a, s);
    else launch_w4_act<ACT, false, 8, 0>(a, s);
  }
}

int64_t fold_stat_slots(int64_t M, int64_t N) {
  const int width = w4_tile_width(M, N, w4_cu_count() & ~7, dense_rows_dev() != nullptr);
  return 4 * (N / width);
}

bool mfma_bf16_dense_w4_fold_supported(const DenseFold &f, int act, int combine,
                                       const void *other, int64_t N) {
  // (mode 1 with an activation: b' and c as fp32 LDS images of 16 KiB each, N <= 4096; not
  // instantiated for the tanh GELU, which no post-norm stack uses)
  if (f.mode == 1 && (act == GNLP_ACT_GELU_TANH || (act != GNLP_ACT_IDENTITY && N > 4096))) return false;
  if (f.mode == 1)
    return !other && f.rstats && f.c && f.b && ((uintptr_t)f.c | (uintptr_t)f.b) % 16 == 0 &&
           (uintptr_t)f.rstats % 8 == 0;
  if (f.mode == 2)   // (N <= 2048: the fp32 bias + beta / gamma images share the 16 KiB bias region)
    return N <= 2048 && other && combine == GNLP_COMBINE_ADD && act == GNLP_ACT_IDENTITY && f.rstats &&
           f.stats_part && (uintptr_t)f.stats_part % 16 == 0 &&
           f.ln_g && f.ln_b && ((uintptr_t)f.ln_g | (uintptr_t)f.ln_b) % 16 == 0 &&
           (uintptr_t)f.rstats % 8 == 0;
  return f.mode == 0;
}

void launch_mfma_bf16_dense_w4(void *out, const void *x, const void *w, const void *bias,
                               const void *other, int combine, int act, int64_t M, int64_t N,
                               int64_t K, int64_t ldx, int64_t ldo, int64_t ldother,
                               hipStream_t s, const DenseFold *fold) {
  DenseW4Args a;
  a.x = (const bf16 *)x; a.w = (const bf16 *)w; a.bias = (const bf16 *)bias;
  a.other = (const bf16 *)other; a.out = (bf16 *)out;
  a.ldx = ldx; a.ldo = ldo; a.ldother = ldother;
  a.K = (int)K; a.nk = (int)(K >> 6); a.combine = combine;
  a.tiles_m = (int)(M / 256);
  a.rows_dev = dense_rows_dev();
  const int mode = fold ? fold->mode : 0;
  a.rstats = mode ? fold->rstats : nullptr;
  a.fold_c = mode == 1 ? fold->c : nullptr;
  a.fold_b = mode == 1 ? fold->b : nullptr;
  a.ln_g = mode == 2 ? (const bf16 *)fold->ln_g : nullptr;
  a.ln_b = mode == 2 ? (const bf16 *)fold->ln_b : nullptr;
  a.stats_part = mode == 2 ? fold->stats_part : nullptr;
  const int width = w4_tile_width(M, N, w4_cu_count() & ~7, a.rows_dev != nullptr);
  a.tiles_n = (int)(N / width);
  a.group_m = knobs().group_m;
  if (mode == 2) {   // residual = LayerNorm(other): identity activation, add
    if (width == 192) launch_w4_act<GNLP_ACT_IDENTITY, true, 6, 2>(a, s);
    else launch_w4_act<GNLP_ACT_IDENTITY, true, 8, 2>(a, s);
    return;
  }
  switch (act) {
    case GNLP_ACT_RELU: launch_w4_other<GNLP_ACT_RELU>(a, width, mode, s); break;
    case GNLP_ACT_GELU_ERF: launch_w4_other<GNLP_ACT_GELU_ERF>(a, width, mode, s); break;
    case GNLP_ACT_GELU_TANH:   // (no FOLD 1 instantiation: mfma_bf16_dense_w4_fold_supported)
      launch_w4_other<GNLP_ACT_GELU_TANH>(a, width, 0, s);
      break;
    default: launch_w4_other<GNLP_ACT_IDENTITY>(a, width, mode, s); break;
  }
}

}  // namespace gnlp

#ifdef W4_STAMP
extern "C" int gnlp_debug_w4_read(unsigned *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gnlp::w4_dbg), (size_t)n * sizeof(unsigned));
}
#endif
