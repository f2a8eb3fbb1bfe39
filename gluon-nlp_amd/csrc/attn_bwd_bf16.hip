// Backward pass of multi-head dot-product attention for bf16 tensors with 64-wide heads, on the
// bf16 matrix cores (v_mfma_f32_32x32x16_bf16) -- the shape the fused forward kernel of
// attn_bf16.hip serves.  Same mathematics, work decomposition, masks and outputs as the general
// kernels of attn_bwd.hip (read its header first); what differs:
//   * Q, K, V, dO are staged in LDS as bf16 row-major images of 64 rows x 128 bytes by the
//     global -> LDS DMA (no registers, no vector ALU), double-buffered: the next tile lands while
//     the current one is multiplied, one barrier per tile.  The 16-byte chunk c of row r sits at
//     chunk c ^ swz(r) (swz permutes bits 1..3 of r): both the 16-byte row reads of 16
//     consecutive rows and the transposing reads below touch every bank once;
//   * products whose A operand runs along the head dimension (S^T = K Q^T, dP^T = V dO^T;
//     S = Q K^T, dP = dO V^T) read 16-byte row fragments; those whose A operand runs along the
//     sequence (dQ^T += K^T dS^T; dV^T += dO^T P, dK^T += Q^T dS) read the SAME image through the
//     transposing ds_read_b64_tr_b16, as the forward kernel does for V;
//   * P and dS, computed in fp32 from the fp32 accumulators, are rounded to bf16 to become the B
//     operand of the second products (as the forward kernel does with P): accumulator registers
//     8 s .. 8 s + 7 of a lane hold the keys (queries) {16 s + 4 hh + 0..3, 16 s + 8 + 4 hh + 0..3}
//     of a 32-block, which is the order the transposing read delivers;
//   * 32-blocks whose elements are all switched on (no bias, no dense mask, inside valid_length and
//     below the causal diagonal) take a path without per-element predicates; statistics are kept
//     in base-2 units (exp2 of one fused multiply-add); 1 / temperature is applied once to dQ / dK;
//   * when the forward ran the fused kernel for a caller that wants gradients it left lse and the
//     residual of ctx's bf16 rounding (FusedAttn::lse, o_lo): kernel A then skips its statistics
//     pass and takes D = <dO, ctx + ctx_lo>.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace gnlp {

namespace {

constexpr float BB_NEG = -1.0e30f;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr int BT = 64;          // keys (kernel A) / queries (kernel B) per staged tile
constexpr int IMGB = BT * 128;  // bytes per image
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void wave_lds_sync_bb() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int row_of_bb(int e, int hh) { return 8 * (e >> 2) + 4 * hh + (e & 3); }

// chunk swizzle of image row r: bits (1, 2, 3) of r -> bits (2, 0, 1)
__device__ __forceinline__ int swz(int r) {
  return (((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1);
}

// rows r0 .. r0 + 63 of a (rows, 64) bf16 matrix with row stride rs -> image: 4 waves, two
// 1 KiB pieces (8 rows) each, 16 bytes per lane; rows >= nrows repeat row nrows - 1 (finite
// values that only ever meet P = dS = 0).  A full tile costs no vector ALU: uniform tile base +
// the lane's 32-bit element offset inside the tile.
struct TileSrc {
  const bf16 *src;
  int64_t rs, nrows;
  unsigned off[2];   // (row in tile) * rs + 8 * chunk for the lane's two pieces
  bool small;        // 63 * rs + 56 fits 32 bits
};
__device__ __forceinline__ TileSrc tile_src(const bf16 *src, int64_t rs, int64_t nrows, int wave,
                                            int lane) {
  TileSrc t;
  t.src = src;
  t.rs = rs;
  t.nrows = nrows;
  t.small = rs >= 0 && rs < (1 << 24);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = (wave * 2 + u) * 8 + (lane >> 3);
    t.off[u] = (unsigned)row * (unsigned)rs + 8u * (unsigned)((lane & 7) ^ swz(row));
  }
  return t;
}
__device__ __forceinline__ void tile_dma(char *img, const TileSrc &ts, int64_t r0, int wave,
                                         int lane) {
  if (ts.small && r0 + BT <= ts.nrows) {   // uniform
    const bf16 *tb = ts.src + r0 * ts.rs;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)ts.off[u]),
                                       (lptr_t)(img + (wave * 2 + u) * 1024), 16, 0, 0);
    return;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int piece = wave * 2 + u;
    const int row = piece * 8 + (lane >> 3);
    int64_t gr = r0 + row;
    gr = gr < ts.nrows ? gr : ts.nrows - 1;
    const int chunk = (lane & 7) ^ swz(row);
    __builtin_amdgcn_global_load_lds((gptr_t)(ts.src + gr * ts.rs + chunk * 8),
                                     (lptr_t)(img + piece * 1024), 16, 0, 0);
  }
}

// per-lane byte offsets of the fragment reads inside a 32-row block of an image
struct FragOff {
  int rm[4];      // row read: row n, chunk 2 s + hh
  int tr[2][2];   // transposing read: [head-dim half d][rows + 8]
};
__device__ __forceinline__ FragOff frag_offsets(int lane) {
  const int n = lane & 31, hh = lane >> 5;
  FragOff f;
#pragma unroll
  for (int s = 0; s < 4; ++s) f.rm[s] = n * 128 + (((2 * s + hh) ^ swz(n)) << 4);
  // ds_read_b64_tr_b16: inside each group of 16 lanes, lane i reads 4 bf16 of row (i >> 2) at
  // columns 4 (i & 3) .. + 3 and receives column i of rows 0..3
  const int q4 = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1;
  const int base = (4 * hh + q4) * 128 + (p & 1) * 8;
  const int c0 = (2 * g1 + (p >> 1)) ^ (((q4 >> 1) << 2) | hh);
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) f.tr[d][hi] = base + hi * 1024 + ((c0 ^ (4 * d) ^ (2 * hi)) << 4);
  return f;
}
// A fragment along the head dimension: row blk * 32 + n, head dims 16 s + 8 hh .. + 7
__device__ __forceinline__ bf16x8 frag_rm(const char *img, int blk, int s, const FragOff &f) {
  return *reinterpret_cast<const bf16x8 *>(img + blk * 4096 + f.rm[s]);
}
// A fragment along the sequence: lane (n, hh) receives image[blk * 32 + {16 s + 4 hh + 0..3,
// 16 s + 8 + 4 hh + 0..3}][d * 32 + n].  (The compiler puts s_waitcnt vmcnt(0) in front of these
// reads -- it cannot tell them from the LDS side of the DMA of the next tile; issuing them as
// inline assembly removes the wait and changes nothing measurable: the tile has landed by then.)
__device__ __forceinline__ bf16x8 frag_tr(const char *img, int blk, int d, int s, const FragOff &f) {
  typedef __attribute__((address_space(3))) bf16x4 *lds_bf16x4_p;
  const char *p = img + blk * 4096 + s * 2048;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_p)(p + f.tr[d][0]));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_p)(p + f.tr[d][1]));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// accumulator registers 8 s .. 8 s + 7 -> the B operand of k-step s
__device__ __forceinline__ bf16x8 pack_b(const f32x16 &x, int s) {
  bf16x8 b;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = (bf16)x[8 * s + i];
  return b;
}
// bit e set iff 8 (e >> 2) + (e & 3) < t : the registers whose position inside the 32-block
// (less the lane's 4 hh) lies below a per-lane limit
__device__ __forceinline__ unsigned bits_below(int t) {
  unsigned on = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    int c = t - 8 * g;
    c = c < 0 ? 0 : (c > 4 ? 4 : c);
    on |= ((1u << c) - 1u) << (4 * g);
  }
  return on;
}

// acc[d][e] = X[r0 + n][d * 32 + row_of(e, hh)] / scale of this wave's 32 rows -> bf16 rows of
// `out`: the wave transposes its 32 x 64 block through a private 4 KiB LDS image (8-byte slot
// index XOR (row & 15)) so that the global stores are full 128-byte rows, 16 bytes per lane
__device__ __forceinline__ void store_rows(char *ow, const f32x16 (&acc)[2], float scale, bf16 *out,
                                           int64_t rs, int64_t r0, int64_t nrows, int lane) {
  const int n = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[d][g * 4 + e] * scale);
      const int slot = d * 8 + g * 2 + hh;
      *reinterpret_cast<bf16x4 *>(ow + n * 128 + ((slot ^ (n & 15)) << 3)) = o;
    }
  wave_lds_sync_bb();
  const int oc = lane & 7, orow0 = lane >> 3;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + orow0;
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(ow + row * 128 + (((2 * oc) ^ (row & 15)) << 3));
    const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(ow + row * 128 + (((2 * oc + 1) ^ (row & 15)) << 3));
    const int64_t rr = r0 + row;
    if (rr < nrows)
      *reinterpret_cast<bf16x8 *>(out + rr * rs + oc * 8) =
          __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  wave_lds_sync_bb();
}

typedef std::integral_constant<bool, true> full_t;
typedef std::integral_constant<bool, false> part_t;

// ------------------------------------------------------------------------------------------
// kernel A: row statistics (lse in base-2 units, D) and dQ.  256 threads = 128 queries.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
attn_bwd_q_bf16_kernel(GenAttnBwd a) {
  extern __shared__ __attribute__((aligned(16))) char smem_bb[];
  char *img = smem_bb;                    // [2 buffers][K, V][IMGB]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Sw = (float *)(img + 4 * IMGB) + wave * (32 * 33);
  const int n = lane & 31, hh = lane >> 5;
  const FragOff fo = frag_offsets(lane);

  const int64_t nqb = (a.Lq + 127) / 128;
  int64_t head, qb;
  xcd_head_map(blockIdx.x, nqb, a.B * a.H, head, qb);
  const int64_t h = head % a.H, b = head / a.H;
  const int64_t q0 = qb * 128 + wave * 32;
  const int64_t my_q = q0 + n;
  const bool wave_live = q0 < a.Lq;

  const bf16 *qg = (const bf16 *)a.q + b * a.qs[0] + h * a.qs[1];
  const bf16 *kg = (const bf16 *)a.k + b * a.ks[0] + h * a.ks[1];
  const bf16 *vg = (const bf16 *)a.v + b * a.vs[0] + h * a.vs[1];
  const bf16 *gg = (const bf16 *)a.g + b * a.gs[0] + h * a.gs[1];

  int64_t vlen = -1, qlen = -1;   // valid keys / valid queries
  if (!a.mask && a.vl) load_len_pair(a.vl, a.vl_dtype, a.vl_pair, a.B, b, vlen, qlen);
  const bool gen = a.bias != nullptr || a.mask != nullptr;   // uniform: per-element bias / mask
  // without a dense mask: key j of this lane's query is on iff j < klim
  const int lim = (int)((vlen >= 0 && vlen < a.Lm) ? vlen : a.Lm);
  int klim = 0;
  if (my_q < a.Lq && (qlen < 0 || my_q < qlen))
    klim = (a.causal && my_q + 1 < lim) ? (int)(my_q + 1) : lim;

  bf16x8 qf[4], gf[4];   // B operands: query my_q, head dims 16 s + 8 hh .. + 7
  {
    const int64_t qrow = my_q < a.Lq ? my_q : a.Lq - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8 *>(qg + qrow * a.qs[2] + 16 * s + 8 * hh);
      gf[s] = *reinterpret_cast<const bf16x8 *>(gg + qrow * a.gs[2] + 16 * s + 8 * hh);
    }
  }

  // sacc[e] = <K[key(e)], Q[my_q]>,  dpacc[e] = <V[key(e)], dO[my_q]>
  auto dot_blocks = [&](const char *Kr, const char *Vr, int kb, f32x16 &sacc, f32x16 &dpacc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = dpacc[e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Kr, kb, s, fo), qf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Vr, kb, s, fo), gf[s], dpacc, 0, 0, 0);
    }
  };

  // + bias; on-mask bits of the 16 registers (global accesses run along the key axis through
  // the per-wave transpose image, as in attn_gen.hip)
  auto bias_mask_block = [&](int jb, f32x16 &sacc) -> unsigned {
    if (a.bias) {
#pragma unroll 2
      for (int it = 0; it < 16; ++it) {
        const int64_t i = q0 + 2 * it + hh, j = jb + n;
        float bv = 0.0f;
        if (i < a.Lq && j < a.Lm)
          bv = (float)((const bf16 *)a.bias)[b * a.bs[0] + h * a.bs[1] + i * a.bs[2] + j * a.bs[3]];
        Sw[(2 * it + hh) * 33 + n] = bv;
      }
      wave_lds_sync_bb();
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] += Sw[n * 33 + row_of_bb(e, hh)];
      wave_lds_sync_bb();
    }
    if (!a.mask) return bits_below(klim - jb - 4 * hh);
    unsigned on = 0;
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t i = q0 + 2 * it + hh, j = jb + n;
      float mv = 0.0f;
      if (i < a.Lq && j < a.Lm)
        mv = a.mask[b * a.ms[0] + i * a.ms[1] + j * a.ms[2]] != 0 ? 1.0f : 0.0f;
      Sw[(2 * it + hh) * 33 + n] = mv;
    }
    wave_lds_sync_bb();
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (Sw[n * 33 + row_of_bb(e, hh)] != 0.0f) on |= 1u << e;
    wave_lds_sync_bb();
    return on;
  };

  const float inv_t = a.inv_temperature;
  const float c_scale = inv_t * LOG2E;   // > 0 (attn_bwd_bf16_supported)
  // keys at or beyond kv_end are off for every row of the workgroup
  int64_t kv_end = a.mask ? a.Lm : lim;
  if (a.causal && !a.mask && (qb + 1) * 128 < kv_end) kv_end = (qb + 1) * 128;
  const int ntiles = (int)((kv_end + BT - 1) / BT);

  const TileSrc ksrc = tile_src(kg, a.ks[2], a.Lm, wave, lane);
  const TileSrc vsrc = tile_src(vg, a.vs[2], a.Lm, wave, lane);
  // one sweep over the key tiles; body(Kr, Vr, kb, jb, full)
  auto sweep = [&](auto &&body) {
    if (ntiles == 0) return;
    tile_dma(img, ksrc, 0, wave, lane);
    tile_dma(img + IMGB, vsrc, 0, wave, lane);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
      if (t + 1 < ntiles) {   // lands under the products of tile t
        tile_dma(img + (cur ^ 1) * 2 * IMGB, ksrc, (int64_t)(t + 1) * BT, wave, lane);
        tile_dma(img + (cur ^ 1) * 2 * IMGB + IMGB, vsrc, (int64_t)(t + 1) * BT, wave, lane);
      }
      const char *Kr = img + cur * 2 * IMGB, *Vr = Kr + IMGB;
      if (wave_live) {
#pragma unroll 1
        for (int kb = 0; kb < BT / 32; ++kb) {
          const int jb = t * BT + kb * 32;
          if (jb >= kv_end) break;
          if (!gen && __all(klim >= jb + 32)) body(Kr, Vr, kb, jb, full_t());
          else body(Kr, Vr, kb, jb, part_t());
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  };

  // ---- row statistics: lse (base-2 units) and D = sum_j P_j dP_j ----
  float lse2 = 0.0f, dsum = 0.0f;
  if (a.fwd_lse) {   // uniform.  Saved by the forward kernel: its lse, and D = <dO, O> with O
                     // to 16 bits (the bf16 output + the residual of its rounding)
    const int64_t qrow = my_q < a.Lq ? my_q : a.Lq - 1;
    const int64_t ooff = b * a.os[0] + h * a.os[1] + qrow * a.os[2];
    const bf16 *og = (const bf16 *)a.fwd_o + ooff, *ol = (const bf16 *)a.fwd_o_lo + ooff;
    float acc = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 o8 = *reinterpret_cast<const bf16x8 *>(og + 16 * s + 8 * hh);
      const bf16x8 l8 = *reinterpret_cast<const bf16x8 *>(ol + 16 * s + 8 * hh);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf((float)gf[s][i], (float)o8[i] + (float)l8[i], acc);
    }
    dsum = acc + __shfl_xor(acc, 32, 64);
    lse2 = a.fwd_lse[(b * a.H + h) * a.Lq + qrow];
  } else {
    // pass 1: m, l and sum_j exp(x_j - m) dP_j, rescaled together
    float m_run = BB_NEG, l_run = 0.0f, d_run = 0.0f;
    sweep([&](const char *Kr, const char *Vr, int kb, int jb, auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      f32x16 sacc, dpacc;
      dot_blocks(Kr, Vr, kb, sacc, dpacc);
      unsigned on = 0xffffu;
      if (!FULL) on = gen ? bias_mask_block(jb, sacc) : bits_below(klim - jb - 4 * hh);
      float mloc = BB_NEG;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (FULL || (on & (1u << e))) mloc = fmaxf(mloc, sacc[e]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mz = mloc > -1.0e29f ? mloc * c_scale : BB_NEG;
      if (mz > m_run) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - mz);   // m_run = BB_NEG: both sums are 0
        l_run *= alpha;
        d_run *= alpha;
        m_run = mz;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float w = __builtin_amdgcn_exp2f(fmaf(sacc[e], c_scale, -m_run));
        if (!FULL) w = (on & (1u << e)) ? w : 0.0f;
        l_run += w;
        d_run = fmaf(w, dpacc[e], d_run);
      }
    });
    l_run += __shfl_xor(l_run, 32, 64);
    d_run += __shfl_xor(d_run, 32, 64);
    const bool any_on = l_run > 0.0f;
    lse2 = any_on ? m_run + __builtin_amdgcn_logf(l_run) : 0.0f;   // fully masked: P = 0
    dsum = any_on ? d_run / l_run : 0.0f;
  }
  if (wave_live && hh == 0 && my_q < a.Lq) {
    float *st = a.stats + ((b * a.H + h) * a.Lq + my_q) * 2;
    st[0] = lse2;
    st[1] = dsum;
  }

  // ---- pass 2: dQ^T += K^T dS^T (dS without the 1 / temperature) ----
  f32x16 dq[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[d][e] = 0.0f;
  sweep([&](const char *Kr, const char *Vr, int kb, int jb, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    f32x16 sacc, dpacc;
    dot_blocks(Kr, Vr, kb, sacc, dpacc);
    unsigned on = 0xffffu;
    if (!FULL) on = gen ? bias_mask_block(jb, sacc) : bits_below(klim - jb - 4 * hh);
    f32x16 ds;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float p = __builtin_amdgcn_exp2f(fmaf(sacc[e], c_scale, -lse2));
      if (!FULL) p = (on & (1u << e)) ? p : 0.0f;
      ds[e] = p * (dpacc[e] - dsum);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 dsb = pack_b(ds, s);
#pragma unroll
      for (int d = 0; d < 2; ++d)
        dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Kr, kb, d, s, fo), dsb, dq[d],
                                                        0, 0, 0);
    }
  });
  if (!wave_live) return;

  // dq[d][e] = dQ[my_q][d * 32 + row_of(e)] * temperature
  store_rows((char *)Sw, dq, inv_t, (bf16 *)a.dq + b * a.dqs[0] + h * a.dqs[1], a.dqs[2], q0, a.Lq,
             lane);
}

// ------------------------------------------------------------------------------------------
// kernel B: dK, dV, dE.  256 threads = 128 keys.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
attn_bwd_kv_bf16_kernel(GenAttnBwd a) {
  extern __shared__ __attribute__((aligned(16))) char smem_bb[];
  char *img = smem_bb;                    // [2 buffers][Q, dO][IMGB]
  float *LD = (float *)(img + 4 * IMGB);  // [2 buffers][lse2[64], D[64]]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Sw = LD + 4 * BT + wave * (32 * 33);
  const int n = lane & 31, hh = lane >> 5;
  const FragOff fo = frag_offsets(lane);

  const int64_t nkb = (a.Lm + 127) / 128;
  int64_t head, kbi;
  xcd_head_map(blockIdx.x, nkb, a.B * a.H, head, kbi);
  const int64_t h = head % a.H, b = head / a.H;
  const int64_t k0 = kbi * 128 + wave * 32;   // first key of this wave
  const int64_t my_k = k0 + n;
  const bool wave_live = k0 < a.Lm;

  const bf16 *qg = (const bf16 *)a.q + b * a.qs[0] + h * a.qs[1];
  const bf16 *kg = (const bf16 *)a.k + b * a.ks[0] + h * a.ks[1];
  const bf16 *vg = (const bf16 *)a.v + b * a.vs[0] + h * a.vs[1];
  const bf16 *gg = (const bf16 *)a.g + b * a.gs[0] + h * a.gs[1];

  int64_t vlen = -1, qlen = -1;   // valid keys / valid queries
  if (!a.mask && a.vl) load_len_pair(a.vl, a.vl_dtype, a.vl_pair, a.B, b, vlen, qlen);
  const bool gen = a.bias != nullptr || a.mask != nullptr || a.de != nullptr;   // uniform
  const bool col_on = my_k < a.Lm && (vlen < 0 || my_k < vlen);
  // without a dense mask: query i is on for this lane's key iff qlo <= i < qhi
  const int qlim = (int)((qlen >= 0 && qlen < a.Lq) ? qlen : a.Lq);
  const int qhi = col_on ? qlim : 0;
  const int qlo = a.causal ? (int)my_k : 0;

  bf16x8 kf[4], vf[4];   // B operands: key my_k, head dims 16 s + 8 hh .. + 7
  {
    const int64_t krow = my_k < a.Lm ? my_k : a.Lm - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8 *>(kg + krow * a.ks[2] + 16 * s + 8 * hh);
      vf[s] = *reinterpret_cast<const bf16x8 *>(vg + krow * a.vs[2] + 16 * s + 8 * hh);
    }
  }

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[d][e] = dv[d][e] = 0.0f;

  const float inv_t = a.inv_temperature;
  const float c_scale = inv_t * LOG2E;
  const int64_t q_end = a.mask ? a.Lq : qlim;   // query rows beyond valid_length are off
  const int ntiles = (int)((q_end + BT - 1) / BT);
  // causal predicate: query tiles that end before the workgroup's first key see none of its keys
  const int t0 = (a.causal && !a.mask) ? (int)((kbi * 128) / BT) : 0;

  auto block = [&](const char *Qr, const char *Gr, const float *Ls, const float *Ds, int qb, int ib,
                   auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    // sacc[e] = <Q[query(e)], K[my_k]>,  dpacc[e] = <dO[query(e)], V[my_k]>
    f32x16 sacc, dpacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = dpacc[e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Qr, qb, s, fo), kf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Gr, qb, s, fo), vf[s], dpacc, 0, 0, 0);
    }
    f32x16 p, ds;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const f32x4 *>(Ls + qb * 32 + 8 * g + 4 * hh);
      const f32x4 d4 = *reinterpret_cast<const f32x4 *>(Ds + qb * 32 + 8 * g + 4 * hh);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = 4 * g + c;
        const int i = ib + 8 * g + 4 * hh + c;
        float x = sacc[e];
        bool on = true;
        if (!FULL) {
          on = i >= qlo && i < qhi;
          if (gen) {
            on = col_on && i < a.Lq;
            if (on) {
              if (a.bias)
                x += (float)((const bf16 *)a.bias)[b * a.bs[0] + h * a.bs[1] + i * a.bs[2] +
                                                    my_k * a.bs[3]];
              if (a.mask) on = a.mask[b * a.ms[0] + i * a.ms[1] + my_k * a.ms[2]] != 0;
              else on = i >= qlo && i < qhi;
            }
          }
        }
        float pe = __builtin_amdgcn_exp2f(fmaf(x, c_scale, -l4[c]));
        if (!FULL) pe = on ? pe : 0.0f;
        p[e] = pe;
        ds[e] = pe * (dpacc[e] - d4[c]);
        if (!FULL && a.de && on)
          atomicAdd(a.de + b * a.des[0] + h * a.des[1] + i * a.des[2] + my_k * a.des[3],
                    ds[e] * inv_t);
      }
    }
    // dV^T += dO^T P,  dK^T += Q^T dS
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pb = pack_b(p, s), dsb = pack_b(ds, s);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gr, qb, d, s, fo), pb, dv[d],
                                                        0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qr, qb, d, s, fo), dsb, dk[d],
                                                        0, 0, 0);
      }
    }
  };

  if (t0 < ntiles) {
    float streg = 0.0f;
    const TileSrc qsrc = tile_src(qg, a.qs[2], a.Lq, wave, lane);
    const TileSrc gsrc = tile_src(gg, a.gs[2], a.Lq, wave, lane);
    auto dma = [&](int buf, int t) {
      tile_dma(img + buf * 2 * IMGB, qsrc, (int64_t)t * BT, wave, lane);
      tile_dma(img + buf * 2 * IMGB + IMGB, gsrc, (int64_t)t * BT, wave, lane);
    };
    auto load_stats = [&](int t) {   // threads 0..63: lse2, 64..127: D
      if (threadIdx.x < 2 * BT) {
        const int64_t i = (int64_t)t * BT + (threadIdx.x & (BT - 1));
        streg = i < a.Lq ? a.stats[((b * a.H + h) * a.Lq + i) * 2 + (threadIdx.x >> 6)] : 0.0f;
      }
    };
    auto store_stats = [&](int buf) {
      if (threadIdx.x < 2 * BT) LD[buf * 2 * BT + threadIdx.x] = streg;
    };
    dma(0, t0);
    load_stats(t0);
    store_stats(0);
    __syncthreads();
    int cur = 0;
    for (int t = t0; t < ntiles; ++t) {
      const bool more = t + 1 < ntiles;
      if (more) {   // lands under the products of tile t
        dma(cur ^ 1, t + 1);
        load_stats(t + 1);
      }
      const char *Qr = img + cur * 2 * IMGB, *Gr = Qr + IMGB;
      const float *Ls = LD + cur * 2 * BT, *Ds = Ls + BT;
      if (wave_live) {
#pragma unroll 1
        for (int qb = 0; qb < BT / 32; ++qb) {
          const int ib = t * BT + qb * 32;   // first query of the block
          if (ib >= q_end) break;
          if (!gen && __all(qlo <= ib && qhi >= ib + 32)) block(Qr, Gr, Ls, Ds, qb, ib, full_t());
          else block(Qr, Gr, Ls, Ds, qb, ib, part_t());
        }
      }
      if (more) store_stats(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  if (!wave_live) return;

  // dk[d][e] = dK[my_k][d * 32 + row_of(e)] * temperature,  dv[d][e] = dV[my_k][d * 32 + row_of(e)]
  store_rows((char *)Sw, dk, inv_t, (bf16 *)a.dk_out + b * a.dks[0] + h * a.dks[1], a.dks[2], k0,
             a.Lm, lane);
  store_rows((char *)Sw, dv, 1.0f, (bf16 *)a.dv_out + b * a.dvs[0] + h * a.dvs[1], a.dvs[2], k0,
             a.Lm, lane);
}

constexpr size_t kLdsQ = (size_t)4 * IMGB + (size_t)4 * 32 * 33 * sizeof(float);
constexpr size_t kLdsKV = kLdsQ + (size_t)4 * BT * sizeof(float);
static_assert(kLdsKV <= 64 * 1024, "static LDS limit");

}  // namespace

// bf16, dk = dv = 64, 16-byte aligned rows of the inputs and the outputs (16-byte accesses),
// positive temperature (the running maximum is taken before scaling), sequence lengths that fit
// the 32-bit position arithmetic
bool attn_bwd_bf16_supported(const GenAttnBwd &a) {
  if (a.dk != 64 || a.dv != 64) return false;
  if (a.B < 1 || a.H < 1 || a.Lq < 1 || a.Lm < 1) return false;
  if (a.Lq > (1 << 30) || a.Lm > (1 << 30)) return false;
  if (!(a.inv_temperature > 0.0f)) return false;
  const int64_t *ss[7] = {a.qs, a.ks, a.vs, a.gs, a.dqs, a.dks, a.dvs};
  for (int t = 0; t < 7; ++t)
    for (int e = 0; e < 3; ++e)
      if (ss[t][e] % 8 != 0) return false;
  if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.g | (uintptr_t)a.dq |
       (uintptr_t)a.dk_out | (uintptr_t)a.dv_out) % 16 != 0)
    return false;
  const int64_t nb = ((a.Lq > a.Lm ? a.Lq : a.Lm) + 127) / 128;
  return nb * a.H * a.B <= 0x7fffffffLL;
}

void launch_attn_bwd_bf16(const GenAttnBwd &a_in, hipStream_t s) {
  GenAttnBwd a = a_in;
  if (a.fwd_lse && (!a.fwd_o || !a.fwd_o_lo || ((uintptr_t)a.fwd_o | (uintptr_t)a.fwd_o_lo) % 16 != 0 ||
                    a.os[0] % 8 != 0 || a.os[1] % 8 != 0 || a.os[2] % 8 != 0)) {
    a.fwd_lse = nullptr;   // rows of the saved output are not 16-byte aligned: recompute
    a.fwd_o = a.fwd_o_lo = nullptr;
  }
  const int64_t nqb = (a.Lq + 127) / 128, nkb = (a.Lm + 127) / 128;
  attn_bwd_q_bf16_kernel<<<dim3((unsigned)(nqb * a.H * a.B)), 256, kLdsQ, s>>>(a);
  attn_bwd_kv_bf16_kernel<<<dim3((unsigned)(nkb * a.H * a.B)), 256, kLdsKV, s>>>(a);
}

}  // namespace gnlp
