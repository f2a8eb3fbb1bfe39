// Backward pass of multi-head dot-product attention for bf16 tensors with 64-wide heads, on the
// bf16 matrix cores (v_mfma_f32_32x32x16_bf16) -- the shape the fused forward kernel of
// attn_bf16.hip serves.  Same mathematics, work decomposition, masks and outputs as the general
// kernels of attn_bwd.hip (read its header first); what differs:
//   * Q, K, V, dO go through LDS as bf16: a row-major image ([64 rows][72]) for the products whose
//     A operand runs along the head dimension (S^T = K Q^T, dP^T = V dO^T; S = Q K^T, dP = dO V^T)
//     and a transposed image ([64 dims][72]) for those whose A operand runs along the sequence
//     (dQ^T += K^T dS^T; dV^T += dO^T P, dK^T += Q^T dS); the B operands are registers;
//   * P and dS, computed in fp32 from the fp32 accumulators, are rounded to bf16 to become the B
//     operand of the second products (as the forward kernel does with P): accumulator registers
//     8 s .. 8 s + 7 of a lane hold the keys (queries) {16 s + 4 hh + 0..3, 16 s + 8 + 4 hh + 0..3}
//     of a 32-block, and the A fragment gathers the transposed image in that same order.
// 16 x the matrix rate of the fp32 kernels: 6.9 ms -> see DESIGN.md at the BERT-large shape.
#include "common.h"
#include "kernels.h"

namespace gnlp {

namespace {

constexpr float BB_NEG = -1.0e30f;
constexpr int BT = 64;          // keys (kernel A) / queries (kernel B) per staged tile
constexpr int LDW = 72;         // bf16 per image row: 64 + 8 pad (144 B, 16-byte aligned rows)

__device__ __forceinline__ void wave_lds_sync_bb() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int row_of_bb(int e, int hh) { return 8 * (e >> 2) + 4 * hh + (e & 3); }

// rows r0 .. r0 + 63 of a (rows, 64) bf16 matrix with row stride `rs` -> row-major image `rm`
// ([64][LDW]) and, if `tr`, transposed image ([64 dims][LDW]); rows >= nrows are zeros
__device__ __forceinline__ void stage_tile(bf16 *rm, bf16 *tr, const bf16 *src, int64_t rs,
                                           int64_t r0, int64_t nrows) {
  for (int idx = threadIdx.x; idx < BT * 8; idx += blockDim.x) {
    const int r = idx >> 3, c8 = idx & 7;
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)0.0f;
    if (r0 + r < nrows) v = *reinterpret_cast<const bf16x8 *>(src + (r0 + r) * rs + c8 * 8);
    if (rm) *reinterpret_cast<bf16x8 *>(rm + r * LDW + c8 * 8) = v;
    if (tr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) tr[(c8 * 8 + i) * LDW + r] = v[i];
    }
  }
}

// A fragment from a row-major image: row blk * 32 + n, head dims 16 s + 8 hh .. + 7
__device__ __forceinline__ bf16x8 frag_rm(const bf16 *rm, int blk, int n, int hh, int s) {
  return *reinterpret_cast<const bf16x8 *>(rm + (blk * 32 + n) * LDW + 16 * s + 8 * hh);
}
// A fragment from a transposed image: dim, sequence positions blk * 32 + {16 s + 4 hh + 0..3,
// 16 s + 8 + 4 hh + 0..3}
__device__ __forceinline__ bf16x8 frag_tr(const bf16 *tr, int dim, int blk, int hh, int s) {
  const bf16 *p = tr + dim * LDW + blk * 32 + 16 * s + 4 * hh;
  const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(p);
  const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(p + 8);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// accumulator registers 8 s .. 8 s + 7 -> the B operand of k-step s
__device__ __forceinline__ bf16x8 pack_b(const f32x16 &x, int s) {
  bf16x8 b;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = (bf16)x[8 * s + i];
  return b;
}

// ------------------------------------------------------------------------------------------
// kernel A: row statistics (lse, D) and dQ
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
attn_bwd_q_bf16_kernel(GenAttnBwd a) {
  extern __shared__ __attribute__((aligned(16))) char smem_bb[];
  bf16 *Kr = (bf16 *)smem_bb;             // [64 keys][LDW]
  bf16 *Vr = Kr + BT * LDW;               // [64 keys][LDW]
  bf16 *Kt = Vr + BT * LDW;               // [64 dims][LDW]
  const int nwaves = blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Sw = (float *)(Kt + BT * LDW) + wave * (32 * 33);
  const int n = lane & 31, hh = lane >> 5;

  const int64_t nqb = (a.Lq + 32 * nwaves - 1) / (32 * nwaves);
  const int64_t lin = blockIdx.x;
  const int64_t qb = lin % nqb;
  const int64_t h = (lin / nqb) % a.H, b = lin / (nqb * a.H);
  const int64_t q0 = qb * 32 * nwaves + wave * 32;
  const int64_t my_q = q0 + n;
  const bool wave_live = q0 < a.Lq;

  const bf16 *qg = (const bf16 *)a.q + b * a.qs[0] + h * a.qs[1];
  const bf16 *kg = (const bf16 *)a.k + b * a.ks[0] + h * a.ks[1];
  const bf16 *vg = (const bf16 *)a.v + b * a.vs[0] + h * a.vs[1];
  const bf16 *gg = (const bf16 *)a.g + b * a.gs[0] + h * a.gs[1];

  int64_t vlen = -1;
  if (!a.mask && a.vl) {
    vlen = load_len(a.vl, a.vl_dtype, b);
    vlen = vlen < 0 ? 0 : vlen;
  }
  const bool row_on = vlen < 0 || my_q < vlen;

  bf16x8 qf[4], gf[4];   // B operands: query my_q, head dims 16 s + 8 hh .. + 7
  {
    const int64_t qrow = my_q < a.Lq ? my_q : a.Lq - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8 *>(qg + qrow * a.qs[2] + 16 * s + 8 * hh);
      gf[s] = *reinterpret_cast<const bf16x8 *>(gg + qrow * a.gs[2] + 16 * s + 8 * hh);
    }
  }

  auto dot_blocks = [&](int kb, f32x16 &sacc, f32x16 &dpacc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = dpacc[e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Kr, kb, n, hh, s), qf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Vr, kb, n, hh, s), gf[s], dpacc, 0, 0, 0);
    }
  };

  auto bias_mask_block = [&](int64_t jb, f32x16 &sacc) -> unsigned {
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
    unsigned on = 0;
    if (a.mask) {
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
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t j = jb + row_of_bb(e, hh);
        if (j < a.Lm && row_on && (vlen < 0 || j < vlen) && (!a.causal || j <= my_q)) on |= 1u << e;
      }
    }
    return on;
  };

  const float inv_t = a.inv_temperature;
  int64_t kv_end = a.Lm;
  if (a.causal && !a.mask) {
    const int64_t qe = (qb + 1) * 32 * nwaves;
    kv_end = qe < a.Lm ? qe : a.Lm;
  }
  if (vlen >= 0 && vlen < kv_end) kv_end = vlen;   // keys beyond valid_length are off for every row
  const int ntiles = (int)((kv_end + BT - 1) / BT);

  // ---- pass 1 ----
  float m_run = BB_NEG, l_run = 0.0f, d_run = 0.0f;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile(Kr, nullptr, kg, a.ks[2], (int64_t)t * BT, a.Lm);
    stage_tile(Vr, nullptr, vg, a.vs[2], (int64_t)t * BT, a.Lm);
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll 1
    for (int kb = 0; kb < BT / 32; ++kb) {
      const int64_t jb = (int64_t)t * BT + kb * 32;
      if (jb >= a.Lm) break;
      f32x16 sacc, dpacc;
      dot_blocks(kb, sacc, dpacc);
      const unsigned on = bias_mask_block(jb, sacc);
      float mloc = BB_NEG;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (on & (1u << e)) mloc = fmaxf(mloc, sacc[e] * inv_t);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      if (mloc > m_run) {
        const float alpha = __expf(m_run - mloc);
        l_run *= alpha;
        d_run *= alpha;
        m_run = mloc;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (on & (1u << e)) {
          const float w = __expf(sacc[e] * inv_t - m_run);
          l_run += w;
          d_run += w * dpacc[e];
        }
    }
  }
  l_run += __shfl_xor(l_run, 32, 64);
  d_run += __shfl_xor(d_run, 32, 64);
  const bool any_on = l_run > 0.0f;
  const float lse = any_on ? m_run + __logf(l_run) : 0.0f;
  const float dsum = any_on ? d_run / l_run : 0.0f;
  if (wave_live && hh == 0 && my_q < a.Lq) {
    float *st = a.stats + ((b * a.H + h) * a.Lq + my_q) * 2;
    st[0] = lse;
    st[1] = dsum;
  }

  // ---- pass 2: dQ^T += K^T dS^T ----
  f32x16 dq[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[d][e] = 0.0f;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile(Kr, Kt, kg, a.ks[2], (int64_t)t * BT, a.Lm);
    stage_tile(Vr, nullptr, vg, a.vs[2], (int64_t)t * BT, a.Lm);
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll 1
    for (int kb = 0; kb < BT / 32; ++kb) {
      const int64_t jb = (int64_t)t * BT + kb * 32;
      if (jb >= a.Lm) break;
      f32x16 sacc, dpacc;
      dot_blocks(kb, sacc, dpacc);
      const unsigned on = bias_mask_block(jb, sacc);
      f32x16 ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = (on & (1u << e)) ? __expf(sacc[e] * inv_t - lse) : 0.0f;
        ds[e] = p * (dpacc[e] - dsum) * inv_t;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 dsb = pack_b(ds, s);
#pragma unroll
        for (int d = 0; d < 2; ++d)
          dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Kt, d * 32 + n, kb, hh, s), dsb,
                                                          dq[d], 0, 0, 0);
      }
    }
  }
  if (!wave_live) return;

  bf16 *ob = (bf16 *)a.dq + b * a.dqs[0] + h * a.dqs[1];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Sw[n * 33 + row_of_bb(e, hh)] = dq[d][e];
    wave_lds_sync_bb();
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t i = q0 + 2 * it + hh;
      if (i < a.Lq) ob[i * a.dqs[2] + d * 32 + n] = (bf16)Sw[(2 * it + hh) * 33 + n];
    }
    wave_lds_sync_bb();
  }
}

// ------------------------------------------------------------------------------------------
// kernel B: dK, dV, dE
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
attn_bwd_kv_bf16_kernel(GenAttnBwd a) {
  extern __shared__ __attribute__((aligned(16))) char smem_bb[];
  bf16 *Qr = (bf16 *)smem_bb;             // [64 queries][LDW]
  bf16 *Gr = Qr + BT * LDW;               // [64 queries][LDW]   (dO)
  bf16 *Qt = Gr + BT * LDW;               // [64 dims][LDW]
  bf16 *Gt = Qt + BT * LDW;               // [64 dims][LDW]
  float *Ls = (float *)(Gt + BT * LDW);   // [64] lse
  float *Ds = Ls + BT;                    // [64] D
  const int nwaves = blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Sw = Ds + BT + wave * (32 * 33);
  const int n = lane & 31, hh = lane >> 5;

  const int64_t nkb = (a.Lm + 32 * nwaves - 1) / (32 * nwaves);
  const int64_t lin = blockIdx.x;
  const int64_t kbi = lin % nkb;
  const int64_t h = (lin / nkb) % a.H, b = lin / (nkb * a.H);
  const int64_t k0 = kbi * 32 * nwaves + wave * 32;
  const int64_t my_k = k0 + n;
  const bool wave_live = k0 < a.Lm;

  const bf16 *qg = (const bf16 *)a.q + b * a.qs[0] + h * a.qs[1];
  const bf16 *kg = (const bf16 *)a.k + b * a.ks[0] + h * a.ks[1];
  const bf16 *vg = (const bf16 *)a.v + b * a.vs[0] + h * a.vs[1];
  const bf16 *gg = (const bf16 *)a.g + b * a.gs[0] + h * a.gs[1];

  int64_t vlen = -1;
  if (!a.mask && a.vl) {
    vlen = load_len(a.vl, a.vl_dtype, b);
    vlen = vlen < 0 ? 0 : vlen;
  }
  const bool col_on = my_k < a.Lm && (vlen < 0 || my_k < vlen);

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
  int64_t q_end = a.Lq;
  if (vlen >= 0 && vlen < q_end) q_end = vlen;   // query rows beyond valid_length are off
  const int ntiles = (int)((q_end + BT - 1) / BT);
  const int t0 = (a.causal && !a.mask) ? (int)((kbi * 32 * nwaves) / BT) : 0;
  for (int t = t0; t < ntiles; ++t) {
    __syncthreads();
    stage_tile(Qr, Qt, qg, a.qs[2], (int64_t)t * BT, a.Lq);
    stage_tile(Gr, Gt, gg, a.gs[2], (int64_t)t * BT, a.Lq);
    for (int r = threadIdx.x; r < BT; r += blockDim.x) {
      const int64_t i = (int64_t)t * BT + r;
      float l = 0.0f, dd = 0.0f;
      if (i < a.Lq) {
        const float *st = a.stats + ((b * a.H + h) * a.Lq + i) * 2;
        l = st[0];
        dd = st[1];
      }
      Ls[r] = l;
      Ds[r] = dd;
    }
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll 1
    for (int qb = 0; qb < BT / 32; ++qb) {
      const int64_t ib = (int64_t)t * BT + qb * 32;
      if (ib >= a.Lq) break;
      f32x16 sacc, dpacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] = dpacc[e] = 0.0f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Qr, qb, n, hh, s), kf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(Gr, qb, n, hh, s), vf[s], dpacc, 0, 0, 0);
      }
      f32x16 p, ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = qb * 32 + row_of_bb(e, hh);
        const int64_t i = (int64_t)t * BT + r;
        bool on = col_on && i < a.Lq;
        float x = sacc[e];
        if (on) {
          if (a.bias)
            x += (float)((const bf16 *)a.bias)[b * a.bs[0] + h * a.bs[1] + i * a.bs[2] +
                                                my_k * a.bs[3]];
          if (a.mask) on = a.mask[b * a.ms[0] + i * a.ms[1] + my_k * a.ms[2]] != 0;
          else on = (vlen < 0 || i < vlen) && (!a.causal || my_k <= i);
        }
        p[e] = on ? __expf(x * inv_t - Ls[r]) : 0.0f;
        ds[e] = p[e] * (dpacc[e] - Ds[r]) * inv_t;
        if (a.de && on)
          atomicAdd(a.de + b * a.des[0] + h * a.des[1] + i * a.des[2] + my_k * a.des[3], ds[e]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = pack_b(p, s), dsb = pack_b(ds, s);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gt, d * 32 + n, qb, hh, s), pb,
                                                          dv[d], 0, 0, 0);
          dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qt, d * 32 + n, qb, hh, s), dsb,
                                                          dk[d], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_live) return;

  bf16 *okb = (bf16 *)a.dk_out + b * a.dks[0] + h * a.dks[1];
  bf16 *ovb = (bf16 *)a.dv_out + b * a.dvs[0] + h * a.dvs[1];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Sw[n * 33 + row_of_bb(e, hh)] = dk[d][e];
    wave_lds_sync_bb();
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t j = k0 + 2 * it + hh;
      if (j < a.Lm) okb[j * a.dks[2] + d * 32 + n] = (bf16)Sw[(2 * it + hh) * 33 + n];
    }
    wave_lds_sync_bb();
#pragma unroll
    for (int e = 0; e < 16; ++e) Sw[n * 33 + row_of_bb(e, hh)] = dv[d][e];
    wave_lds_sync_bb();
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t j = k0 + 2 * it + hh;
      if (j < a.Lm) ovb[j * a.dvs[2] + d * 32 + n] = (bf16)Sw[(2 * it + hh) * 33 + n];
    }
    wave_lds_sync_bb();
  }
}

}  // namespace

// bf16, dk = dv = 64, 16-byte aligned rows (the vector loads of the staging and of the B operands)
bool attn_bwd_bf16_supported(const GenAttnBwd &a) {
  if (a.dk != 64 || a.dv != 64) return false;
  if (a.B < 1 || a.H < 1 || a.Lq < 1 || a.Lm < 1) return false;
  const int64_t *ss[4] = {a.qs, a.ks, a.vs, a.gs};
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 3; ++e)
      if (ss[t][e] % 8 != 0) return false;
  if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.g) % 16 != 0) return false;
  const int64_t nb = ((a.Lq > a.Lm ? a.Lq : a.Lm) + 31) / 32;
  return nb * a.H * a.B <= 0x7fffffffLL;
}

void launch_attn_bwd_bf16(const GenAttnBwd &a, hipStream_t s) {
  {
    const int nwaves = a.Lq <= 32 ? 1 : (a.Lq <= 64 ? 2 : 4);
    const size_t lds = (size_t)3 * BT * LDW * sizeof(bf16) + (size_t)nwaves * 32 * 33 * sizeof(float);
    const int64_t nqb = (a.Lq + 32 * nwaves - 1) / (32 * nwaves);
    attn_bwd_q_bf16_kernel<<<dim3((unsigned)(nqb * a.H * a.B)), nwaves * 64, lds, s>>>(a);
  }
  {
    const int nwaves = a.Lm <= 32 ? 1 : (a.Lm <= 64 ? 2 : 4);
    const size_t lds = (size_t)4 * BT * LDW * sizeof(bf16) + 2 * BT * sizeof(float) +
                       (size_t)nwaves * 32 * 33 * sizeof(float);
    const int64_t nkb = (a.Lm + 32 * nwaves - 1) / (32 * nwaves);
    attn_bwd_kv_bf16_kernel<<<dim3((unsigned)(nkb * a.H * a.B)), nwaves * 64, lds, s>>>(a);
  }
}

}  // namespace gnlp
