// Backward pass of multi-head dot-product attention on the f32 matrix cores, for gfx950.
// Reference: the gradients mx.autograd takes through multi_head_dot_attn
// (attention_cell.py:332-524; driven by tests/test_attention_cell.py:82-151 for query / key /
// value / rel_scores across the NKT / NTK / TNK layouts).  With
//   P  = masked_softmax((Q K^T + E) / t, mask)          O = P V           dO = cotangent of O
//   dV = P^T dO       dP = dO V^T       D_i = sum_j P_ij dP_ij  ( = <dO_i, O_i> )
//   dS = P * (dP - D) / t        dQ = dS K        dK = dS^T Q        dE = dS (summed over the
//                                                                    axes E was broadcast along)
// Nothing of size (B, H, Lq, Lm) is read or written: P is recomputed from Q, K, E and the mask.
//
// Two kernels, both mirroring attn_gen.hip (v_mfma_f32_32x32x2_f32 everywhere, bf16 widened to
// fp32 while staged, statistics in fp32):
//   A  "lane owns a query": one workgroup = up to 128 queries of one (batch, head); K / V tiles
//      staged in LDS.  Pass 1 over the keys: S^T = K Q^T, dP^T = V dO^T, running row maximum, row
//      sum and D (rescaled together) -> stats (lse_i, D_i) to a small workspace.  Pass 2:
//      dS^T from the recomputed blocks, dQ^T += K^T dS^T.
//   B  "lane owns a key": one workgroup = up to 128 keys of one (batch, head); Q / dO tiles and
//      their (lse, D) staged in LDS.  S = Q K^T and dP = dO V^T land with the lane's key in the
//      column, so bias / mask / dE are addressed along the key axis by consecutive lanes
//      (coalesced, no transpose).  dV^T += dO^T P, dK^T += Q^T dS, dE += dS (fp32 atomics into a
//      zero-filled buffer whose strides are 0 along the broadcast axes).
// HBM traffic per (b, h): Q, K, V, dO read once by each kernel per opposite-axis block, bias and
// mask twice, dQ / dK / dV written once.
#include "common.h"
#include "kernels.h"

namespace gnlp {

namespace {

constexpr float BWD_NEG = -1.0e30f;

template <typename T>
__device__ __forceinline__ float ldf(const void *p, int64_t i) { return to_f32(((const T *)p)[i]); }

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// key / query index inside a 32-block held by accumulator register e of lane half hh
__device__ __forceinline__ int row_of(int e, int hh) { return 8 * (e >> 2) + 4 * hh + (e & 3); }

// ------------------------------------------------------------------------------------------
// kernel A: row statistics (lse, D) and dQ
// ------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256, D <= 64 ? 2 : 1)
attn_bwd_q_kernel(GenAttnBwd a) {
  constexpr int BK = D <= 64 ? 64 : 32;
  constexpr int LD = D + 1;
  constexpr int NDB = D / 32;
  extern __shared__ __attribute__((aligned(16))) float smem_bwd[];
  float *Ks = smem_bwd;
  float *Vs = Ks + BK * LD;
  const int nwaves = blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Sw = Vs + BK * LD + wave * (32 * 33);
  const int n = lane & 31, hh = lane >> 5;

  const int64_t nqb = (a.Lq + 32 * nwaves - 1) / (32 * nwaves);
  int64_t head, qb;
  xcd_head_map(blockIdx.x, nqb, a.B * a.H, head, qb);
  const int64_t h = head % a.H, b = head / a.H;
  const int64_t q0 = qb * 32 * nwaves + wave * 32;
  const int64_t my_q = q0 + n;
  const bool wave_live = q0 < a.Lq;

  const int64_t qoff = b * a.qs[0] + h * a.qs[1], koff = b * a.ks[0] + h * a.ks[1],
                voff = b * a.vs[0] + h * a.vs[1], goff = b * a.gs[0] + h * a.gs[1];

  int64_t vlen = -1, qlen = -1;   // valid keys / valid queries
  if (!a.mask && a.vl) load_len_pair(a.vl, a.vl_dtype, a.vl_pair, a.B, b, vlen, qlen);
  const bool row_on = qlen < 0 || my_q < qlen;

  float qf[D / 2], gf[D / 2];
  {
    const int64_t qrow = my_q < a.Lq ? my_q : a.Lq - 1;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) {
      const int dd = hh * (D / 2) + s;
      qf[s] = (wave_live && dd < a.dk) ? ldf<T>(a.q, qoff + qrow * a.qs[2] + dd) : 0.0f;
      gf[s] = (wave_live && dd < a.dv) ? ldf<T>(a.g, goff + qrow * a.gs[2] + dd) : 0.0f;
    }
  }

  auto stage = [&](int64_t j0) {
    for (int idx = threadIdx.x; idx < BK * D; idx += blockDim.x) {
      const int key = idx / D, c = idx - key * D;
      const int64_t j = j0 + key;
      float kv = 0.0f, vv = 0.0f;
      if (j < a.Lm) {
        if (c < a.dk) kv = ldf<T>(a.k, koff + j * a.ks[2] + c);
        if (c < a.dv) vv = ldf<T>(a.v, voff + j * a.vs[2] + c);
      }
      Ks[key * LD + c] = kv;
      Vs[key * LD + c] = vv;
    }
  };

  // sacc[e] = <K[key(e)], Q[my_q]>,  dpacc[e] = <V[key(e)], dO[my_q]>
  auto dot_blocks = [&](int kb, f32x16 &sacc, f32x16 &dpacc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = dpacc[e] = 0.0f;
    const float *kr = Ks + (kb * 32 + n) * LD + hh * (D / 2);
    const float *vr = Vs + (kb * 32 + n) * LD + hh * (D / 2);
#pragma unroll
    for (int s = 0; s < D / 2; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[s], qf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[s], gf[s], dpacc, 0, 0, 0);
    }
  };

  // + bias; on-mask bits of the 16 registers (global accesses run along the key axis through
  // the per-wave transpose image, as in attn_gen.hip)
  auto bias_mask_block = [&](int64_t jb, f32x16 &sacc) -> unsigned {
    if (a.bias) {
#pragma unroll 2
      for (int it = 0; it < 16; ++it) {
        const int64_t i = q0 + 2 * it + hh, j = jb + n;
        float bv = 0.0f;
        if (i < a.Lq && j < a.Lm)
          bv = ldf<T>(a.bias, b * a.bs[0] + h * a.bs[1] + i * a.bs[2] + j * a.bs[3]);
        Sw[(2 * it + hh) * 33 + n] = bv;
      }
      wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] += Sw[n * 33 + row_of(e, hh)];
      wave_lds_sync();
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
      wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (Sw[n * 33 + row_of(e, hh)] != 0.0f) on |= 1u << e;
      wave_lds_sync();
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t j = jb + row_of(e, hh);
        if (j < a.Lm && row_on && (vlen < 0 || j < vlen) && (!a.causal || j <= my_q)) on |= 1u << e;
      }
    }
    return on;
  };

  const float inv_t = a.inv_temperature;
  int64_t kv_end = a.Lm;   // causal predicate: key tiles above the workgroup's diagonal are skipped
  if (a.causal && !a.mask) {
    const int64_t qe = (qb + 1) * 32 * nwaves;
    kv_end = qe < a.Lm ? qe : a.Lm;
  }
  const int ntiles = (int)((kv_end + BK - 1) / BK);

  // ---- pass 1: m, l and sum_j exp(x_j - m) dP_j, rescaled together ----
  float m_run = BWD_NEG, l_run = 0.0f, d_run = 0.0f;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage((int64_t)t * BK);
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll 1
    for (int kb = 0; kb < BK / 32; ++kb) {
      const int64_t jb = (int64_t)t * BK + kb * 32;
      if (jb >= a.Lm) break;
      f32x16 sacc, dpacc;
      dot_blocks(kb, sacc, dpacc);
      const unsigned on = bias_mask_block(jb, sacc);
      float mloc = BWD_NEG;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (on & (1u << e)) mloc = fmaxf(mloc, sacc[e] * inv_t);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      if (mloc > m_run) {
        const float alpha = __expf(m_run - mloc);   // m_run = BWD_NEG: both sums are still 0
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
  const float lse = any_on ? m_run + __logf(l_run) : 0.0f;   // fully masked row: P = 0 (no bits on)
  const float dsum = any_on ? d_run / l_run : 0.0f;
  if (wave_live && hh == 0 && my_q < a.Lq) {
    float *st = a.stats + ((b * a.H + h) * a.Lq + my_q) * 2;
    st[0] = lse;
    st[1] = dsum;
  }

  // ---- pass 2: dQ^T += K^T dS^T ----
  f32x16 dq[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[d][e] = 0.0f;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage((int64_t)t * BK);
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll 1
    for (int kb = 0; kb < BK / 32; ++kb) {
      const int64_t jb = (int64_t)t * BK + kb * 32;
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
      for (int d = 0; d < NDB; ++d) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float kf = Ks[(kb * 32 + row_of(e, hh)) * LD + d * 32 + n];
          dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, ds[e], dq[d], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_live) return;

  // dq[d][e] = dQ[my_q][d * 32 + row_of(e)]
  T *ob = (T *)a.dq + b * a.dqs[0] + h * a.dqs[1];
#pragma unroll
  for (int d = 0; d < NDB; ++d) {
    if (d * 32 >= a.dk) break;
#pragma unroll
    for (int e = 0; e < 16; ++e) Sw[n * 33 + row_of(e, hh)] = dq[d][e];
    wave_lds_sync();
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t i = q0 + 2 * it + hh;
      const int c = d * 32 + n;
      if (i < a.Lq && c < a.dk) ob[i * a.dqs[2] + c] = from_f32<T>(Sw[(2 * it + hh) * 33 + n]);
    }
    wave_lds_sync();
  }
}

// ------------------------------------------------------------------------------------------
// kernel B: dK, dV, dE
// ------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256, D <= 32 ? 2 : 1)
attn_bwd_kv_kernel(GenAttnBwd a) {
  constexpr int BQ = D <= 64 ? 64 : 32;   // queries per staged tile
  constexpr int LD = D + 1;
  constexpr int NDB = D / 32;
  extern __shared__ __attribute__((aligned(16))) float smem_bwd[];
  float *Qs = smem_bwd;                   // [BQ][LD]
  float *Gs = Qs + BQ * LD;               // [BQ][LD]  (dO)
  float *Ls = Gs + BQ * LD;               // [BQ] lse
  float *Ds = Ls + BQ;                    // [BQ] D
  const int nwaves = blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Sw = Ds + BQ + wave * (32 * 33);
  const int n = lane & 31, hh = lane >> 5;

  const int64_t nkb = (a.Lm + 32 * nwaves - 1) / (32 * nwaves);
  int64_t head, kbi;
  xcd_head_map(blockIdx.x, nkb, a.B * a.H, head, kbi);
  const int64_t h = head % a.H, b = head / a.H;
  const int64_t k0 = kbi * 32 * nwaves + wave * 32;   // first key of this wave
  const int64_t my_k = k0 + n;
  const bool wave_live = k0 < a.Lm;

  const int64_t qoff = b * a.qs[0] + h * a.qs[1], koff = b * a.ks[0] + h * a.ks[1],
                voff = b * a.vs[0] + h * a.vs[1], goff = b * a.gs[0] + h * a.gs[1];

  int64_t vlen = -1, qlen = -1;   // valid keys / valid queries
  if (!a.mask && a.vl) load_len_pair(a.vl, a.vl_dtype, a.vl_pair, a.B, b, vlen, qlen);
  const bool col_on = my_k < a.Lm && (vlen < 0 || my_k < vlen);

  float kf[D / 2], vf[D / 2];
  {
    const int64_t krow = my_k < a.Lm ? my_k : a.Lm - 1;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) {
      const int dd = hh * (D / 2) + s;
      kf[s] = (wave_live && dd < a.dk) ? ldf<T>(a.k, koff + krow * a.ks[2] + dd) : 0.0f;
      vf[s] = (wave_live && dd < a.dv) ? ldf<T>(a.v, voff + krow * a.vs[2] + dd) : 0.0f;
    }
  }

  auto stage = [&](int64_t i0) {   // queries i0 .. i0 + BQ - 1 -> Qs, Gs, Ls, Ds
    for (int idx = threadIdx.x; idx < BQ * D; idx += blockDim.x) {
      const int r = idx / D, c = idx - r * D;
      const int64_t i = i0 + r;
      float qv = 0.0f, gv = 0.0f;
      if (i < a.Lq) {
        if (c < a.dk) qv = ldf<T>(a.q, qoff + i * a.qs[2] + c);
        if (c < a.dv) gv = ldf<T>(a.g, goff + i * a.gs[2] + c);
      }
      Qs[r * LD + c] = qv;
      Gs[r * LD + c] = gv;
    }
    for (int r = threadIdx.x; r < BQ; r += blockDim.x) {
      const int64_t i = i0 + r;
      float l = 0.0f, dd = 0.0f;
      if (i < a.Lq) {
        const float *st = a.stats + ((b * a.H + h) * a.Lq + i) * 2;
        l = st[0];
        dd = st[1];
      }
      Ls[r] = l;
      Ds[r] = dd;
    }
  };

  f32x16 dk[NDB], dv[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[d][e] = dv[d][e] = 0.0f;

  const float inv_t = a.inv_temperature;
  const int ntiles = (int)((a.Lq + BQ - 1) / BQ);
  // causal predicate: query tiles that end before the workgroup's first key see none of its keys
  const int t0 = (a.causal && !a.mask) ? (int)((kbi * 32 * nwaves) / BQ) : 0;
  for (int t = t0; t < ntiles; ++t) {
    __syncthreads();
    stage((int64_t)t * BQ);
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll 1
    for (int qb = 0; qb < BQ / 32; ++qb) {
      const int64_t ib = (int64_t)t * BQ + qb * 32;   // first query of the block
      if (ib >= a.Lq) break;
      // sacc[e] = <Q[query(e)], K[my_k]>,  dpacc[e] = <dO[query(e)], V[my_k]>
      f32x16 sacc, dpacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] = dpacc[e] = 0.0f;
      {
        const float *qr = Qs + (qb * 32 + n) * LD + hh * (D / 2);
        const float *gr = Gs + (qb * 32 + n) * LD + hh * (D / 2);
#pragma unroll
        for (int s = 0; s < D / 2; ++s) {
          sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[s], kf[s], sacc, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[s], vf[s], dpacc, 0, 0, 0);
        }
      }
      f32x16 p, ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = qb * 32 + row_of(e, hh);
        const int64_t i = (int64_t)t * BQ + r;
        bool on = col_on && i < a.Lq;
        float x = sacc[e];
        if (on) {
          if (a.bias) x += ldf<T>(a.bias, b * a.bs[0] + h * a.bs[1] + i * a.bs[2] + my_k * a.bs[3]);
          if (a.mask) on = a.mask[b * a.ms[0] + i * a.ms[1] + my_k * a.ms[2]] != 0;
          else on = (qlen < 0 || i < qlen) && (!a.causal || my_k <= i);
        }
        p[e] = on ? __expf(x * inv_t - Ls[r]) : 0.0f;
        ds[e] = p[e] * (dpacc[e] - Ds[r]) * inv_t;
        if (a.de && on)
          atomicAdd(a.de + b * a.des[0] + h * a.des[1] + i * a.des[2] + my_k * a.des[3], ds[e]);
      }
      // dV^T += dO^T P,  dK^T += Q^T dS : k-step e pairs query(e, 0) with query(e, 1)
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = qb * 32 + row_of(e, hh);
          const float gv = Gs[r * LD + d * 32 + n];
          const float qv = Qs[r * LD + d * 32 + n];
          dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, p[e], dv[d], 0, 0, 0);
          dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, ds[e], dk[d], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_live) return;

  // dk[d][e] = dK[my_k][d * 32 + row_of(e)], likewise dv
  T *okb = (T *)a.dk_out + b * a.dks[0] + h * a.dks[1];
  T *ovb = (T *)a.dv_out + b * a.dvs[0] + h * a.dvs[1];
#pragma unroll
  for (int d = 0; d < NDB; ++d) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Sw[n * 33 + row_of(e, hh)] = dk[d][e];
    wave_lds_sync();
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t j = k0 + 2 * it + hh;
      const int c = d * 32 + n;
      if (j < a.Lm && c < a.dk) okb[j * a.dks[2] + c] = from_f32<T>(Sw[(2 * it + hh) * 33 + n]);
    }
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < 16; ++e) Sw[n * 33 + row_of(e, hh)] = dv[d][e];
    wave_lds_sync();
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t j = k0 + 2 * it + hh;
      const int c = d * 32 + n;
      if (j < a.Lm && c < a.dv) ovb[j * a.dvs[2] + c] = from_f32<T>(Sw[(2 * it + hh) * 33 + n]);
    }
    wave_lds_sync();
  }
}

template <typename T, int D>
void launch_bwd_d(const GenAttnBwd &a, hipStream_t s) {
  constexpr int BK = D <= 64 ? 64 : 32;
  static PerDeviceFlag attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void *)attn_bwd_q_kernel<T, D>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((2 * BK * (D + 1) + 4 * 32 * 33) * sizeof(float)));
    (void)hipFuncSetAttribute((const void *)attn_bwd_kv_kernel<T, D>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((2 * BK * (D + 1) + 2 * BK + 4 * 32 * 33) * sizeof(float)));
  });
  {
    const int nwaves = a.Lq <= 32 ? 1 : (a.Lq <= 64 ? 2 : 4);
    const size_t lds = (size_t)(2 * BK * (D + 1) + nwaves * 32 * 33) * sizeof(float);
    const int64_t nqb = (a.Lq + 32 * nwaves - 1) / (32 * nwaves);
    attn_bwd_q_kernel<T, D><<<dim3((unsigned)(nqb * a.H * a.B)), nwaves * 64, lds, s>>>(a);
  }
  {
    const int nwaves = a.Lm <= 32 ? 1 : (a.Lm <= 64 ? 2 : 4);
    const size_t lds = (size_t)(2 * BK * (D + 1) + 2 * BK + nwaves * 32 * 33) * sizeof(float);
    const int64_t nkb = (a.Lm + 32 * nwaves - 1) / (32 * nwaves);
    attn_bwd_kv_kernel<T, D><<<dim3((unsigned)(nkb * a.H * a.B)), nwaves * 64, lds, s>>>(a);
  }
}

template <typename T>
void launch_bwd_t(const GenAttnBwd &a, int64_t dmax, hipStream_t s) {
  if (dmax <= 32) launch_bwd_d<T, 32>(a, s);
  else if (dmax <= 64) launch_bwd_d<T, 64>(a, s);
  else launch_bwd_d<T, 128>(a, s);
}

}  // namespace

bool attn_bwd_supported(const GenAttnBwd &a) {
  if (a.dk < 1 || a.dv < 1 || a.dk > 128 || a.dv > 128) return false;
  if (a.B < 1 || a.H < 1 || a.Lq < 1 || a.Lm < 1) return false;
  const int64_t nb = ((a.Lq > a.Lm ? a.Lq : a.Lm) + 31) / 32;
  return nb * a.H * a.B <= 0x7fffffffLL;   // grid.x
}

void launch_attn_bwd(const GenAttnBwd &a, int dtype, hipStream_t s) {
  const int64_t dmax = a.dk > a.dv ? a.dk : a.dv;
  if (dtype == GNLP_BF16 && knobs().attn_bwd_bf16 && attn_bwd_bf16_supported(a)) {
    launch_attn_bwd_bf16(a, s);
    return;
  }
  if (dtype == GNLP_BF16) launch_bwd_t<bf16>(a, dmax, s);
  else launch_bwd_t<float>(a, dmax, s);
}

}  // namespace gnlp
