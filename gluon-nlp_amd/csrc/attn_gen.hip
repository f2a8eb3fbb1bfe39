// General multi-head dot-product attention on the f32 matrix cores, for gfx950.
// Reference: multi_head_dot_attn, attention_cell.py:332-524 --
//   scores = Q K^T (+ edge_scores)                        :452-454  (returned un-tempered, :524)
//   w      = masked_softmax(scores, mask, temperature)    :455
//   ctx    = w V                                          :462-463
// for everything the bf16 / d = 64 kernel of attn_bf16.hip does not take: float32 tensors, head
// dims other than 64 (the reference tests use d_k = 8, d_v = 6), and calls that want `scores` /
// `attn_weights` back.  (B, H, Lq, Lm) is written only when the caller asks for it and is never
// read back: no workspace.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 for QK^T and PV (an exact fp32 fma chain; bf16 inputs are
// widened to fp32 while they are staged, so products are exact and P stays fp32), softmax
// statistics in fp32.  The matrix pipe runs fp32 at 1/16 of its bf16 rate -- this is the
// tolerance / generality path, the bf16 hot path is attn_bf16.hip.
//
// Work decomposition: one workgroup = up to 128 query rows of one (batch, head), a wave owns 32.
// K / V tiles (BK keys, widened to fp32, rows padded to D + 1 words so that column reads are
// bank-conflict free) are staged in LDS by the whole workgroup.
//   S^T = K . Q^T : a lane owns ONE query (column l & 31) and 16 of the 32 keys of a key block
//         (key = 8 (e / 4) + 4 (l >> 5) + e % 4 for accumulator register e), Q stays in registers.
//   O^T = V^T . P^T : register e of the S^T accumulator IS the B operand of the e-th k-step (lane
//         half hh supplies key(e, hh)); the A operand is one LDS word of V per lane.
// Weights wanted: two passes over the keys (row max / sum first, then the normalised weights and
// PV), so that what is written is exactly what multiplies V.  Otherwise one pass with the usual
// running-max rescale.  Bias, dense mask, `scores` and `attn_weights` move between global memory
// and the lane-owns-a-query layout through a per-wave 32 x 33 LDS image, so that global accesses
// run along the key axis (coalesced) whatever the strides.
#include "common.h"
#include "kernels.h"

namespace gnlp {

namespace {

constexpr float GEN_NEG = -1.0e30f;

template <typename T>
__device__ __forceinline__ float ldf(const void *p, int64_t i) { return to_f32(((const T *)p)[i]); }

// wave-private LDS hand-off: LDS serves a wave's accesses in order; the wait makes the data
// visible to the other lanes' following reads, the scheduling barriers keep the order as written
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int D, bool TWO_PASS>
__global__ void __launch_bounds__(256, D <= 64 ? 2 : 1)
attn_gen_kernel(GenAttn a) {
  constexpr int BK = D <= 64 ? 64 : 32;   // keys per staged tile
  constexpr int LD = D + 1;               // padded row (words)
  constexpr int NDB = D / 32;             // 32-wide blocks of the value dim
  extern __shared__ __attribute__((aligned(16))) float smem_gen[];
  float *Ks = smem_gen;                   // [BK][LD]
  float *Vs = Ks + BK * LD;               // [BK][LD]
  const int nwaves = blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *Sw = Vs + BK * LD + wave * (32 * 33);   // per-wave 32 x 33 transpose image
  const int n = lane & 31, hh = lane >> 5;

  const int64_t nqb = (a.Lq + 32 * nwaves - 1) / (32 * nwaves);
  const int64_t lin = blockIdx.x;
  const int64_t qb = lin % nqb;
  const int64_t h = (lin / nqb) % a.H, b = lin / (nqb * a.H);
  const int64_t q0 = qb * 32 * nwaves + wave * 32;   // first query of this wave
  const int64_t my_q = q0 + n;
  const bool wave_live = q0 < a.Lq;                  // (a dead wave still helps staging)

  const int64_t qoff = b * a.qs[0] + h * a.qs[1], koff = b * a.ks[0] + h * a.ks[1],
                voff = b * a.vs[0] + h * a.vs[1];

  // valid_length predicate (i < qlen) & (j < vlen) when no dense mask (qlen = vlen = vl[b] for a
  // self-attention mask, two lengths for a memory mask)
  int64_t vlen = -1, qlen = -1;
  if (!a.mask && a.vl) load_len_pair(a.vl, a.vl_dtype, a.vl_pair, a.B, b, vlen, qlen);
  const bool row_on = qlen < 0 || my_q < qlen;

  // ---- Q fragments: qf[s] = Q[my_q][hh * D/2 + s] (zero beyond d_k / Lq) ----
  float qf[D / 2];
  {
    const int64_t qrow = my_q < a.Lq ? my_q : a.Lq - 1;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) {
      const int dd = hh * (D / 2) + s;
      qf[s] = (wave_live && dd < a.dk) ? ldf<T>(a.q, qoff + qrow * a.qs[2] + dd) : 0.0f;
    }
  }

  auto stage = [&](int64_t j0) {   // keys j0 .. j0 + BK - 1 -> Ks, Vs (fp32, zero padded)
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

  // S^T block kb of the staged tile: sacc[e] = <K[key(e)], Q[my_q]>
  auto qk_block = [&](int kb, f32x16 &sacc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.0f;
    const float *kr = Ks + (kb * 32 + n) * LD + hh * (D / 2);
#pragma unroll
    for (int s = 0; s < D / 2; ++s)
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[s], qf[s], sacc, 0, 0, 0);
  };

  // `scores` block: + bias, optional store of QK^T + bias; returns on-mask bits for the 16 regs
  auto bias_mask_block = [&](int64_t jb, f32x16 &sacc, bool store_scores) -> unsigned {
    // jb = global index of the block's first key
    if (a.bias) {
#pragma unroll 2
      for (int it = 0; it < 16; ++it) {      // rows 2 it + hh of the wave, key jb + n
        const int64_t i = q0 + 2 * it + hh, j = jb + n;
        float bv = 0.0f;
        if (i < a.Lq && j < a.Lm)
          bv = ldf<T>(a.bias, b * a.bs[0] + h * a.bs[1] + i * a.bs[2] + j * a.bs[3]);
        Sw[(2 * it + hh) * 33 + n] = bv;
      }
      wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] += Sw[n * 33 + 8 * (e >> 2) + 4 * hh + (e & 3)];
      wave_lds_sync();
    }
    if (store_scores) {
#pragma unroll
      for (int e = 0; e < 16; ++e) Sw[n * 33 + 8 * (e >> 2) + 4 * hh + (e & 3)] = sacc[e];
      wave_lds_sync();
      T *so = (T *)a.scores_out + ((b * a.H + h) * a.Lq) * a.Lm;
#pragma unroll 2
      for (int it = 0; it < 16; ++it) {
        const int64_t i = q0 + 2 * it + hh, j = jb + n;
        if (i < a.Lq && j < a.Lm) so[i * a.Lm + j] = from_f32<T>(Sw[(2 * it + hh) * 33 + n]);
      }
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
        if (Sw[n * 33 + 8 * (e >> 2) + 4 * hh + (e & 3)] != 0.0f) on |= 1u << e;
      wave_lds_sync();
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t j = jb + 8 * (e >> 2) + 4 * hh + (e & 3);
        const bool ok = j < a.Lm && row_on && (vlen < 0 || j < vlen) && (!a.causal || j <= my_q);
        if (ok) on |= 1u << e;
      }
    }
    return on;
  };

  const float inv_t = a.inv_temperature;
  // causal predicate and no (B, H, Lq, Lm) output to fill: the key tiles above the diagonal of the
  // workgroup's last query row are skipped
  int64_t kv_end = a.Lm;
  if (a.causal && !a.mask && !a.scores_out && !a.weights_out) {
    const int64_t qe = (qb + 1) * 32 * nwaves;
    kv_end = qe < a.Lm ? qe : a.Lm;
  }
  const int ntiles = (int)((kv_end + BK - 1) / BK);
  constexpr bool two_pass = TWO_PASS;
  float m_run = GEN_NEG, l_run = 0.0f;

  if constexpr (two_pass) {   // ---- pass 1: row maximum and sum (and the `scores` output) ----
    for (int t = 0; t < ntiles; ++t) {
      __syncthreads();
      stage((int64_t)t * BK);
      __syncthreads();
      if (!wave_live) continue;
#pragma unroll 1
      for (int kb = 0; kb < BK / 32; ++kb) {
        const int64_t jb = (int64_t)t * BK + kb * 32;
        if (jb >= a.Lm) break;
        f32x16 sacc;
        qk_block(kb, sacc);
        const unsigned on = bias_mask_block(jb, sacc, a.scores_out != nullptr);
        float mloc = GEN_NEG;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (on & (1u << e)) mloc = fmaxf(mloc, sacc[e] * inv_t);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (mloc > m_run) {
          l_run *= __expf(m_run - mloc);   // m_run = GEN_NEG: l_run is still 0
          m_run = mloc;
        }
        float ps = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (on & (1u << e)) ps += __expf(sacc[e] * inv_t - m_run);
        l_run += ps;
      }
    }
    l_run += __shfl_xor(l_run, 32, 64);
  }
  const float inv_l2 = l_run > 0.0f ? 1.0f / l_run : 0.0f;   // two-pass normaliser

  f32x16 oacc[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.0f;
  float m1 = GEN_NEG, l1 = 0.0f;   // one-pass running statistics

  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage((int64_t)t * BK);
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll 1
    for (int kb = 0; kb < BK / 32; ++kb) {
      const int64_t jb = (int64_t)t * BK + kb * 32;
      if (jb >= a.Lm) break;
      f32x16 sacc;
      qk_block(kb, sacc);
      const unsigned on = bias_mask_block(jb, sacc, !two_pass && a.scores_out != nullptr);
      f32x16 p;
      if constexpr (two_pass) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          p[e] = (on & (1u << e)) ? __expf(sacc[e] * inv_t - m_run) * inv_l2 : 0.0f;
        // attn_weights block (B, H, Lq, Lm), through the transpose image
#pragma unroll
        for (int e = 0; e < 16; ++e) Sw[n * 33 + 8 * (e >> 2) + 4 * hh + (e & 3)] = p[e];
        wave_lds_sync();
        T *wo = (T *)a.weights_out + ((b * a.H + h) * a.Lq) * a.Lm;
#pragma unroll 2
        for (int it = 0; it < 16; ++it) {
          const int64_t i = q0 + 2 * it + hh, j = jb + n;
          if (i < a.Lq && j < a.Lm) wo[i * a.Lm + j] = from_f32<T>(Sw[(2 * it + hh) * 33 + n]);
        }
        wave_lds_sync();
      } else {
        float mloc = GEN_NEG;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (on & (1u << e)) mloc = fmaxf(mloc, sacc[e] * inv_t);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (__any(mloc > m1)) {   // wave-uniform branch around the O-wide multiply
          const float m_new = fmaxf(m1, mloc);
          const float alpha = __expf(m1 - m_new);   // 1 when unchanged; m1 = GEN_NEG: O, l are 0
          m1 = m_new;
          l1 *= alpha;
#pragma unroll
          for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
        }
        float ps = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          p[e] = (on & (1u << e)) ? __expf(sacc[e] * inv_t - m1) : 0.0f;
          ps += p[e];
        }
        l1 += ps;
      }
      // O^T += V^T P^T : k-step e pairs key(e, 0) with key(e, 1)
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float vf = Vs[(kb * 32 + 8 * (e >> 2) + 4 * hh + (e & 3)) * LD + d * 32 + n];
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, p[e], oacc[d], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_live) return;

  // ---- write ctx[b, i, h, :] : oacc[d][e] = O[my_q][d * 32 + 8 (e / 4) + 4 hh + e % 4] ----
  float scale = 1.0f;
  if constexpr (!two_pass) {
    const float l_tot = l1 + __shfl_xor(l1, 32, 64);
    scale = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  }
  T *ob = (T *)a.o + b * a.os[0] + h * a.os[1];
#pragma unroll
  for (int d = 0; d < NDB; ++d) {
    if (d * 32 >= a.dv) break;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      Sw[n * 33 + 8 * (e >> 2) + 4 * hh + (e & 3)] = oacc[d][e] * scale;
    wave_lds_sync();
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int64_t i = q0 + 2 * it + hh;
      const int c = d * 32 + n;
      if (i < a.Lq && c < a.dv) ob[i * a.os[2] + c] = from_f32<T>(Sw[(2 * it + hh) * 33 + n]);
    }
    wave_lds_sync();
  }
}

template <typename T, int D, bool TP>
void launch_gen_tp(const GenAttn &a, hipStream_t s) {
  constexpr int BK = D <= 64 ? 64 : 32;
  const int nwaves = a.Lq <= 32 ? 1 : (a.Lq <= 64 ? 2 : 4);
  const size_t lds = (size_t)(2 * BK * (D + 1) + nwaves * 32 * 33) * sizeof(float);
  static PerDeviceFlag attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void *)attn_gen_kernel<T, D, TP>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((2 * BK * (D + 1) + 4 * 32 * 33) * sizeof(float)));
  });
  const int64_t nqb = (a.Lq + 32 * nwaves - 1) / (32 * nwaves);
  attn_gen_kernel<T, D, TP><<<dim3((unsigned)(nqb * a.H * a.B)), nwaves * 64, lds, s>>>(a);
}

template <typename T, int D>
void launch_gen_d(const GenAttn &a, hipStream_t s) {
  if (a.weights_out) launch_gen_tp<T, D, true>(a, s);
  else launch_gen_tp<T, D, false>(a, s);
}

template <typename T>
void launch_gen_t(const GenAttn &a, int64_t dmax, hipStream_t s) {
  if (dmax <= 32) launch_gen_d<T, 32>(a, s);
  else if (dmax <= 64) launch_gen_d<T, 64>(a, s);
  else launch_gen_d<T, 128>(a, s);
}

}  // namespace

bool gen_attn_supported(const GenAttn &a) {
  if (a.dk < 1 || a.dv < 1 || a.dk > 128 || a.dv > 128) return false;
  if (a.B < 1 || a.H < 1 || a.Lq < 1 || a.Lm < 1) return false;
  const int64_t nqb = (a.Lq + 31) / 32;
  return nqb * a.H * a.B <= 0x7fffffffLL;   // grid.x
}

void launch_gen_attn(const GenAttn &a, int dtype, hipStream_t s) {
  const int64_t dmax = a.dk > a.dv ? a.dk : a.dv;
  if (dtype == GNLP_BF16) launch_gen_t<bf16>(a, dmax, s);
  else launch_gen_t<float>(a, dmax, s);
}

}  // namespace gnlp
