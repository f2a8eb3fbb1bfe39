// Round 3 probe: why do the MFMA and vector-ALU phases of the fused attention kernel not overlap
// across the waves of a SIMD, when tools/coissue_probe2.hip (C) shows that independent phases do?
// One pass = one K / V tile of attn_bf16.hip as a wave sees it, built up feature by feature:
//   8 MFMAs (S^T = K Q^T) -> row maximum (16 dependent v_max3, lane ^ 32 swap, wave-uniform branch)
//   -> 32 x (fma, exp2), 32 adds, 16 cvt_pk -> 8 MFMAs (O^T += V^T P^T) [-> workgroup barrier]
// FLAGS: 1 the softmax reads the MFMA results (else loop-carried stand-ins)
//        2 the second MFMA block reads the softmax's output (else constants)
//        4 s_barrier per pass (4 waves = 4 SIMDs)
//        8 K fragments (8 ds_read_b128) and V^T fragments (16 ds_read_b64_tr_b16) come from LDS
//       16 no exponentials (a multiply instead)
//       32 no vector-ALU work at all (calibration)
//       64 s_setprio 3 around the MFMA blocks (implies 128)
//      128 the MFMA blocks stay blocks (sched_barrier around them)
//      256 the next tile's K / V pieces are requested at the top of the pass (4 global_load ... lds
//          per wave from an L2-resident buffer, per-lane addresses advanced as in the kernel)
//      512 the scores leave the MFMAs as exp2 arguments (no fma; 2 more MFMAs in the first block)
//     1024 the row sums come from 4 more MFMAs against a ones fragment (no adds)
// W workgroups per CU (LDS-limited), all resident at once; output: SIMD cycles per pass.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/build/attn_chain_probe tools/attn_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) bf16x4 *lds_bf16x4_ptr;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int FLAGS>
__global__ void __launch_bounds__(256, 3) probe(float *out, long long *cyc, int iters, float c_scale, const char *gsrc) {
  constexpr bool DEP_MV = FLAGS & 1, DEP_VM = FLAGS & 2, BAR = FLAGS & 4, LDS = FLAGS & 8,
                 NOEXP = FLAGS & 16, NOVALU = FLAGS & 32, PRIO = FLAGS & 64, BLOCKS = FLAGS & 128, DMA = FLAGS & 256, NOFMA = FLAGS & 512, SUMM = FLAGS & 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  for (int i = threadIdx.x; i < 4 * 8192 / 2; i += 256)
    reinterpret_cast<__bf16 *>(smem)[i] = (__bf16)(0.01f * (float)((i * 37) % 19 - 9));
  __syncthreads();
  bf16x8 qf[4], kc[2], vc[2], pc[4];
  for (int s = 0; s < 4; ++s)
    for (int e = 0; e < 8; ++e) {
      qf[s][e] = (__bf16)(0.02f * (float)((lane + 3 * s + e) % 7 - 3));
      pc[s][e] = (__bf16)(0.1f + 0.01f * (float)((lane + s + e) % 5));
    }
  for (int d = 0; d < 2; ++d)
    for (int e = 0; e < 8; ++e) {
      kc[d][e] = (__bf16)(0.03f * (float)((lane + d + e) % 5 - 2));
      vc[d][e] = (__bf16)(0.05f * (float)((lane + 2 * d + e) % 9 - 4));
    }
  const int k_off = r * 128, k_sw = (r >> 1) & 7;
  const int v_q4 = (lane & 15) >> 2, v_p = lane & 3, v_g1 = (lane >> 4) & 1;
  const int v_row0 = 4 * hh + v_q4, v_colb = (v_g1 * 16 + 4 * v_p) * 2, v_sw = ((v_row0 >> 1) & 1) << 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *kp0[2], *vp0[2];
  for (int t = 0; t < 2; ++t) {
    const int row = (wave * 2 + t) * 8 + (lane >> 3);
    kp0[t] = gsrc + (size_t)(blockIdx.x & 511) * 65536 + row * 128 + (lane & 7) * 16;
    vp0[t] = kp0[t] + 8192;
  }
  float m_run = -1.0e30f, l_run = 0.0f;
  f32x16 oacc[2], fs[2], lacc;
  for (int e = 0; e < 16; ++e) lacc[e] = 0.0f;
  for (int d = 0; d < 2; ++d)
    for (int e = 0; e < 16; ++e) { oacc[d][e] = 0.0f; fs[d][e] = 0.01f * (float)((lane + e) % 11); }
  int cur = 0;
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const char *kt = smem + cur * 8192, *vt = smem + (2 + cur) * 8192;
    if (DMA) {   // (a 64 KiB window per workgroup, walked tile by tile, wrapping)
      const long long step = (long long)(it & 3) * 16384;   // uniform
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int piece = wave * 2 + t;
        __builtin_amdgcn_global_load_lds((gptr_t)(kp0[t] + step), (lptr_t)(smem + (cur ^ 1) * 8192 + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vp0[t] + step), (lptr_t)(smem + (2 + (cur ^ 1)) * 8192 + piece * 1024), 16, 0, 0);
      }
    }
    f32x16 sacc[2];
    {
      bf16x8 kf[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          if (LDS)
            kf[s][kb] = *reinterpret_cast<const bf16x8 *>(kt + kb * 4096 + k_off + (((2 * s + hh) ^ k_sw) << 4));
          else
            kf[s][kb] = kc[kb];
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.0f;
      if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][kb], qf[s], sacc[kb], 0, 0, 0);
      if (NOFMA) {   // (ones x (-m): one more k-step per key block)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[kb], qf[kb], sacc[kb], 0, 0, 0);
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      if (PRIO || BLOCKS) __builtin_amdgcn_sched_barrier(0);
    }
    if (NOVALU) {   // calibration: the 16 MFMAs alone
      asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int d = 0; d < 2; ++d)
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[d], pc[s], oacc[d], 0, 0, 0);
      continue;
    }
    f32x16 sv[2];
    if (DEP_MV) { sv[0] = sacc[0]; sv[1] = sacc[1]; }
    else {
      sv[0] = fs[0]; sv[1] = fs[1];
      asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));   // the MFMAs stay
      fs[0][it & 15] += sacc[0][0] * 0.0f;                 // (a use that orders nothing much)
    }
    float mloc = -1.0e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2) mloc = fmaxf(fmaxf(mloc, sv[kb][e]), sv[kb][e + 1]);
    {
      unsigned ma = __builtin_bit_cast(unsigned, mloc), mb = ma;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
      mloc = fmaxf(__builtin_bit_cast(float, ma), __builtin_bit_cast(float, mb));
    }
    const float mz = mloc > -1.0e29f ? mloc * c_scale : -1.0e30f;
    if (__any(mz > m_run + 8.0f)) {
      const float m_new = fmaxf(m_run, mz);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
      if (SUMM) {
#pragma unroll
        for (int e = 0; e < 16; ++e) lacc[e] *= alpha;
      }
    }
    bf16x8 pf[4];
    float nm = -m_run, ps0 = 0.0f, ps1 = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        float p0, p1;
        if (NOFMA) { p0 = sv[kb][e]; p1 = sv[kb][e + 1]; }
        else { p0 = __builtin_fmaf(sv[kb][e], c_scale, nm); p1 = __builtin_fmaf(sv[kb][e + 1], c_scale, nm); }
        if (NOEXP) { p0 *= 0.5f; p1 *= 0.5f; }
        else { p0 = __builtin_amdgcn_exp2f(p0); p1 = __builtin_amdgcn_exp2f(p1); }
        if (!SUMM) { ps0 += p0; ps1 += p1; }
        pf[2 * kb + (e >> 3)][e & 7] = (__bf16)p0;
        pf[2 * kb + (e >> 3)][(e & 7) + 1] = (__bf16)p1;
        if ((e & 7) == 6) asm volatile("" : "+v"(nm), "+v"(ps0), "+v"(ps1));
      }
    l_run += ps0 + ps1;
    if (!DEP_VM) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        asm volatile("" : "+v"(pf[s]));   // computed, then dropped
        pf[s] = pc[s];
      }
    }
    if (PRIO || BLOCKS) __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        bf16x8 vf;
        if (LDS) {
          const int off0 = (16 * s + v_row0) * 128 + ((d * 64 + v_colb) ^ v_sw);
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vt + off0));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vt + off0 + 8 * 128));
          vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
          vf = vc[d];
        }
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oacc[d], 0, 0, 0);
      }
    if (SUMM) {
#pragma unroll
      for (int s = 0; s < 4; ++s) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[0], pf[s], lacc, 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (PRIO || BLOCKS) __builtin_amdgcn_sched_barrier(0);
    if (BAR) __syncthreads();
    cur ^= 1;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    cyc[4096 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r0;
    cyc[8192 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r1;
  }
  float s = l_run + m_run + lacc[0] + lacc[5];
  for (int d = 0; d < 2; ++d)
    for (int e = 0; e < 16; ++e) s += oacc[d][e] + fs[d][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}


// ---- the same pass as ONE software-pipelined stream per wave: the 8 S^T MFMAs of tile t + 1 and the
// 8 O^T MFMAs of tile t each followed by a chunk (8-14 instructions) of tile t's softmax, order
// pinned with sched_barrier; K fragments of t + 1 are read in the tail, V^T fragments of t between
// the first MFMAs.  LDSF: fragments come from LDS (else constants); BARF: s_barrier per pass.
#define SB() __builtin_amdgcn_sched_barrier(0)
template <bool LDSF, bool BARF, int WPE>
__global__ void __launch_bounds__(256, WPE) probe_pipe(float *out, long long *cyc, int iters, float c_scale, const char *gsrc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  for (int i = threadIdx.x; i < 4 * 8192 / 2; i += 256)
    reinterpret_cast<__bf16 *>(smem)[i] = (__bf16)(0.01f * (float)((i * 37) % 19 - 9));
  __syncthreads();
  bf16x8 qf[4], kc[2], vc[2];
  for (int s = 0; s < 4; ++s)
    for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)(0.02f * (float)((lane + 3 * s + e) % 7 - 3));
  for (int d = 0; d < 2; ++d)
    for (int e = 0; e < 8; ++e) {
      kc[d][e] = (__bf16)(0.03f * (float)((lane + d + e) % 5 - 2));
      vc[d][e] = (__bf16)(0.05f * (float)((lane + 2 * d + e) % 9 - 4));
    }
  const int k_off = r * 128, k_sw = (r >> 1) & 7;
  const int v_q4 = (lane & 15) >> 2, v_p = lane & 3, v_g1 = (lane >> 4) & 1;
  const int v_row0 = 4 * hh + v_q4, v_colb = (v_g1 * 16 + 4 * v_p) * 2, v_sw = ((v_row0 >> 1) & 1) << 6;
  float m_run = -1.0e30f, l_run = 0.0f;
  f32x16 oacc[2], sA[2], sB[2];
  for (int d = 0; d < 2; ++d)
    for (int e = 0; e < 16; ++e) { oacc[d][e] = 0.0f; sA[d][e] = 0.01f * (float)((lane + e) % 11); }
  bf16x8 kf[4][2];
  auto read_k = [&](const char *kt, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (LDSF) kf[s][kb] = *reinterpret_cast<const bf16x8 *>(kt + kb * 4096 + k_off + (((2 * s + hh) ^ k_sw) << 4));
      else kf[s][kb] = kc[kb];
    }
  };
#pragma unroll
  for (int s = 0; s < 4; ++s) read_k(smem, s);
  // one pass: sv = S(t) (complete), sn = S(t + 1) (written here)
  auto pass = [&](f32x16 (&sv)[2], f32x16 (&sn)[2], const char *kt_next, const char *vt) __attribute__((always_inline)) {
    bf16x8 vf[4][2], pf[4];
    auto read_v = [&](int s) __attribute__((always_inline)) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        if (LDSF) {
          const int off0 = (16 * s + v_row0) * 128 + ((d * 64 + v_colb) ^ v_sw);
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vt + off0));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vt + off0 + 8 * 128));
          vf[s][d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else vf[s][d] = vc[d];
      }
    };
    auto qk = [&](int s, int kb, bool first) __attribute__((always_inline)) {
      if (first) {
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.0f;
        sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][kb], qf[s], z, 0, 0, 0);
      } else sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][kb], qf[s], sn[kb], 0, 0, 0);
    };
    float m0 = -1.0e30f, m1 = -1.0e30f;
    float nm, ps0 = 0.0f, ps1 = 0.0f;
    auto frag_half = [&](int f, int h) __attribute__((always_inline)) {   // 4 of fragment f's 8 values
      const int kb = f >> 1, e0 = (f & 1) * 8 + h * 4;
      float p[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[kb][e0 + e], c_scale, nm));
      ps0 += p[0]; ps1 += p[1]; ps0 += p[2]; ps1 += p[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) pf[f][h * 4 + e] = (__bf16)p[e];
    };
    // M0..M7: S(t+1); chunks: two max chains, the exchange, fragment halves 0..4
    qk(0, 0, true); SB();
#pragma unroll
    for (int e = 0; e < 8; e += 2) { m0 = fmaxf(fmaxf(m0, sv[0][e]), sv[0][e + 1]); m1 = fmaxf(fmaxf(m1, sv[1][e]), sv[1][e + 1]); }
    read_v(0);
    SB(); qk(0, 1, true); SB();
#pragma unroll
    for (int e = 8; e < 16; e += 2) { m0 = fmaxf(fmaxf(m0, sv[0][e]), sv[0][e + 1]); m1 = fmaxf(fmaxf(m1, sv[1][e]), sv[1][e + 1]); }
    read_v(1);
    SB(); qk(1, 0, false); SB();
    float mloc = fmaxf(m0, m1);
    {
      unsigned ma = __builtin_bit_cast(unsigned, mloc), mb = ma;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
      mloc = fmaxf(__builtin_bit_cast(float, ma), __builtin_bit_cast(float, mb));
    }
    const float mz = mloc > -1.0e29f ? mloc * c_scale : -1.0e30f;
    read_v(2);
    SB();
    if (__any(mz > m_run + 8.0f)) {
      const float m_new = fmaxf(m_run, mz);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
    }
    nm = -m_run;
    SB(); qk(1, 1, false); SB();
    frag_half(0, 0); read_v(3);
    SB(); qk(2, 0, false); SB();
    frag_half(0, 1);
    SB(); qk(2, 1, false); SB();
    frag_half(1, 0);
    SB(); qk(3, 0, false); SB();
    frag_half(1, 1);
    SB(); qk(3, 1, false); SB();
    frag_half(2, 0);
    // M8..M15: O(t)
    SB(); oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0], pf[0], oacc[0], 0, 0, 0); SB();
    frag_half(2, 1);
    SB(); oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][1], pf[0], oacc[1], 0, 0, 0); SB();
    frag_half(3, 0);
    SB(); oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][0], pf[1], oacc[0], 0, 0, 0); SB();
    frag_half(3, 1);
    l_run += ps0 + ps1;
    SB(); oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][1], pf[1], oacc[1], 0, 0, 0); SB();
    read_k(kt_next, 0); read_k(kt_next, 1);
    SB(); oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2][0], pf[2], oacc[0], 0, 0, 0); SB();
    read_k(kt_next, 2); read_k(kt_next, 3);
    SB(); oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2][1], pf[2], oacc[1], 0, 0, 0);
    oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3][0], pf[3], oacc[0], 0, 0, 0);
    oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3][1], pf[3], oacc[1], 0, 0, 0);
    SB();
    if (BARF) __syncthreads();
  };
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 2) {
    pass(sA, sB, smem + 8192, smem + 2 * 8192);
    pass(sB, sA, smem, smem + 3 * 8192);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    cyc[4096 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r0;
    cyc[8192 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r1;
  }
  float s = l_run + m_run;
  for (int d = 0; d < 2; ++d)
    for (int e = 0; e < 16; ++e) s += oacc[d][e] + sA[d][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float *out; static long long *cyc; static char *gsrc; static long long h[3 * 4096];
static const int iters = 4000;

typedef void (*kern_t)(float *, long long *, int, float, const char *);
static void run_k(kern_t k, int w);
template <int FLAGS> static void run(int w) { run_k(probe<FLAGS>, w); }
static void run_k(kern_t kfn, int w) {
  const int blocks = 256 * w;
  const int lds = w == 1 ? 100 * 1024 : w == 2 ? 64 * 1024 : 52 * 1024;
  hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kfn, 256, lds);
  if (occ != w) printf(" [occupancy API says %d] ", occ);
  for (int rep = 0; rep < 2; ++rep) { kfn<<<blocks, 256, lds>>>(out, cyc, iters, 0.18f, gsrc); hipDeviceSynchronize(); }
  hipMemcpy(h, cyc, 3 * 4096 * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
  // (100 MHz wall clock: the span of the launch against the mean life of a wave -- 1.0 when every
  // workgroup was resident from start to end)
  long long lo = h[4096], hi = h[8192]; double life = 0;
  for (int i = 0; i < blocks * 4; ++i) {
    if (h[4096 + i] < lo) lo = h[4096 + i];
    if (h[8192 + i] > hi) hi = h[8192 + i];
    life += (double)(h[8192 + i] - h[4096 + i]);
  }
  life /= blocks * 4;
  const double mhz = s / (blocks * 4) / life * 100.0;
  // (the oldest wave of a SIMD is served first: waves that share a SIMD do not live equally long,
  // so the SIMD's cost per pass is the launch's span over iters * W, not a wave's life over W)
  printf("  %d WG/CU: wave %6.0f, SIMD %6.1f (span/life %.2f, %4.0f MHz)", w, s / (blocks * 4) / iters,
         (double)(hi - lo) * mhz / 100.0 / iters / w, (double)(hi - lo) / life, mhz);
}
template <int FLAGS> static void row(const char *what) {
  printf("%-58s", what);
  for (int w = 1; w <= 3; ++w) run<FLAGS>(w);
  printf("\n");
  fflush(stdout);
}

int main() {
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  hipMalloc(&cyc, 3 * 4096 * sizeof(long long));
  hipMalloc(&gsrc, 64 << 20);
  hipMemset(gsrc, 0, 64 << 20);
  row<32>("16 MFMAs alone");
  printf("%-58s", "pipelined stream"); for (int w = 1; w <= 2; ++w) run_k(probe_pipe<false, false, 2>, w); printf("\n");
  printf("%-58s", "pipelined stream + LDS reads"); for (int w = 1; w <= 2; ++w) run_k(probe_pipe<true, false, 2>, w); printf("\n");
  printf("%-58s", "pipelined stream + LDS reads + barrier"); for (int w = 1; w <= 2; ++w) run_k(probe_pipe<true, true, 2>, w); printf("\n");
  fflush(stdout);
  row<0>("independent phases");
  row<1>("softmax reads the MFMA results");
  row<2>("second MFMA block reads the softmax output");
  row<3>("both dependencies");
  row<3 | 128>("both, MFMAs in two blocks of 8");
  row<3 | 64>("both, blocks, s_setprio 3 around the blocks");
  row<3 | 4 | 8 | 64>("both + barrier + LDS reads, blocks, s_setprio");
  row<3 | 4>("both + barrier");
  row<3 | 8>("both + LDS fragment reads");
  row<3 | 4 | 8>("both + barrier + LDS reads");
  row<3 | 4 | 8 | 256>("both + barrier + LDS reads + DMA");
  row<3 | 4 | 256>("both + barrier + DMA (no LDS reads)");
  row<3 | 4 | 8 | 256 | 512>("everything, no fma (+2 MFMAs)");
  row<3 | 4 | 8 | 256 | 1024>("everything, row sums by 4 MFMAs (no adds)");
  row<3 | 4 | 8 | 256 | 512 | 1024>("everything, no fma, no adds (+6 MFMAs)");
  row<3 | 512 | 1024>("both dependencies, no fma, no adds (+6 MFMAs)");
  row<3 | 4 | 8 | 16 | 256>("both + barrier + LDS reads + DMA, no exps");
  row<3 | 4 | 8 | 16>("both + barrier + LDS reads, no exps");
  row<3 | 16>("both dependencies, no exps");
  row<16>("independent phases, no exps");
  return 0;
}
