// Fused multi-head dot-product attention forward for gfx950 (bf16, d_k = d_v = 64).
// Reference: multi_head_dot_attn, attention_cell.py:332-524 --
//   scores = Q K^T (+ edge_scores);  w = masked_softmax(scores, mask, temperature);  ctx = w V
// computed flash-style: the (B, H, Lq, Lm) tensor is never written to HBM.
//
// Work decomposition: one 256-thread workgroup = 128 query rows of one (batch, head); each of the
// 4 waves owns 32 query rows.  K / V tiles of 64 keys are staged into double-buffered LDS with
// 16-byte direct-to-LDS loads and shared by the 4 waves.
//   S^T = K . Q^T  with v_mfma_f32_32x32x16_bf16 (A = K rows via ds_read_b128 from an XOR-swizzled
//         image, B = Q held in registers): a lane owns ONE query row (column l&31) and 16 of the
//         32 keys of each key block, so the softmax needs a single cross-lane exchange (lane^32).
//   O^T = V^T . P^T: the S^T accumulator, converted pairwise to bf16, is directly the B operand
//         (no LDS round trip); V^T fragments come from the row-major V image with the hardware
//         transposing read ds_read_b64_tr_b16.
// The NKT / NTK / TNK layouts and the fused-QKV buffer are stride descriptors.  The mask is either
// the valid_length predicate (tiles beyond valid_length are skipped entirely, padded query rows
// produce exact zeros), the causal predicate (j <= i) & (i < valid_length) (tiles above the block's
// diagonal are skipped) or a dense byte mask; edge_scores (T5 bias) is added before the
// temperature (attention_cell.py:453-455).
#include "common.h"
#include "kernels.h"

namespace gnlp {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef __attribute__((address_space(3))) bf16x4 *lds_bf16x4_ptr;

constexpr int KV_TILE = 64;
constexpr int TILE_BYTES = KV_TILE * 128;   // 64 keys x 64 bf16
constexpr float NEG_BIG = -1.0e30f;

#ifdef ATTN_STAMP   // diagnostic build only (tools/attn_stamp.py): cycles and clock of a workgroup
__device__ unsigned long long attn_dbg[4096 * 4];
#endif

// (three workgroups per CU -- at most 168 registers -- wherever the variant allows: with two the
// kernel is no longer issue-bound; measured on the causal variant, 198 -> 168 registers)
template <bool HAS_BIAS, bool DENSE_MASK, bool CAUSAL, bool BIAS_F32 = false>
__global__ void __launch_bounds__(256, (HAS_BIAS && !BIAS_F32) || DENSE_MASK ? 2 : 3)
attn_fwd_bf16_d64(FusedAttn a) {
#ifdef ATTN_STAMP
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // K0 K1 V0 V1 [, per wave: 32 x 64 mask bytes] [, per wave: 32 x 64 bf16 bias values]
  constexpr bool BIAS_ST = HAS_BIAS && !BIAS_F32;
  constexpr int MST_STRIDE = 80, BST_STRIDE = 144;   // row strides of the two staging images
  constexpr int MST_BYTES = DENSE_MASK ? 4 * 32 * MST_STRIDE : 0;
  constexpr int BST_BYTES = BIAS_ST ? 4 * 32 * BST_STRIDE : 0;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES + MST_BYTES + BST_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  // Workgroups are placed on the 8 XCDs round-robin by dispatch order.  Remap so that the query
  // blocks of one (batch, head) share an XCD and are dispatched together: K / V of a head are then
  // fetched into one L2 once instead of once per query block (measured 576 -> 192+ MB per launch
  // at L = 512).  Bijective when B * H is a multiple of 8; identity otherwise.
  int64_t b = blockIdx.z, h = blockIdx.y;
  int64_t qb = blockIdx.x;
  {
    const unsigned nx = gridDim.x, nh = gridDim.y * gridDim.z;
    if ((nh & 7u) == 0) {
      const unsigned lin = blockIdx.x + nx * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned grp = lin / (8 * nx), i = lin - grp * 8 * nx;
      const unsigned head = grp * 8 + (i & 7);
      qb = i >> 3;
      b = head / gridDim.y;
      h = head - (unsigned)b * gridDim.y;
    }
  }
  // causal predicate: the last query blocks see the most keys -- dispatch them first
  if (!DENSE_MASK && CAUSAL) qb = (int64_t)gridDim.x - 1 - qb;
  const int64_t q0 = qb * 128;

  int64_t kv_len = a.Lm;
  int64_t q_valid = a.Lq;
  constexpr bool causal = !DENSE_MASK && CAUSAL;   // (j <= i) & (i < vl): attention_cell.py:168-180
  const bool use_pred = !DENSE_MASK && (a.vl != nullptr || causal);
  if (use_pred) {
    if (a.vl) {
      int64_t n = load_len(a.vl, a.vl_dtype, b);
      n = n < 0 ? 0 : n;
      kv_len = n < a.Lm ? n : a.Lm;
      if (a.vl_pair) {   // memory mask: the queries have their own lengths
        n = load_len(a.vl, a.vl_dtype, a.B + b);
        n = n < 0 ? 0 : n;
      }
      q_valid = n < a.Lq ? n : a.Lq;
    }
    // key tiles entirely above the diagonal of this block's last row are never staged
    if (causal && q0 + 128 < kv_len) kv_len = q0 + 128;
  }
  const int64_t my_q = q0 + wave * 32 + r;  // this lane's query row
  // packed ragged batch: sequence b starts at token row cu[b] (batch strides are then 0) and rows
  // beyond its valid length belong to the next sequence: they are never written
  const int64_t tok0 = a.cu ? (int64_t)a.cu[b] : 0;
  const int64_t q_store = a.cu ? q_valid : a.Lq;
  bf16 *const obase = (bf16 *)a.o + b * a.os[0] + h * a.os[1] + tok0 * a.os[2];
  const int ntiles = (int)((kv_len + KV_TILE - 1) / KV_TILE);

  if (q0 >= q_valid || ntiles == 0) {  // block-uniform: every query row of the block is padded
    if (my_q < q_store) {
      bf16x4 z = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
      for (int g = 0; g < 8; ++g)
        *reinterpret_cast<bf16x4 *>(obase + my_q * a.os[2] + g * 8 + hh * 4) = z;
      if (a.lse) {   // training (never packed): statistics and rounding residual of a zero row
        if (hh == 0) a.lse[(b * a.H + h) * a.Lq + my_q] = 0.0f;
        bf16 *const lbase = (bf16 *)a.o_lo + b * a.os[0] + h * a.os[1];
#pragma unroll
        for (int g = 0; g < 8; ++g)
          *reinterpret_cast<bf16x4 *>(lbase + my_q * a.os[2] + g * 8 + hh * 4) = z;
      }
    }
    return;
  }

  // ---- Q fragments (B operand of S^T = K Q^T): Q[my_q][16 s + 8 hh .. +8] ----
  const int64_t qrow = my_q < a.Lq ? my_q : a.Lq - 1;
  const bf16 *qp = (const bf16 *)a.q + b * a.qs[0] + h * a.qs[1] + (tok0 + qrow) * a.qs[2];
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8 *>(qp + 16 * s + 8 * hh);

  // ---- staging: each wave moves 2 K pieces and 2 V pieces (8 rows x 128 B each) per tile ----
  const int srow = lane >> 3, cpos = lane & 7;
  const bf16 *kbase = (const bf16 *)a.k + b * a.ks[0] + h * a.ks[1] + tok0 * a.ks[2];
  const bf16 *vbase = (const bf16 *)a.v + b * a.vs[0] + h * a.vs[1] + tok0 * a.vs[2];
  // Lane pointers of tile 0; a full tile adds a uniform offset (scalar ALU) -- the per-lane
  // 64-bit row * stride products cost 74 of the 248 vector-ALU issue slots per tile (PMC +
  // disassembly) -- only the last, partial tile recomputes clamped rows.
  const bf16 *kp0[2], *vp0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = (wave * 2 + t) * 8 + srow;
    kp0[t] = kbase + row * a.ks[2] + (cpos ^ ((row >> 1) & 7)) * 8;          // K image: b128 row reads
    vp0[t] = vbase + row * a.vs[2] + (cpos ^ (((row >> 1) & 1) << 2)) * 8;   // V image: transposed b64 reads
  }
  auto stage = [&](int buf, int tile) {
    const int64_t adv = (int64_t)tile * KV_TILE;           // uniform
    if (adv + KV_TILE <= a.Lm) {
      const int64_t kstep = adv * a.ks[2], vstep = adv * a.vs[2];   // uniform
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int piece = wave * 2 + t;
        __builtin_amdgcn_global_load_lds((gptr_t)(kp0[t] + kstep),
                                         (lptr_t)(smem + buf * TILE_BYTES + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vp0[t] + vstep),
                                         (lptr_t)(smem + (2 + buf) * TILE_BYTES + piece * 1024), 16,
                                         0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int piece = wave * 2 + t;
        const int row = piece * 8 + srow;
        int64_t key = adv + row;
        if (key > a.Lm - 1) key = a.Lm - 1;
        const int ck = cpos ^ ((row >> 1) & 7);
        const int cv = cpos ^ (((row >> 1) & 1) << 2);
        __builtin_amdgcn_global_load_lds((gptr_t)(kbase + key * a.ks[2] + ck * 8),
                                         (lptr_t)(smem + buf * TILE_BYTES + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vbase + key * a.vs[2] + cv * 8),
                                         (lptr_t)(smem + (2 + buf) * TILE_BYTES + piece * 1024), 16,
                                         0, 0);
      }
    }
  };

  // K fragment (A operand): row = kb*32 + r, 16-byte chunk 2s + hh
  const int k_off = r * 128;
  const int k_sw = (r >> 1) & 7;
  // V^T fragment via ds_read_b64_tr_b16: lane (i = l&15 -> q4 = i>>2, p = i&3; G = l>>4):
  //   key = 16 s + 4 hh + q4 (+8), column = dvb*32 + (G&1)*16 + 4 p
  const int v_q4 = (lane & 15) >> 2, v_p = lane & 3, v_g1 = (lane >> 4) & 1;
  const int v_row0 = 4 * hh + v_q4;                      // + 16 s (+ 8)
  const int v_colb = (v_g1 * 16 + 4 * v_p) * 2;          // byte column inside a 64-byte half
  const int v_sw = ((v_row0 >> 1) & 1) << 6;             // same for row0 + 8 and + 16 s

  const float c_scale = a.inv_temperature * 1.44269504088896340736f;
  float m_run = NEG_BIG, l_run = 0.0f;
  f32x16 oacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.0f;

  const bool row_ok = my_q < q_valid;
  const bool rows_all_ok = __all(row_ok);
  constexpr float RESCALE_THR = 8.0f;
  const int64_t boff = HAS_BIAS ? b * a.bs[0] + h * a.bs[1] + qrow * a.bs[2] : 0;
  const float *browf = (HAS_BIAS && BIAS_F32) ? (const float *)a.bias + boff : nullptr;

  // A lane owns a query ROW: read straight from memory, the 64 mask bytes / bias values of a
  // tile would be 64 scattered lines per load instruction (measured at B = 64, L = 512: dense
  // mask 568 us, bf16 bias 229 us against 101 us for the predicate).  Instead the wave fetches its
  // 32 x 64 block with 16-byte loads along the key axis, one tile ahead, and hands it over through
  // a private LDS image (row strides 80 / 144 bytes: a row's reads fall in distinct banks).
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  char *const mst = smem + 4 * TILE_BYTES + wave * (32 * MST_STRIDE);
  char *const bst = smem + 4 * TILE_BYTES + MST_BYTES + wave * (32 * BST_STRIDE);
  constexpr bool mask_st = DENSE_MASK;               // (the host guarantees unit key stride)
  u32x4 mreg[2];
  bf16x8 breg[4];
  auto mask_fetch = [&](int tile) {
    const int64_t j0 = (int64_t)tile * KV_TILE + (lane & 3) * 16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int64_t gq = q0 + wave * 32 + (lane >> 2) + 16 * u;
      gq = gq < a.Lq ? gq : a.Lq - 1;
      const uint8_t *p = a.mask + b * a.ms[0] + gq * a.ms[1] + j0;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (j0 + 16 <= a.Lm) {
        v = *reinterpret_cast<const u32x4 *>(p);
      } else if (j0 < a.Lm) {   // the row ends inside this piece
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (j0 + i < a.Lm) v[i >> 2] |= (unsigned)p[i] << (8 * (i & 3));
      }
      mreg[u] = v;
    }
  };
  auto mask_store = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      *reinterpret_cast<u32x4 *>(mst + ((lane >> 2) + 16 * u) * MST_STRIDE + (lane & 3) * 16) = mreg[u];
  };
  auto bias_fetch = [&](int tile) {
    const int64_t j0 = (int64_t)tile * KV_TILE + (lane & 7) * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t gq = q0 + wave * 32 + (lane >> 3) + 8 * u;
      gq = gq < a.Lq ? gq : a.Lq - 1;
      const bf16 *p = (const bf16 *)a.bias + b * a.bs[0] + h * a.bs[1] + gq * a.bs[2] + j0;
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (bf16)0.0f;
      if (j0 + 8 <= a.Lm) {
        v = *reinterpret_cast<const bf16x8 *>(p);
      } else if (j0 < a.Lm) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (j0 + i < a.Lm) v[i] = p[i];
      }
      breg[u] = v;
    }
  };
  auto bias_store = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<bf16x8 *>(bst + ((lane >> 3) + 8 * u) * BST_STRIDE + (lane & 7) * 16) = breg[u];
  };

  if (mask_st) mask_fetch(0);
  if (BIAS_ST) bias_fetch(0);
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int tile = 0; tile < ntiles; ++tile) {
    // (the block's loads come BEFORE the next tile's DMA in program order: waiting for them never
    // waits for the DMA)
    if (mask_st) mask_store();
    if (BIAS_ST) bias_store();
    if (tile + 1 < ntiles) {
      if (mask_st) mask_fetch(tile + 1);
      if (BIAS_ST) bias_fetch(tile + 1);
      stage(cur ^ 1, tile + 1);
    }
    const char *kt = smem + cur * TILE_BYTES;
    const char *vt = smem + (2 + cur) * TILE_BYTES;
    // dense mask: this lane's 64 bytes of the tile; a wave whose 32 x 64 block is switched off
    // entirely skips the tile (what the causal predicate does with the tiles above the diagonal),
    // one whose block is switched on entirely skips the per-element selects
    unsigned mw[2][4];
    bool mask_full = true;
    if (DENSE_MASK) {
      unsigned any = 0;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned w = *reinterpret_cast<const unsigned *>(mst + r * MST_STRIDE + kb * 32 + g * 8 + hh * 4);
          mw[kb][g] = w;
          any |= w;
          mask_full = mask_full && ((w - 0x01010101u) & ~w & 0x80808080u) == 0;   // no zero byte
        }
      if (__all(any == 0)) {   // wave-uniform
        __syncthreads();
        cur ^= 1;
        continue;
      }
      mask_full = __all(mask_full);
    }

    // ---- S^T = K Q^T : sacc[kb][reg] = score(key = kb*32 + (reg&3) + 8*(reg>>2) + 4*hh, my_q) ----
    f32x16 sacc[2];
    {
      bf16x8 kf[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          kf[s][kb] = *reinterpret_cast<const bf16x8 *>(kt + kb * 4096 + k_off +
                                                        (((2 * s + hh) ^ k_sw) << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.0f;
      if (HAS_BIAS) {   // the accumulators START from the bias (attention_cell.py:453-455: added
        // before the temperature): no add after the MFMAs, the loads fly under the K reads.
        // Host guarantees unit key stride; loads may be unaligned.
        const int64_t kb0 = (int64_t)tile * KV_TILE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int64_t j0 = kb0 + kb * 32 + g * 8 + hh * 4;
            if (BIAS_F32) {   // a float32 copy of the bias: the loads ARE the accumulator init
              if (j0 + 4 <= a.Lm) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(browf + j0);
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc[kb][g * 4 + e] = b4[e];
              } else if (j0 < a.Lm) {
#pragma unroll
                for (int e = 0; e < 3; ++e)
                  if (j0 + e < a.Lm) sacc[kb][g * 4 + e] = browf[j0 + e];
              }
            } else {   // from the wave's staged block (zeros beyond the row's end)
              const bf16x4 b4 = *reinterpret_cast<const bf16x4 *>(
                  bst + r * BST_STRIDE + (kb * 32 + g * 8 + hh * 4) * 2);
#pragma unroll
              for (int e = 0; e < 4; ++e) sacc[kb][g * 4 + e] = (float)b4[e];
            }
          }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][kb], qf[s], sacc[kb], 0, 0, 0);
    }

    // ---- bias, mask, online softmax (one query row per lane, keys split with lane^32) ----
    // Scores stay in raw units; the temperature is folded into the exp2 argument
    // (p = exp2(s*c - m)).  The running max is only raised -- and O / l rescaled -- when some row of
    // the wave grew by more than RESCALE_THR (log2 units), so that most tiles skip the O-wide
    // multiply; p is then bounded by 2^THR, harmless in bf16 (same relative precision).
    const int64_t kbase_idx = (int64_t)tile * KV_TILE;
    // (causal: the tile's last key lies beyond the wave's first row -> some (i, j) have j > i)
    const bool tail = (DENSE_MASK && !mask_full) || !rows_all_ok || (kbase_idx + KV_TILE > kv_len) ||
                      (causal && tile * KV_TILE + KV_TILE - 1 > (int)q0 + wave * 32);
    if (tail) {   // wave-uniform: only the tile straddling kv_len, dense masks, padded rows
      // keys of this tile that may be on for this lane's row: those below kv_len, up to the
      // diagonal under the causal predicate, none for a padded row -- one 32-bit limit per lane
      // (32-bit: fused_attn_supported bounds the sequence lengths)
      const int kb32 = tile * KV_TILE;
      int lim = (int)kv_len - kb32;
      if (!DENSE_MASK && causal) lim = min(lim, (int)my_q - kb32 + 1);
      lim = (row_ok ? max(min(lim, KV_TILE), 0) : 0) - hh * 4;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bool on = kb * 32 + g * 8 + e < lim;          // (key index inside the tile, less 4 hh)
            if (DENSE_MASK) on = on && ((mw[kb][g] >> (8 * e)) & 0xffu) != 0;
            sacc[kb][g * 4 + e] = on ? sacc[kb][g * 4 + e] : NEG_BIG;
          }
        }
    }
    float mloc = NEG_BIG;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2)            // fmax(fmax(m, a), b) = one v_max3_f32
        mloc = fmaxf(fmaxf(mloc, sacc[kb][e]), sacc[kb][e + 1]);
    {   // lane ^ 32 exchange in the vector ALU (v_permlane32_swap) instead of an LDS round trip
      // As inline assembly: through __builtin_amdgcn_permlane32_swap hipcc 7.2 drops the maximum
      // of the two results (no v_max after the swap in the ISA, whether the operands are the same
      // value or not) and the running maximum then only sees the keys of the lower lane half --
      // harmless while both halves hold similar scores, an overflow of exp2 (inf / inf = NaN) when
      // the largest score of a row sits in the upper half, e.g. when every key of the lower half
      // is masked.  tests/test_gpu_encoder.py::test_fused_attention_max_over_both_lane_halves
      unsigned ma = __builtin_bit_cast(unsigned, mloc), mb = ma;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
      mloc = fmaxf(__builtin_bit_cast(float, ma), __builtin_bit_cast(float, mb));
    }
    const float mz = mloc > -1.0e29f ? mloc * c_scale : NEG_BIG;   // scaled units
    if (__any(mz > m_run + RESCALE_THR)) {                          // wave-uniform
      const float m_new = fmaxf(m_run, mz);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
    }
    bf16x8 pf[4];  // B operand fragments of O^T = V^T P^T, k-step s = 2 kb + t
    // Single-issue v_fma_f32 / v_add_f32 on purpose (the file is built with -fno-slp-vectorize):
    // a packed f32 instruction does not run beside another wave's MFMAs at all
    // (tools/coissue_probe2.hip: 112 v_pk_fma_f32 beside 16 MFMAs of the SIMD partner take
    // 576 + 512 cycles, 112 v_fma_f32 653), and three waves share a SIMD here.
    float nm = -m_run;
    float ps0 = 0.0f, ps1 = 0.0f;
    if (tail) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const float s0 = sacc[kb][e], s1 = sacc[kb][e + 1];
          float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0, c_scale, nm));
          float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1, c_scale, nm));
          p0 = s0 > -1.0e29f ? p0 : 0.0f;
          p1 = s1 > -1.0e29f ? p1 : 0.0f;
          ps0 += p0;
          ps1 += p1;
          pf[2 * kb + (e >> 3)][e & 7] = (bf16)p0;
          pf[2 * kb + (e >> 3)][(e & 7) + 1] = (bf16)p1;
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][e], c_scale, nm));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][e + 1], c_scale, nm));
          ps0 += p0;
          ps1 += p1;
          pf[2 * kb + (e >> 3)][e & 7] = (bf16)p0;
          pf[2 * kb + (e >> 3)][(e & 7) + 1] = (bf16)p1;
          // (one fragment at a time -- the next one's multiply-adds depend on this opaque copy of
          // nm: all 32 of them hoisted ahead of the exponentials cost 30 registers and spills)
          if ((e & 7) == 6) asm volatile("" : "+v"(nm), "+v"(ps0), "+v"(ps1));
        }
    }
    l_run += ps0 + ps1;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int off0 = (16 * s + v_row0) * 128 + ((d * 64 + v_colb) ^ v_sw);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vt + off0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vt + off0 + 8 * 128));
        bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oacc[d], 0, 0, 0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

#ifdef ATTN_STAMP
  if (threadIdx.x == 0) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin < 4096) {
      attn_dbg[lin * 4 + 0] = __builtin_amdgcn_s_memtime() - st_c0;
      attn_dbg[lin * 4 + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
      attn_dbg[lin * 4 + 2] = (unsigned long long)ntiles;
      attn_dbg[lin * 4 + 3] = st_r0;
    }
  }
#endif
  // ---- finalize: O / l.  A lane holds ctx[my_q][d*32 + 8 g + 4 hh .. +3]; the wave transposes its
  // 32 x 64 block through a private 4 KiB LDS image (8-byte slot index XOR (row & 15)) so that the
  // global stores are full 128-byte rows, 16 bytes per lane.
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (a.lse && hh == 0 && my_q < a.Lq)   // training: what the backward kernels rebuild P from
    a.lse[(b * a.H + h) * a.Lq + my_q] = l_tot > 0.0f ? m_run + __builtin_amdgcn_logf(l_tot) : 0.0f;
  char *ow = smem + wave * 4096;   // K/V tiles are dead: the loop ended with a barrier
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(oacc[d][g * 4 + e] * inv_l);
      const int slot = d * 8 + g * 2 + hh;
      *reinterpret_cast<bf16x4 *>(ow + r * 128 + ((slot ^ (r & 15)) << 3)) = o;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const int oc = lane & 7, orow0 = lane >> 3;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + orow0;
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(ow + row * 128 + (((2 * oc) ^ (row & 15)) << 3));
    const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(ow + row * 128 + (((2 * oc + 1) ^ (row & 15)) << 3));
    const bf16x8 o8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    const int64_t qq = q0 + wave * 32 + row;
    if (qq < q_store) *reinterpret_cast<bf16x8 *>(obase + qq * a.os[2] + oc * 8) = o8;
  }
  if (a.lse) {   // uniform.  Training: the residual of ctx's bf16 rounding, same route
    bf16 *const lbase = (bf16 *)a.o_lo + b * a.os[0] + h * a.os[1];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = oacc[d][g * 4 + e] * inv_l;
          o[e] = (bf16)(x - (float)(bf16)x);
        }
        const int slot = d * 8 + g * 2 + hh;
        *reinterpret_cast<bf16x4 *>(ow + r * 128 + ((slot ^ (r & 15)) << 3)) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + orow0;
      const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(ow + row * 128 + (((2 * oc) ^ (row & 15)) << 3));
      const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(ow + row * 128 + (((2 * oc + 1) ^ (row & 15)) << 3));
      const bf16x8 o8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      const int64_t qq = q0 + wave * 32 + row;
      if (qq < a.Lq) *reinterpret_cast<bf16x8 *>(lbase + qq * a.os[2] + oc * 8) = o8;
    }
  }
}

__global__ void __launch_bounds__(256) widen_bf16_kernel(float *out, const bf16 *in, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
void launch_widen_bf16(float *out, const void *in, int64_t n, hipStream_t s) {
  if (n > 0) widen_bf16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(out, (const bf16 *)in, n);
}

bool fused_attn_supported(const FusedAttn &a, int64_t dk, int64_t dv) {
  if (dk != 64 || dv != 64) return false;
  if (a.B < 1 || a.H < 1 || a.Lq < 1 || a.Lm < 1) return false;
  if (a.H > 65535 || a.B > 65535) return false;
  if (a.Lq > (1 << 30) || a.Lm > (1 << 30)) return false;   // 32-bit position arithmetic
  // 16-byte aligned rows for q / k / v, 8-byte for the output
  const int64_t *ss[3] = {a.qs, a.ks, a.vs};
  for (int t = 0; t < 3; ++t)
    for (int e = 0; e < 3; ++e)
      if (ss[t][e] % 8 != 0) return false;
  for (int e = 0; e < 3; ++e)
    if (a.os[e] % 8 != 0) return false;
  if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) % 16 != 0) return false;
  if (a.bias) {  // vector loads of 4 consecutive keys of the bias row: unit key stride; any row /
    // head / batch stride (the Toeplitz table of the T5 stack has row stride -1: its 8-byte loads
    // are unaligned, which global memory allows); the last 1-3 keys of a row are read one by one
    if (a.bs[3] != 1 || (uintptr_t)a.bias % (a.bias_f32 ? 4 : 2) != 0) return false;
  }
  if ((uintptr_t)a.o % 16 != 0) return false;
  return true;
}

void launch_fused_attn_bf16(const FusedAttn &a, hipStream_t s) {
  dim3 grid((unsigned)((a.Lq + 127) / 128), (unsigned)a.H, (unsigned)a.B);
  // (the causal predicate is a template argument: as a run-time flag it cost the full-length
  // non-causal case 20 %: 102 -> 123 us per launch at the BERT-large shape)
  if (a.bias) {
    if (a.bias_f32) attn_fwd_bf16_d64<true, false, false, true><<<grid, 256, 0, s>>>(a);
    else if (a.mask) attn_fwd_bf16_d64<true, true, false><<<grid, 256, 0, s>>>(a);
    else if (a.causal) attn_fwd_bf16_d64<true, false, true><<<grid, 256, 0, s>>>(a);
    else attn_fwd_bf16_d64<true, false, false><<<grid, 256, 0, s>>>(a);
  } else {
    if (a.mask) attn_fwd_bf16_d64<false, true, false><<<grid, 256, 0, s>>>(a);
    else if (a.causal) attn_fwd_bf16_d64<false, false, true><<<grid, 256, 0, s>>>(a);
    else attn_fwd_bf16_d64<false, false, false><<<grid, 256, 0, s>>>(a);
  }
}

}  // namespace gnlp

#ifdef ATTN_STAMP
extern "C" int gnlp_debug_attn_read(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gnlp::attn_dbg), (size_t)n * sizeof(unsigned long long));
}
#endif
