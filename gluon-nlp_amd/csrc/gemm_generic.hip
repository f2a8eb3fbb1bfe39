// Shape-agnostic batched strided GEMM on the vector ALU (fp32 accumulate).
// Used where the MFMA kernels' shape constraints do not hold (tiny head dims such as the
// reference tests' d_k = 8 / d_v = 6, K not a multiple of 32, unaligned rows) and by the unfused
// attention path that materialises `scores` / `attn_weights` (attention_cell.py:452-463,524).
//   C[b0,b1,m,n] = combine(act(alpha * sum_k A[b0,b1,m,k] * B[b0,b1,(n,k)|(k,n)] + bias[n]), other)
// 64x64 output tile per 256-thread block, 4x4 outputs per thread, 16-deep K slices through LDS.
#include "common.h"
#include "kernels.h"

namespace gnlp {

template <typename TI>
__device__ __forceinline__ float ld(const void *p, int64_t i) { return to_f32(((const TI *)p)[i]); }

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
generic_gemm_kernel(GenericGemm g) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tid = threadIdx.x;
  // grid.y / grid.z are capped at 65535: a block walks the batches / row tiles beyond the cap
  // (block-uniform loops, so the barriers inside stay legal)
  const int64_t nbatch = g.nb0 * g.nb1, mtiles = (g.M + 63) / 64;
  for (int64_t bz = blockIdx.z; bz < nbatch; bz += gridDim.z)
  for (int64_t mt = blockIdx.y; mt < mtiles; mt += gridDim.y) {
  const int64_t b0 = bz / g.nb1, b1 = bz % g.nb1;
  const int64_t m0 = mt * 64, n0 = (int64_t)blockIdx.x * 64;
  const TI *A = (const TI *)g.A + b0 * g.sAb0 + b1 * g.sAb1;
  const TI *Bm = (const TI *)g.Bm + b0 * g.sBb0 + b1 * g.sBb1;
  const int tx = tid & 15, ty = tid >> 4;  // thread owns rows ty*4.., cols tx*4..
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

  for (int64_t k0 = 0; k0 < g.K; k0 += 16) {
    // A tile: 64 rows x 16 k  -> As[k][m]
    for (int e = tid; e < 64 * 16; e += 256) {
      const int kk = e & 15, mm = e >> 4;
      const int64_t m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < g.M && k < g.K) ? to_f32(A[m * g.sAm + k]) : 0.0f;
    }
    if (g.trans_b) {  // B is (N, K): element (n, k) at n*sBn + k
      for (int e = tid; e < 64 * 16; e += 256) {
        const int kk = e & 15, nn = e >> 4;
        const int64_t n = n0 + nn, k = k0 + kk;
        Bs[kk][nn] = (n < g.N && k < g.K) ? to_f32(Bm[n * g.sBn + k]) : 0.0f;
      }
    } else {          // B is (K, N): element (k, n) at k*sBn + n
      for (int e = tid; e < 64 * 16; e += 256) {
        const int nn = e & 63, kk = e >> 6;
        const int64_t n = n0 + nn, k = k0 + kk;
        Bs[kk][nn] = (n < g.N && k < g.K) ? to_f32(Bm[k * g.sBn + n]) : 0.0f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

  TO *C = (TO *)g.C + b0 * g.sCb0 + b1 * g.sCb1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      float v = acc[i][j] * g.alpha;
      if (g.bias) v += (g.in_dtype == GNLP_BF16) ? ld<bf16>(g.bias, n) : ld<float>(g.bias, n);
      v = apply_act<true>(v, g.act);
      if (g.other) {
        const int64_t oi = b0 * g.sOb0 + b1 * g.sOb1 + m * g.sOm + n * g.sOn;
        const float o = (g.other_dtype == GNLP_BF16) ? ld<bf16>(g.other, oi) : ld<float>(g.other, oi);
        v = (g.combine == GNLP_COMBINE_MUL) ? v * o : v + o;
      }
      C[m * g.sCm + n] = from_f32<TO>(v);
    }
  }
  }
}

void launch_generic_gemm(const GenericGemm &g, hipStream_t s) {
  if (g.M == 0 || g.N == 0 || g.nb0 * g.nb1 == 0) return;
  const int64_t mt = (g.M + 63) / 64, nb = g.nb0 * g.nb1;
  dim3 grid((unsigned)((g.N + 63) / 64), (unsigned)(mt < 65535 ? mt : 65535),
            (unsigned)(nb < 65535 ? nb : 65535));
  if (g.in_dtype == GNLP_BF16 && g.out_dtype == GNLP_BF16)
    generic_gemm_kernel<bf16, bf16><<<grid, 256, 0, s>>>(g);
  else if (g.in_dtype == GNLP_BF16)
    generic_gemm_kernel<bf16, float><<<grid, 256, 0, s>>>(g);
  else if (g.out_dtype == GNLP_BF16)
    generic_gemm_kernel<float, bf16><<<grid, 256, 0, s>>>(g);
  else
    generic_gemm_kernel<float, float><<<grid, 256, 0, s>>>(g);
}

}  // namespace gnlp
