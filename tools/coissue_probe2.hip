// Round 3 re-run of the co-issue question with SINGLE-ISSUE fillers, hand-placed (inline asm):
// what does vector-ALU work cost beside v_mfma_f32_32x32x16_bf16 on gfx950
//   (A) in the partner wave of the same SIMD (2 waves per SIMD, one MFMA-only, one VALU-only),
//   (B) in the gaps of one wave's own MFMA stream (1 wave per SIMD, n fillers after each MFMA),
//   (C) in W waves per SIMD that each alternate a 16-MFMA phase and a V-instruction VALU phase
//       without barriers (the shape of the fused attention kernel: 3 workgroups of 4 waves per CU).
// Filler kinds: v_fma_f32, v_exp_f32, v_pk_fma_f32, and the softmax mix (2 fma, 2 exp, 2 add, 1
// cvt_pk per 7).  Times are shader cycles (s_memtime) per MFMA or per phase pair.
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/coissue_probe2 tools/coissue_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p1), "v"(p2))

// one filler of kind K, using register slot i (0..7) so that consecutive fillers are independent
template <int K>
__device__ __forceinline__ void filler(int i, float (&x)[8], f32x2 (&px)[4], unsigned &d, float k1,
                                       float k2, f32x2 p1, f32x2 p2) {
  if (K == 0) FMA(x[i & 7]);
  if (K == 1) EXP(x[i & 7]);
  if (K == 2) PKFMA(px[i & 3]);
  if (K == 3) {   // softmax mix per 7: fma fma exp exp add add cvt -> at most one exp per 3-4 slots
    switch (i % 7) {
      case 0: FMA(x[0]); break;
      case 1: EXP(x[1]); break;
      case 2: FMA(x[2]); break;
      case 3: ADD(x[4], x[5]); break;
      case 4: EXP(x[3]); break;
      case 5: ADD(x[6], x[7]); break;
      default: CVT(d, x[4], x[6]); break;
    }
  }
}

// (A) partners: waves 0-3 MFMA-only, waves 4-7 VALU-only (512 threads: two waves per SIMD)
template <int K>
__global__ void __launch_bounds__(512) probe_a(float *out, long long *cyc, int mode, int iters) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 c0 = {0}, c1 = {0};
  float x[8]; f32x2 px[4]; unsigned d = 0;
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int i = 0; i < 4; ++i) px[i] = f32x2{threadIdx.x * 1e-3f + i, i * 0.25f};
  const float k1 = 0.999f, k2 = -0.001f; const f32x2 p1 = {k1, k1}, p2 = {k2, k2};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4 && (mode & 1)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { MFMA(c0); MFMA(c1); }
    }
  }
  if (wave >= 4 && (mode & 2)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 112; ++u) filler<K>(u, x, px, d, k1, k2, p1, p2);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  float s = c0[0] + c1[3] + (float)d;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += px[i][0] + px[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// (B) one wave per SIMD: N fillers of kind K after every MFMA
template <int K, int N>
__global__ void __launch_bounds__(256) probe_b(float *out, long long *cyc, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 c0 = {0}, c1 = {0};
  float x[8]; f32x2 px[4]; unsigned d = 0;
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int i = 0; i < 4; ++i) px[i] = f32x2{threadIdx.x * 1e-3f + i, i * 0.25f};
  const float k1 = 0.999f, k2 = -0.001f; const f32x2 p1 = {k1, k1}, p2 = {k2, k2};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 14; ++u) {
      if (u & 1) MFMA(c1); else MFMA(c0);
#pragma unroll
      for (int f = 0; f < N; ++f) filler<K>(u * N + f, x, px, d, k1, k2, p1, p2);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  float s = c0[0] + c1[3] + (float)d;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += px[i][0] + px[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (C) W waves per SIMD, each: 16 MFMAs back to back, then V fillers of kind K, no barrier;
// wave w starts in the VALU phase when (w / 4) is odd, so that the phases of a SIMD's waves differ
template <int K, int V>
__global__ void __launch_bounds__(256) probe_c(float *out, long long *cyc, int iters, int stagger) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 c0 = {0}, c1 = {0};
  float x[8]; f32x2 px[4]; unsigned d = 0;
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int i = 0; i < 4; ++i) px[i] = f32x2{threadIdx.x * 1e-3f + i, i * 0.25f};
  const float k1 = 0.999f, k2 = -0.001f; const f32x2 p1 = {k1, k1}, p2 = {k2, k2};
  if (stagger && (blockIdx.x & 1)) {   // half of the workgroups start with the other phase
#pragma unroll
    for (int u = 0; u < V; ++u) filler<K>(u, x, px, d, k1, k2, p1, p2);
  }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { MFMA(c0); MFMA(c1); }
#pragma unroll
    for (int u = 0; u < V; ++u) filler<K>(u, x, px, d, k1, k2, p1, p2);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float s = c0[0] + c1[3] + (float)d;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += px[i][0] + px[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float *out; static long long *cyc; static long long h[4096 * 8];
static const int iters = 2000;
static const char *kinds[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "softmax mix"};

template <int K> static void run_a() {
  double r[4][2];
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) { probe_a<K><<<256, 512>>>(out, cyc, mode, iters); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, 8 * sizeof(long long), hipMemcpyDeviceToHost);
    r[mode][0] = (double)h[0] / iters; r[mode][1] = (double)h[4] / iters;
  }
  printf("A %-13s 16 MFMA alone %7.1f | 112 fillers alone %7.1f (%.2f each) | together: mfma wave %7.1f, valu wave %7.1f\n",
         kinds[K], r[1][0], r[2][1], r[2][1] / 112, r[3][0], r[3][1]);
}
template <int K, int N> static void run_b() {
  for (int rep = 0; rep < 2; ++rep) { probe_b<K, N><<<256, 256>>>(out, cyc, iters); hipDeviceSynchronize(); }
  hipMemcpy(h, cyc, 8 * sizeof(long long), hipMemcpyDeviceToHost);
  printf("B %-13s %d fillers per gap: %6.1f cycles per MFMA\n", kinds[K], N, (double)h[0] / iters / 14);
}
template <int K, int V> static void run_c(int wgs_per_cu, int stagger) {
  const int blocks = 256 * wgs_per_cu;
  for (int rep = 0; rep < 2; ++rep) { probe_c<K, V><<<blocks, 256>>>(out, cyc, iters, stagger); hipDeviceSynchronize(); }
  hipMemcpy(h, cyc, blocks * 4 * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
  printf("C %-13s %d waves per SIMD%s, 16 MFMA + %3d fillers per pass: %7.1f cycles per pass per wave -> %7.1f SIMD cycles per pass\n",
         kinds[K], wgs_per_cu, stagger ? " staggered" : "", V, s / (blocks * 4) / iters,
         s / (blocks * 4) / iters / wgs_per_cu);
}

int main() {
  hipMalloc(&out, 4096 * 512 * sizeof(float));
  hipMalloc(&cyc, 4096 * 8 * sizeof(long long));
  run_a<0>(); run_a<1>(); run_a<2>(); run_a<3>();
  run_b<0, 0>(); run_b<0, 2>(); run_b<0, 4>(); run_b<0, 5>(); run_b<0, 6>(); run_b<0, 8>();
  run_b<1, 1>(); run_b<1, 2>(); run_b<1, 3>();
  run_b<2, 1>(); run_b<2, 2>(); run_b<2, 4>();
  run_b<3, 3>(); run_b<3, 5>(); run_b<3, 7>(); run_b<3, 10>();
  for (int w = 1; w <= 3; ++w) {
    run_c<3, 112>(w, 0); run_c<3, 112>(w, 1);
    run_c<3, 160>(w, 1);
    run_c<2, 80>(w, 1);
    run_c<0, 160>(w, 1);
  }
  return 0;
}
