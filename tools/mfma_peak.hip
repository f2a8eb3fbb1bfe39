// Register-only MFMA issue-rate probe: what the matrix pipes sustain with no memory traffic at all
// (the practical ceiling the GEMM kernels are measured against).
//   mfma_peak <waves_per_simd 1|2> <data 0=zeros 1=normal-like> <iters> <reps> <0|1: 32x32x16>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// same flops per iteration as probe(): 2 x (2x2 tiles of 32x32, K = 32) = 16 MFMA 32x32x16
__global__ void __launch_bounds__(512) probe32(const bf16x8 *src, float *out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(threadIdx.x * 8 + i) & 4095];
    b[i] = src[(threadIdx.x * 8 + 4 + i) & 4095];
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[j * 2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i * 2 + s], b[j * 2 + s],
                                                                     acc[j * 2 + i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) t += acc[i][e];
  if (t == 12345.678f) out[0] = t;
}

__global__ void __launch_bounds__(512) probe(const bf16x8 *src, float *out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(threadIdx.x * 8 + i) & 4095];
    b[i] = src[(threadIdx.x * 8 + 4 + i) & 4095];
  }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[j * 4 + i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

int main(int argc, char **argv) {
  int wps = argc > 1 ? atoi(argv[1]) : 2, data = argc > 2 ? atoi(argv[2]) : 1;
  int iters = argc > 3 ? atoi(argv[3]) : 2000, reps = argc > 4 ? atoi(argv[4]) : 10;
  int big = argc > 5 ? atoi(argv[5]) : 0;   // 1: 32x32x16 instead of 16x16x32
  unsigned short *h = (unsigned short *)malloc(4096 * 16);
  unsigned x = 12345;
  for (int i = 0; i < 4096 * 8; ++i) {
    x = x * 1664525u + 1013904223u;
    // bf16 with random sign/mantissa, exponent near 1.0 (normal-like toggling) or all zeros
    h[i] = data ? (unsigned short)(((x >> 16) & 0x80ff) | (0x3f00 + ((x >> 8) & 0x80))) : 0;
  }
  void *d; float *o;
  (void)hipMalloc(&d, 4096 * 16); (void)hipMalloc((void **)&o, 4);
  (void)hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int threads = wps * 256;
  probe<<<256, threads>>>((const bf16x8 *)d, o, 10);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(e0);
    if (big) probe32<<<256, threads>>>((const bf16x8 *)d, o, iters);
    else probe<<<256, threads>>>((const bf16x8 *)d, o, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * (threads / 64) * (double)iters * 32 * 16384.0;
    printf("%s wps=%d data=%d iters=%d: %.3f ms  %.0f TFLOP/s\n", big ? "32x32x16" : "16x16x32", wps, data, iters, ms, flops / ms / 1e9);
  }
  return 0;
}
