// Do MFMA and vector-ALU instructions of two waves that share a SIMD overlap on gfx950?
// 512-thread workgroup: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (their SIMD partners)
// a softmax-like VALU loop (v_pk_fma_f32 + v_exp_f32 + v_cvt_pk_bf16_f32).  Each mode is timed with
// the shader clock per wave: MFMA alone, VALU alone, both together.
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/coissue_probe tools/archive/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ void __launch_bounds__(512) probe(float *out, long long *cyc, int mode, int iters) {
  const int wave = threadIdx.x >> 6, grp = wave >> 2;
  const bool do_mfma = grp == 0 ? (mode & 1) : (mode & 4);
  const bool do_valu = grp == 1 ? (mode & 2) : (mode & 8);
  float sink = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (do_mfma) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    if (SHAPE == 0) {
      f32x16 c0 = {0}, c1 = {0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
      }
      sink += c0[0] + c1[5];
    } else {
      f32x4 c[4] = {{0}, {0}, {0}, {0}};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {     // same MACs per iteration: 32 x 16x16x32 = 16 x 32x32x16
#pragma unroll
          for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
        }
      }
      sink += c[0][0] + c[1][1] + c[2][2] + c[3][3];
    }
  }
  if (do_valu) {
    f32x2 x[16];
    for (int i = 0; i < 16; ++i) x[i] = f32x2{threadIdx.x * 1e-3f + i, i * 0.25f};
    const f32x2 c2 = {0.999f, 0.999f}, m2 = {-0.001f, -0.001f};
    f32x2 acc = {0.f, 0.f};
    unsigned pk = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {        // one "tile" of the softmax: 16 pk_fma, 32 exp, 16 pk_add, 16 cvt
        const f32x2 t = x[i] * c2 + m2;
        f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        acc += p;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 q = {(__bf16)p[0], (__bf16)p[1]};
        pk ^= *reinterpret_cast<unsigned *>(&q);
        x[i] = p;
      }
    }
    sink += acc[0] + acc[1] + (float)pk;
  }
  if (grp == 0 && (mode & 32)) {   // one wave, both kinds, interleaved 1 MFMA : 5 VALU by the scheduler
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 c0 = {0}, c1 = {0};
    f32x2 x[16];
    for (int i = 0; i < 16; ++i) x[i] = f32x2{threadIdx.x * 1e-3f + i, i * 0.25f};
    const f32x2 c2 = {0.999f, 0.999f}, m2 = {-0.001f, -0.001f};
    f32x2 acc = {0.f, 0.f};
    unsigned pk = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        const f32x2 t = x[i] * c2 + m2;
        f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        acc += p;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 q = {(__bf16)p[0], (__bf16)p[1]};
        pk ^= *reinterpret_cast<unsigned *>(&q);
        x[i] = p;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
      }
    }
    sink += c0[0] + c1[5] + acc[0] + acc[1] + (float)pk;
  }
  const long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = sink;
}

int main() {
  const int blocks = 256, iters = 2000;
  float *out; long long *cyc;
  hipMalloc(&out, blocks * 512 * sizeof(float));
  hipMalloc(&cyc, blocks * 8 * sizeof(long long));
  long long h[8];
  const char *names[] = {"", "mfma alone (waves 0-3)", "valu alone (waves 4-7)", "mfma + valu partners",
                         "", "mfma on both partners", "", "", "", "", "valu on both partners", "", "", "", "", "",
                         "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "",
                         "one wave, interleaved"};
  for (int shape = 0; shape < 2; ++shape) {
    printf("== %s ==\n", shape == 0 ? "v_mfma_f32_32x32x16_bf16 (16 per iteration)" : "v_mfma_f32_16x16x32_bf16 (32 per iteration)");
    for (int mode : {1, 2, 3, 5, 10, 32}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (shape == 0) probe<0><<<blocks, 512>>>(out, cyc, mode, iters);
        else probe<1><<<blocks, 512>>>(out, cyc, mode, iters);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      printf("%-26s cycles/iteration: mfma-wave %8.1f  valu-wave %8.1f\n", names[mode],
             (double)h[0] / iters, (double)h[4] / iters);
    }
  }
  return 0;
}
