// Developer probe: issue cost (cycles per wave64 instruction, one wave per SIMD) of the vector-ALU
// instructions the epilogues and the softmax are made of.  s_memtime around 64 x 32 independent ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP32(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x
template <int OP>
__global__ void probe(unsigned long long *out, float *sink, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
        a6 = a0 + 6, a7 = a0 + 7;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; ++it) {
    if (OP == 0) { REP32(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 1) { REP32(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 2) { REP32(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
    if (OP == 3) { REP32(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 4) { REP32(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 5) { REP32(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 6) { REP32(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
    if (OP == 7) { REP32(asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %2, %2, %2, %2\n v_exp_f32 %1, %1\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(p2), "+v"(p3));) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1];
}
int main() {
  unsigned long long *d, h;
  float *sink;
  hipMalloc(&d, 8); hipMalloc(&sink, 4096);
  const char *names[] = {"v_exp_f32", "v_rcp_f32", "v_pk_fma_f32", "v_fma_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_pk_mul/add_f32", "exp + pk_fma interleaved"};
#define RUN(OP) probe<OP><<<1, 64>>>(d, sink, 0.5f); hipDeviceSynchronize(); probe<OP><<<1, 64>>>(d, sink, 0.5f); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf("%-28s %.2f memtime ticks / instruction\n", names[OP], (double)h / (64.0 * 32 * 4));
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  // s_memtime ticks at a fixed 100 MHz; convert with the shader clock: run a known-cost loop
  return 0;
}
