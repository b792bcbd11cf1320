// fp32 MFMA peak microbenchmark: 4 independent accumulators per wave, no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = float __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 c0{}, c1{}, c2{}, c3{};
  if (b < 0.6f) {   // randomised operands: per-lane, changing every iteration (realistic toggling)
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < iters; ++i) {
      s = s * 1664525u + 1013904223u;
      float ra = __uint_as_float((s & 0x007fffffu) | 0x3f800000u) - 1.5f;
      float rb = __uint_as_float(((s >> 9) & 0x007fffffu) | 0x3f800000u) - 1.5f;
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra, rb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb, ra, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra, ra, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb, rb, c3, 0, 0, 0);
    }
  } else
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  for (int blocks : {256, 512, 1024, 2048}) {
    int iters = 20000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * 2 * 32 * 32 * 2;
    printf("random operands blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("const  operands blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
  }
  return 0;
}
