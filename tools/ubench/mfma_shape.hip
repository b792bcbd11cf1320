// mfma_shape.hip — does the MFMA shape of wave A change how fast a neighbour wave B (same SIMD) gets VALU / LDS issue slots?
//   wave A (waves 0-3, one per SIMD): a stream of fp32 MFMAs of one shape, same FLOPs per iteration for every shape
//   wave B (waves 4-7): 288 dependent v_fma (test V) or 96 ds_read_b128 + 96 ds_write_b128 (test L), optionally s_setprio 3
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
constexpr int IT = 200;

// SHAPE 0: 16x16x4 (96 per iter), 1: 32x32x2 (48 per iter: same FLOPs), 2: 4x4x1 (16 blocks; 64 FLOP*... ) skipped
template <int SHAPE, int BK>   // BK: 0 = B idle, 1 = VALU chain, 2 = LDS traffic
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int prio, int a_on) {
  __shared__ float4 lds[512];
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-3f, b = 1.0001f, v = 0.5f;
  f32x4 acc4[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  f32x16 acc16[2] = {};
  float4 x = make_float4(a, b, v, 1.f);
  lds[threadIdx.x] = x;
  const bool is_a = wave < 4;
  if (!is_a && prio) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < IT; ++it) {
    if (is_a) {
      if (a_on) {
        if (SHAPE == 0) {
#pragma unroll
          for (int i = 0; i < 96; ++i) acc4[i % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i % 3], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 48; ++i) acc16[i % 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[i % 2], 0, 0, 0);
        }
      }
    } else {
      if (BK == 1) {
#pragma unroll
        for (int i = 0; i < 288; ++i) v = __builtin_fmaf(v, 1.0001f, 0.001f);
      } else if (BK == 2) {
#pragma unroll
        for (int i = 0; i < 96; ++i) {
          x = lds[(threadIdx.x + i) & 511];
          x.x += 1.f;
          lds[(threadIdx.x + 64 + i) & 511] = x;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc4[0][0] + acc4[1][1] + acc4[2][2] + acc16[0][0] + acc16[1][5] + v + x.x;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = (t1 - t0) / IT;
}

template <int SHAPE, int BK> void run(const char* name, int prio, int a_on, float* out, unsigned long long* cyc) {
  hipMemset(cyc, 0, 64);
  hipLaunchKernelGGL((k<SHAPE, BK>), dim3(256), dim3(512), 0, 0, out, cyc, prio, a_on);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-64s A %6llu  B %6llu\n", name, h[0], h[4]);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  run<0, 1>("A idle, B = 288 dependent fma", 0, 0, out, cyc);
  run<0, 2>("A idle, B = 96 x (ds_read_b128, add, ds_write_b128)", 0, 0, out, cyc);
  run<0, 0>("A = 96 x 16x16x4, B idle", 0, 1, out, cyc);
  run<1, 0>("A = 48 x 32x32x2, B idle", 0, 1, out, cyc);
  run<0, 1>("A = 16x16x4, B = fma chain", 0, 1, out, cyc);
  run<1, 1>("A = 32x32x2, B = fma chain", 0, 1, out, cyc);
  run<0, 1>("A = 16x16x4, B = fma chain, B at s_setprio 3", 1, 1, out, cyc);
  run<1, 1>("A = 32x32x2, B = fma chain, B at s_setprio 3", 1, 1, out, cyc);
  run<0, 2>("A = 16x16x4, B = LDS traffic", 0, 1, out, cyc);
  run<1, 2>("A = 32x32x2, B = LDS traffic", 0, 1, out, cyc);
  run<0, 2>("A = 16x16x4, B = LDS traffic, B at s_setprio 3", 1, 1, out, cyc);
  run<1, 2>("A = 32x32x2, B = LDS traffic, B at s_setprio 3", 1, 1, out, cyc);
  return 0;
}
