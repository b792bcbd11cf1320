// mfma_valu.hip — how much VALU / LDS work hides under fp32 MFMAs on one SIMD (gfx950)?
//   test 0: one wave per SIMD, NM MFMAs (3 independent accumulators) per iteration, nothing else
//   test 1: same wave also runs a DEPENDENT chain of KV v_fma between the MFMAs (one every MFMA)
//   test 2: two waves per SIMD: wave A = MFMA stream, wave B = dependent VALU chain only; report both times
//   test 3: two waves per SIMD, both MFMA streams (pipe sharing)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = float __attribute__((ext_vector_type(4)));
constexpr int NM = 96, IT = 200;

template <int KV>   // VALU ops interleaved per MFMA in the same wave
__global__ __launch_bounds__(512) void k_same(float* out, unsigned long long* cyc, int role_mask) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  float a = threadIdx.x * 1e-3f, b = 1.0001f, v = 0.5f;
  const bool do_mfma = (role_mask >> (wave >> 2)) & 1;        // bit 0: waves 0-3, bit 1: waves 4-7
  const bool do_valu = (role_mask >> (2 + (wave >> 2))) & 1;  // bit 2: waves 0-3, bit 3: waves 4-7
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < IT; ++it) {
    if (do_mfma && do_valu) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < KV; ++k) v = __builtin_fmaf(v, 1.0001f, 0.001f);
      }
    } else if (do_mfma) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 3], 0, 0, 0);
    } else if (do_valu) {
#pragma unroll
      for (int i = 0; i < NM; ++i)
#pragma unroll
        for (int k = 0; k < (KV > 0 ? KV : 1); ++k) v = __builtin_fmaf(v, 1.0001f, 0.001f);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + v;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = (t1 - t0) / IT;
}

template <int KV> void run(const char* name, int threads, int mask, float* out, unsigned long long* cyc) {
  hipMemset(cyc, 0, 64);
  hipLaunchKernelGGL(k_same<KV>, dim3(256), dim3(threads), 0, 0, out, cyc, mask);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-58s wave0 %6llu  wave4 %6llu  ticks/iter (96 MFMA = 3072 cycles of pipe)\n", name, h[0], h[4]);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  run<0>("1 wave/SIMD: MFMA only", 256, 0x1, out, cyc);
  run<1>("1 wave/SIMD: MFMA + 1 dependent fma per MFMA", 256, 0x5, out, cyc);
  run<3>("1 wave/SIMD: MFMA + 3 dependent fma per MFMA", 256, 0x5, out, cyc);
  run<6>("1 wave/SIMD: MFMA + 6 dependent fma per MFMA", 256, 0x5, out, cyc);
  run<3>("1 wave/SIMD: VALU chain only (288 dependent fma)", 256, 0x4, out, cyc);
  run<3>("2 waves/SIMD: A = MFMA, B = VALU chain (288 fma)", 512, 0x1 | 0x8, out, cyc);
  run<0>("2 waves/SIMD: both MFMA", 512, 0x3, out, cyc);
  run<3>("2 waves/SIMD: both MFMA + 3 fma per MFMA", 512, 0xf, out, cyc);
  return 0;
}
