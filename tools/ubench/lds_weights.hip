// lds_weights.hip — can a GRU workgroup hold a SECOND weight slice (W_ih of its 16 units: 48 x 512 fp32 = 96 KB) in LDS and
// feed v_mfma_f32_16x16x4_f32 from there at the rate it feeds the first one (W_hh) from registers?
// (Design question for a fused multi-layer sweep: DESIGN.md, next-round candidate 1.)
// One workgroup = 4 waves, 16 rows x 48 gate columns, K = 512 split over the waves (as gru_persistent.hip does).
//   mode 0: B operand from registers (96 floats per lane, loaded once)          -- today's recurrent product
//   mode 1: B operand from LDS (ds_read_b128 per 16-k chunk and gate)           -- the candidate input product
//   mode 2: both per step (registers for one product, LDS for the other)        -- the fused step's MFMA work
// The A operand is regenerated in registers (no global traffic): this isolates the operand feed.  Prints cycles per step.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_weights lds_weights.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using f32x4 = float __attribute__((ext_vector_type(4)));
constexpr int H = 512, NCH = H / 16 / 4;   // 16-wide k chunks per wave (K split over 4 waves): 8

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ w, float* __restrict__ out, long long* cyc, int steps) {
  extern __shared__ __attribute__((aligned(16))) float lw[];      // [3][16 units][H + 4]: pitch 516 floats (conflict-free b128 rows)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  constexpr int P = H + 4;
  if (MODE >= 1) for (int i = tid; i < 3 * 16 * H; i += 256) lw[(i / H) * P + i % H] = w[i];
  f32x4 wr[3][NCH];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int c = 0; c < NCH; ++c) wr[g][c] = *reinterpret_cast<const f32x4*>(w + (g * 16 + j) * H + (wave * NCH + c) * 16 + 4 * q);
  __syncthreads();
  f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  float a0 = (float)lane * 1e-3f;
  const long long t0 = clock64();
  for (int s = 0; s < steps; ++s) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const f32x4 a = f32x4{a0, a0 + 1.f, a0 + 2.f, a0 + 3.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], wr[g][c][0], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], wr[g][c][1], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], wr[g][c][2], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], wr[g][c][3], acc[g], 0, 0, 0);
        }
      }
    }
    if (MODE >= 1) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const f32x4 a = f32x4{a0 + 4.f, a0 + 5.f, a0 + 6.f, a0 + 7.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(&lw[(g * 16 + j) * P + (wave * NCH + c) * 16 + 4 * q]);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[g], 0, 0, 0);
        }
      }
    }
    a0 = acc[0][0] * 1e-30f + (float)lane * 1e-3f;      // the next step depends on this one (as a recurrence does)
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + tid] = acc[0][0] + acc[1][1] + acc[2][2];
}

int main() {
  float* w; float* out; long long* cyc;
  CK(hipMalloc(&w, 3 * 16 * H * 4)); CK(hipMemset(w, 0, 3 * 16 * H * 4));
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  const int steps = 2000;
  const size_t lds = 3 * 16 * (H + 4) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int grid : {1, 256}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, w, out, cyc, steps);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), lds, 0, w, out, cyc, steps);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), lds, 0, w, out, cyc, steps);
        CK(hipDeviceSynchronize());
      }
      long long h[256]; CK(hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost));
      long long mx = 0; for (int i = 0; i < grid; ++i) mx = h[i] > mx ? h[i] : mx;
      printf("grid %3d, %s: %.0f clock64 ticks per step (MFMA issue floor: %d x 32 passes-cycles = %d cycles per product)\n", grid,
             mode == 0 ? "W from registers      " : mode == 1 ? "W from LDS            " : "both (fused step)     ", (double)mx / steps, NCH * 12, NCH * 12 * 32);
    }
  }
  return 0;
}
