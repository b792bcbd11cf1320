// Per-CU cost of pulling a [16 rows x 512 floats] operand block (32 KB) per iteration through the vector memory path,
// as the persistent GRU sweeps do, for different lane->address mappings:
//   mode 0: MFMA-fragment pattern: one 16-byte load per lane, 4 lanes cover 64 B of a row, 16 rows per instruction
//   mode 1: row-contiguous pattern: one instruction = 1 KB contiguous (half a row), 8 full 128-byte lines
//   mode 2: like 0 but dword loads (4 B per lane: 16 lanes cover 64 B of a row, 4 rows per instruction)
// 256 threads per workgroup, one workgroup per CU, sc1 or plain loads, data resident in L2/MALL (64 blocks cycle).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
template <int AUX>
__device__ __forceinline__ u32x4 ld16(const float* base, unsigned off) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ unsigned ld4(const float* base, unsigned off) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX);
}
template <int MODE, int AUX>
__global__ __launch_bounds__(256) void k(const float* data, int iters, int nslab, unsigned long long* cyc, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  unsigned acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const float* slab = data + (size_t)((it + blockIdx.x) % nslab) * 16 * 512;   // [16][512] floats
    if (MODE == 0) {
      u32x4 v[8];
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) v[ci] = ld16<AUX>(slab, (unsigned)((j * 512 + (wave * 8 + ci) * 16 + 4 * q) * 4));
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) acc += v[ci].x ^ v[ci].w;
    } else if (MODE == 1) {
      u32x4 v[8];
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) v[ci] = ld16<AUX>(slab, (unsigned)(((wave * 8 + ci) * 256 + lane * 4) * 4));
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) acc += v[ci].x ^ v[ci].w;
    } else if (MODE == 3) {   // same bytes per instruction as mode 0 (16 rows x 64 B), lanes permuted: a quad = 64 contiguous bytes
      u32x4 v[8];
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) v[ci] = ld16<AUX>(slab, (unsigned)(((lane >> 2) * 512 + (wave * 8 + ci) * 16 + 4 * (lane & 3)) * 4));
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) acc += v[ci].x ^ v[ci].w;
    } else {
      unsigned v[32];
#pragma unroll
      for (int ci = 0; ci < 32; ++ci) v[ci] = ld4<AUX>(slab, (unsigned)((((ci & 3) * 4 + q) * 512 + (wave * 8 + (ci >> 2)) * 16 + j) * 4));
#pragma unroll
      for (int ci = 0; ci < 32; ++ci) acc += v[ci];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x12345) sink[0] = acc;
}
template <int MODE, int AUX> void run(const char* name, const float* d, int nslab, int wgs) {
  unsigned long long* c; unsigned* s; hipMalloc(&c, wgs * 8); hipMalloc(&s, 64);
  const int iters = 400;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE, AUX>), dim3(wgs), dim3(256), 0, 0, d, iters, nslab, c, s); hipDeviceSynchronize(); }
  unsigned long long* h = (unsigned long long*)malloc(wgs * 8); hipMemcpy(h, c, wgs * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < wgs; ++i) sum += (double)h[i];
  printf("%-44s %4d WGs: %7.0f cycles per 32 KB block  (%.1f B/clk/CU)\n", name, wgs, sum / wgs / iters, 32768.0 / (sum / wgs / iters));
  hipFree(c); hipFree(s); free(h);
}
int main() {
  const int nslab = 64; float* d; hipMalloc(&d, (size_t)nslab * 16 * 512 * 4); hipMemset(d, 0, (size_t)nslab * 16 * 512 * 4);
  for (int wgs : {1, 256}) {
    run<0, 16>("fragment pattern, 16 B/lane, sc1", d, nslab, wgs);
    run<0, 0>("fragment pattern, 16 B/lane, plain", d, nslab, wgs);
    run<1, 16>("row-contiguous, 16 B/lane, sc1", d, nslab, wgs);
    run<1, 0>("row-contiguous, 16 B/lane, plain", d, nslab, wgs);
    run<2, 16>("fragment pattern, 4 B/lane, sc1", d, nslab, wgs);
    run<3, 16>("fragment bytes, quad-contiguous lanes, sc1", d, nslab, wgs);
  }
  return 0;
}
