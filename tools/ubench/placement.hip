// Where does the workgroup dispatcher put the workgroups of concurrent 128-workgroup kernels?
// Each workgroup records (XCC, SE, CU) and then spins ~200 us so kernels on different streams are co-resident.
// Dynamic LDS sets the per-CU slot count like the sweep kernels' register footprint does (48 KB -> 3 per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
__device__ __forceinline__ unsigned hw_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x)); return x; }
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long ticks) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    lds[0] = 1.f;
    out[blockIdx.x] = (xcc_id() << 16) | (hw_id() & 0xffff);
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
}
int main(int argc, char** argv) {
  const int nk = argc > 1 ? atoi(argv[1]) : 2, wgs = argc > 2 ? atoi(argv[2]) : 128, lds = argc > 3 ? atoi(argv[3]) : 48 * 1024;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  unsigned* d; hipMalloc(&d, nk * wgs * 4);
  hipStream_t s[16];
  int lo = -1, hi = -1, stride = 1;   // optional CU mask: bits lo, lo+stride, ... < hi  (argv[4..6])
  if (argc > 5) { lo = atoi(argv[4]); hi = atoi(argv[5]); if (argc > 6) stride = atoi(argv[6]); }
  for (int i = 0; i < nk; ++i) {
    if (lo < 0) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    else {
      uint32_t m[8] = {0}; for (int b = lo; b < hi; b += stride) m[b >> 5] |= 1u << (b & 31);
      hipError_t e = hipExtStreamCreateWithCUMask(&s[i], 8, m);
      if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 1; }
    }
  }
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, s[i], d + i * wgs, 20000ull /* 100 MHz -> 200 us */);
    hipDeviceSynchronize();
  }
  unsigned* h = (unsigned*)malloc(nk * wgs * 4); hipMemcpy(h, d, nk * wgs * 4, hipMemcpyDeviceToHost);
  // key = xcc(4b) se(3b) sh(1b) cu(4b)
  static int cnt[16][16 * 8 * 2 * 16];
  int total[16 * 8 * 2 * 16]; memset(total, 0, sizeof total); memset(cnt, 0, sizeof cnt);
  for (int i = 0; i < nk; ++i) for (int b = 0; b < wgs; ++b) {
    unsigned v = h[i * wgs + b]; unsigned xcc = v >> 16, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
    int key = ((xcc * 8 + se) * 2 + sh) * 16 + cu; cnt[i][key]++; total[key]++;
  }
  int hist[16] = {0}, used = 0; for (int kx = 0; kx < 16 * 8 * 2 * 16; ++kx) if (total[kx]) { used++; hist[total[kx] > 15 ? 15 : total[kx]]++; }
  printf("%d kernels x %d workgroups, %d B LDS: %d distinct CUs used; CUs by resident workgroups:", nk, wgs, lds, used);
  for (int c = 1; c < 16; ++c) if (hist[c]) printf("  %dx:%d", c, hist[c]);
  printf("\n");
  if (lo >= 0) {   // list the CUs used, grouped
    printf(" mask bits [%d,%d) stride %d -> CUs:", lo, hi, stride);
    for (int kx = 0; kx < 16 * 8 * 2 * 16; ++kx) if (total[kx]) printf(" %d.%d.%d(%d)", kx >> 8, (kx >> 5) & 7, kx & 15, total[kx]);
    printf("\n");
  }
  for (int i = 0; i < nk && i < 3; ++i) {
    printf(" kernel %d first 16 WGs (xcc.se.sh.cu):", i);
    for (int b = 0; b < 16; ++b) { unsigned v = h[i * wgs + b]; printf(" %u.%u.%u.%u", v >> 16, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 0xf); }
    printf("\n");
  }
  return 0;
}
