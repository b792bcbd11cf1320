// Does a CU-masked stream confine a kernel's workgroups to some XCDs, and does such a kernel finish while another kernel fills the
// other XCDs?  (Static round-robin dispatch: workgroup i of a launch goes to XCD i % 8 and WAITS there if that XCD is full -- so a GEMM
// launched beside the layer wavefront, which fills XCDs 0-4, cannot finish before the sweep does.)
// hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__device__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
__global__ void where_kernel(unsigned* hist) { if (threadIdx.x == 0) atomicAdd(hist + xcc_id(), 1u); }
// fills the CUs of XCDs < nx for `ticks` of the 100 MHz clock (160 KB of LDS: one workgroup per CU); others leave
__global__ __launch_bounds__(256, 1) void blocker_kernel(int nx, long long ticks, unsigned* started) {
  extern __shared__ char lds[];
  // (all 512 registers of a lane: nothing else fits the CU's SIMDs while this workgroup is resident -- as under the layer wavefront)
  asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
  if ((int)xcc_id() >= nx) return;
  if (threadIdx.x == 0) { atomicAdd(started, 1u); lds[0] = 1; }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void work_kernel(float* x, int iters) {   // some work per workgroup (~20 us)
  float v = x[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.000001f + 0.5f;
  x[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("CUs %d\n", p.multiProcessorCount);
  unsigned* hist; CK(hipMalloc(&hist, 64 * 4));
  float* x; CK(hipMalloc(&x, 4096 * 256 * 4)); CK(hipMemset(x, 0, 4096 * 256 * 4));
  unsigned* started; CK(hipMalloc(&started, 4));
  CK(hipFuncSetAttribute((const void*)blocker_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int words = 16;
  struct M { const char* name; std::vector<uint32_t> m; };
  std::vector<M> masks;
  { M a{"all", std::vector<uint32_t>(words, 0xffffffffu)}; masks.push_back(a); }
  { M a{"bits k %8 >= 5", std::vector<uint32_t>(words, 0)}; for (int k = 0; k < words * 32; ++k) if (k % 8 >= 5) a.m[k / 32] |= 1u << (k % 32); masks.push_back(a); }
  { M a{"bits [160,256)", std::vector<uint32_t>(words, 0)}; for (int k = 160; k < 256; ++k) a.m[k / 32] |= 1u << (k % 32); masks.push_back(a); }
  { M a{"bits [0,96)", std::vector<uint32_t>(words, 0)}; for (int k = 0; k < 96; ++k) a.m[k / 32] |= 1u << (k % 32); masks.push_back(a); }
  { M a{"bits k %8 < 3", std::vector<uint32_t>(words, 0)}; for (int k = 0; k < words * 32; ++k) if (k % 8 < 3) a.m[k / 32] |= 1u << (k % 32); masks.push_back(a); }
  hipStream_t blk; CK(hipStreamCreateWithFlags(&blk, hipStreamNonBlocking));
  for (auto& mk : masks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mk.m.size(), mk.m.data());
    if (e != hipSuccess) { printf("mask '%s': create failed: %s\n", mk.name, hipGetErrorString(e)); continue; }
    CK(hipMemsetAsync(hist, 0, 64 * 4, s));
    hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 0, s, hist);
    CK(hipStreamSynchronize(s));
    unsigned h[16]; CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost));
    printf("mask '%-16s' workgroups by XCD:", mk.name); for (int i = 0; i < 8; ++i) printf(" %u", h[i]); printf("\n");
    // beside a blocker that fills XCDs 0-4 for 3 ms: when does a 1024-workgroup kernel on this stream finish?
    for (int nx : {0, 5}) {
      CK(hipMemset(started, 0, 4));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      if (nx) hipLaunchKernelGGL(blocker_kernel, dim3(256), dim3(256), 160 * 1024 - 64, blk, nx, 300000LL, started);
      if (nx) { unsigned st = 0; while (st < (unsigned)(nx * 32)) { CK(hipMemcpy(&st, started, 4, hipMemcpyDeviceToHost)); } }
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(work_kernel, dim3(1024), dim3(256), 0, s, x, 20000);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipDeviceSynchronize());
      printf("    work kernel (1024 workgroups) %s: %.3f ms\n", nx ? "beside a 3 ms blocker on XCDs 0-4" : "alone", ms);
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
