// Ping-pong hand-off latency between two workgroups: same XCD vs cross XCD, plain vs sc1 granule stores.
// Granule = 8 bytes {value, tag}; consumer spins with sc1 loads (L1 bypass) until the tag matches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned long long u64;
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }

__global__ void map_kernel(unsigned* xcc) { if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id(); }

// blocks A and B ping-pong ITER times; other blocks exit. mode 0: sc1 stores, 1: plain stores
__global__ void pingpong(u64* ga, u64* gb, int blkA, int blkB, int iters, int mode, u64* cycles) {
  if (threadIdx.x != 0) return;
  const int me = blockIdx.x == blkA ? 0 : (blockIdx.x == blkB ? 1 : -1);
  if (me < 0) return;
  u64* mine = me == 0 ? ga : gb;     // I write here
  u64* theirs = me == 0 ? gb : ga;   // I poll here
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      u64 g = ((u64)i << 32) | (unsigned)i;
      if (mode == 0) __hip_atomic_store(mine, g, RLX_AGENT);
      else if (mode == 1) *(volatile u64*)mine = g;
      else asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(mine), "v"(g) : "memory");   // truly plain store
      unsigned spins = 0;
      while ((__hip_atomic_load(theirs, RLX_AGENT) >> 32) != (u64)i) { if (++spins > (1u << 18)) { cycles[2] = 1; return; } }
    } else {
      unsigned spins = 0;
      while ((__hip_atomic_load(theirs, RLX_AGENT) >> 32) != (u64)i) { if (++spins > (1u << 18)) { cycles[2] = 1; return; } }
      u64 g = ((u64)i << 32) | (unsigned)i;
      if (mode == 0) __hip_atomic_store(mine, g, RLX_AGENT);
      else if (mode == 1) *(volatile u64*)mine = g;
      else asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(mine), "v"(g) : "memory");
    }
  }
  if (me == 0) cycles[0] = __builtin_amdgcn_s_memtime() - t0;
}

int main() {
  unsigned* dx; hipMalloc(&dx, 4096); unsigned hx[256];
  hipLaunchKernelGGL(map_kernel, dim3(256), dim3(64), 0, 0, dx); hipMemcpy(hx, dx, 1024, hipMemcpyDeviceToHost);
  int ok = 0; for (int b = 0; b < 256; ++b) ok += (hx[b] == (unsigned)(b % 8));
  printf("blocks with xcc == b%%8: %d / 256   first 16:", ok); for (int b = 0; b < 16; ++b) printf(" %u", hx[b]); printf("\n");
  u64 *g, *cyc; hipMalloc(&g, 1 << 20); hipMalloc(&cyc, 64);
  const int iters = 2000;
  int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
  const char* names[3] = {"same XCD (b0,b8)", "cross XCD (b0,b1)", "cross XCD (b0,b4)"};
  for (int p = 0; p < 3; ++p) for (int mode = 0; mode < 3; ++mode) {
    hipMemset(g, 0, 1 << 20); hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, g, g + 4096, pairs[p][0], pairs[p][1], iters, mode, cyc);
    hipDeviceSynchronize();
    u64 h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    printf("%-20s %-6s stores: %s  round trip %.0f cycles (one-way hand-off ~%.0f)\n", names[p], mode == 0 ? "sc1" : (mode == 1 ? "volat." : "plain"),
           h[2] ? "TIMEOUT" : "ok", (double)h[0] / iters, (double)h[0] / iters / 2);
  }
  return 0;
}
