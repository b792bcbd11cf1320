// Probe: hipStreamWriteValue32 / hipStreamWaitValue32 against a resident kernel.
//   stream 1: kernel P (persistent, 1 WG): spins until *go >= 1, then for i in 1..N: does ~20 us of work, then stores i to *prog
//   stream 2: hipStreamWriteValue32(go, 1)  (host-enqueued, command-processor executed)
//   stream 3: for i in 1..N: hipStreamWaitValue32(prog >= i) ; tiny kernel stamps the time it ran
// Reports the latency from P's store of i (device timestamp) to the start of the dependent kernel on stream 3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void persistent(volatile unsigned* go, unsigned* prog, int n, unsigned long long* tstore) {
  if (threadIdx.x != 0) return;
  while (__hip_atomic_load((unsigned*)go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < 1u) __builtin_amdgcn_s_sleep(8);
  for (int i = 1; i <= n; ++i) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000ull) __builtin_amdgcn_s_sleep(8);   // 20 us at 100 MHz
    tstore[i] = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(prog, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void stamp(unsigned long long* t, int i) { if (threadIdx.x == 0) t[i] = __builtin_amdgcn_s_memrealtime(); }
int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned *go, *prog;
  // can signal memory be an array?  (one allocation, words 8 bytes apart)
  unsigned long long* arr = nullptr;
  hipError_t ea = hipExtMallocWithFlags((void**)&arr, 8 * 64, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(512 B, hipMallocSignalMemory): %s\n", hipGetErrorString(ea));
  if (ea != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc((void**)&arr, 8 * 64)); printf("falling back to plain hipMalloc for the flag array\n"); }
  CK(hipMemset(arr, 0, 8 * 64));
  go = (unsigned*)(arr + 5); prog = (unsigned*)(arr + 9);
  const int n = 50;
  unsigned long long *ts, *tk; CK(hipMalloc(&ts, (n + 1) * 8)); CK(hipMalloc(&tk, (n + 1) * 8));
  hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  hipLaunchKernelGGL(persistent, dim3(1), dim3(64), 0, s1, go, prog, n, ts);
  for (int i = 1; i <= n; ++i) {
    CK(hipStreamWaitValue32(s3, prog, (unsigned)i, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s3, tk, i);
  }
  CK(hipStreamWriteValue32(s2, go, 1u, 0));
  CK(hipDeviceSynchronize());
  unsigned long long hs[64], hk[64]; CK(hipMemcpy(hs, ts, (n + 1) * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hk, tk, (n + 1) * 8, hipMemcpyDeviceToHost));
  double sum = 0, mx = 0; for (int i = 1; i <= n; ++i) { double d = (double)(long long)(hk[i] - hs[i]) / 100.0; sum += d; if (d > mx) mx = d; }
  printf("device store -> dependent kernel start: avg %.1f us, max %.1f us over %d hops\n", sum / n, mx, n);
  return 0;
}
