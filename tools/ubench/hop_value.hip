// hop_value.hip — cost of one cross-stream dependency: HIP events vs stream memory operations
// (hipStreamWriteValue32 after the producer, hipStreamWaitValue32 >= in front of the consumer) on signal memory.
// Chain of N tiny kernels alternating between two streams; per-hop time from HIP events around the chain.
// build: hipcc --offload-arch=gfx950 -O3 -o hop_value hop_value.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(unsigned* p) { if (threadIdx.x == 0) p[0] += 1; }
int main() {
  unsigned* scratch; CK(hipMalloc(&scratch, 256)); CK(hipMemset(scratch, 0, 256));
  unsigned long long* sig = nullptr;
  if (hipExtMallocWithFlags((void**)&sig, 8 * 1024, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc((void**)&sig, 8 * 1024)); printf("plain memory for flags\n"); }
  CK(hipMemset(sig, 0, 8 * 1024));
  const int NS = 6;
  hipStream_t st[NS];
  for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  const int N = 200;
  hipEvent_t e0, e1, ev[N];
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  for (int pair = 0; pair < 2; ++pair) {
    hipStream_t a = st[0], b = st[pair == 0 ? 1 : 4];   // different pipes / same pipe (stream index mod 4)
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(sig, 0, 8 * 1024)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, a));
        hipStream_t cur = a, other = b;
        for (int i = 0; i < N; ++i) {
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, cur, scratch);
          if (mode == 1) { CK(hipEventRecord(ev[i], cur)); CK(hipStreamWaitEvent(other, ev[i], 0)); }
          if (mode == 2) { CK(hipStreamWriteValue32(cur, sig + i, 1u, 0)); CK(hipStreamWaitValue32(other, sig + i, 1u, hipStreamWaitValueGte, 0xffffffffu)); }
          if (mode != 0) { hipStream_t t = cur; cur = other; other = t; }
        }
        CK(hipEventRecord(e1, cur)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%s, %s: %.2f us per kernel\n", pair == 0 ? "streams 0,1" : "streams 0,4", mode == 0 ? "one stream" : mode == 1 ? "events" : "write/wait value", best * 1e3 / N);
    }
  }
  return 0;
}
