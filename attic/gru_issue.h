// gru_issue.h — assembly-issued memory operations and hand-counted waits for the software-pipelined GRU sweeps
// (gru_pipeline.hip, gru_stack.hip): loads carried around a loop must be invisible to the compiler's wait insertion.
#pragma once
#include "gru_cell.h"
#include "gru_sync.h"

namespace b2t {

constexpr int PAUX = 16;   // sc1 payload accesses, as in mode 1

// The prefetch of the NEXT item must stay in flight while the current item's MFMAs run.  The compiler's own wait
// insertion drains everything (s_waitcnt vmcnt(0)) in front of the MFMAs once loads are carried around the loop, so
// the loads of the pipeline are issued from inline assembly (invisible to that pass) and waited for by hand:
// vector-memory operations of a wave retire in order, so "at most N outstanding" (vmcnt(N)) with N = the number of
// operations issued after the one needed is exact.  Every value produced by such a load is passed through the
// matching wait_vm<>() so that no use can be scheduled above the wait.
using u32x4s = unsigned int __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4s make_rsrc(const float* base_uniform) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base_uniform);
  u32x4s r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r.z = 0x7fffffffu;
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void issue_load_sc1_x4(f32x4& dst, u32x4s rsrc, unsigned byte_off) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "=v"(dst) : "v"(byte_off), "s"(rsrc) : "memory");
}
template <int OFF>   // the same with an immediate byte offset (0..4095)
__device__ __forceinline__ void issue_load_sc1_x4_imm(f32x4& dst, u32x4s rsrc, unsigned byte_off) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3 sc1" : "=v"(dst) : "v"(byte_off), "s"(rsrc), "n"(OFF) : "memory");
}
__device__ __forceinline__ void issue_store_x4(u32x4s rsrc, unsigned byte_off, f32x4 v) {   // ordinary 16-byte store
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(v), "v"(byte_off), "s"(rsrc) : "memory");
}
// 16-byte stores through per-lane pointers (lanes of one instruction may address different buffers)
__device__ __forceinline__ void issue_store_sc1_x4_ptr(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void issue_store_x4_ptr(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void issue_load_x4_ptr(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void issue_load_buf_f32(float& dst, u32x4s rsrc, unsigned byte_off) {
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(byte_off), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void issue_load_f32(float& dst, const float* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void issue_poll(unsigned& dst, const unsigned* p) {
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
// Plain store issued from assembly: the compiler would otherwise protect the data registers of its own outstanding
// stores with s_waitcnt vmcnt(n) computed WITHOUT the invisible loads above, i.e. far too strong.
__device__ __forceinline__ void issue_store_f32(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void issue_store_sc1_x4(u32x4s rsrc, unsigned byte_off, f32x4 v) {
#ifdef B2T_EXPERIMENT_PLAIN_TILE_STORE   // timing experiment only: NOT visible across XCDs
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(v), "v"(byte_off), "s"(rsrc) : "memory");
#elif defined(B2T_EXPERIMENT_NO_TILE_STORE)
  asm volatile("" ::"v"(v), "v"(byte_off), "s"(rsrc) : "memory");
#else
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1" ::"v"(v), "v"(byte_off), "s"(rsrc) : "memory");
#endif
}
// The source registers of an assembly-issued store must stay untouched until the store has completed (this target has
// no interlock: the compiler protects its OWN stores with vmcnt waits, it cannot see ours).  keep_until_here() is a
// fake use: placed behind the drain that covers the store, it keeps the register allocator from reusing them earlier.
__device__ __forceinline__ void keep_until_here(const f32x4& v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void keep_until_here(float v) { asm volatile("" ::"v"(v)); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Full drain that the compiler's wait-insertion pass can also see (a real S_WAITCNT vmcnt(0), expcnt/lgkmcnt
// untouched): after it the pass knows none of ITS loads is pending and adds no waits of its own downstream.
__device__ __forceinline__ void drain_vm() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// ties a value to the preceding wait (no instruction; the value cannot be read before this point)
__device__ __forceinline__ void after_wait(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void after_wait(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void after_wait(unsigned& v) { asm volatile("" : "+v"(v)); }

// Poll value of a counter: every lane loads the same word (one request), made wave-uniform for the branch.
__device__ __forceinline__ unsigned poll_once(const unsigned* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(const_cast<unsigned*>(p), RLX_AGENT));
}

// Block until *p >= target (rare: the prefetched poll normally already saw it).  Every wave polls for itself.
__device__ __forceinline__ void poll_until(const unsigned* p, unsigned target, unsigned seen, unsigned* err) {
  unsigned spins = 0;
  while (seen < target) {
    __builtin_amdgcn_s_sleep(1);
    seen = poll_once(p);
    if ((++spins & 255u) == 0u) {
      if (poll_once(err) != 0u) break;
      if (spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
    }
  }
}

}  // namespace b2t
