// gru_sync.h — inter-workgroup hand-off pieces shared by the persistent GRU sweeps (gru_persistent.hip,
// gru_pipeline.hip): agent-scope counters with bounded spins, sc1 payload accesses, the self-cleaning counter sets.
#pragma once
#include "gru_cell.h"

namespace b2t {

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned SPIN_LIMIT = 4u << 20;
constexpr int CSTRIDE = 1;   // words between counters (one line per counter was measured: no effect)
constexpr int SETW = 64 * 1024;  // counters per set; two sets alternate between calls (self-cleaning, no memset)  // ~seconds; a healthy hand-off takes microseconds

// Thread 0 polls until *p >= target (or the error word is set / the spin limit is hit), then barrier.
__device__ __forceinline__ void wait_count(unsigned* p, unsigned target, unsigned* err) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, RLX_AGENT) < target) {
      ++spins;
      if ((spins & 255u) == 0u) {
        if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
        if (spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// All waves have issued their sc1 payload stores: drain, barrier, one lane publishes.
__device__ __forceinline__ void publish_count(unsigned* p) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, 1u, RLX_AGENT);
}

// 16-byte sc1 (L1-bypassing) load through a buffer descriptor based at a wave-uniform pointer.
template <int AUX>   // AUX 16 = sc1 (device scope: never served from a stale per-XCD L2 line), 0 = ordinary cached load
__device__ __forceinline__ float4 load_f4(const float* base_uniform, unsigned byte_off) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, AUX);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}

__device__ __forceinline__ float fast_sigmoid(float x) {
  return __frcp_rn(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2/(1+e^{2x}); saturates correctly: e^{2x} -> inf gives 1, -> 0 gives -1
  return 1.0f - 2.0f * __frcp_rn(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

__device__ __forceinline__ void store_sc1(float* p, float v) { __hip_atomic_store(p, v, RLX_AGENT); }

// 16-byte sc1 (write-through) store through a buffer descriptor based at a wave-uniform pointer.
// Scalar sc1 stores are one fabric write each (~6x the cost per byte of a 16-byte one), so the 16x16
// tile a workgroup produces per step is staged through LDS and written as 64 x 16 B.
template <int AUX>   // AUX 16 = sc1 write-through to memory, 0 = ordinary store (lands in this XCD's L2)
__device__ __forceinline__ void store_f4(float* base_uniform, unsigned byte_off, float4 v) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
  u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, byte_off, 0, AUX);
}

// End of a call: the LAST workgroup to finish flips the counter-set parity (word 1) and re-arms the
// finish counter (word 2).  It must be the last one: row groups are independent recurrences, so any fixed
// workgroup (say block 0) can finish all T steps before a late-dispatched workgroup of another row group has
// read word 1 -- that workgroup would then count in (and clear) the wrong set and the call would never finish.
// (Seen as rare hand-off timeouts when several sweeps and GEMMs shared the chip.)
__device__ __forceinline__ void finish_call(unsigned* sync, unsigned pset) {
  if (threadIdx.x == 0) {
    const unsigned total = gridDim.x * gridDim.y;
    const unsigned done = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == total - 1u) {
      __hip_atomic_store(sync + 2, 0u, RLX_AGENT);
      __hip_atomic_store(sync + 1, 1u - pset, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Which 16-wide K chunk wave `w` contracts in its ci-th slot.  Contiguous per wave (default): the wave's consecutive
// 64-byte operand reads of a row fall into the same 128-byte lines.  B2T_KCHUNK_STRIDED: chunks dealt round-robin.
#ifdef B2T_KCHUNK_STRIDED
#define KCHUNK(w, ci, n) ((w) + 4 * (ci))
#else
#define KCHUNK(w, ci, n) ((w) * (n) + (ci))
#endif

constexpr int TP = 20;  // LDS pitch (floats) of a staged 16x16 tile: 16-byte aligned rows, conflict-light


// Layout of the sync workspace (unsigned words): [0] sticky error flag, [1] counter-set parity, [2] finish counter,
// [8..31] timing scratch, then two counter sets of SETW words each.  Counter of (row group rg, step t): rg*T + t.
__device__ __forceinline__ unsigned* counter_set(unsigned* sync, unsigned pset) { return sync + 32 + (size_t)pset * SETW; }

}  // namespace b2t
