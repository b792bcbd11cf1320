// gru_sync.h — inter-workgroup hand-off pieces shared by the persistent GRU sweeps (gru_persistent.hip): agent-scope counters with bounded spins, sc1 payload accesses, the self-cleaning counter sets.
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

// XCD-LOCAL hand-off (all workgroups of a row group run on ONE XCD, i.e. share one L2): the counter lives in that L2 and is
// polled with a returning L2 atomic (an ordinary load could keep hitting a stale line in the CU's vector cache), payloads are
// ordinary stores / loads (write-through to L2; every payload address is read once per kernel, after the kernel-start
// invalidate, so the vector cache holds no stale copy).  Nothing of the hand-off then crosses the fabric.
__device__ __forceinline__ unsigned l2_atomic_read(unsigned* p) {
  unsigned v; const unsigned zero = 0u;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
  return v;
}
__device__ __forceinline__ void wait_count_local(unsigned* p, unsigned target, unsigned* err) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (l2_atomic_read(p) < target) {
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
__device__ __forceinline__ void l2_atomic_inc(unsigned* p) {
  const unsigned one = 1u;
  asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(one) : "memory");
}
__device__ __forceinline__ void publish_count_local(unsigned* p) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) l2_atomic_inc(p);
}
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }

// Role of a workgroup under the XCD-local hand-off: row group r runs on XCD (2 r + par) & 7 (sweeps of layers with different
// parity use different XCDs, so two to four concurrent sweeps spread evenly); workgroups are dealt to the XCDs round-robin,
// the first G that arrive on a wanted XCD take its tiles (ticket order), the others leave.  Returns false for those.
__device__ __forceinline__ bool local_role(unsigned* tickets, int ngroups, unsigned G, int par, int& rg, int& tile) {
  __shared__ int role[2];
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    int r = -1;
    for (int i = 0; i < ngroups; ++i) if ((unsigned)((2 * i + par) & 7) == x) r = i;
    int tk = -1;
    if (r >= 0) {
      tk = (int)__hip_atomic_fetch_add(tickets + x, 1u, RLX_AGENT);
      if (tk >= (int)G) r = -1;
    }
    role[0] = r; role[1] = tk;
  }
  __syncthreads();
  rg = role[0]; tile = role[1];
  return rg >= 0;
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
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));   // v_rcp_f32: 1 ulp, one instruction
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2/(1+e^{2x}); saturates correctly: e^{2x} -> inf gives 1, -> 0 gives -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
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
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned done = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == total - 1u) {
      __hip_atomic_store(sync + 2, 0u, RLX_AGENT);
      __hip_atomic_store(sync + 1, 1u - pset, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Which 16-wide K chunk wave `w` contracts in its ci-th slot: contiguous per wave (the operand staging below relies
// on it: a wave reads 32 consecutive floats of a row per load).
#define KCHUNK(w, ci, n) ((w) * (n) + (ci))

// ---- operand block -> MFMA A fragments --------------------------------------------------------------------------
// The A fragment of v_mfma_f32_16x16x4_f32 wants lane (j = lane & 15, q = lane >> 4) to hold row j, 4 consecutive k.
// Loading it directly (one 16-byte load per lane) makes every 4 CONSECUTIVE lanes touch 4 different rows, i.e. 4
// different cache lines, and the texture addresser then accepts only 16 B/clock per CU (tools/ubench/tcp_pattern.hip:
// 2048 cycles per 32 KB block, whatever the cache policy; 515 cycles when 4 consecutive lanes read 64 consecutive
// bytes; 571 under full-chip load when 8 consecutive lanes read one whole 128-byte line).  So a wave loads the block
// line by line -- instruction (p, rh): rows 8 rh .. 8 rh + 7 of the row group, 32 floats (one line) each, starting at
// the wave's column col0 + 32 p -- and transposes each pair of instructions into two fragments through a private LDS
// slot pair (the same wave writes and reads: LDS operations of a wave execute in order, no barrier).
// Slot: 8 rows x 36 floats (144-byte pitch: the fragment reads of 16 lanes then hit 16 distinct 4-bank groups, and
// the second slot of a pair starts 1152 B = 32 banks later, interleaving with the first).
constexpr int SLOT_F = 8 * 36;

template <int N, int AUX>
__device__ __forceinline__ void issue_block_loads(float4 (&v)[N], const float* slab_uniform, int m0, int B, int ld,
                                                  int col0, int ncol, int lane) {
  const int r8 = lane >> 3, p8 = lane & 7;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int row = m0 + (i & 1) * 8 + r8, col = col0 + (i >> 1) * 32 + p8 * 4;
    // clamped, branch-free: rows beyond the batch are never stored, columns beyond the operand meet zero weights
    const long long off = (long long)(row < B ? row : B - 1) * ld + (col < ncol ? col : ncol - 4);
    v[i] = load_f4<AUX>(slab_uniform, (unsigned)(off * 4));
  }
}

// v0 = instruction (p, 0), v1 = instruction (p, 1)  ->  a0 = fragment of chunk 2p, a1 = fragment of chunk 2p + 1
__device__ __forceinline__ void transpose_pair(float* slot_pair, float4 v0, float4 v1, float4& a0, float4& a1, int lane) {
  const int r8 = lane >> 3, p8 = lane & 7;
  *reinterpret_cast<float4*>(&slot_pair[r8 * 36 + p8 * 4]) = v0;
  *reinterpret_cast<float4*>(&slot_pair[SLOT_F + r8 * 36 + p8 * 4]) = v1;
  const int j = lane & 15, q = lane >> 4;
  const float* s = slot_pair + (j >> 3) * SLOT_F + (j & 7) * 36 + 4 * q;
  a0 = *reinterpret_cast<const float4*>(s);
  a1 = *reinterpret_cast<const float4*>(s + 16);
}

// BF16 = true: the recurrent products take bf16 operands (the reference's autocast regime, opt-in).  v_mfma_f32_16x16x16_bf16
// wants exactly the fragment the fp32 path builds -- lane (j, q) holds 4 consecutive k of row j -- so the 4 floats are
// rounded to bf16 (nearest-even) and ONE MFMA replaces the four 16x16x4 fp32 ones; the weight slice is kept as bf16
// (half the registers).  Accumulation, gates and everything stored stay fp32.
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 to_bf16x4(float4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
  return r;
}
template <bool BF16> struct WFrag { using type = float4; };
template <> struct WFrag<true> { using type = bf16x4; };
template <bool BF16> __device__ __forceinline__ typename WFrag<BF16>::type make_wfrag(float4 v) {
  if constexpr (BF16) return to_bf16x4(v); else return v;
}
template <bool BF16>
__device__ __forceinline__ f32x4 mfma_chunk16(float4 a, typename WFrag<BF16>::type w, f32x4 acc) {
  if constexpr (BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(to_bf16x4(a), w, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
  }
}

// ---- bf16 hand-off ring (round 4; sweeps with bf16 operands and 32-unit workgroups) --------------------------------
// With bf16 operands the peers of a row group need a step's tile only as MFMA A fragments of v_mfma_f32_16x16x16_bf16: lane
// (j, q) holds 4 consecutive k of row j as 4 bf16.  The fp32 hand-off made every consumer load 16 B per lane line-wise, transpose
// each pair of loads through LDS and round to bf16 (98 KB per workgroup and step in the backward sweep at H = 512, two
// workgroups per CU: ~3 k cycles of a CU's L2 read port).  So the PRODUCER rounds its tile once and stores it in fragment
// order: per (ring slot, row group, chunk PAIR) 64 lanes x 16 B = {4 bf16 of chunk 2p, 4 bf16 of chunk 2p + 1} -- one 1 KB
// contiguous store per pair; a consumer wave loads one pair per instruction (1 KB contiguous) straight into the registers the
// MFMA reads: half the bytes, no transpose, no conversion.  The values are the ones the consumers rounded before (same
// nearest-even rounding of the same fp32 numbers): results are bit-identical.  The fp32 tile is still written for the GEMMs
// and the other pass, but AFTER the counter increment: it is nobody's dependency inside the sweep any more.
// The tiles of the call's steps go to slots of a buffer behind the counter sets in the sync workspace (RING_BYTES).  A call
// whose steps all fit -- every call of the training plans: 24.6 MB for a 125-step backward chunk at B = 64, H = 512 -- uses
// every address ONCE, which is what the XCD-local hand-off's ordinary loads need (they may hit the CU's vector cache, and that
// is invalidated at kernel start only; buffer_inv sc1 per step also drops the L2's lines: measured 10.5 -> 18.3 ms per step).
// Longer calls wrap around (slot = t mod depth, depth = RING_BYTES / bytes per step >= 42): safe for the counters' protocol
// from depth 3 on (a workgroup that writes step t has seen all peers publish the neighbouring step, i.e. consume the tiles two
// steps away), and between two uses of a slot a CU streams >= 42 steps x 16-144 KB of tiles through its 32 KB vector cache;
// the device-scope loads (sc1) never hit that cache anyway.
#ifndef B2T_HANDOFF16
#define B2T_HANDOFF16 1
#endif
constexpr size_t RING_BYTES = (size_t)32 << 20;   // >= 42 steps of the largest sweep (3 x 256 pairs of 1 KB per step)
__device__ __forceinline__ unsigned ring_depth(unsigned step_bytes, int T) {
  const unsigned fit = (unsigned)(RING_BYTES / step_bytes);
  return fit < (unsigned)T ? fit : (unsigned)T;
}
constexpr size_t SYNC_WORDS = (size_t)2 * SETW + 64;
using u32x2 = unsigned int __attribute__((ext_vector_type(2)));
__device__ __forceinline__ char* ring_base(unsigned* sync) { return reinterpret_cast<char*>(sync + SYNC_WORDS); }
template <int AUX>
__device__ __forceinline__ u32x4 load_u4(const char* base_uniform, unsigned byte_off) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void store_u4(char* base_uniform, unsigned byte_off, u32x4 v) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, byte_off, 0, AUX);
}
__device__ __forceinline__ u32x4 pack_frag_pair(float4 c0, float4 c1) {   // {4 bf16 of chunk 2p, 4 bf16 of chunk 2p + 1}
  const u32x2 a = __builtin_bit_cast(u32x2, to_bf16x4(c0)), b = __builtin_bit_cast(u32x2, to_bf16x4(c1));
  return u32x4{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ bf16x4 frag_lo(u32x4 v) { return __builtin_bit_cast(bf16x4, u32x2{v.x, v.y}); }
__device__ __forceinline__ bf16x4 frag_hi(u32x4 v) { return __builtin_bit_cast(bf16x4, u32x2{v.z, v.w}); }

constexpr int TP = 20;  // LDS pitch (floats) of a staged 16x16 tile: 16-byte aligned rows, conflict-light


// Layout of the sync workspace (unsigned words): [0] sticky error flag, [1] counter-set parity, [2] finish counter,
// [8..31] timing scratch, then two counter sets of SETW words each.  Counter of (row group rg, step t): rg*T + t.
__device__ __forceinline__ unsigned* counter_set(unsigned* sync, unsigned pset) { return sync + 32 + (size_t)pset * SETW; }

}  // namespace b2t
