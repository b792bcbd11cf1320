// gru_wave.hip — the step-granular LAYER WAVEFRONT of the GRU stack for gfx950 (round 6; bf16 operands, the reference's
// `use_amp: true` regime: rnn_args.yaml:19, rnn_trainer.py:527; replaces nn.GRU(num_layers = L) at rnn_model.py:65-72,126).
//
// Until round 5 the L layers overlapped at time-CHUNK granularity: layer l + 1 swept chunk c while layer l swept chunk c + 1, a
// projection GEMM and two queue hops between them, so the serial chain of a pass was (chunks + L - 1) stages of (steps + ~100 us).
// Here ALL L sweeps of a direction are resident at once in ONE launch and layer l + 1 runs one or two time STEPS behind layer l:
// the chain is T + ~1.5 (L - 1) steps.
//
// Decomposition.  A workgroup = (layer l, 16 hidden units) for ALL batch rows: wave w of its four owns row group w (16 rows) and
// runs its recurrence on its own -- there is NO workgroup barrier in the step loop and no cross-wave reduction: a wave contracts
// the whole K range itself, its slice of W_hh (forward: 48 x H; backward: W_hh^T, 16 x 3H) as bf16 B fragments in ITS registers
// (H = 768: 288 registers per lane; one workgroup per CU, 512 registers per lane).  L x H / 16 workgroups: 160 at H = 512, 240 at
// H = 768 -- every layer of the stack fits the chip at once at both bench shapes.
//   * own recurrence (the chain): h_t of the layer's H / 16 workgroups is exchanged as bf16 MFMA A fragments of
//     v_mfma_f32_16x16x32_bf16 through a ring in HBM/L2 (one slot per step; sc1 stores -> vmcnt(0) -> one agent-scope counter
//     increment per (layer, row group, step); consumers poll that word, then sc1 loads straight into the MFMA's registers).
//   * the NEXT layer's input projection is done by the CONSUMER: workgroup (l + 1, units) keeps its 48 x H slice of W_ih[l + 1] in
//     LDS (72 KB at H = 768) and contracts layer l's ring fragments of step t with it -- between publishing its own h_{t-1} and
//     the arrival of its peers' tiles, i.e. in the hand-off's shadow.  gi of the layers >= 1 never exists in memory; the projection
//     GEMMs, their pack passes and every queue hop between the layers are gone.  nn.GRU's inter-layer dropout is applied by the
//     PRODUCER (it holds the fp32 values: same Philox draws and the same rounding as dropout-then-pack), which publishes a second,
//     dropped ring for the layer above and writes the dropped fp32 values the backward pass's weight-gradient GEMM reads.
//   * backward is the mirror image: the chain contracts dGh_{t+1} (3H long) with the W_hh^T slice in registers; the gradient
//     wrt the layer's output, dY[l]_t = dGi[l + 1]_t W_ih[l + 1], is made by the consumer from layer l + 1's ring with its W_ih^T
//     slice in LDS (the input-gradient GEMMs of the layers >= 1 are gone too); fp32 dG is still written for the weight gradients.
// K order: lane (j, q) of pair p holds hidden units 32 p + 8 q .. + 7 in both operands -- one 16-byte load is one MFMA operand.
// Placement-independent: device-scope hand-off everywhere; all L x H / 16 workgroups must be resident (checked: <= CU count).
#include "gru_cell.h"
#include "gru_sync.h"
#include <stdlib.h>

namespace b2t {

using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 cvt8(float4 a, float4 b) {
  bf16x8 r;
  r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
  r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
  return r;
}
// 8 floats -> 8 bf16, all-zero where !ok: an unconditional load (clamped address) + a mask, so that no branch sits around the load
__device__ __forceinline__ u32x4 masked8(float4 a, float4 b, bool ok) {
  const u32x4 v = __builtin_bit_cast(u32x4, cvt8(a, b));
  const unsigned m = ok ? 0xffffffffu : 0u;
  return u32x4{v.x & m, v.y & m, v.z & m, v.w & m};
}
__device__ __forceinline__ bf16x8 zero8() { return __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u}); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Every lane of the wave polls the same word (one request per poll): no barrier follows -- the wave is its own recurrence.
// LOCAL: the counter lives in this XCD's L2 (all the workgroups that bump and poll it run on this XCD): polled by one lane with a
// returning L2 atomic (an ordinary load could keep hitting a stale line in the CU's vector cache), as gru_sync.h's local hand-off.
template <bool LOCAL>
__device__ __forceinline__ void wave_wait(const unsigned* p, unsigned target, unsigned* err, int lane) {
  unsigned spins = 0;
  for (;;) {
    unsigned v;
    if constexpr (LOCAL) {
      // a SCALAR load that bypasses the scalar cache (glc): it counts in lgkmcnt, so the poll does not wait for the vector loads
      // and stores this wave has in flight (a vector poll's vmcnt(0) put the step's HBM prefetches -- 12 to 24 loads -- and the
      // previous step's stores in front of every first look at the counter: 2-5 k cycles per step, R6.2).  The counter is bumped
      // by L2 atomics of the same XCD, which is where the scalar load reads it.
      (void)lane;
      // (the address is wave-uniform but derives from the thread index: readfirstlane makes it an SGPR pair for the assembler)
      const unsigned long long pa = reinterpret_cast<unsigned long long>(p);
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pa >> 32));
      const unsigned long long ps = ((unsigned long long)hi << 32) | (unsigned long long)lo;
      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ps) : "memory");
    } else {
      v = __hip_atomic_load(p, RLX_AGENT);
    }
    if (v >= target) break;
    if ((++spins & 255u) == 0u) {
      if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
      if (spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
template <bool LOCAL>
__device__ __forceinline__ void wave_bump(unsigned* p, int lane) {
  if (lane == 0) { if constexpr (LOCAL) l2_atomic_inc(p); else __hip_atomic_fetch_add(p, 1u, RLX_AGENT); }
}
__device__ __forceinline__ void wave_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Which (layer, 16-unit slice) a workgroup is.  Placement-independent form: by block index.  LOCAL form (a layer's H / 16 <= 32
// workgroups, one per CU, fill ONE XCD -- layer l on XCD l -- so that the layer's own hand-off never leaves that XCD's L2): by the
// XCD the workgroup finds itself on and a ticket; the launch is 8 x 32 workgroups dealt round-robin (verified once per process by
// gru_xcd_dispatch_ok), those without a role leave.
template <bool LOCAL>
__device__ __forceinline__ bool wave_role(unsigned* tickets, int L, int G, bool top_first, int& layer, int& slice) {
  if constexpr (LOCAL) {
    __shared__ int role[2];
    if (threadIdx.x == 0) {
      const unsigned x = xcc_id();
      int l = x < (unsigned)L ? (int)x : -1, tk = -1;
      if (l >= 0) { tk = (int)__hip_atomic_fetch_add(tickets + x, 1u, RLX_AGENT); if (tk >= G) l = -1; }
      role[0] = l; role[1] = tk;
    }
    __syncthreads();
    layer = role[0]; slice = role[1];
    return layer >= 0;
  } else {
    const int k = (int)blockIdx.x / G;
    layer = top_first ? L - 1 - k : k; slice = (int)blockIdx.x % G;
    return layer >= 0 && layer < L;
  }
}

// LDS reads the compiler does not see as LDS reads: it may neither merge them nor put its own `s_waitcnt lgkmcnt(0)` in front of
// the MFMA that uses one (it folded a four-deep software pipeline back to two and left ~100 cycles of LDS latency per MFMA exposed:
// R6.2).  The caller waits with lds_wait<N>() -- "at most N of my LDS reads still in flight" (reads return in order).
__device__ __forceinline__ u32x4 lds_read_async(const void* p) {
  u32x4 v;
  const unsigned a = (unsigned)reinterpret_cast<unsigned long long>(p);      // low 32 bits of a flat LDS address = the LDS offset
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}
template <int N> __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }
// make `v` (filled by an asynchronous read that lds_wait has covered) usable: ties the wait to the value for the compiler
__device__ __forceinline__ u32x4 lds_use(u32x4 v) { asm volatile("" : "+v"(v)); return v; }

constexpr int WTP = 20;                    // pitch (floats) of a wave's staged 16 x 16 tile
constexpr int WTILE_F = 16 * WTP;          // one tile
constexpr int WAVE_TILES = 4;              // tiles per wave (backward: dr, dz, dn r, dn; forward: h, dropped h)

// C layout of a 16 x 16 tile (lane (j, q): rows 4 q + i, column j) -> the wave's LDS tile
__device__ __forceinline__ void tile_put(float* tile, const f32x4& v, int j, int q) {
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[(4 * q + i) * WTP + j] = v[i];
}
// ... and back as fragment halves: lane L' < 32 = (row L' & 15, unit group L' >> 4) reads 8 consecutive units of its row
__device__ __forceinline__ u32x4 tile_frag(const float* tile, int lane) {
  const float* s = tile + (lane & 15) * WTP + 8 * ((lane >> 4) & 1);
  return __builtin_bit_cast(u32x4, cvt8(ld4(s), ld4(s + 4)));
}

// pair p of a row group's fragments: one 16-byte load per lane = one MFMA A operand (units 32 p + 8 q .. + 7 of row j)
// Cache policy of the fragment loads.  The tiles are written through to memory (sc1 stores, acknowledged before the counter moves)
// and every ring address is written ONCE per launch and read only after its counter is complete: a line a consumer's L2 / vector
// cache fetches is final, and neither cache can hold an older copy (both are invalidated when the kernel starts).  So ORDINARY loads
// are safe across XCDs -- the first consumer on an XCD brings the line into that L2, the other 31 CUs hit it at L2 speed -- where
// sc1 loads go to memory every time at ~10 B per clock and CU (measured: 6.6 / 12.7 us per step at C2, NOTES.md R6.2).
__device__ __forceinline__ u32x4 load_frag(const char* ring, unsigned base, int p, int P, int H, int lane, int q, bool plain = true) {
  const int pc = p < P ? p : P - 1;
  // units past H (odd H / 16: the last pair's upper half) meet zero weights: those lanes re-read the lower half (finite data)
  const unsigned lo = (32 * pc + 8 * q < H) ? (unsigned)lane * 16u : (unsigned)(lane & 31) * 16u;
  (void)plain;   // (the sc1 A/B of R6.2 is over: a run-time choice made the compiler issue BOTH loads per fragment)
  return load_u4<0>(ring, base + (unsigned)pc * 1024u + lo);
}
// One operand stream: every pair's load goes out before the first MFMA when the stream is at most 12 pairs long (48 registers);
// longer streams keep 12 in flight and refill a slot as soon as its MFMAs are issued.  NO run-time branch inside (a `p < P` test per
// pair made the compiler wait for every refill with vmcnt(0): 430 cycles per pair, R6.2): pairs beyond P re-read the last valid pair
// and meet zero weights.  BODY sees `av` (the A operand of pair `p`).
// (slot p of a stream holds pair rotp(p): the workgroups of a layer read the SAME fragments, and in the same order they all asked for
// the same line at the same moment and all sat out its fetch -- rotated by the workgroup's slice, a line's first reader fetches it
// and the others hit the L2; the weight fragments are loaded in the same rotated order.  frot_ = 0 unless every pair exists, NP == P.)
#define B2T_ROTP(p) (((p) + frot_ >= NP) ? (p) + frot_ - NP : (p) + frot_)
#define B2T_WAVE_STREAM(RING, BASE, BODY) B2T_WAVE_STREAM2(RING, BASE, {}, BODY)
#define B2T_WAVE_STREAM2(RING, BASE, EXTRA, BODY)                                                                      \
  {                                                                                                                    \
    constexpr int LB_ = NP > 12 ? 12 : NP;                                                                             \
    u32x4 v_[LB_];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < LB_; ++p) v_[p] = load_frag(RING, BASE, B2T_ROTP(p), P, H, lane, q, plain_);          \
    EXTRA                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                                   \
      const bf16x8 av = __builtin_bit_cast(bf16x8, v_[p % LB_]);                                                       \
      if (p + LB_ < NP) v_[p % LB_] = load_frag(RING, BASE, B2T_ROTP(p + LB_), P, H, lane, q, plain_);                           \
      BODY                                                                                                             \
      if (NP > LB_) __builtin_amdgcn_sched_barrier(0);   /* refills stay where they are: hoisted, they would all be in flight (spills) */ \
    }                                                                                                                  \
  }

// The same stream in the row-group form (RGF): the workgroup's four waves all need the SAME row group's fragments, so each wave
// fetches a quarter of the pairs, the stream goes through LDS (SBUF: NP KB) and every wave reads every pair from there -- a
// quarter of the L2 -> L1 traffic of four waves fetching four different row groups.  One workgroup barrier per stream.
#define B2T_WAVE_STREAM_LDS(SBUF, RING, BASE, BODY)                                                                    \
  {                                                                                                                    \
    static_assert(NP % 4 == 0, "row-group form: pairs split over four waves");                                         \
    u32x4 v_[NP / 4];                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NP / 4; ++i) v_[i] = load_frag(RING, BASE, wave + 4 * i, P, H, lane, q, plain_); \
    _Pragma("unroll") for (int i = 0; i < NP / 4; ++i)                                                                 \
      *reinterpret_cast<u32x4*>((SBUF) + (unsigned)(wave + 4 * i) * 1024u + (unsigned)lane * 16u) = v_[i];             \
    __syncthreads();                                                                                                   \
    u32x4 f_[4];                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) f_[i] = lds_read_async((SBUF) + (unsigned)i * 1024u + (unsigned)lane * 16u); \
    _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                                   \
      if (p + 4 <= NP) lds_wait<3>(); else if (p + 3 == NP) lds_wait<2>(); else if (p + 2 == NP) lds_wait<1>(); else lds_wait<0>(); \
      const bf16x8 av = __builtin_bit_cast(bf16x8, lds_use(f_[p % 4]));                                                \
      if (p + 4 < NP) f_[p % 4] = lds_read_async((SBUF) + (unsigned)(p + 4) * 1024u + (unsigned)lane * 16u);           \
      BODY                                                                                                             \
    }                                                                                                                  \
  }

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// RGF (row-group form, H % 64 == 0, H <= 512): a workgroup = (layer, ROW GROUP, 64 units) -- wave w owns units 64 ub + 16 w .. + 15 of
// that one row group -- instead of (layer, 16 units) for all row groups.  Same hand-off protocol, same tiles, same counters; but the
// four waves now consume the SAME operand streams (B2T_WAVE_STREAM_LDS) and each holds BOTH of its weight slices in registers
// (W_hh and W_ih: 384 at H = 512, pinned to register classes), since four different W_ih slices do not fit LDS.
template <int NP, bool DROP, bool LOC, bool RGF = false>   // NP: pairs of 16-unit chunks per row (H <= 32 NP); LOC: a layer = one XCD (wave_role)
__global__ __launch_bounds__(256, 1) void gru_wave_fwd_kernel(const WaveFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wave_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int H = a.H, B = a.B, T = a.T, G = H / 16, P = (G + 1) / 2, ngrp = (B + 15) / 16, L = a.L;
  const bool plain_ = LOC || (a.flags & 1) == 0;
  int layer, slice, rg_of_wg = 0;
  if constexpr (RGF) {
    int k;
    if (!wave_role<LOC>(a.tickets, L, ngrp * (H / 64), false, layer, k)) return;
    rg_of_wg = k / (H / 64);
    slice = 4 * (k % (H / 64)) + wave;
  } else {
    if (!wave_role<LOC>(a.tickets, L, G, false, layer, slice)) return;
  }
  const int u0 = slice * 16, unit = u0 + j;
  const int frot_ = (!RGF && P == NP && H == 32 * NP) ? slice % NP : 0;
  // LDS: fat form [3][NP][64] W_ih slice as B fragments; row-group form two stream buffers of NP KB; then the waves' tiles
  u32x4* wl = reinterpret_cast<u32x4*>(wave_lds);
  char* sbuf_c = wave_lds;
  char* sbuf_p = wave_lds + (size_t)NP * 1024;
  float* tiles = reinterpret_cast<float*>(wave_lds + (size_t)(RGF ? 2 : 3) * NP * 1024) + wave * (WAVE_TILES * WTILE_F);
  (void)wl; (void)sbuf_c; (void)sbuf_p;

  bf16x8 w[3][NP];
  {
    const float* whh = a.w_hh[layer];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int k0 = 32 * B2T_ROTP(p) + 8 * q;
      const bool ok = k0 < H;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float* src = whh + ((long long)g * H + unit) * H + (ok ? k0 : 0);
        w[g][p] = __builtin_bit_cast(bf16x8, masked8(ld4(src), ld4(src + 4), ok));
        // H > 512: 288 registers of weights.  Left to itself the allocator spilled 63 of the 72 fragments to scratch and re-read them
        // in front of every MFMA; pinned -- two gates in accumulation registers (192 of the 256), the third in architectural ones --
        // nothing spills (R6.2)
        if constexpr (NP > 16) { if (g < 2) asm volatile("" : "+a"(w[g][p])); else asm volatile("" : "+v"(w[g][p])); }
        else if constexpr (NP == 16 && !RGF) asm volatile("" : "+a"(w[g][p]));
        if constexpr (RGF) asm volatile("" : "+a"(w[g][p]));      // row-group form: W_hh in accumulation registers, W_ih (below) mostly in architectural ones
      }
      __builtin_amdgcn_sched_barrier(0);   // (a pair's six loads are converted before the next pair's go out: hoisted, all 6 NP loads would be live)
    }
  }
  bf16x8 w2[RGF ? 3 : 1][RGF ? NP : 1];      // row-group form: the W_ih slice
  if constexpr (RGF) {
    const float* wih = a.w_ih[layer > 0 ? layer : 1 < L ? 1 : 0];     // (layer 0 projects nothing: any valid matrix, never used)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int k0 = 32 * p + 8 * q;
      const bool ok = k0 < H && layer > 0;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float* src = (layer > 0 ? wih : a.w_hh[layer]) + ((long long)g * H + unit) * H + (k0 < H ? k0 : 0);
        w2[g][p] = __builtin_bit_cast(bf16x8, masked8(ld4(src), ld4(src + 4), ok));
        if (g == 0 && p < (NP * 3) / 4) asm volatile("" : "+a"(w2[g][p])); else asm volatile("" : "+v"(w2[g][p]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (layer > 0) {
    const float* wih = a.w_ih[layer];
    for (int idx = wave; idx < 3 * NP; idx += 4) {
      const int g = idx / NP, p = idx % NP, k0 = 32 * B2T_ROTP(p) + 8 * q;
      const bool ok = k0 < H;
      const float* src = wih + ((long long)g * H + unit) * H + (ok ? k0 : 0);
      wl[idx * 64 + lane] = masked8(ld4(src), ld4(src + 4), ok);
    }
  }
  __syncthreads();
  if (!RGF && wave >= ngrp) return;
  __builtin_amdgcn_s_setprio(3);

  // Two rings per layer when the layer above must not read the own-recurrence ring: with dropout (it reads the DROPPED states) and
  // under LOC (it sits on another XCD: the own ring is written with ordinary stores into this XCD's L2, the copy for the layer above
  // is written through to memory).  The cross ring's counter of slot s moves one step late -- when the NEXT own publish has drained
  // this wave's stores anyway -- so that no step waits for a memory acknowledgement of its own.
  constexpr bool XRING = DROP || LOC;
  const int rg = RGF ? rg_of_wg : wave, m0 = rg * 16;
  unsigned* err = a.err;
  const size_t cstride = (size_t)(T + 1);
  unsigned* cnt_own = a.cnt + ((size_t)(layer * 2 + 0) * ngrp + rg) * cstride;
  unsigned* cnt_x = a.cnt + ((size_t)(layer * 2 + 1) * ngrp + rg) * cstride;
  const bool feeds = layer + 1 < L;                 // a layer above reads this layer's outputs
  const bool dropping = DROP && feeds;              // ... through nn.GRU's dropout
  const unsigned* cnt_in = layer > 0 ? a.cnt + ((size_t)((layer - 1) * 2 + (XRING ? 1 : 0)) * ngrp + rg) * cstride : nullptr;
  char* ring = a.ring[layer];
  char* ringx = a.ringd[layer];
  const char* ring_in = layer > 0 ? (XRING ? a.ringd[layer - 1] : a.ring[layer - 1]) : nullptr;
  const unsigned slot_bytes = (unsigned)ngrp * (unsigned)P * 1024u, rg_off = (unsigned)rg * (unsigned)P * 1024u;
  const unsigned my_frag = (unsigned)(slice >> 1) * 1024u + (unsigned)(slice & 1) * 512u + (unsigned)(lane & 31) * 16u;
  const float bhr = a.b_hh[layer][unit], bhz = a.b_hh[layer][H + unit], bhn = a.b_hh[layer][2 * H + unit];
  float bi[3] = {0.f, 0.f, 0.f};
  if (layer > 0) { bi[0] = a.b_ih[layer][unit]; bi[1] = a.b_ih[layer][H + unit]; bi[2] = a.b_ih[layer][2 * H + unit]; }
  bool live[4];
  f32x4 hp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + 4 * q + i;
    live[i] = row < B;
    hp[i] = live[i] ? a.h_init[layer][(long long)row * H + unit] : 0.f;
  }
  // slot 0 of the ring = the initial state
  tile_put(tiles, hp, j, q);
  {
    const u32x4 f = tile_frag(tiles, lane);
    if (lane < 32) store_u4<LOC ? 0 : 16>(ring, rg_off + my_frag, f);
    wave_drain();
    wave_bump<LOC>(cnt_own, lane);
  }
  int pending_x = -1;      // slot of the cross ring whose store is in flight (its counter moves at the next drain)

#ifdef B2T_WAVE_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define WSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; }
#else
#define WSTAMP(i)
#endif
  f32x4 gi[3];
  // gi of step t from layer - 1's (dropped) h_t = slot t + 1 of its ring, W_ih slice from LDS
  auto project = [&](int t) {
    wave_wait<false>(cnt_in + (t + 1), (unsigned)G, err, lane);
    WSTAMP(5)   // wait for the layer below
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned pbase = (unsigned)(t + 1) * slot_bytes + rg_off;
    if constexpr (RGF) {
      B2T_WAVE_STREAM_LDS(sbuf_p, ring_in, pbase, {
        _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w2[g][p], acc[g], 0, 0, 0);
      })
    } else {
      // the B fragments come from LDS: the NEXT pair's three are requested before this pair's MFMAs (asynchronous reads, above)
      u32x4 b_[2][3];
      _Pragma("unroll") for (int g = 0; g < 3; ++g) b_[0][g] = lds_read_async(&wl[(g * NP + 0) * 64 + lane]);
      B2T_WAVE_STREAM(ring_in, pbase, {
        if (p + 1 < NP) { _Pragma("unroll") for (int g = 0; g < 3; ++g) b_[(p + 1) & 1][g] = lds_read_async(&wl[(g * NP + p + 1) * 64 + lane]); }
        if (p + 1 < NP) lds_wait<3>(); else lds_wait<0>();
        _Pragma("unroll") for (int g = 0; g < 3; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, lds_use(b_[p & 1][g])), acc[g], 0, 0, 0);
      })
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) gi[g][i] = acc[g][i] + bi[g];
    WSTAMP(6)   // projection
  };
  if (layer > 0) project(0);

  // layer 0's input projection comes from memory (HBM): the loads go out before the wait and land during it (local form: the scalar
  // poll does not wait for them; behind the recurrent product's first loads instead they cost the placement-independent form 2.5 k
  // cycles per step, R6.2)
  auto load_gi0 = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* g3 = a.gi0 + ((long long)t * B + (m0 + 4 * q + i)) * 3 * H + unit;
#pragma unroll
      for (int g = 0; g < 3; ++g) gi[g][i] = live[i] ? __builtin_nontemporal_load(g3 + (long long)g * H) : 0.f;
    }
  };
  for (int t = 0; t < T; ++t) {
    if (layer == 0) load_gi0(t);
    wave_wait<LOC>(cnt_own + t, (unsigned)G, err, lane);
    WSTAMP(0)   // wait for the peers' h_{t-1}
    f32x4 gh[3];
    {
#pragma unroll
      for (int g = 0; g < 3; ++g) gh[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned cbase = (unsigned)t * slot_bytes + rg_off;
      if constexpr (RGF) {
        B2T_WAVE_STREAM_LDS(sbuf_c, ring, cbase, {
          _Pragma("unroll") for (int g = 0; g < 3; ++g) gh[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[g][p], gh[g], 0, 0, 0);
        })
      } else {
        B2T_WAVE_STREAM2(ring, cbase, {}, {
          _Pragma("unroll") for (int g = 0; g < 3; ++g) gh[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[g][p], gh[g], 0, 0, 0);
        })
      }
    }
    asm volatile("s_nop 0" :: "v"(gh[0][0]), "v"(gh[1][0]), "v"(gh[2][0]));
    WSTAMP(1)   // operand loads + recurrent product
    f32x4 sr, sz, sn, sg, h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ghn = gh[2][i] + bhn;
      const float r = fast_sigmoid(gi[0][i] + gh[0][i] + bhr);
      const float z = fast_sigmoid(gi[1][i] + gh[1][i] + bhz);
      const float nn = fast_tanh(gi[2][i] + r * ghn);
      h[i] = (1.0f - z) * nn + z * hp[i];
      sr[i] = r; sz[i] = z; sn[i] = nn; sg[i] = ghn;
    }
    hp = h;
    // publish h_t: slot t + 1
    tile_put(tiles, h, j, q);
    const u32x4 f = tile_frag(tiles, lane);
    if (lane < 32) store_u4<LOC ? 0 : 16>(ring, (unsigned)(t + 1) * slot_bytes + rg_off + my_frag, f);
    WSTAMP(2)   // gates + tile
    wave_drain();
    wave_bump<LOC>(cnt_own + (t + 1), lane);
    if (XRING && pending_x >= 0) wave_bump<false>(cnt_x + pending_x, lane);   // (drained above)
    WSTAMP(3)   // store acknowledged, counters
    // fp32 copy (backward, weight gradients, head): lane (row L & 15, unit group L >> 4) stores 4 units
    const int rrow = m0 + (lane & 15), kg = lane >> 4;
    const float4 hv = ld4(tiles + (lane & 15) * WTP + 4 * kg);
    if (XRING && feeds) {
      u32x4 fx = f;
      if (dropping) {
        // nn.GRU's dropout on the way to the layer above: flat element (t, row, unit) of this layer's output, as b2t_dropout_f32
        const long long e = a.elem0 + ((long long)t * B + (rrow < B ? rrow : 0)) * H + u0 + 4 * kg;
        const float4 u = Philox::uniform4(a.seed[layer], (uint64_t)(e >> 2), 2u);
        float4 hd;
        hd.x = u.x >= a.drop_p ? hv.x * a.drop_scale : 0.f; hd.y = u.y >= a.drop_p ? hv.y * a.drop_scale : 0.f;
        hd.z = u.z >= a.drop_p ? hv.z * a.drop_scale : 0.f; hd.w = u.w >= a.drop_p ? hv.w * a.drop_scale : 0.f;
        float* td = tiles + WTILE_F;
        *reinterpret_cast<float4*>(td + (lane & 15) * WTP + 4 * kg) = hd;
        fx = tile_frag(td, lane);
        if (rrow < B) *reinterpret_cast<float4*>(a.outd[layer] + ((long long)t * B + rrow) * H + u0 + 4 * kg) = hd;
      }
      if (lane < 32) store_u4<16>(ringx, (unsigned)(t + 1) * slot_bytes + rg_off + my_frag, fx);
      pending_x = t + 1;
    }
    if (rrow < B) *reinterpret_cast<float4*>(a.out[layer] + ((long long)t * B + rrow) * H + u0 + 4 * kg) = hv;
    if (a.reserve[layer]) {
      // (r, z, n, gh_n) row-major through the wave's four tiles: 4 x 16-byte stores per lane instead of 16 x 4-byte ones
      tile_put(tiles + 0 * WTILE_F, sr, j, q); tile_put(tiles + 1 * WTILE_F, sz, j, q);
      tile_put(tiles + 2 * WTILE_F, sn, j, q); tile_put(tiles + 3 * WTILE_F, sg, j, q);
      if (rrow < B) {
        float* rs = a.reserve[layer] + ((long long)t * B + rrow) * 4 * H + u0 + 4 * kg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v4 = ld4(tiles + g * WTILE_F + (lane & 15) * WTP + 4 * kg);
          __builtin_nontemporal_store(v4.x, rs + (long long)g * H); __builtin_nontemporal_store(v4.y, rs + (long long)g * H + 1);
          __builtin_nontemporal_store(v4.z, rs + (long long)g * H + 2); __builtin_nontemporal_store(v4.w, rs + (long long)g * H + 3);
        }
      }
    }
    WSTAMP(4)   // cross ring, fp32 stores
    if (layer > 0 && t + 1 < T) project(t + 1);
  }
  if (XRING && pending_x >= 0) { wave_drain(); wave_bump<false>(cnt_x + pending_x, lane); }
#ifdef B2T_WAVE_TIMING
  if (slice == 0 && wave == 0 && lane == 0 && a.timing)
    for (int i = 0; i < 8; ++i) a.timing[layer * 8 + i] = (unsigned)(tacc[i] / (unsigned long long)T);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  Ring of layer l, slot t: the gate gradients of step t as fragments, 4 arrays (dr, dz, dn r, dn) x P pairs.
// ---------------------------------------------------------------------------------------------------------------------
template <int NP, bool DROP, bool LOC, bool RGF = false>      // RGF: the row-group form (see the forward kernel)
__global__ __launch_bounds__(256, 1) void gru_wave_bwd_kernel(const WaveBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wave_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int H = a.H, B = a.B, T = a.T, G = H / 16, P = (G + 1) / 2, ngrp = (B + 15) / 16, L = a.L;
  const bool plain_ = LOC || (a.flags & 1) == 0;
  // (placement-independent form: the TOP layer starts the wavefront and gets the first workgroups)
  int layer, slice, rg_of_wg = 0;
  if constexpr (RGF) {
    int k;
    if (!wave_role<LOC>(a.tickets, L, ngrp * (H / 64), true, layer, k)) return;
    rg_of_wg = k / (H / 64);
    slice = 4 * (k % (H / 64)) + wave;
  } else {
    if (!wave_role<LOC>(a.tickets, L, G, true, layer, slice)) return;
  }
  const int u0 = slice * 16, unit = u0 + j;
  const int frot_ = (!RGF && P == NP && H == 32 * NP) ? slice % NP : 0;
  // LDS: fat form [3][NP][64] W_ih[layer + 1]^T slice; row-group form two stream buffers of 3 NP KB (three arrays each); then the tiles
  u32x4* wl = reinterpret_cast<u32x4*>(wave_lds);
  char* sbuf_c = wave_lds;
  char* sbuf_p = wave_lds + (size_t)3 * NP * 1024;
  float* tiles = reinterpret_cast<float*>(wave_lds + (size_t)(RGF ? 6 : 3) * NP * 1024) + wave * (WAVE_TILES * WTILE_F);
  (void)wl; (void)sbuf_c; (void)sbuf_p;
  const bool has_up = layer + 1 < L;

  bf16x8 w[3][NP];     // W_hh^T slice: column `unit`, k = array * H + 32 p + 8 q .. + 7
  {
    const float* wt = a.w_hh_t[layer] + (long long)unit * 3 * H;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int k0 = 32 * B2T_ROTP(p) + 8 * q;
      const bool ok = k0 < H;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float* src = wt + (long long)g * H + (ok ? k0 : 0);
        w[g][p] = __builtin_bit_cast(bf16x8, masked8(ld4(src), ld4(src + 4), ok));
        // H > 512: 288 registers of weights.  Left to itself the allocator spilled 63 of the 72 fragments to scratch and re-read them
        // in front of every MFMA; pinned -- two gates in accumulation registers (192 of the 256), the third in architectural ones --
        // nothing spills (R6.2)
        if constexpr (NP > 16) { if (g < 2) asm volatile("" : "+a"(w[g][p])); else asm volatile("" : "+v"(w[g][p])); }
        else if constexpr (NP == 16 && !RGF) asm volatile("" : "+a"(w[g][p]));
        if constexpr (RGF) asm volatile("" : "+a"(w[g][p]));
      }
      __builtin_amdgcn_sched_barrier(0);   // (a pair's six loads are converted before the next pair's go out: hoisted, all 6 NP loads would be live)
    }
  }
  bf16x8 w2[RGF ? 3 : 1][RGF ? NP : 1];      // row-group form: the W_ih[layer + 1]^T slice
  if constexpr (RGF) {
    const float* wt = (has_up ? a.w_ih_t[layer + 1] : a.w_hh_t[layer]) + (long long)unit * 3 * H;     // (top layer: any valid matrix, masked to zero)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int k0 = 32 * p + 8 * q;
      const bool ok = k0 < H && has_up;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float* src = wt + (long long)g * H + (k0 < H ? k0 : 0);
        w2[g][p] = __builtin_bit_cast(bf16x8, masked8(ld4(src), ld4(src + 4), ok));
        if (g == 0 && p < (NP * 3) / 4) asm volatile("" : "+a"(w2[g][p])); else asm volatile("" : "+v"(w2[g][p]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (has_up) {
    const float* wt = a.w_ih_t[layer + 1] + (long long)unit * 3 * H;
    for (int idx = wave; idx < 3 * NP; idx += 4) {
      const int g = idx / NP, p = idx % NP, k0 = 32 * B2T_ROTP(p) + 8 * q;
      const bool ok = k0 < H;
      const float* src = wt + (long long)g * H + (ok ? k0 : 0);
      wl[idx * 64 + lane] = masked8(ld4(src), ld4(src + 4), ok);
    }
  }
  __syncthreads();
  if (!RGF && wave >= ngrp) return;
  __builtin_amdgcn_s_setprio(3);

  // LOC: the own-recurrence ring stays in this XCD's L2 (ordinary stores, L2 counters); the layer BELOW (another XCD) reads a
  // second, written-through copy whose counter moves one step late (see the forward kernel)
  constexpr bool XRING = LOC;
  const int rg = RGF ? rg_of_wg : wave, m0 = rg * 16;
  unsigned* err = a.err;
  const size_t cstride = (size_t)T;
  unsigned* cnt_own = a.cnt + ((size_t)(layer * 2 + 0) * ngrp + rg) * cstride;
  unsigned* cnt_x = a.cnt + ((size_t)(layer * 2 + 1) * ngrp + rg) * cstride;
  const bool feeds = layer > 0;                     // a layer below reads this layer's gate gradients
  const unsigned* cnt_up = has_up ? a.cnt + ((size_t)((layer + 1) * 2 + (XRING ? 1 : 0)) * ngrp + rg) * cstride : nullptr;
  char* ring = a.ring[layer];
  char* ringx = a.ringx[layer];
  const char* ring_up = has_up ? (XRING ? a.ringx[layer + 1] : a.ring[layer + 1]) : nullptr;
  const unsigned arr_bytes = (unsigned)P * 1024u, rg_bytes = 4u * arr_bytes, slot_bytes = (unsigned)ngrp * rg_bytes, rg_off = (unsigned)rg * rg_bytes;
  const unsigned my_frag = (unsigned)(slice >> 1) * 1024u + (unsigned)(slice & 1) * 512u + (unsigned)(lane & 31) * 16u;
  bool live[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) live[i] = m0 + 4 * q + i < B;
  const int rrow = m0 + (lane & 15), kg = lane >> 4;   // the (row, 4-unit group) this lane handles in row-major passes
  int pending_x = -1;

#ifdef B2T_WAVE_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
  f32x4 dy = f32x4{0.f, 0.f, 0.f, 0.f};
  // dY[layer]_t = dGi[layer + 1]_t . W_ih[layer + 1][:, units] (arrays dr, dz, dn of the ring above), then the dropout mask
  auto project = [&](int t) {
    wave_wait<false>(cnt_up + t, (unsigned)G, err, lane);
    WSTAMP(5)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RGF) {
      // the three arrays (dr, dz, dn) of the layer above, a quarter of the pairs per wave, through LDS: ONE barrier
      static_assert(!RGF || NP % 4 == 0, "row-group form: pairs split over four waves");
#pragma unroll
      for (int g = 0; g < 3; ++g) {      // (array by array: NP / 4 loads in flight per wave, not 3 NP / 4 -- registers)
        u32x4 v_[NP / 4];
#pragma unroll
        for (int i = 0; i < NP / 4; ++i)
          v_[i] = load_frag(ring_up, (unsigned)t * slot_bytes + rg_off + (unsigned)(g == 2 ? 3 : g) * arr_bytes, wave + 4 * i, P, H, lane, q, plain_);
#pragma unroll
        for (int i = 0; i < NP / 4; ++i)
          *reinterpret_cast<u32x4*>(sbuf_p + (unsigned)(g * NP + wave + 4 * i) * 1024u + (unsigned)lane * 16u) = v_[i];
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      {
        // LDS reads four fragments ahead, two accumulators (a read's ~128 cycles and a dependent MFMA's ~40 were the loop: 150 per pair)
        f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 f_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f_[i] = lds_read_async(sbuf_p + (unsigned)i * 1024u + (unsigned)lane * 16u);
#pragma unroll
        for (int k = 0; k < 3 * NP; ++k) {
          if (k + 4 <= 3 * NP) lds_wait<3>(); else if (k + 3 == 3 * NP) lds_wait<2>(); else if (k + 2 == 3 * NP) lds_wait<1>(); else lds_wait<0>();
          const bf16x8 av = __builtin_bit_cast(bf16x8, lds_use(f_[k % 4]));
          if (k + 4 < 3 * NP) f_[k % 4] = lds_read_async(sbuf_p + (unsigned)(k + 4) * 1024u + (unsigned)lane * 16u);
          if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w2[k / NP][k % NP], acc1, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w2[k / NP][k % NP], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += acc1[i];
      }
    } else {
    // (two accumulators, pairs alternating: one accumulator is a chain of dependent MFMAs at ~40 cycles each instead of 17)
    f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const unsigned pbase = (unsigned)t * slot_bytes + rg_off + (unsigned)(g == 2 ? 3 : g) * arr_bytes;
      B2T_WAVE_STREAM(ring_up, pbase, {
        if (p & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, wl[(g * NP + p) * 64 + lane]), acc1, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, wl[(g * NP + p) * 64 + lane]), acc, 0, 0, 0);
      })
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += acc1[i];
    }
    if (DROP) {
      // the mask of the forward's dropout on out[layer]: row-major through the wave's tile (one Philox block per lane)
      float* td = tiles;
      tile_put(td, acc, j, q);
      float4 v4 = ld4(td + (lane & 15) * WTP + 4 * kg);
      const long long e = a.elem0 + ((long long)t * B + (rrow < B ? rrow : 0)) * H + u0 + 4 * kg;
      const float4 u = Philox::uniform4(a.seed[layer], (uint64_t)(e >> 2), 2u);
      v4.x = u.x >= a.drop_p ? v4.x * a.drop_scale : 0.f; v4.y = u.y >= a.drop_p ? v4.y * a.drop_scale : 0.f;
      v4.z = u.z >= a.drop_p ? v4.z * a.drop_scale : 0.f; v4.w = u.w >= a.drop_p ? v4.w * a.drop_scale : 0.f;
      *reinterpret_cast<float4*>(td + (lane & 15) * WTP + 4 * kg) = v4;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = td[(4 * q + i) * WTP + j];
    }
    dy = acc;
    WSTAMP(6)
  };
  if (has_up) project(T - 1);

  f32x4 dzterm = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 r, z, nv, ghn, hprev;
  // the step's elementwise operands (saved gates, h_{t-1}, the top layer's dY) come from memory (HBM): issued before the poll,
  // they land during the wait (the local form's scalar poll does not even wait for them)
  auto prefetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[i] = z[i] = nv[i] = ghn[i] = hprev[i] = 0.f;
      if (live[i] && t >= 0) {
        const int row = m0 + 4 * q + i;
        const float* rs = a.reserve[layer] + ((long long)t * B + row) * 4 * H + unit;
        r[i] = __builtin_nontemporal_load(rs); z[i] = __builtin_nontemporal_load(rs + H);
        nv[i] = __builtin_nontemporal_load(rs + 2 * H); ghn[i] = __builtin_nontemporal_load(rs + 3 * H);
        hprev[i] = t > 0 ? a.out[layer][((long long)(t - 1) * B + row) * H + unit] : a.h_init[layer][(long long)row * H + unit];
        if (!has_up) dy[i] = __builtin_nontemporal_load(a.dY_top + ((long long)t * B + row) * H + unit);
      }
    }
  };
  for (int t = T - 1; t >= -1; --t) {
    prefetch(t);
    f32x4 carry = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < T - 1) {
      wave_wait<LOC>(cnt_own + (t + 1), (unsigned)G, err, lane);
      WSTAMP(0)
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (RGF) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          u32x4 v_[NP / 4];
#pragma unroll
          for (int i = 0; i < NP / 4; ++i)
            v_[i] = load_frag(ring, (unsigned)(t + 1) * slot_bytes + rg_off + (unsigned)g * arr_bytes, wave + 4 * i, P, H, lane, q, plain_);
#pragma unroll
          for (int i = 0; i < NP / 4; ++i)
            *reinterpret_cast<u32x4*>(sbuf_c + (unsigned)(g * NP + wave + 4 * i) * 1024u + (unsigned)lane * 16u) = v_[i];
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        {
          f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
          u32x4 f_[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) f_[i] = lds_read_async(sbuf_c + (unsigned)i * 1024u + (unsigned)lane * 16u);
#pragma unroll
          for (int k = 0; k < 3 * NP; ++k) {
            if (k + 4 <= 3 * NP) lds_wait<3>(); else if (k + 3 == 3 * NP) lds_wait<2>(); else if (k + 2 == 3 * NP) lds_wait<1>(); else lds_wait<0>();
          const bf16x8 av = __builtin_bit_cast(bf16x8, lds_use(f_[k % 4]));
            if (k + 4 < 3 * NP) f_[k % 4] = lds_read_async(sbuf_c + (unsigned)(k + 4) * 1024u + (unsigned)lane * 16u);
            if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[k / NP][k % NP], acc1, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[k / NP][k % NP], acc, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += acc1[i];
        }
      } else {
      f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const unsigned cbase = (unsigned)(t + 1) * slot_bytes + rg_off + (unsigned)g * arr_bytes;
        B2T_WAVE_STREAM2(ring, cbase, {}, {
          if (p & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[g][p], acc1, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[g][p], acc, 0, 0, 0);
        })
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += acc1[i];
      }
      asm volatile("s_nop 0" :: "v"(acc[0]));
      WSTAMP(1)
#pragma unroll
      for (int i = 0; i < 4; ++i) carry[i] = acc[i] + dzterm[i];
    } else if (a.dh_last[layer]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) if (live[i]) carry[i] = a.dh_last[layer][(long long)(m0 + 4 * q + i) * H + unit];
    }
    if (t < 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) if (live[i]) a.dh_init[layer][(long long)(m0 + 4 * q + i) * H + unit] = carry[i];
      break;
    }
    f32x4 g0, g1, g2, g3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = dy[i] + carry[i];
      const float dn = d * (1.0f - z[i]);
      const float dz = d * (hprev[i] - nv[i]);
      const float dn_pre = dn * (1.0f - nv[i] * nv[i]);
      const float dz_pre = dz * z[i] * (1.0f - z[i]);
      const float dr_pre = dn_pre * ghn[i] * r[i] * (1.0f - r[i]);
      g0[i] = dr_pre; g1[i] = dz_pre; g2[i] = dn_pre * r[i]; g3[i] = dn_pre;
      dzterm[i] = d * z[i];
    }
    tile_put(tiles + 0 * WTILE_F, g0, j, q); tile_put(tiles + 1 * WTILE_F, g1, j, q);
    tile_put(tiles + 2 * WTILE_F, g2, j, q); tile_put(tiles + 3 * WTILE_F, g3, j, q);
    // lanes 0-31 publish arrays 0 and 2, lanes 32-63 arrays 1 and 3
    const int a0 = lane >> 5;
    const u32x4 f0 = tile_frag(tiles + a0 * WTILE_F, lane), f1 = tile_frag(tiles + (a0 + 2) * WTILE_F, lane);
    const unsigned base = (unsigned)t * slot_bytes + rg_off + my_frag;
    store_u4<LOC ? 0 : 16>(ring, base + (unsigned)a0 * arr_bytes, f0);
    store_u4<LOC ? 0 : 16>(ring, base + (unsigned)(a0 + 2) * arr_bytes, f1);
    WSTAMP(2)
    wave_drain();
    wave_bump<LOC>(cnt_own + t, lane);
    if (XRING && pending_x >= 0) wave_bump<false>(cnt_x + pending_x, lane);
    // progress for consumers OUTSIDE the launch (the gated weight-gradient GEMMs, wave_gate_kernel): the drain above acknowledged
    // every store of the steps > t, so G x (T - t) bumps of this word certify dG[t + 1 .. T) in memory (written through below)
    if (a.prog) wave_bump<false>(a.prog + layer * ngrp + rg, lane);
    WSTAMP(3)
    if (XRING && feeds) {
      store_u4<16>(ringx, base + (unsigned)a0 * arr_bytes, f0);
      store_u4<16>(ringx, base + (unsigned)(a0 + 2) * arr_bytes, f1);
      pending_x = t;
    }
    if (rrow < B) {
      float* dgl = a.dG[layer] + (long long)t * B * 4 * H;
      const unsigned off = (unsigned)(((long long)rrow * 4 * H + u0 + 4 * kg) * 4);
      if (a.flags & 4) {      // (written through only for readers inside the sweep's lifetime: see gru_wave_ks.h)
#pragma unroll
        for (int g = 0; g < 4; ++g) store_f4<16>(dgl, off + (unsigned)g * (unsigned)H * 4u, ld4(tiles + g * WTILE_F + (lane & 15) * WTP + 4 * kg));
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) store_f4<0>(dgl, off + (unsigned)g * (unsigned)H * 4u, ld4(tiles + g * WTILE_F + (lane & 15) * WTP + 4 * kg));
      }
    }
    WSTAMP(4)
    if (has_up && t > 0) project(t - 1);
  }
  if (XRING && pending_x >= 0) { wave_drain(); wave_bump<false>(cnt_x + pending_x, lane); }
  if (a.prog) { wave_drain(); wave_bump<false>(a.prog + layer * ngrp + rg, lane); }     // G x (T + 1): everything is in memory
#ifdef B2T_WAVE_TIMING
  if (slice == 0 && wave == 0 && lane == 0 && a.timing)
    for (int i = 0; i < 8; ++i) a.timing[layer * 8 + i] = (unsigned)(tacc[i] / (unsigned long long)T);
#endif
}

// A consumer outside the launch waits for the sweep's progress: one wave, lane r polls word r (row group r of the layer) until
// it reaches `target` = G x (T - t0 + 1) -- dG[t0 .. T) of that layer is then in memory.  Enqueued in front of a gated GEMM on the
// GEMM's queue; it runs on a CU the sweeps leave free.  Bounded like every spin here (sets the sticky error word).
__global__ void wave_gate_kernel(const unsigned* prog, int n, unsigned target, unsigned* err) {
  const int lane = threadIdx.x;
  if (lane >= n) return;
  unsigned spins = 0;
  while (__hip_atomic_load(prog + lane, RLX_AGENT) < target) {
    if ((++spins & 255u) == 0u) {
      if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
      if (spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

#include "gru_wave_ks.h"

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static int wave_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    n = p.multiProcessorCount;
  }
  return n;
}

size_t gru_wave_lds_bytes_rgf_bwd(int H);
static int wave_np(int H) { return H <= 128 ? 4 : H <= 256 ? 8 : H <= 512 ? 16 : 24; }   // the kernel template's pair count
size_t gru_wave_lds_bytes(int H) { return (size_t)3 * wave_np(H) * 1024 + (size_t)4 * WAVE_TILES * WTILE_F * sizeof(float); }
size_t gru_wave_lds_bytes_rgf_bwd(int H) { return (size_t)6 * wave_np(H) * 1024 + (size_t)4 * WAVE_TILES * WTILE_F * sizeof(float); }   // two stream buffers of three arrays
size_t gru_wave_ring_bytes_fwd(int T, int B, int H) { return (size_t)(T + 1 > KS_D ? T + 1 : KS_D) * ((B + 15) / 16) * ((H / 16 + 1) / 2) * 1024; }   // (>= KS_D slots: the K-split form's own ring)
size_t gru_wave_ring_bytes_bwd(int T, int B, int H) { return (size_t)(T > KS_D ? T : KS_D) * ((B + 15) / 16) * 4 * ((H / 16 + 1) / 2) * 1024; }
size_t gru_wave_cnt_words_fwd(int L, int T, int B) { return 16 + (size_t)L * 2 * ((B + 15) / 16) * (T + 1); }   // 16: the XCD tickets of the local form
size_t gru_wave_cnt_words_bwd(int L, int T, int B) { return 16 + 64 + (size_t)L * 2 * ((B + 15) / 16) * T; }   // tickets, progress words [L][row groups] (<= 32), counters

// Shapes the wavefront serves; `why` (optional) receives the reason when it does not.
bool gru_wave_ok(int L, int T, int B, int H, const char** why) {
  const char* w = nullptr;
  if (L < 1 || L > B2T_MAX_LAYERS) w = "layer count";
  else if (H % 16 != 0 || H < 16 || H > 768) w = "H must be a multiple of 16, <= 768";
  else if (B < 1 || B > 64) w = "B must be <= 64 (one wave per row group)";
  else if (T < 1) w = "T";
  else if ((long long)L * (H / 16) > wave_cus()) w = "L x H / 16 workgroups exceed the CUs (all layers must be resident)";
  else if (gru_wave_ring_bytes_bwd(T, B, H) >= ((size_t)1 << 31) || gru_wave_ring_bytes_fwd(T, B, H) >= ((size_t)1 << 31)) w = "ring of one layer must stay below 2 GB";
  if (why) *why = w;
  return w == nullptr;
}

template <typename K> static int wave_lds_attr(K kernel, size_t bytes) {
  return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "gru_wave: LDS size");
}

static int wave_flags() {   // read per call (A/B in one process): bit 0 = sc1 (memory-side) fragment loads in the placement-independent form
  const char* e = getenv("B2T_WAVE_SC1_LOADS");
  return (e && atoi(e) != 0) ? 1 : 0;
}
// the local form: a layer's H / 16 <= 32 workgroups on one XCD (B2T_WAVE_LOCAL=0, read per call: the placement-independent form)
bool gru_wave_local(int L, int H) {
  const char* e = getenv("B2T_WAVE_LOCAL");
  if (e && atoi(e) == 0) return false;
  return L <= 8 && H / 16 <= 32 && wave_cus() >= 256 && gru_xcd_dispatch_ok();
}

template <int NPV, bool DR, bool LC, bool RG = false> static int wave_launch_fwd(const WaveFwdArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool at = false;
  if (!at) { const int rc = wave_lds_attr(gru_wave_fwd_kernel<NPV, DR, LC, RG>, gru_wave_lds_bytes(32 * NPV)); if (rc) return rc; at = true; }
  hipLaunchKernelGGL((gru_wave_fwd_kernel<NPV, DR, LC, RG>), grid, dim3(256), lds, s, a);
  return 0;
}
// the row-group form serves H % 64 == 0, H <= 256 (both weight slices of a wave in registers); B2T_WAVE_RGF=1 (read per call) selects it
// (round 6, late: superseded by the K-split form and OFF unless B2T_WAVE_RGF=1; H <= 256 only -- at H = 512 both weight slices of a
// wave plus the stream buffers do not fit 512 registers and the allocator spilled MFMA operand tuples: see gru_wave_ks.h)
bool gru_wave_rgf(int H) {
  const char* e = getenv("B2T_WAVE_RGF");
  return e && atoi(e) != 0 && H % 64 == 0 && H <= 256;
}
// the K-split form (gru_wave_ks.h): local placement, H % 128 == 0, H <= 512, a layer's workgroups fit one XCD; B2T_WAVE_KS=0 (read per call): off
bool gru_wave_ks(int L, int T, int B, int H) {
  const char* e = getenv("B2T_WAVE_KS");
  if (e && atoi(e) == 0) return false;
  const int nrh = ((B + 15) / 16 + 1) / 2;
  if ((unsigned long long)T * B * 4 * H * sizeof(float) >= (1ull << 31)) return false;      // 32-bit offsets into the saved gates
  return gru_wave_local(L, H) && H % 128 == 0 && H <= 512 && nrh * (H / 32) <= 32;
}
static size_t ks_lds_bytes(bool backward) { return (size_t)2 * 4 * (backward ? 8 : 16) * 1024 + (size_t)4 * WAVE_TILES * WTILE_F * sizeof(float); }
static int ks_arm(char* const* ring, int L, size_t slot_bytes, hipStream_t s) {
  KsArmArgs k;
  for (int l = 0; l < B2T_MAX_LAYERS; ++l) k.ring[l] = l < L ? ring[l] : nullptr;
  k.vec16 = (unsigned)((size_t)KS_D * slot_bytes / 16);
  hipLaunchKernelGGL(ks_arm_kernel, dim3(32, L), dim3(256), 0, s, k);
  return check_hip(hipGetLastError(), "gru_wave: arm");
}
template <int NPQ, bool DR> static int ks_launch_fwd(const WaveFwdArgs& a, hipStream_t s) {
  static bool at = false;
  if (!at) { const int rc = wave_lds_attr(gru_ks_fwd_kernel<NPQ, DR>, ks_lds_bytes(false)); if (rc) return rc; at = true; }
  hipLaunchKernelGGL((gru_ks_fwd_kernel<NPQ, DR>), dim3(256), dim3(256), ks_lds_bytes(false), s, a);
  return 0;
}
template <int NPQ, bool DR> static int ks_launch_bwd(const WaveBwdArgs& a, hipStream_t s) {
  static bool at = false;
  if (!at) { const int rc = wave_lds_attr(gru_ks_bwd_kernel<NPQ, DR>, ks_lds_bytes(true)); if (rc) return rc; at = true; }
  hipLaunchKernelGGL((gru_ks_bwd_kernel<NPQ, DR>), dim3(256), dim3(256), ks_lds_bytes(true), s, a);
  return 0;
}
template <int NPV, bool DR, bool LC, bool RG = false> static int wave_launch_bwd(const WaveBwdArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool at = false;
  if (!at) { const int rc = wave_lds_attr(gru_wave_bwd_kernel<NPV, DR, LC, RG>, RG ? gru_wave_lds_bytes_rgf_bwd(32 * NPV) : gru_wave_lds_bytes(32 * NPV)); if (rc) return rc; at = true; }
  hipLaunchKernelGGL((gru_wave_bwd_kernel<NPV, DR, LC, RG>), grid, dim3(256), lds, s, a);
  return 0;
}

int gru_wave_fwd(const WaveFwdArgs& a_in, hipStream_t s) {
  WaveFwdArgs a = a_in;
  a.flags = wave_flags();
  const char* why = nullptr;
  if (!gru_wave_ok(a.L, a.T, a.B, a.H, &why)) { set_error("gru_wave_fwd: unsupported shape L=%d T=%d B=%d H=%d (%s)", a.L, a.T, a.B, a.H, why); return 2; }
  const bool drop = a.drop_p > 0.f && a.L > 1, loc = gru_wave_local(a.L, a.H);
  const size_t lds = gru_wave_lds_bytes(a.H);
  a.tickets = a.cnt; a.cnt = a.cnt + 16;
  int rc = check_hip(hipMemsetAsync(a.tickets, 0, gru_wave_cnt_words_fwd(a.L, a.T, a.B) * sizeof(unsigned), s), "gru_wave_fwd: counters");
  if (rc) return rc;
  if (gru_wave_ks(a.L, a.T, a.B, a.H)) {
    rc = ks_arm(a.ring, a.L, (size_t)((a.B + 15) / 16) * (a.H / 32) * 1024, s);
    if (rc) return rc;
    switch (a.H / 128) {
      case 1: rc = drop ? ks_launch_fwd<1, true>(a, s) : ks_launch_fwd<1, false>(a, s); break;
      case 2: rc = drop ? ks_launch_fwd<2, true>(a, s) : ks_launch_fwd<2, false>(a, s); break;
      case 3: rc = drop ? ks_launch_fwd<3, true>(a, s) : ks_launch_fwd<3, false>(a, s); break;
      default: rc = drop ? ks_launch_fwd<4, true>(a, s) : ks_launch_fwd<4, false>(a, s); break;
    }
    if (rc) return rc;
    return check_hip(hipGetLastError(), "gru_wave_fwd");
  }
  const bool rgf = gru_wave_rgf(a.H);
  const int ngrp_ = (a.B + 15) / 16;
  const dim3 grid(loc ? 256 : (rgf ? a.L * ngrp_ * (a.H / 64) : a.L * (a.H / 16)));
#define B2T_WAVE_FWD(NPV)                                                                                              \
  do {                                                                                                                 \
    if (rgf && loc) rc = drop ? wave_launch_fwd<NPV, true, true, true>(a, grid, lds, s) : wave_launch_fwd<NPV, false, true, true>(a, grid, lds, s); \
    else if (rgf) rc = drop ? wave_launch_fwd<NPV, true, false, true>(a, grid, lds, s) : wave_launch_fwd<NPV, false, false, true>(a, grid, lds, s); \
    else if (loc) rc = drop ? wave_launch_fwd<NPV, true, true>(a, grid, lds, s) : wave_launch_fwd<NPV, false, true>(a, grid, lds, s);   \
    else rc = drop ? wave_launch_fwd<NPV, true, false>(a, grid, lds, s) : wave_launch_fwd<NPV, false, false>(a, grid, lds, s);     \
  } while (0)
  if (a.H <= 128) B2T_WAVE_FWD(4);
  else if (a.H <= 256) B2T_WAVE_FWD(8);
  else if (a.H <= 512) {
    if (loc) rc = drop ? wave_launch_fwd<16, true, true>(a, grid, lds, s) : wave_launch_fwd<16, false, true>(a, grid, lds, s);
    else rc = drop ? wave_launch_fwd<16, true, false>(a, grid, lds, s) : wave_launch_fwd<16, false, false>(a, grid, lds, s);
  } else rc = drop ? wave_launch_fwd<24, true, false>(a, grid, lds, s) : wave_launch_fwd<24, false, false>(a, grid, lds, s);
#undef B2T_WAVE_FWD
  if (rc) return rc;
  return check_hip(hipGetLastError(), "gru_wave_fwd");
}

int gru_wave_bwd_clear(unsigned* cnt, int L, int T, int B, hipStream_t s) {
  return check_hip(hipMemsetAsync(cnt, 0, gru_wave_cnt_words_bwd(L, T, B) * sizeof(unsigned), s), "gru_wave_bwd: counters");
}
// gate of a consumer of layer `layer`'s dG[t0 .. T) (see wave_gate_kernel); cnt = the block gru_wave_bwd counts in
int gru_wave_gate(unsigned* cnt, int layer, int t0, int T, int B, int H, unsigned* err, hipStream_t s) {
  const int ngrp = (B + 15) / 16, G = H / 16;
  hipLaunchKernelGGL(wave_gate_kernel, dim3(1), dim3(64), 0, s, cnt + 16 + layer * ngrp, ngrp, (unsigned)G * (unsigned)(T - t0 + 1), err);
  return check_hip(hipGetLastError(), "gru_wave_gate");
}

int gru_wave_bwd(const WaveBwdArgs& a_in, hipStream_t s) {
  WaveBwdArgs a = a_in;
  a.flags = wave_flags() | (a_in.flags & 4);     // bit 2: write dG through (consumers inside the sweep's lifetime: gated GEMMs)
  if (a_in.flags & 8) {      // bit 3 (K-split form only): w_hh_t / w_ih_t point at the UNtransposed matrices
    if (!gru_wave_ks(a.L, a.T, a.B, a.H)) { set_error("gru_wave_bwd: untransposed weights need the K-split form"); return 2; }
    a.flags |= 8;
  }
  const char* why = nullptr;
  if (!gru_wave_ok(a.L, a.T, a.B, a.H, &why)) { set_error("gru_wave_bwd: unsupported shape L=%d T=%d B=%d H=%d (%s)", a.L, a.T, a.B, a.H, why); return 2; }
  const bool drop = a.drop_p > 0.f && a.L > 1, loc = gru_wave_local(a.L, a.H);
  const size_t lds = gru_wave_lds_bytes(a.H);
  int rc = 0;
  if (!(a_in.flags & 2)) { rc = gru_wave_bwd_clear(a.cnt, a.L, a.T, a.B, s); if (rc) return rc; }   // (bit 1: the caller cleared them, gated consumers are already waiting)
  a.tickets = a.cnt; a.prog = a_in.prog ? a.cnt + 16 : nullptr; a.cnt = a.cnt + 16 + 64;
  if (gru_wave_ks(a.L, a.T, a.B, a.H)) {
    rc = ks_arm(a.ring, a.L, (size_t)((a.B + 15) / 16) * 4 * (a.H / 32) * 1024, s);
    if (rc) return rc;
    switch (a.H / 128) {
      case 1: rc = drop ? ks_launch_bwd<1, true>(a, s) : ks_launch_bwd<1, false>(a, s); break;
      case 2: rc = drop ? ks_launch_bwd<2, true>(a, s) : ks_launch_bwd<2, false>(a, s); break;
      case 3: rc = drop ? ks_launch_bwd<3, true>(a, s) : ks_launch_bwd<3, false>(a, s); break;
      default: rc = drop ? ks_launch_bwd<4, true>(a, s) : ks_launch_bwd<4, false>(a, s); break;
    }
    if (rc) return rc;
    return check_hip(hipGetLastError(), "gru_wave_bwd");
  }
  const bool rgf = gru_wave_rgf(a.H);
  const int ngrp_ = (a.B + 15) / 16;
  const dim3 grid(loc ? 256 : (rgf ? a.L * ngrp_ * (a.H / 64) : a.L * (a.H / 16)));
  const size_t lds_r = gru_wave_lds_bytes_rgf_bwd(a.H);
#define B2T_WAVE_BWD(NPV)                                                                                              \
  do {                                                                                                                 \
    if (rgf && loc) rc = drop ? wave_launch_bwd<NPV, true, true, true>(a, grid, lds_r, s) : wave_launch_bwd<NPV, false, true, true>(a, grid, lds_r, s); \
    else if (rgf) rc = drop ? wave_launch_bwd<NPV, true, false, true>(a, grid, lds_r, s) : wave_launch_bwd<NPV, false, false, true>(a, grid, lds_r, s); \
    else if (loc) rc = drop ? wave_launch_bwd<NPV, true, true>(a, grid, lds, s) : wave_launch_bwd<NPV, false, true>(a, grid, lds, s);   \
    else rc = drop ? wave_launch_bwd<NPV, true, false>(a, grid, lds, s) : wave_launch_bwd<NPV, false, false>(a, grid, lds, s);     \
  } while (0)
  if (a.H <= 128) B2T_WAVE_BWD(4);
  else if (a.H <= 256) B2T_WAVE_BWD(8);
  else if (a.H <= 512) {
    if (loc) rc = drop ? wave_launch_bwd<16, true, true>(a, grid, lds, s) : wave_launch_bwd<16, false, true>(a, grid, lds, s);
    else rc = drop ? wave_launch_bwd<16, true, false>(a, grid, lds, s) : wave_launch_bwd<16, false, false>(a, grid, lds, s);
  } else rc = drop ? wave_launch_bwd<24, true, false>(a, grid, lds, s) : wave_launch_bwd<24, false, false>(a, grid, lds, s);
#undef B2T_WAVE_BWD
  if (rc) return rc;
  return check_hip(hipGetLastError(), "gru_wave_bwd");
}

}  // namespace b2t

using namespace b2t;

// ---- C ABI: the stack's sweeps as one launch per direction (include/b2t.h) --------------------------------------------------
static size_t wave_align(size_t v) { return (v + 255) / 256 * 256; }

// 0: not held; 1: the 16-unit form; 2: the K-split form (gru_wave_ks.h: this device, this placement, B2T_WAVE_KS) -- callers use the
// difference to decide whether the BACKWARD pass goes on the wavefront too (it pays in the K-split form only: NOTES.md R6.2)
extern "C" int b2t_gru_wave_supported(int L, int T, int B, int H) { return !gru_wave_ok(L, T, B, H, nullptr) ? 0 : gru_wave_ks(L, T, B, H) ? 2 : 1; }

extern "C" size_t b2t_gru_wave_ws_bytes(int L, int T, int B, int H, int backward, int dropout) {
  if (L < 1 || L > B2T_MAX_LAYERS || T < 1 || B < 1 || H < 16) return 0;
  const size_t cnt = wave_align((backward ? gru_wave_cnt_words_bwd(L, T, B) : gru_wave_cnt_words_fwd(L, T, B)) * sizeof(unsigned) + 64);
  const size_t ring = wave_align(backward ? gru_wave_ring_bytes_bwd(T, B, H) : gru_wave_ring_bytes_fwd(T, B, H));
  (void)dropout;
  return cnt + (size_t)L * ring * 2;     // the own-recurrence ring + the copy the neighbouring layer reads (dropout / the local form)
}

// carve(ws): [counters][ring per layer][second ring per layer]
static void wave_carve(char* ws, int L, int T, int B, int H, bool backward, unsigned*& cnt, char** ring, char** ring2) {
  cnt = reinterpret_cast<unsigned*>(ws);
  const size_t cb = wave_align((backward ? gru_wave_cnt_words_bwd(L, T, B) : gru_wave_cnt_words_fwd(L, T, B)) * sizeof(unsigned) + 64);
  const size_t rb = wave_align(backward ? gru_wave_ring_bytes_bwd(T, B, H) : gru_wave_ring_bytes_fwd(T, B, H));
  char* p = ws + cb;
  for (int l = 0; l < L; ++l) { ring[l] = p; p += rb; }
  for (int l = 0; l < L; ++l) { ring2[l] = p; p += rb; }
}

extern "C" int b2t_gru_wave_fwd_f32(const b2t_wave_t* d, void* ws, unsigned* err_word, void* stream) {
  B2T_REQUIRE(d && ws && err_word, "gru_wave_fwd: null argument");
  const int L = d->L, T = d->T, B = d->B, H = d->H;
  B2T_REQUIRE(gru_wave_ok(L, T, B, H, nullptr), "gru_wave_fwd: unsupported shape L=%d T=%d B=%d H=%d", L, T, B, H);
  B2T_REQUIRE(d->gi0 && d->drop_p >= 0.f && d->drop_p < 1.f, "gru_wave_fwd: gi0 / dropout");
  WaveFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.L = L; a.T = T; a.B = B; a.H = H; a.gi0 = d->gi0;
  const bool drop = d->drop_p > 0.f && L > 1;
  wave_carve(reinterpret_cast<char*>(ws), L, T, B, H, false, a.cnt, a.ring, a.ringd);
  a.err = err_word;
#ifdef B2T_WAVE_TIMING
  a.timing = err_word + 16;      // (timing build: the caller's error buffer holds 16 + 8 L words)
#endif
  for (int l = 0; l < L; ++l) {
    B2T_REQUIRE(d->w_hh[l] && d->b_hh[l] && d->h_init[l] && d->out[l] && (l == 0 || (d->w_ih[l] && d->b_ih[l])), "gru_wave_fwd: null tensor of layer %d", l);
    B2T_REQUIRE(!drop || l + 1 == L || d->outd[l], "gru_wave_fwd: dropout needs outd[%d]", l);
    a.w_hh[l] = d->w_hh[l]; a.b_hh[l] = d->b_hh[l]; a.w_ih[l] = d->w_ih[l]; a.b_ih[l] = d->b_ih[l];
    a.h_init[l] = d->h_init[l]; a.out[l] = d->out[l]; a.outd[l] = d->outd[l]; a.reserve[l] = d->reserve[l];
    a.seed[l] = d->seed[l];
  }
  a.drop_p = drop ? d->drop_p : 0.f; a.drop_scale = drop ? 1.0f / (1.0f - d->drop_p) : 1.f; a.elem0 = d->elem0;
  return gru_wave_fwd(a, as_stream(stream));
}

extern "C" int b2t_gru_wave_bwd_f32(const b2t_wave_t* d, void* ws, unsigned* err_word, void* stream) {
  B2T_REQUIRE(d && ws && err_word, "gru_wave_bwd: null argument");
  const int L = d->L, T = d->T, B = d->B, H = d->H;
  B2T_REQUIRE(gru_wave_ok(L, T, B, H, nullptr), "gru_wave_bwd: unsupported shape L=%d T=%d B=%d H=%d", L, T, B, H);
  B2T_REQUIRE(d->dY_top && d->dh_init && d->drop_p >= 0.f && d->drop_p < 1.f, "gru_wave_bwd: dY_top / dh_init / dropout");
  WaveBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.L = L; a.T = T; a.B = B; a.H = H; a.dY_top = d->dY_top;
  for (int l = 0; l < L; ++l) { a.dh_last[l] = d->dh_last ? d->dh_last + (size_t)l * B * H : nullptr; a.dh_init[l] = d->dh_init + (size_t)l * B * H; }
  const bool drop = d->drop_p > 0.f && L > 1;
  wave_carve(reinterpret_cast<char*>(ws), L, T, B, H, true, a.cnt, a.ring, a.ringx);
  a.err = err_word;
#ifdef B2T_WAVE_TIMING
  a.timing = err_word + 16 + 8 * B2T_MAX_LAYERS;
#endif
  for (int l = 0; l < L; ++l) {
    B2T_REQUIRE(d->w_hh_t[l] && d->h_init[l] && d->out[l] && d->reserve[l] && d->dG[l] && (l == 0 || d->w_ih_t[l]), "gru_wave_bwd: null tensor of layer %d", l);
    a.w_hh_t[l] = d->w_hh_t[l]; a.w_ih_t[l] = d->w_ih_t[l]; a.h_init[l] = d->h_init[l]; a.out[l] = d->out[l];
    a.reserve[l] = d->reserve[l]; a.dG[l] = d->dG[l]; a.seed[l] = d->seed[l];
  }
  a.drop_p = drop ? d->drop_p : 0.f; a.drop_scale = drop ? 1.0f / (1.0f - d->drop_p) : 1.f; a.elem0 = d->elem0;
  return gru_wave_bwd(a, as_stream(stream));
}
