// gru_wave_ks.h — the K-SPLIT form of the layer wavefront (included by gru_wave.hip between its device helpers and its host side;
// same hand-off data, same results' meaning, same launch geometry class as the local form: layer l on XCD l).
//
// Why.  In the 16-unit ("fat") form every wave of a workgroup contracts the WHOLE K range for its own row group, so a CU pulls
// all four row groups' fragments through its vector cache every step: 128 KB forward, 384 KB backward per step and CU at H = 512
// against ~55 B per clock and CU from the L2 -- 2.3 k / 7 k cycles of a ~10 k / ~21 k cycle step, with the counter hand-off's four
// dependent L2 round trips (store -> acknowledge -> atomic -> poll -> load) on top (NOTES.md R6.2).  Here
//   * a workgroup = (layer, TWO row groups, TWO 16-unit slices = one fragment pair) and its four waves split K: wave w contracts
//     the pairs [NPQ w, NPQ (w + 1)) of both row groups for all of the workgroup's output columns, with a QUARTER of the weight
//     slices in its registers (96 registers per matrix at H = 512: W_hh AND W_ih both fit, nothing in LDS); the four partial tiles
//     are summed through LDS (one workgroup barrier per product); then wave w = (row group w >> 1, slice w & 1) runs the gates of
//     its own 16 x 16 tile exactly as before.  Per CU and step: 64 KB forward, 192 KB backward;
//   * the own-recurrence hand-off is DATA-POLLED: the ring slot is armed with a sentinel (an all-ones dword -- the producers clamp
//     their bf16 pairs below it, which only touches a NaN payload), consumers load the fragments they need with L1-bypassing
//     (sc1) loads straight into the MFMA's operand registers and retry while any dword is still the sentinel.  No acknowledge
//     wait, no counter, no separate poll: one store -> visible -> load per step.  The ring is KS_D slots deep and lives in the
//     XCD's L2; a producer re-arms the slot two publishes ahead (the retry loop's vmcnt(0) orders that store before the data of
//     the next publish; depth >= 4 makes the re-armed slot dead: everything up to two publishes back has been consumed by
//     every peer that published the previous one);
//   * the neighbouring layer (another XCD) still reads a second, written-through ring behind agent-scope counters.
// Serves H % 128 == 0, H <= 512 under the local placement; other shapes keep the 16-unit form.
#pragma once
#include <type_traits>

constexpr int KS_D = 8;                          // own-ring depth (slots)
constexpr unsigned KS_MAXDATA = 0xFFFEFFFFu;     // data dwords are clamped to this; anything above is "not yet written"

__device__ __forceinline__ u32x4 ks_clamp(u32x4 v) {
  return u32x4{v.x < KS_MAXDATA ? v.x : KS_MAXDATA, v.y < KS_MAXDATA ? v.y : KS_MAXDATA, v.z < KS_MAXDATA ? v.z : KS_MAXDATA, v.w < KS_MAXDATA ? v.w : KS_MAXDATA};
}
__device__ __forceinline__ unsigned ks_max(unsigned m, u32x4 v) {
  const unsigned a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
  const unsigned c = a > b ? a : b;
  return m > c ? m : c;
}
// ring accesses with the wave-uniform part of the offset (slot, array) in the instruction's SCALAR offset: the per-lane part is then
// loop-invariant and the fragment index folds into the immediate -- one address register per stream instead of one per load
// (added to the per-lane offset, the slot base made every load's address its own VGPR: 56 of them in the backward kernel)
template <int AUX>
__device__ __forceinline__ u32x4 ks_load(const char* base_uniform, unsigned voff, unsigned soff) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, AUX);
}
template <int AUX>
__device__ __forceinline__ void ks_store(char* base_uniform, unsigned voff, unsigned soff, u32x4 v) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, AUX);
}
// one float through a buffer descriptor: per-lane byte offset (a lane whose row does not exist passes KS_DEAD: out of range, reads 0 --
// no branch around the load) + a wave-uniform scalar offset.  Tensors below 2 GB (gru_wave_ks checks).
constexpr unsigned KS_DEAD = 0x80000000u;
template <int AUX>
__device__ __forceinline__ float ks_ldf(const float* base_uniform, unsigned voff, unsigned soff) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, AUX));
}
__device__ __forceinline__ u32x4 ks_sentinel() { return u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; }

// both row groups' counters of the neighbouring layer's ring slot: even lanes look at the first, odd lanes at the second
__device__ __forceinline__ void ks_wait2(const unsigned* c0, const unsigned* c1, unsigned target, unsigned* err, int lane) {
  const unsigned* p = (lane & 1) ? c1 : c0;
  unsigned spins = 0;
  for (;;) {
    const unsigned v = __hip_atomic_load(p, RLX_AGENT);
    if (__all(v >= target)) break;
    if ((++spins & 255u) == 0u) {
      if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
      if (spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Workgroup barrier for the partial tiles: LDS writes done (lgkmcnt), NOT the vector memory queue -- __syncthreads() also waits for
// vmcnt(0), i.e. for every fp32 store (written through: a fabric round trip) and every HBM prefetch in flight: 1-2 k cycles per
// barrier, two barriers per step (NOTES.md R6.2)
__device__ __forceinline__ void ks_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum of the four waves' partial tiles `tile` (of NT per wave) for this lane: C layout in, C layout out
template <int NT>
__device__ __forceinline__ f32x4 ks_reduce(const float4* part, int tile, int lane) {
  float4 s = part[(0 * NT + tile) * 64 + lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 v = part[(w * NT + tile) * 64 + lane];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  return f32x4{s.x, s.y, s.z, s.w};
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <int NPQ, bool DROP>      // NPQ = H / 128: fragment pairs per K quarter
__global__ __launch_bounds__(256, 1) void gru_ks_fwd_kernel(const WaveFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wave_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int H = a.H, B = a.B, T = a.T, G = H / 16, P = G / 2, ngrp = (B + 15) / 16, L = a.L, nrh = (ngrp + 1) / 2;
  int layer, k;
  if (!wave_role<true>(a.tickets, L, nrh * P, false, layer, k)) return;
  const int rh = k / P, ub = k % P;                       // row half (row groups 2 rh, 2 rh + 1), unit block (slices 2 ub, 2 ub + 1 = pair ub)
  const int rgl = wave >> 1, us = wave & 1;               // the tile this wave finishes
  const int rg = 2 * rh + rgl, slice = 2 * ub + us, u0 = slice * 16, unit = u0 + j, m0 = rg * 16;
  const bool rg_live = rg < ngrp;
  const int rgc0 = 2 * rh, rgc1 = (2 * rh + 1 < ngrp) ? 2 * rh + 1 : 2 * rh;     // the row groups whose fragments this workgroup contracts
  // partial tiles per wave: (row group 2) x (slice 2) x (gate 3) of the step's sums + (row group 2) x (slice 2) of gi_n.  ONE reduction
  // per step: the projection of step t + 1 (made at the end of step t) stays UNREDUCED in registers; its r and z tiles seed the next
  // product's accumulators (the gates only need gi + gh), its n tile (tanh(gi_n + r gh_n) needs the two apart) rides along as a
  // fourth tile.  Two buffers by step parity (nothing else separates one step's reads from the next step's writes).
  constexpr int NT = 16;
  float4* part_p = reinterpret_cast<float4*>(wave_lds);
  float4* part_q = part_p + 4 * NT * 64;
  float* tiles = reinterpret_cast<float*>(wave_lds + (size_t)2 * 4 * NT * 1024) + wave * (WAVE_TILES * WTILE_F);

  // this wave's quarter of the K range of the workgroup's W_hh and W_ih slices, as B fragments in accumulation registers.
  // The PROJECTION walks its fragments in an order rotated by the unit block: the workgroups of a row half all read the SAME
  // fragments of the neighbouring layer's ring, from another XCD -- in the same order they asked for the same line at the same
  // moment, every one of them sat out the fabric latency on every line (24 loads took 7-12 k cycles to issue in the backward
  // pass: the CU's miss queue, ~64 lines, times ~2 us); rotated, a line's first reader fetches it and the others hit the L2.
  const int rot = ub % NPQ;
  bf16x8 w[2][3][NPQ], w2[2][3][NPQ];
  {
    const float* whh = a.w_hh[layer];
    const float* wih = layer > 0 ? a.w_ih[layer] : a.w_hh[layer];
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int k0 = 32 * (NPQ * wave + i) + 8 * q;
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const long long row = (long long)g * H + 16 * (2 * ub + u2) + j;
          const float* s1 = whh + row * H + k0;
          const float* s2 = wih + row * H + 32 * (NPQ * wave + (i + rot) % NPQ) + 8 * q;      // (slot i of the projection = pair (i + rot) % NPQ)
          w[u2][g][i] = cvt8(ld4(s1), ld4(s1 + 4));
          w2[u2][g][i] = __builtin_bit_cast(bf16x8, masked8(ld4(s2), ld4(s2 + 4), layer > 0));
          asm volatile("" : "+a"(w[u2][g][i]));
          asm volatile("" : "+a"(w2[u2][g][i]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __builtin_amdgcn_s_setprio(3);

  unsigned* err = a.err;
  const size_t cstride = (size_t)(T + 1);
  unsigned* cnt_x = a.cnt + ((size_t)(layer * 2 + 1) * ngrp + (rg_live ? rg : 0)) * cstride;
  const bool feeds = layer + 1 < L, dropping = DROP && feeds;
  const unsigned* cnt_in0 = layer > 0 ? a.cnt + ((size_t)((layer - 1) * 2 + 1) * ngrp + rgc0) * cstride : nullptr;
  const unsigned* cnt_in1 = layer > 0 ? a.cnt + ((size_t)((layer - 1) * 2 + 1) * ngrp + rgc1) * cstride : nullptr;
  char* ring = a.ring[layer];
  char* ringx = a.ringd[layer];
  const char* ring_in = layer > 0 ? a.ringd[layer - 1] : nullptr;
  const unsigned slot_bytes = (unsigned)ngrp * (unsigned)P * 1024u, rg_off = (unsigned)rg * (unsigned)P * 1024u;
  const unsigned my_frag = (unsigned)ub * 1024u + (unsigned)us * 512u + (unsigned)(lane & 31) * 16u;
  const unsigned in_off0 = (unsigned)rgc0 * (unsigned)P * 1024u + (unsigned)(NPQ * wave) * 1024u + (unsigned)lane * 16u;
  const unsigned in_off1 = (unsigned)rgc1 * (unsigned)P * 1024u + (unsigned)(NPQ * wave) * 1024u + (unsigned)lane * 16u;
  const float bhr = a.b_hh[layer][unit], bhz = a.b_hh[layer][H + unit], bhn = a.b_hh[layer][2 * H + unit];
  float bi[3] = {0.f, 0.f, 0.f};
  if (layer > 0) { bi[0] = a.b_ih[layer][unit]; bi[1] = a.b_ih[layer][H + unit]; bi[2] = a.b_ih[layer][2 * H + unit]; }
  bool live[4];
  f32x4 hp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + 4 * q + i;
    live[i] = rg_live && row < B;
    hp[i] = live[i] ? a.h_init[layer][(long long)row * H + unit] : 0.f;
  }
  // index 0 of the ring = the initial state (the slots were armed by the host's arm kernel)
  if (rg_live) {
    tile_put(tiles, hp, j, q);
    const u32x4 f = ks_clamp(tile_frag(tiles, lane));
    if (lane < 32) store_u4<0>(ring, rg_off + my_frag, f);
  }
  int pending_x = -1;

#ifdef B2T_WAVE_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
  // layers >= 1: pj[r][u2][g] = this wave's K quarter of gi for the NEXT product (unreduced)
  f32x4 pj[2][2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
      for (int g = 0; g < 3; ++g) pj[r][u2][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  // acc += A fragments v[2][NPQ] x B fragments wt (this wave's K quarter, all 12 tiles of the workgroup)
  auto mma = [&](const u32x4 (&v)[2][NPQ], const bf16x8 (&wt)[2][3][NPQ], f32x4 (&acc)[2][2][3]) {
#pragma unroll
    for (int i = 0; i < NPQ; ++i)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, v[r][i]);
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[r][u2][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wt[u2][g][i], acc[r][u2][g], 0, 0, 0);
      }
  };
  // gi of step t from layer - 1's (dropped) h_t = slot t + 1 of its written-through ring -> pj (no LDS, no barrier).  Split-phase: the
  // slot's counters are REQUESTED two steps ahead and looked at one step ahead (right after the own poll, so that neither request sits
  // in front of a poll's loads for long: vector memory returns in order); if they were complete, the fragments are requested then and
  // there and are in registers when the projection starts.  Otherwise (the layer below is less than two steps ahead) the blocking path.
  unsigned csnap = 0u;
  auto project = [&](int t, bool loaded, u32x4 (&xq)[2][NPQ]) {
    if (!loaded) {
      ks_wait2(cnt_in0 + (t + 1), cnt_in1 + (t + 1), (unsigned)G, err, lane);
      const unsigned base = (unsigned)(t + 1) * slot_bytes;
#pragma unroll
      for (int i = 0; i < NPQ; ++i) {
        xq[0][i] = ks_load<0>(ring_in, in_off0, base + (unsigned)((i + rot) % NPQ) * 1024u);
        xq[1][i] = ks_load<0>(ring_in, in_off1, base + (unsigned)((i + rot) % NPQ) * 1024u);
      }
    }
    WSTAMP(5)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int g = 0; g < 3; ++g) pj[r][u2][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma(xq, w2, pj);
#ifdef B2T_WAVE_TIMING
    asm volatile("s_nop 0" :: "v"(pj[0][0][0][0]), "v"(pj[1][1][2][0]));
#endif
    WSTAMP(6)
  };
  // after the own poll of step t: fragments of slot t + 2 if last step's look at its counters found them complete; request the counters
  // of slot t + 3.  BOTH unconditionally (the fragments from a harmless address when the counters were not complete, the counter index
  // clamped): behind a branch the compiler merged the loaded registers with the not-loaded path through copies -- and waited for the
  // loads (fabric: ~2 k cycles) right where they were issued (ISA of the timing build, NOTES.md R6.2b)
  auto prefetch_in = [&](int t, u32x4 (&xq)[2][NPQ]) -> bool {
    const bool ready = t + 1 < T && t > 0 && __all(csnap >= (unsigned)G);
    const unsigned base = (unsigned)(ready ? t + 2 : 0) * slot_bytes;      // (slot 0: the initial state, complete since project(0))
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      xq[0][i] = ks_load<0>(ring_in, in_off0, base + (unsigned)((i + rot) % NPQ) * 1024u);
      xq[1][i] = ks_load<0>(ring_in, in_off1, base + (unsigned)((i + rot) % NPQ) * 1024u);
    }
    csnap = __hip_atomic_load(((lane & 1) ? cnt_in1 : cnt_in0) + (t + 3 <= T ? t + 3 : T), RLX_AGENT);
    return ready;
  };
  unsigned vo_gi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) vo_gi[i] = live[i] ? ((unsigned)(m0 + 4 * q + i) * 3u * (unsigned)H + (unsigned)unit) * 4u : KS_DEAD;
  const unsigned gi_step = (unsigned)B * 3u * (unsigned)H * 4u;
  auto load_gi0 = [&](int t, f32x4 (&dst)[3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 3; ++g) dst[g][i] = ks_ldf<2>(a.gi0, vo_gi[i], (unsigned)t * gi_step + (unsigned)g * (unsigned)H * 4u);
  };
  f32x4 giA[3], giB[3];
  if (layer > 0) { u32x4 xq[2][NPQ]; project(0, false, xq); } else load_gi0(0, giA);

  // the step loop, instantiated twice (layer 0: gi from memory, no projection; layers >= 1): in one body the two paths shared
  // registers through copies the compiler waited on (NOTES.md R6.2b)
  // (layer 0: gi of a step comes from memory, requested one step ahead into the OTHER of two buffers -- the loop runs two steps per
  // iteration so that no copy exists: a copy at the end of the step was scheduled right behind the loads and waited for HBM there)
  auto run = [&](auto l0c) {
  constexpr bool L0 = decltype(l0c)::value;
  auto step = [&](int t, f32x4 (&gcur)[3], f32x4 (&gnxt)[3]) {
    // h_{t-1} of both row groups, this wave's K quarter: load until no dword is the sentinel
    u32x4 v[2][NPQ];
    {
      const unsigned base = (unsigned)(t % KS_D) * slot_bytes;
      unsigned spins = 0;
#ifdef KS_WITNESS
      // a look at ONE fragment until it is there (an eighth of the bytes per look: 128 waves of an XCD poll at once), then all of them
      for (;;) {
        const u32x4 wv = ks_load<16>(ring, in_off1 + (unsigned)(NPQ - 1) * 1024u, base);
        if (!__any(ks_max(0u, wv) > KS_MAXDATA)) break;
        if ((++spins & 255u) == 0u) {
          if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
          if (spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
        }
      }
#endif
      for (;;) {
#pragma unroll
        for (int i = 0; i < NPQ; ++i) {
          v[0][i] = ks_load<16>(ring, in_off0 + (unsigned)i * 1024u, base);
          v[1][i] = ks_load<16>(ring, in_off1 + (unsigned)i * 1024u, base);
        }
        unsigned m = 0u;
#pragma unroll
        for (int i = 0; i < NPQ; ++i) { m = ks_max(m, v[0][i]); m = ks_max(m, v[1][i]); }
        if (!__any(m > KS_MAXDATA)) break;
        if ((++spins & 63u) == 0u) {
          if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
          if (spins > (SPIN_LIMIT >> 2)) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
        }
      }
    }
    wave_drain();      // (already empty: the loads above were waited for -- and with them every earlier store of this wave)
    if (pending_x >= 0) { wave_bump<false>(cnt_x + pending_x, lane); pending_x = -1; }
    WSTAMP(0)   // the peers' h_{t-1} is here
    u32x4 xq[2][NPQ];
    bool in_loaded = false;
    if (L0) { if (t + 1 < T) load_gi0(t + 1, gnxt); }     // (lands during the product)
    else in_loaded = prefetch_in(t, xq);
    // the step's sums: accumulators seeded with the projection's r and z tiles (layers >= 1), product on top, ONE reduction
    f32x4 gh[3], gin_;
    {
      f32x4 acc[2][2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
          acc[r][u2][0] = !L0 ? pj[r][u2][0] : f32x4{0.f, 0.f, 0.f, 0.f};
          acc[r][u2][1] = !L0 ? pj[r][u2][1] : f32x4{0.f, 0.f, 0.f, 0.f};
          acc[r][u2][2] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      mma(v, w, acc);
#ifdef B2T_WAVE_TIMING
      asm volatile("s_nop 0" :: "v"(acc[0][0][0][0]), "v"(acc[1][1][2][0]));
      WSTAMP(3)   // (timing build: the product's MFMAs alone)
#endif
      float4* part = (t & 1) ? part_q : part_p;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const f32x4 c = acc[r][u2][g];
            part[(wave * NT + (r * 2 + u2) * 3 + g) * 64 + lane] = float4{c[0], c[1], c[2], c[3]};
          }
          if (!L0) {
            const f32x4 c = pj[r][u2][2];
            part[(wave * NT + 12 + r * 2 + u2) * 64 + lane] = float4{c[0], c[1], c[2], c[3]};
          }
        }
      ks_barrier();
#ifdef B2T_WAVE_TIMING
      WSTAMP(7)   // (timing build: partial tiles written + the barrier)
#endif
#pragma unroll
      for (int g = 0; g < 3; ++g) gh[g] = ks_reduce<NT>(part, (rgl * 2 + us) * 3 + g, lane);
      gin_ = !L0 ? ks_reduce<NT>(part, 12 + rgl * 2 + us, lane) : gcur[2];
    }
#ifdef B2T_WAVE_TIMING
    asm volatile("s_nop 0" :: "v"(gh[0][0]), "v"(gh[1][0]), "v"(gh[2][0]), "v"(gin_[0]));
#endif
    WSTAMP(1)   // recurrent product + reduction
    if (rg_live) {
      f32x4 sr, sz, sn, sg, h;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (layers >= 1: gh[0], gh[1] already hold gi + gh of the r and z gates; layer 0: gi of memory is added here)
        const float ghn = gh[2][i] + bhn;
        const float r = fast_sigmoid((!L0 ? bi[0] : gcur[0][i]) + gh[0][i] + bhr);
        const float z = fast_sigmoid((!L0 ? bi[1] : gcur[1][i]) + gh[1][i] + bhz);
        const float nn = fast_tanh(gin_[i] + bi[2] + r * ghn);
        h[i] = (1.0f - z) * nn + z * hp[i];
        sr[i] = r; sz[i] = z; sn[i] = nn; sg[i] = ghn;
      }
      hp = h;
      // publish h_t = index t + 1; re-arm the slot of index t + 3
      tile_put(tiles, h, j, q);
      const u32x4 f = tile_frag(tiles, lane);
      if (lane < 32) {
        ks_store<0>(ring, rg_off + my_frag, (unsigned)((t + 1) % KS_D) * slot_bytes, ks_clamp(f));
        ks_store<0>(ring, rg_off + my_frag, (unsigned)((t + 3) % KS_D) * slot_bytes, ks_sentinel());
      }
      WSTAMP(2)   // gates + publish
      const int rrow = m0 + (lane & 15), kg = lane >> 4;
      const float4 hv = ld4(tiles + (lane & 15) * WTP + 4 * kg);
      if (feeds) {
        u32x4 fx = f;
        if (dropping) {
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
        if (lane < 32) ks_store<16>(ringx, rg_off + my_frag, (unsigned)(t + 1) * slot_bytes, fx);
        pending_x = t + 1;
      }
      if (rrow < B) *reinterpret_cast<float4*>(a.out[layer] + ((long long)t * B + rrow) * H + u0 + 4 * kg) = hv;
      if (a.reserve[layer]) {
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
      WSTAMP(4)   // neighbour's ring, fp32 stores
    }
    if (!L0) { if (t + 1 < T) project(t + 1, in_loaded, xq); }
  };
  for (int t = 0; t < T; t += 2) {
    step(t, giA, giB);
    if (t + 1 < T) step(t + 1, giB, giA);
  }
  };
  if (layer == 0) run(std::true_type{}); else run(std::false_type{});
  if (pending_x >= 0) { wave_drain(); wave_bump<false>(cnt_x + pending_x, lane); }
#ifdef B2T_WAVE_TIMING
  if (k == 0 && wave == 0 && lane == 0 && a.timing)
    for (int i = 0; i < 8; ++i) a.timing[layer * 8 + i] = (unsigned)(tacc[i] / (unsigned long long)T);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  Ring slot: the gate gradients of one step as fragments, 4 arrays (dr, dz, dn r, dn) x P pairs per row group.
// Index n of the own ring = step T - 1 - n.
// The gradient wrt the layer BELOW's outputs, dY[layer - 1]_t = dGi[layer]_t . W_ih[layer], is made HERE, by the layer that owns dGi
// (producer side): its fragments are the ones this layer polls for its own recurrence anyway (dr, dz shared; dn instead of dn r: one
// more array from the own ring, in the XCD's L2), so nothing but a 16 x 16 fp32 tile per wave and step crosses to the layer below --
// the consumer-side form read 96 KB per CU and step across XCDs at ~27 B per clock: 3.5-4.5 k cycles of a 13 k cycle step (R6.2b).
// ---------------------------------------------------------------------------------------------------------------------
template <int NPQ, bool DROP>
__global__ __launch_bounds__(256, 1) void gru_ks_bwd_kernel(const WaveBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wave_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int H = a.H, B = a.B, T = a.T, G = H / 16, P = G / 2, ngrp = (B + 15) / 16, L = a.L, nrh = (ngrp + 1) / 2;
  int layer, k;
  if (!wave_role<true>(a.tickets, L, nrh * P, true, layer, k)) return;
  const int rh = k / P, ub = k % P;
  const int rgl = wave >> 1, us = wave & 1;
  const int rg = 2 * rh + rgl, slice = 2 * ub + us, u0 = slice * 16, unit = u0 + j, m0 = rg * 16;
  const bool rg_live = rg < ngrp;
  const int rgc0 = 2 * rh, rgc1 = (2 * rh + 1 < ngrp) ? 2 * rh + 1 : 2 * rh;
  constexpr int NT = 8;                                   // partial tiles per wave: (row group 2) x (slice 2) of the recurrent product, then of the projection
  float4* part_p = reinterpret_cast<float4*>(wave_lds);
  float4* part_q = part_p + 4 * NT * 64;
  float* tiles = reinterpret_cast<float*>(wave_lds + (size_t)2 * 4 * NT * 1024) + wave * (WAVE_TILES * WTILE_F);
  const bool has_up = layer + 1 < L, feeds = layer > 0;

  // this wave's K quarter (pairs [NPQ w, NPQ (w + 1)) of each array) of the W_hh^T and W_ih[layer]^T column slices
  bf16x8 w[2][3][NPQ], w2[2][3][NPQ];
  {
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int k0 = 32 * (NPQ * wave + i) + 8 * q;
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2) {
        const long long col = 16 * (2 * ub + u2) + j;
        const float* t1 = a.w_hh_t[layer] + col * 3 * H;
        const float* t2 = (feeds ? a.w_ih_t[layer] : a.w_hh_t[layer]) + col * 3 * H;      // (layer 0 projects nothing: any valid matrix, masked to zero)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          float4 h0, h1, i0, i1;
          if (a.flags & 8) {
            // the matrices as the model holds them ([3H][H], flags bit 3): element (k, col) at k H + col -- eight strided loads per
            // fragment, once per launch, instead of a transpose kernel per layer in front of the sweep (and the queue hop behind it)
            const float* m1 = a.w_hh_t[layer] + ((long long)g * H + k0) * H + col;
            const float* m2 = (feeds ? a.w_ih_t[layer] : a.w_hh_t[layer]) + ((long long)g * H + k0) * H + col;
            h0 = float4{m1[0], m1[H], m1[2 * H], m1[3 * H]}; h1 = float4{m1[4 * H], m1[5 * H], m1[6 * H], m1[7 * H]};
            i0 = float4{m2[0], m2[H], m2[2 * H], m2[3 * H]}; i1 = float4{m2[4 * H], m2[5 * H], m2[6 * H], m2[7 * H]};
          } else {
            const float* s1 = t1 + (long long)g * H + k0;
            const float* s2 = t2 + (long long)g * H + k0;
            h0 = ld4(s1); h1 = ld4(s1 + 4); i0 = ld4(s2); i1 = ld4(s2 + 4);
          }
          w[u2][g][i] = cvt8(h0, h1);
          w2[u2][g][i] = __builtin_bit_cast(bf16x8, masked8(i0, i1, feeds));
          asm volatile("" : "+a"(w[u2][g][i]));
          asm volatile("" : "+a"(w2[u2][g][i]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __builtin_amdgcn_s_setprio(3);

  unsigned* err = a.err;
  const size_t cstride = (size_t)T;
  unsigned* cnt_x = a.cnt + ((size_t)(layer * 2 + 1) * ngrp + (rg_live ? rg : 0)) * cstride;      // dY[layer - 1] tiles of (row group, step) published
  const unsigned* cnt_up = has_up ? a.cnt + ((size_t)((layer + 1) * 2 + 1) * ngrp + (rg_live ? rg : 0)) * cstride : nullptr;
  char* ring = a.ring[layer];
  char* dyr = a.ringx[layer];                                    // dY[layer - 1]: [T][row group][slice][64 lanes x 16 B] (C layout of a 16 x 16 tile), written through
  const char* dyr_up = has_up ? a.ringx[layer + 1] : nullptr;
  const unsigned arr_bytes = (unsigned)P * 1024u, rg_bytes = 4u * arr_bytes, slot_bytes = (unsigned)ngrp * rg_bytes, rg_off = (unsigned)rg * rg_bytes;
  const unsigned my_frag = (unsigned)ub * 1024u + (unsigned)us * 512u + (unsigned)(lane & 31) * 16u;
  const unsigned in_off0 = (unsigned)rgc0 * rg_bytes + (unsigned)(NPQ * wave) * 1024u + (unsigned)lane * 16u;
  const unsigned in_off1 = (unsigned)rgc1 * rg_bytes + (unsigned)(NPQ * wave) * 1024u + (unsigned)lane * 16u;
  const unsigned dy_slot = (unsigned)ngrp * (unsigned)G * 1024u, dy_tile = ((unsigned)rg * (unsigned)G + (unsigned)slice) * 1024u + (unsigned)lane * 16u;
  bool live[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) live[i] = rg_live && m0 + 4 * q + i < B;
  const int rrow = m0 + (lane & 15), kg = lane >> 4;
  int pending_x = -1;
  unsigned* prog = (a.prog && rg_live) ? a.prog + layer * ngrp + rg : nullptr;

#ifdef B2T_WAVE_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
  // the step's elementwise operands (saved gates, h_{t-1}, dY of the step: the top layer's from memory, the others' tile from the ring
  // of the layer above) come from memory: requested one step ahead, right after the poll (behind it they would sit in front of the
  // NEXT poll's loads: vector memory returns in order).  dY: only when the counters of its slot were seen complete (requested one
  // step before that); `ok` says so -- otherwise the step waits for them and loads the tile itself.
  struct Elem { f32x4 r, z, nv, ghn, hprev, dyt; bool ok; };
  unsigned vo_res[4], vo_out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = (unsigned)(m0 + 4 * q + i);
    vo_res[i] = live[i] ? (row * 4u * (unsigned)H + (unsigned)unit) * 4u : KS_DEAD;
    vo_out[i] = live[i] ? (row * (unsigned)H + (unsigned)unit) * 4u : KS_DEAD;
  }
  const unsigned res_step = (unsigned)B * 4u * (unsigned)H * 4u, out_step = (unsigned)B * (unsigned)H * 4u, hb = (unsigned)H * 4u;
  unsigned csnap = 0u;
  auto load_dy = [&](int t) -> f32x4 { return __builtin_bit_cast(f32x4, ks_load<0>(dyr_up, dy_tile, (unsigned)t * dy_slot)); };
  auto fetch = [&](int t, Elem& e, bool up_ready) {
    e.ok = up_ready;
    if (t < 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) e.r[i] = e.z[i] = e.nv[i] = e.ghn[i] = e.hprev[i] = e.dyt[i] = 0.f;
      return;
    }
    const unsigned so = (unsigned)t * res_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      e.r[i] = ks_ldf<2>(a.reserve[layer], vo_res[i], so); e.z[i] = ks_ldf<2>(a.reserve[layer], vo_res[i], so + hb);
      e.nv[i] = ks_ldf<2>(a.reserve[layer], vo_res[i], so + 2u * hb); e.ghn[i] = ks_ldf<2>(a.reserve[layer], vo_res[i], so + 3u * hb);
      e.hprev[i] = t > 0 ? ks_ldf<0>(a.out[layer], vo_out[i], (unsigned)(t - 1) * out_step) : ks_ldf<0>(a.h_init[layer], vo_out[i], 0u);
      if (!has_up) e.dyt[i] = ks_ldf<2>(a.dY_top, vo_out[i], (unsigned)t * out_step);
    }
    if (has_up) e.dyt = load_dy(up_ready ? t : T - 1);      // (unconditional -- slot T - 1 is complete since the start: no merge of loaded / not loaded registers)
  };
  // Two buffers used alternately (two steps per loop iteration, no copy), the next step's operands requested behind this step's MFMAs.
  // With dropout (16 more live registers around the Philox draws: the allocator spilled) ONE buffer, requested at the END of the step,
  // when this step's operands are dead -- those HBM loads then sit in front of the next poll's looks (+0.6 us per step alone).
  constexpr bool DB = !DROP;
  Elem ea, eb;
  if (has_up) wave_wait<false>(cnt_up + (T - 1), (unsigned)G, err, lane);
  fetch(T - 1, ea, true);
  f32x4 dzterm = f32x4{0.f, 0.f, 0.f, 0.f};

  // The step body instantiated per role (FD: a layer below to project for; HU: a layer above to take dY from) and two steps per loop
  // iteration on alternating element buffers: one body / a copy made the paths share registers through copies the compiler waited on.
  auto run = [&](auto fdc, auto huc) {
  constexpr bool FD = decltype(fdc)::value, HU = decltype(huc)::value;
  auto step = [&](int t, Elem& cur, Elem& nxt) {
    f32x4 carry = f32x4{0.f, 0.f, 0.f, 0.f};
    bool next_ready = false;
    if (t < T - 1) {
      // dG_{t+1} = index T - 2 - t: arrays dr, dz, dn r of both row groups, this wave's K quarter
      u32x4 v[2][3][NPQ];
      const unsigned base = (unsigned)((T - 2 - t) % KS_D) * slot_bytes;
      unsigned spins = 0;
      {
        // first the array a producer stores LAST (dn r: a third of the bytes per look), then the other two -- every dword of all three
        // is checked (visibility order between a wave's stores is likely, not promised)
        for (;;) {
#pragma unroll
          for (int i = 0; i < NPQ; ++i) {
            const unsigned ao = 2u * arr_bytes;
            v[0][2][i] = ks_load<16>(ring, in_off0 + (unsigned)i * 1024u, base + ao);
            v[1][2][i] = ks_load<16>(ring, in_off1 + (unsigned)i * 1024u, base + ao);
          }
          unsigned m = 0u;
#pragma unroll
          for (int i = 0; i < NPQ; ++i) { m = ks_max(m, v[0][2][i]); m = ks_max(m, v[1][2][i]); }
          if (!__any(m > KS_MAXDATA)) break;
          if ((++spins & 63u) == 0u) {
            if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
            if (spins > (SPIN_LIMIT >> 2)) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
          }
        }
      }
      // (nothing of this wave is in flight here -- the look above was waited for in full: the drain is free)
      wave_drain();
      if (pending_x >= 0) { wave_bump<false>(cnt_x + pending_x, lane); pending_x = -1; }
      // progress for consumers OUTSIDE the launch (the gated weight-gradient GEMMs): every store of the steps > t is acknowledged
      if (prog) wave_bump<false>(prog, lane);
      // Order of the rest (vector memory returns in order: nothing slow may sit in front of a look at the own ring):
      //   request dr, dz -> recurrent MFMAs of dn r (its registers are then free) -> request dn into them (layers that project) ->
      //   check dr, dz (re-request while something is missing) -> their MFMAs (recurrent + projection) -> check dn -> its MFMAs ->
      //   only then the drain / counters and the requests that go to HBM and across XCDs (saved gates, dY tile, counter snapshot).
      f32x4 acc[2][2], accp[2][2];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) { acc[r][u2] = f32x4{0.f, 0.f, 0.f, 0.f}; accp[r][u2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      auto load01 = [&]() {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int i = 0; i < NPQ; ++i) {
            const unsigned ao = (unsigned)g * arr_bytes;
            v[0][g][i] = ks_load<16>(ring, in_off0 + (unsigned)i * 1024u, base + ao);
            v[1][g][i] = ks_load<16>(ring, in_off1 + (unsigned)i * 1024u, base + ao);
          }
      };
      load01();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NPQ; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const bf16x8 av = __builtin_bit_cast(bf16x8, v[r][2][i]);
#pragma unroll
          for (int u2 = 0; u2 < 2; ++u2) acc[r][u2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[u2][2][i], acc[r][u2], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      auto load3 = [&]() {      // (dn into the registers dn r has left)
#pragma unroll
        for (int i = 0; i < NPQ; ++i) {
          v[0][2][i] = ks_load<16>(ring, in_off0 + (unsigned)i * 1024u, base + 3u * arr_bytes);
          v[1][2][i] = ks_load<16>(ring, in_off1 + (unsigned)i * 1024u, base + 3u * arr_bytes);
        }
      };
      if (FD) load3();
      __builtin_amdgcn_sched_barrier(0);
      for (;;) {
        unsigned m = 0u;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int i = 0; i < NPQ; ++i) { m = ks_max(m, v[0][g][i]); m = ks_max(m, v[1][g][i]); }
        if (!__any(m > KS_MAXDATA)) break;
        if ((++spins & 63u) == 0u) {
          if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
          if (spins > (SPIN_LIMIT >> 2)) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
        }
        load01();
      }
      WSTAMP(0)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < NPQ; ++i)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, v[r][g][i]);
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
              acc[r][u2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[u2][g][i], acc[r][u2], 0, 0, 0);
              if (FD) accp[r][u2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w2[u2][g][i], accp[r][u2], 0, 0, 0);
            }
          }
      if (FD) {
        for (;;) {      // (stored with dn r by the same instruction of each producer: normally there at the first look)
          unsigned m = 0u;
#pragma unroll
          for (int i = 0; i < NPQ; ++i) { m = ks_max(m, v[0][2][i]); m = ks_max(m, v[1][2][i]); }
          if (!__any(m > KS_MAXDATA)) break;
          if ((++spins & 63u) == 0u) {
            if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
            if (spins > (SPIN_LIMIT >> 2)) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
          }
          load3();
        }
#pragma unroll
        for (int i = 0; i < NPQ; ++i)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, v[r][2][i]);
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) accp[r][u2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w2[u2][2][i], accp[r][u2], 0, 0, 0);
          }
      }
      // csnap: the counter of slot t - 1 (this row group's dY tiles of the layer above), requested one step ago
      const bool up_ready = HU && t > 0 && t < T - 2 && __all(csnap >= (unsigned)G);
      if (HU) csnap = __hip_atomic_load(cnt_up + (t > 2 ? t - 2 : 0), RLX_AGENT);     // (unconditional: no merge with an old value)
      next_ready = up_ready;
      if (DB) fetch(t - 1, nxt, up_ready);
      WSTAMP(3)
      // ONE reduction for both: two partial buffers by step parity (nothing else separates one step's reads from the next step's writes)
      float4* part = (t & 1) ? part_q : part_p;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
          const f32x4 c = acc[r][u2];
          part[(wave * NT + r * 2 + u2) * 64 + lane] = float4{c[0], c[1], c[2], c[3]};
          if (FD) { const f32x4 d = accp[r][u2]; part[(wave * NT + 4 + r * 2 + u2) * 64 + lane] = float4{d[0], d[1], d[2], d[3]}; }
        }
      ks_barrier();
      WSTAMP(7)
      const f32x4 rec = ks_reduce<NT>(part, rgl * 2 + us, lane);
      if (FD && rg_live) {
        // dY[layer - 1]_{t+1}, this wave's 16 x 16 tile: nn.GRU's dropout mask of out[layer - 1] (same Philox draws as the forward pass),
        // then written through for the layer below; its counter moves at the next drain
        f32x4 py = ks_reduce<NT>(part, 4 + rgl * 2 + us, lane);
        if (DROP) {
          float* td = tiles;
          tile_put(td, py, j, q);
          float4 v4 = ld4(td + (lane & 15) * WTP + 4 * kg);
          const long long e = a.elem0 + ((long long)(t + 1) * B + (rrow < B ? rrow : 0)) * H + u0 + 4 * kg;
          const float4 u = Philox::uniform4(a.seed[layer - 1], (uint64_t)(e >> 2), 2u);
          v4.x = u.x >= a.drop_p ? v4.x * a.drop_scale : 0.f; v4.y = u.y >= a.drop_p ? v4.y * a.drop_scale : 0.f;
          v4.z = u.z >= a.drop_p ? v4.z * a.drop_scale : 0.f; v4.w = u.w >= a.drop_p ? v4.w * a.drop_scale : 0.f;
          *reinterpret_cast<float4*>(td + (lane & 15) * WTP + 4 * kg) = v4;
#pragma unroll
          for (int i = 0; i < 4; ++i) py[i] = td[(4 * q + i) * WTP + j];
        }
        ks_store<16>(dyr, dy_tile, (unsigned)(t + 1) * dy_slot, __builtin_bit_cast(u32x4, py));
        pending_x = t + 1;
      }
#ifdef B2T_WAVE_TIMING
      asm volatile("s_nop 0" :: "v"(rec[0]));
#endif
      WSTAMP(1)
#pragma unroll
      for (int i = 0; i < 4; ++i) carry[i] = rec[i] + dzterm[i];
    } else {
      wave_drain();
      if (prog) wave_bump<false>(prog, lane);
      if (DB) fetch(t - 1, nxt, false);
      if (a.dh_last[layer]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (live[i]) carry[i] = a.dh_last[layer][(long long)(m0 + 4 * q + i) * H + unit];
      }
    }
    if (t < 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) if (live[i]) a.dh_init[layer][(long long)(m0 + 4 * q + i) * H + unit] = carry[i];
      return;
    }
    if (rg_live) {
      f32x4 dy = cur.dyt;
      if (HU && !cur.ok) {      // the layer above was less than two steps ahead when this step's dY was asked for: wait, load it now
        wave_wait<false>(cnt_up + t, (unsigned)G, err, lane);
        dy = load_dy(t);
      }
      WSTAMP(5)
      f32x4 g0, g1, g2, g3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = dy[i] + carry[i];
        const float dn = d * (1.0f - cur.z[i]);
        const float dz = d * (cur.hprev[i] - cur.nv[i]);
        const float dn_pre = dn * (1.0f - cur.nv[i] * cur.nv[i]);
        const float dz_pre = dz * cur.z[i] * (1.0f - cur.z[i]);
        const float dr_pre = dn_pre * cur.ghn[i] * cur.r[i] * (1.0f - cur.r[i]);
        g0[i] = dr_pre; g1[i] = dz_pre; g2[i] = dn_pre * cur.r[i]; g3[i] = dn_pre;
        dzterm[i] = d * cur.z[i];
      }
      tile_put(tiles + 0 * WTILE_F, g0, j, q); tile_put(tiles + 1 * WTILE_F, g1, j, q);
      tile_put(tiles + 2 * WTILE_F, g2, j, q); tile_put(tiles + 3 * WTILE_F, g3, j, q);
      // lanes 0-31 publish arrays 0 and 2, lanes 32-63 arrays 1 and 3
      const int a0 = lane >> 5;
      const u32x4 f0 = tile_frag(tiles + a0 * WTILE_F, lane), f1 = tile_frag(tiles + (a0 + 2) * WTILE_F, lane);
      const int n = T - 1 - t;
      const unsigned base = (unsigned)(n % KS_D) * slot_bytes, arm = (unsigned)((n + 2) % KS_D) * slot_bytes;
      const unsigned vo = rg_off + my_frag + (unsigned)a0 * arr_bytes;      // (per lane, loop-invariant)
      ks_store<0>(ring, vo, base, ks_clamp(f0));
      ks_store<0>(ring, vo, base + 2u * arr_bytes, ks_clamp(f1));
      ks_store<0>(ring, vo, arm, ks_sentinel());
      ks_store<0>(ring, vo, arm + 2u * arr_bytes, ks_sentinel());
      WSTAMP(2)
      if (rrow < B) {
        float* dgl = a.dG[layer] + (long long)t * B * 4 * H;
        const unsigned off = (unsigned)(((long long)rrow * 4 * H + u0 + 4 * kg) * 4);
        // written THROUGH only for readers inside the sweep's lifetime (gated GEMMs, flags bit 2); otherwise ordinary stores (the kernel's
        // end makes them visible)
        if (a.flags & 4) {
#pragma unroll
          for (int g = 0; g < 4; ++g) store_f4<16>(dgl, off + (unsigned)g * (unsigned)H * 4u, ld4(tiles + g * WTILE_F + (lane & 15) * WTP + 4 * kg));
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) store_f4<0>(dgl, off + (unsigned)g * (unsigned)H * 4u, ld4(tiles + g * WTILE_F + (lane & 15) * WTP + 4 * kg));
        }
      }
      WSTAMP(4)
    }
    if (!DB) fetch(t - 1, nxt, next_ready);
  };
  if (DB) {
    for (int t = T - 1; t >= -1; t -= 2) {
      step(t, ea, eb);
      if (t - 1 >= -1) step(t - 1, eb, ea);
    }
  } else {
    for (int t = T - 1; t >= -1; --t) step(t, ea, ea);
  }
  };
  if (feeds) { if (has_up) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}); }
  else { if (has_up) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}); }

  wave_drain();
  if (pending_x >= 0) wave_bump<false>(cnt_x + pending_x, lane);
#ifdef B2T_WAVE_TIMING
  if (k == 0 && wave == 0 && lane == 0 && a.timing)
    for (int i = 0; i < 8; ++i) a.timing[layer * 8 + i] = (unsigned)(tacc[i] / (unsigned long long)T);
#endif
}

// arms the own-recurrence rings (KS_D slots each, all ones) and zeroes nothing else: one launch for all layers
struct KsArmArgs { char* ring[B2T_MAX_LAYERS]; unsigned vec16; };      // vec16: 16-byte words per ring
__global__ void ks_arm_kernel(const KsArmArgs a) {
  u32x4* p = reinterpret_cast<u32x4*>(a.ring[blockIdx.y]);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < a.vec16; i += gridDim.x * blockDim.x) p[i] = ks_sentinel();
}
