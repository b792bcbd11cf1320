// gru_persistent.hip — persistent (single-launch) GRU layer sweeps for gfx950.
//
// Same decomposition as the step-launch kernels in gru.hip — workgroup (js, mb) owns hidden units
// [16js,16js+16) x batch rows [16mb,16mb+16) — but ONE launch runs all T steps:
//   * the workgroup's slice of W_hh (forward: rows of the owned units; backward: columns, read from the
//     transposed copy) is loaded ONCE into VGPRs as MFMA B-operands and stays there for the whole sweep
//     (H=512: 96 VGPRs per lane per wave; the K range is split over the 4 waves);
//   * h_t (forward) / dGh_t (backward) is exchanged between the G = H/16 workgroups of a row group
//     through HBM/L2 with the write-through hand-off of cdna_hip_programming.md §6 Guideline 16 (R1):
//     producers store the payload with sc1 (write-through) stores, every storing wave drains vmcnt(0),
//     barrier, one lane bumps an agent-scope counter; consumers poll that ONE word relaxed and then read
//     the payload with sc1 loads (no L1 hit possible, no fence needed).  Counters are self-cleaning (two sets
//     alternate between calls, finish_call); every spin is bounded and reports through an error word.
//   * row groups are independent recurrences, so the 4 groups of a B=64 batch progress independently.
//   * per step the operand loads (8 x 16 B per lane, ~2700 cycles until the last one lands) are all issued before
//     the first MFMA and consumed chunk by chunk (vmcnt(7), vmcnt(6), ...): loads must be branch-free (clamped)
//     and fenced with sched_barrier, otherwise the compiler either waits for all of them or sinks them.
// The entry point requires grid <= CU count (one row group resident is what correctness needs).
#include <stdlib.h>
#include "gru_cell.h"
#include <algorithm>
#include <stdlib.h>
#include "gru_sync.h"

#ifndef B2T_LOC_ST_AUX
#define B2T_LOC_ST_AUX 16   // payload stores under the XCD-local hand-off: 16 write-through (L2 keeps a clean copy for the peers; ordinary stores, 0, left 1.3 GB of dirty dG per step in the L2s and cost the GEMMs 0.3-1 ms)
#endif
#ifndef B2T_BF16_WIDE_1PERCU
// bf16 32-unit forward sweeps from this many 16-wide K chunks per wave on (H >= 64 * that) are compiled for ONE workgroup per CU:
// at H = 768 the kernel needs ~272 registers and two per CU (256) spilled 13 of them to scratch inside the step loop
#define B2T_BF16_WIDE_1PERCU 12
#endif
#ifndef B2T_LOC_RING_AUX
#define B2T_LOC_RING_AUX 16   // the bf16 fragment tiles under the XCD-local hand-off: 16 write-through, 0 ordinary stores (dirty lines in that XCD's L2; 25 MB per call)
#endif
#ifndef B2T_LOAD_AUX
#define B2T_LOAD_AUX 16   // cache policy of the operand loads: 16 = sc1 (never served from this XCD's L2), 0 = ordinary
#endif

namespace b2t {

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// NT: 16-unit tiles per workgroup (1, or 2 with bf16 operands, whose weight slice is half the registers): 32 units per
// workgroup halve the workgroups of a sweep -- five concurrent sweeps then crowd the CUs half as much.
template <int NCH, bool BF16, int NT = 1, bool LOC = false, bool RING = false>  // NCH: 16-wide K chunks per wave (H <= 64*NCH); LOC: XCD-local hand-off; RING: bf16 fragment hand-off (BF16, NT = 2)
__global__ __launch_bounds__(256, (NT == 2 ? ((BF16 && NCH < B2T_BF16_WIDE_1PERCU) ? 2 : 1) : (NCH <= 8 ? 3 : 1))) void gru_persist_fwd_kernel(const float* __restrict__ gi,
                                                                 const float* __restrict__ w_hh,
                                                                 const float* __restrict__ b_hh,
                                                                 const float* __restrict__ h_init, float* out,
                                                                 float* __restrict__ reserve, int T, int B, int H,
                                                                 unsigned* sync, int par) {
  constexpr int TPN = 16 * NT + 4;    // LDS pitch of the staged tile (= TP for one tile)
  __shared__ __attribute__((aligned(16))) float red[4 * 3 * NT * 4 * 64 + 16 * TPN];
  constexpr int NSLOT_F = NCH < 8 ? NCH : 8;   // staging slots per wave, recycled every NSLOT_F instructions (as in the backward sweep)
  constexpr bool H16 = RING && BF16 && NT == 2;   // hand-off as bf16 MFMA fragments through the ring (gru_sync.h): no staging
  __shared__ __attribute__((aligned(16))) float stage[H16 ? 1 : 4][H16 ? 4 : NSLOT_F * SLOT_F];   // per-wave operand staging (gru_sync.h)
  float* hs = red + 4 * 3 * NT * 4 * 64;   // staged h tile [16 rows][TPN]
  constexpr int AUX = LOC ? B2T_LOC_ST_AUX : 16;   // sc1 payload stores (device scope); under the XCD-local hand-off see B2T_LOC_ST_AUX
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifndef B2T_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);   // the sweep is the critical path: its waves issue ahead of co-resident GEMM waves
#endif
  const unsigned G = (unsigned)H / (16u * NT);
  const int j = lane & 15, q = lane >> 4;
  unsigned* err = sync;  // word 0: error flag (sticky), word 1: which counter set this call uses
  // Two counter sets alternate between calls: this call counts in set p and clears set 1-p for the next call
  // on this workspace (stream order makes that safe), so no memset node is needed in front of the launch.
  const unsigned pset = __hip_atomic_load(sync + 1, RLX_AGENT) & 1u;
  {
    unsigned* other = sync + 32 + (size_t)(1u - pset) * SETW;
    const int nthr = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < SETW; i += nthr) other[i] = 0u;
  }
  int rg = blockIdx.y, tile = blockIdx.x;   // row groups are independent recurrences
  if constexpr (LOC) {
    if (!local_role(sync + 32 + (size_t)pset * SETW + (SETW - 16), (B + 15) / 16, G, par, rg, tile)) { finish_call(sync, pset); return; }
  }
  const int m0 = rg * 16;
  const int j0 = tile * 16 * NT;
  int unit[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) unit[n] = j0 + 16 * n + j;
  const int nch = H / 16;

  typename WFrag<BF16>::type w[3][NT][NCH];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    const int c = KCHUNK(wave, ci, NCH);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        w[g][n][ci] = make_wfrag<BF16>(c < nch ? *reinterpret_cast<const float4*>(w_hh + ((long long)g * H + unit[n]) * H + c * 16 + 4 * q)
                                               : make_float4(0.f, 0.f, 0.f, 0.f));
  }
  float bhr[NT], bhz[NT], bhn[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { bhr[n] = b_hh[unit[n]]; bhz[n] = b_hh[H + unit[n]]; bhn[n] = b_hh[2 * H + unit[n]]; }
  const int row = m0 + 4 * q + wave;
  const bool live = row < B;
  unsigned* cnt = sync + 32 + (size_t)pset * SETW + (size_t)rg * T * CSTRIDE;
  float hp[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) hp[n] = live ? h_init[(long long)row * H + unit[n]] : 0.f;
  char* const ring = ring_base(sync);
  const unsigned ngrp = (unsigned)(B + 15) / 16u;
  const unsigned step_bytes = ngrp * G * 1024u, depth = ring_depth(step_bytes, T);   // (gru_sync.h: one slot per step when the call fits)
  if constexpr (H16) {   // rows beyond the batch are never written: the ring must still carry finite numbers for them
    for (int i = tid; i < 16 * TPN; i += 256) hs[i] = 0.f;
    __syncthreads();
  }

#ifdef B2T_TIMING
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0};
#define TSTAMP(i) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; }
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#else
#define TSTAMP(i)
#endif
  for (int t = 0; t < T; ++t) {
   {
    float gir[NT], giz[NT], gin[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      gir[n] = giz[n] = gin[n] = 0.f;
      if (live) {
        const float* g3 = gi + ((long long)t * B + row) * 3 * H + unit[n];
        // (single-use streams: non-temporal loads keep them from displacing the GEMMs' operand tiles in the L2s)
        gir[n] = __builtin_nontemporal_load(g3); giz[n] = __builtin_nontemporal_load(g3 + H); gin[n] = __builtin_nontemporal_load(g3 + 2 * H);
      }
    }
    const float* hsrc = h_init;
    if (t > 0) {
      if constexpr (LOC) wait_count_local(cnt + (size_t)(t - 1) * CSTRIDE, G, err);
      else wait_count(cnt + (size_t)(t - 1) * CSTRIDE, G, err);
      hsrc = out + (long long)(t - 1) * B * H;
    }
    TSTAMP(0)   // poll + barrier
    f32x4 acc[3 * NT];   // [gate][tile]
#pragma unroll
    for (int g = 0; g < 3 * NT; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (H16) {
      if (t == 0) {
        // the call's first step: h_init is fp32 [B][H]; fragments read directly (4 lanes per row: the slow pattern, once per call)
        const int rj = m0 + j < B ? m0 + j : B - 1;
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
          const int c = KCHUNK(wave, ci, NCH);
          const float4 f = *reinterpret_cast<const float4*>(h_init + (long long)rj * H + (c < nch ? c : nch - 1) * 16 + 4 * q);
#pragma unroll
          for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[g * NT + n] = mfma_chunk16<true>(f, w[g][n][ci], acc[g * NT + n]);
        }
      } else {
        const unsigned base = ((unsigned)(t - 1) % depth) * step_bytes + (unsigned)rg * G * 1024u + (unsigned)lane * 16u;
        u32x4 v[NCH / 2];
#pragma unroll
        for (int p = 0; p < NCH / 2; ++p) {   // pairs beyond the operand meet zero weights: clamped to real (finite) data
          const unsigned pair = (unsigned)(wave * (NCH / 2) + p);
          v[p] = load_u4<LOC ? 0 : B2T_LOAD_AUX>(ring, base + (pair < G ? pair : G - 1u) * 1024u);
        }
        __builtin_amdgcn_sched_barrier(0);   // all loads are in flight before the first MFMA
#pragma unroll
        for (int p = 0; p < NCH / 2; ++p) {
          const bf16x4 a0 = frag_lo(v[p]), a1 = frag_hi(v[p]);
#pragma unroll
          for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              if constexpr (BF16) {
                acc[g * NT + n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, w[g][n][2 * p], acc[g * NT + n], 0, 0, 0);
                acc[g * NT + n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, w[g][n][2 * p + 1], acc[g * NT + n], 0, 0, 0);
              }
            }
        }
      }
    } else {
    // All loads go out first (branch-free, clamped: a conditional load makes the compiler wait for everything), then
    // each pair is transposed and consumed as it lands (vmcnt(6), vmcnt(4), ...).
    float4 v[NCH];
    issue_block_loads<NCH, LOC ? 0 : B2T_LOAD_AUX>(v, hsrc, m0, B, H, wave * NCH * 16, H, lane);
    __builtin_amdgcn_sched_barrier(0);   // all loads are in flight before the first MFMA (the scheduler would sink them)
#ifdef B2T_TIMING_SPLIT_LOADS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSTAMP(6)   // operand loads complete (timing build only: serialises loads and MFMAs)
#endif
#pragma unroll
    for (int p = 0; p < NCH / 2; ++p) {
      float4 a[2];
      transpose_pair(&stage[wave][((2 * p) % NSLOT_F) * SLOT_F], v[2 * p], v[2 * p + 1], a[0], a[1], lane);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int ci = 2 * p + h2;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[g * NT + n] = mfma_chunk16<BF16>(a[h2], w[g][n][ci], acc[g * NT + n]);
      }
    }
    }
    asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));
    TSTAMP(1)   // loads + MFMA
    float gh[3 * NT];
    cross_wave_reduce<3 * NT>(red, acc, gh, wave, lane);
    TSTAMP(2)   // reduce
    float sv_r[NT], sv_z[NT], sv_n[NT], sv_ghn[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      sv_r[n] = sv_z[n] = sv_n[n] = sv_ghn[n] = 0.f;
      if (live) {
      // gate non-linearities on the hardware exp unit (v_exp_f32, ~1 ulp): sigmoid(x) = 1/(1+2^(-x log2 e)),
      // tanh(x) = 1 - 2/(1+2^(2x log2 e)); the precise libm forms cost ~800 cycles per step on the serial chain.
        const float ghn = gh[2 * NT + n] + bhn[n];
        const float r = fast_sigmoid(gir[n] + gh[0 * NT + n] + bhr[n]);
        const float z = fast_sigmoid(giz[n] + gh[1 * NT + n] + bhz[n]);
        const float nn = fast_tanh(gin[n] + r * ghn);
        const float h = (1.0f - z) * nn + z * hp[n];
        hs[(4 * q + wave) * TPN + 16 * n + j] = h;
        sv_r[n] = r; sv_z[n] = z; sv_n[n] = nn; sv_ghn[n] = ghn;
        hp[n] = h;
      }
    }
    TSTAMP(3)   // gates
    __syncthreads();                       // tile staged; also fences `red` for the next iteration
    TSTAMP(4)   // stage barrier (waits for the slowest wave's gates)
    if (wave == 0) {                       // one wave writes the 16x16 tile as 64 x 16 B write-through stores
      if constexpr (H16) {   // the peers read the ring: the tile as one pair of bf16 fragments (1 KB), published before the fp32 tile is stored
        const float4 c0 = *reinterpret_cast<const float4*>(&hs[j * TPN + 4 * q]), c1 = *reinterpret_cast<const float4*>(&hs[j * TPN + 16 + 4 * q]);
        store_u4<(LOC ? B2T_LOC_RING_AUX : 16)>(ring, ((unsigned)t % depth) * step_bytes + ((unsigned)rg * G + (unsigned)tile) * 1024u + (unsigned)lane * 16u, pack_frag_pair(c0, c1));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) { if constexpr (LOC) l2_atomic_inc(cnt + (size_t)t * CSTRIDE); else __hip_atomic_fetch_add(cnt + (size_t)t * CSTRIDE, 1u, RLX_AGENT); }
      }
    }
    // (with the fragment hand-off the fp32 tile is nobody's dependency inside the sweep: ANOTHER wave stores it -- the next
    //  step's poll is wave 0's, and its returning atomic waits for every store the wave has in flight)
    if (wave == (H16 ? 1 : 0)) {
      const int r = lane >> 2;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int c4 = 16 * n + (lane & 3) * 4;
        if (m0 + r < B)
          store_f4<AUX>(out + (long long)t * B * H, (unsigned)(((long long)(m0 + r) * H + j0 + c4) * 4),
                       *reinterpret_cast<const float4*>(&hs[r * TPN + c4]));
      }
      if constexpr (!H16) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) { if constexpr (LOC) l2_atomic_inc(cnt + (size_t)t * CSTRIDE); else __hip_atomic_fetch_add(cnt + (size_t)t * CSTRIDE, 1u, RLX_AGENT); }
      }
    }
    // The gate values saved for the backward sweep are nobody's dependency inside this sweep: store them AFTER the
    // publish so their write acknowledgements are not part of the drain in front of the counter increment.
    if (live && reserve) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float* rs = reserve + ((long long)t * B + row) * 4 * H + unit[n];
        // streaming stores: nobody reads the reserve before the backward pass, it must not push the GEMMs' operands out of the L2s
        __builtin_nontemporal_store(sv_r[n], rs); __builtin_nontemporal_store(sv_z[n], rs + H);
        __builtin_nontemporal_store(sv_n[n], rs + 2 * H); __builtin_nontemporal_store(sv_ghn[n], rs + 3 * H);
      }
    }
    TSTAMP(5)   // tile store + drain + publish
   }
  }
  finish_call(sync, pset);   // next call uses the cleared set
#ifdef B2T_TIMING
  if (threadIdx.x == 0 && blockIdx.z == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 17))
    for (int i = 0; i < 7; ++i) sync[8 + (blockIdx.x ? 8 : 0) + i] = (unsigned)(tacc[i] / (unsigned long long)T);
#endif
}

// ---------------------------------------------------------------------------------------------------
// forward, with the NEXT layer's input projection computed in the sweep's idle matrix-core slots (round 4).
//
// A sweep workgroup uses the matrix cores ~3072 of the ~8-10 k cycles of a step; the rest is the hand-off (store drain,
// counter, poll, first operand load).  The A operand of a step -- the full h_{t-1} row block of this layer, transposed into
// MFMA fragments in the per-wave LDS slots -- is exactly the A operand of the next layer's input projection
// gi'_{t-1} = h_{t-1} W_ih'^T + b_ih' (rnn_model.py:126: nn.GRU applies W_ih of layer l+1 to the outputs of layer l).  So
// after publishing h_t the workgroup contracts the fragments that still sit in its LDS slots with ITS slice of W_ih' (the 48
// gate rows of its 16 units; kept in registers and, for NLDS chunks per wave, in LDS as ready-made B fragments) at LOW wave
// priority, reduces over the four waves and stores its 16 x 48 tile of gi'.  The projection GEMM of layers >= 1
// (0.2 TFLOP per C2 step, and one GEMM + two queue hops on the critical path of every wavefront stage) disappears.
// Time t of the projection is produced during step t+1; the chunk's last one in an epilogue that waits for the peers'
// last tiles like a step does.
// ---------------------------------------------------------------------------------------------------
template <int NCH, int NLDS, bool LOC>   // NLDS: 16-wide K chunks (per wave) of the W_ih' slice kept in LDS, the rest in registers
__global__ __launch_bounds__(256, 2) void gru_persist_fwd_fused_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                                      const float* __restrict__ b_hh, const float* __restrict__ h_init,
                                                                      float* out, float* __restrict__ reserve,
                                                                      const float* __restrict__ w_ih2, const float* __restrict__ b_ih2,
                                                                      float* __restrict__ gi2, int T, int B, int H, unsigned* sync, int par) {
  constexpr int TPN = 20;
  constexpr int NREG = NCH - NLDS;
  __shared__ __attribute__((aligned(16))) float red[4 * 3 * 4 * 64 + 16 * TPN];
  __shared__ __attribute__((aligned(16))) float stage[4][NCH * SLOT_F];   // per-wave operand staging: every slot is kept until the next step (the projection re-reads it)
  __shared__ __attribute__((aligned(16))) float4 w2l[4][NLDS > 0 ? NLDS : 1][3][64];
  float* hs = red + 4 * 3 * 4 * 64;
  constexpr int AUX = LOC ? B2T_LOC_ST_AUX : 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __builtin_amdgcn_s_setprio(3);
  const unsigned G = (unsigned)H / 16u;
  const int j = lane & 15, q = lane >> 4;
  unsigned* err = sync;
  const unsigned pset = __hip_atomic_load(sync + 1, RLX_AGENT) & 1u;
  {
    unsigned* other = sync + 32 + (size_t)(1u - pset) * SETW;
    const int nthr = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < SETW; i += nthr) other[i] = 0u;
  }
  int rg = blockIdx.y, tile = blockIdx.x;
  if constexpr (LOC) {
    if (!local_role(sync + 32 + (size_t)pset * SETW + (SETW - 16), (B + 15) / 16, G, par, rg, tile)) { finish_call(sync, pset); return; }
  }
  const int m0 = rg * 16;
  const int unit = tile * 16 + j;
  const int nch = H / 16;

  float4 w[3][NCH], w2[3][NREG > 0 ? NREG : 1];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    const int c = KCHUNK(wave, ci, NCH);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const bool in = c < nch;
      w[g][ci] = in ? *reinterpret_cast<const float4*>(w_hh + ((long long)g * H + unit) * H + c * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 x2 = in ? *reinterpret_cast<const float4*>(w_ih2 + ((long long)g * H + unit) * H + c * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (ci < NREG) w2[g][ci < NREG ? ci : 0] = x2;
      else w2l[wave][ci - NREG][g][lane] = x2;   // written and read by the same lane of the same wave: no barrier needed
    }
  }
  const float bhr = b_hh[unit], bhz = b_hh[H + unit], bhn = b_hh[2 * H + unit];
  const float b2r = b_ih2[unit], b2z = b_ih2[H + unit], b2n = b_ih2[2 * H + unit];
  const int row = m0 + 4 * q + wave;
  const bool live = row < B;
  unsigned* cnt = sync + 32 + (size_t)pset * SETW + (size_t)rg * T * CSTRIDE;
  float hp = live ? h_init[(long long)row * H + unit] : 0.f;

  // The projection of the h block whose fragments sit in this wave's LDS slots -> gi2[tt].  At LOW wave priority: the slack
  // work yields the matrix pipe to the co-resident workgroup's (another sweep's) recurrent product.  (Measured: splitting it
  // -- most of the chunks here, the rest + the reduction behind the next step's operand loads -- costs registers (15 spilled)
  // and a barrier in front of the recurrent product: one sweep alone 5.9 instead of 5.1 us per step, the C2 step 19.5 instead
  // of 18.6 ms.)
  auto project = [&](int tt) {
    __builtin_amdgcn_s_setprio(1);
    f32x4 acc2[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const float* sp = &stage[wave][(ci & ~1) * SLOT_F] + (j >> 3) * SLOT_F + (j & 7) * 36 + 4 * q + 16 * (ci & 1);
      const float4 a = *reinterpret_cast<const float4*>(sp);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float4 b;
        if (ci < NREG) b = w2[g][ci < NREG ? ci : 0];
        else b = w2l[wave][ci >= NREG ? ci - NREG : 0][g][lane];
        acc2[g] = mfma_chunk16<false>(a, b, acc2[g]);
      }
    }
    float g2[3];
    cross_wave_reduce<3>(red, acc2, g2, wave, lane);
    if (live) {
      float* gp = gi2 + ((long long)tt * B + row) * 3 * H + unit;
      gp[0] = g2[0] + b2r; gp[H] = g2[1] + b2z; gp[2 * H] = g2[2] + b2n;
    }
    __builtin_amdgcn_s_setprio(3);
  };

  for (int t = 0; t < T; ++t) {
    float gir = 0.f, giz = 0.f, gin = 0.f;
    if (live) {
      const float* g3 = gi + ((long long)t * B + row) * 3 * H + unit;
      gir = __builtin_nontemporal_load(g3); giz = __builtin_nontemporal_load(g3 + H); gin = __builtin_nontemporal_load(g3 + 2 * H);
    }
    const float* hsrc = h_init;
    if (t > 0) {
      if constexpr (LOC) wait_count_local(cnt + (size_t)(t - 1) * CSTRIDE, G, err);
      else wait_count(cnt + (size_t)(t - 1) * CSTRIDE, G, err);
      hsrc = out + (long long)(t - 1) * B * H;
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 v[NCH];
    issue_block_loads<NCH, LOC ? 0 : B2T_LOAD_AUX>(v, hsrc, m0, B, H, wave * NCH * 16, H, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NCH / 2; ++p) {
      float4 a[2];
      transpose_pair(&stage[wave][(2 * p) * SLOT_F], v[2 * p], v[2 * p + 1], a[0], a[1], lane);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = mfma_chunk16<false>(a[h2], w[g][2 * p + h2], acc[g]);
    }
    asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));
    float gh[3];
    cross_wave_reduce<3>(red, acc, gh, wave, lane);
    float sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_ghn = 0.f;
    if (live) {
      const float ghn = gh[2] + bhn;
      const float r = fast_sigmoid(gir + gh[0] + bhr);
      const float z = fast_sigmoid(giz + gh[1] + bhz);
      const float nn = fast_tanh(gin + r * ghn);
      const float h = (1.0f - z) * nn + z * hp;
      hs[(4 * q + wave) * TPN + j] = h;
      sv_r = r; sv_z = z; sv_n = nn; sv_ghn = ghn;
      hp = h;
    }
    __syncthreads();                       // tile staged; also fences `red` (every wave has read its sums of this step)
    if (wave == 0) {
      const int r = lane >> 2, c4 = (lane & 3) * 4;
      if (m0 + r < B)
        store_f4<AUX>(out + (long long)t * B * H, (unsigned)(((long long)(m0 + r) * H + tile * 16 + c4) * 4),
                     *reinterpret_cast<const float4*>(&hs[r * TPN + c4]));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) { if constexpr (LOC) l2_atomic_inc(cnt + (size_t)t * CSTRIDE); else __hip_atomic_fetch_add(cnt + (size_t)t * CSTRIDE, 1u, RLX_AGENT); }
    }
    if (live && reserve) {
      float* rs = reserve + ((long long)t * B + row) * 4 * H + unit;
      __builtin_nontemporal_store(sv_r, rs); __builtin_nontemporal_store(sv_z, rs + H);
      __builtin_nontemporal_store(sv_n, rs + 2 * H); __builtin_nontemporal_store(sv_ghn, rs + 3 * H);
    }
    // slack: the LDS slots hold h_{t-1} (this chunk's local time t - 1; at t = 0 the previous chunk's last state, whose
    // projection that chunk's epilogue made).  `red` is free: the barrier above is behind this step's reduction, and the next
    // step's reduction writes it behind the poll barrier, which every wave reaches after it has read the projection's sums.
    if (t > 0) project(t - 1);
  }
  // epilogue: the projection of the chunk's last state h_{T-1}
  {
    if constexpr (LOC) wait_count_local(cnt + (size_t)(T - 1) * CSTRIDE, G, err);
    else wait_count(cnt + (size_t)(T - 1) * CSTRIDE, G, err);
    float4 v[NCH];
    issue_block_loads<NCH, LOC ? 0 : B2T_LOAD_AUX>(v, out + (long long)(T - 1) * B * H, m0, B, H, wave * NCH * 16, H, lane);
#pragma unroll
    for (int p = 0; p < NCH / 2; ++p) {
      float4 a[2];
      transpose_pair(&stage[wave][(2 * p) * SLOT_F], v[2 * p], v[2 * p + 1], a[0], a[1], lane);
      asm volatile("" :: "v"(a[0].x), "v"(a[1].x));
    }
    project(T - 1);
  }
  finish_call(sync, pset);
}

// ---------------------------------------------------------------------------------------------------
// backward.  Workgroup (ks, mb) owns dh columns [16ks,16ks+16): each step it (A) contracts the full
// dGh_{t+1} row block with its register-resident W_hh[:, slice] and (B) forms the gate gradients of
// step t for its slice, publishing them as dG[t] for the other workgroups of the row group.
// ---------------------------------------------------------------------------------------------------
#ifdef B2T_TIMING
__device__ unsigned g_bwd_t[3][512];   // timing build: per workgroup (rg * G + tile) of the LAST backward launch: start, first hand-off received, end (100 MHz wall clock, low word)
#endif
template <int NCB, bool BF16, int NT = 1, bool LOC = false, bool RING = false>  // NCB: 16-wide chunks of the 3H contraction per wave (3H <= 64*NCB); RING: bf16 fragment hand-off (BF16, NT = 2)
__global__ __launch_bounds__(256, 1) void gru_persist_bwd_kernel(const float* __restrict__ dY,
                                                                 const float* __restrict__ dh_last,
                                                                 const float* __restrict__ reserve,
                                                                 const float* __restrict__ out,
                                                                 const float* __restrict__ h_init,
                                                                 const float* __restrict__ w_hh_t, float* dG,
                                                                 float* __restrict__ dh_init, int T, int B, int H,
                                                                 unsigned* sync, int par) {
  constexpr int TPN = 16 * NT + 4;
  __shared__ __attribute__((aligned(16))) float red[4 * NT * 4 * 64 + 4 * 16 * TPN];
  constexpr int NSLOT = NCB < 8 ? NCB : 8;   // staging slots per wave, recycled every NSLOT instructions
  constexpr bool H16 = RING && BF16 && NT == 2;   // hand-off as bf16 MFMA fragments through the ring (gru_sync.h): no staging
  __shared__ __attribute__((aligned(16))) float stage[H16 ? 1 : 4][H16 ? 4 : NSLOT * SLOT_F];
  float* gs = red + 4 * NT * 4 * 64;   // staged gate-gradient tiles [4 arrays][16 rows][TPN]
  constexpr int AUX = LOC ? B2T_LOC_ST_AUX : 16;   // sc1 payload stores (device scope); under the XCD-local hand-off see B2T_LOC_ST_AUX
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifndef B2T_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);   // the sweep is the critical path: its waves issue ahead of co-resident GEMM waves
#endif
  const unsigned G = (unsigned)H / (16u * NT);
  const int j = lane & 15, q = lane >> 4;
  unsigned* err = sync;  // word 0: error flag (sticky), word 1: which counter set this call uses
  // Two counter sets alternate between calls: this call counts in set p and clears set 1-p for the next call
  // on this workspace (stream order makes that safe), so no memset node is needed in front of the launch.
  const unsigned pset = __hip_atomic_load(sync + 1, RLX_AGENT) & 1u;
  {
    unsigned* other = sync + 32 + (size_t)(1u - pset) * SETW;
    const int nthr = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < SETW; i += nthr) other[i] = 0u;
  }
  int rg = blockIdx.y, tile = blockIdx.x;   // row groups are independent recurrences
  if constexpr (LOC) {
    if (!local_role(sync + 32 + (size_t)pset * SETW + (SETW - 16), (B + 15) / 16, G, par, rg, tile)) { finish_call(sync, pset); return; }
  }
  const int m0 = rg * 16;
  const int j0 = tile * 16 * NT;
#ifdef B2T_TIMING
  const unsigned wgid = ((unsigned)rg * G + (unsigned)tile) & 511u;
  if (threadIdx.x == 0) g_bwd_t[0][wgid] = (unsigned)wall_clock64();
#endif
  int unit[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) unit[n] = j0 + 16 * n + j;
  const int nch = 3 * H / 16;

  typename WFrag<BF16>::type w[NT][NCB];
#pragma unroll
  for (int ci = 0; ci < NCB; ++ci) {
    const int c = KCHUNK(wave, ci, NCB);
#pragma unroll
    for (int n = 0; n < NT; ++n)
      w[n][ci] = make_wfrag<BF16>(c < nch ? *reinterpret_cast<const float4*>(w_hh_t + (long long)unit[n] * 3 * H + c * 16 + 4 * q)
                                          : make_float4(0.f, 0.f, 0.f, 0.f));
  }
  const int row = m0 + 4 * q + wave;
  const bool live = row < B;
  unsigned* cnt = sync + 32 + (size_t)pset * SETW + (size_t)rg * T * CSTRIDE;
  float dzterm[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) dzterm[n] = 0.f;
  char* const ring = ring_base(sync);
  const unsigned ngrp = (unsigned)(B + 15) / 16u;
  const unsigned npair = 3u * G;   // chunk pairs of the 3H-long contraction (a 32-unit tile of one gate is one pair)
  const unsigned step_bytes = ngrp * npair * 1024u, depth = ring_depth(step_bytes, T);   // (gru_sync.h: one slot per step when the call fits)
  if constexpr (H16) {   // rows beyond the batch are never written: the ring must still carry finite numbers for them
    for (int i = tid; i < 4 * 16 * TPN; i += 256) gs[i] = 0.f;
    __syncthreads();
  }

#ifdef B2T_TIMING
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  for (int t = T - 1; t >= -1; --t) {
   {
    // operands of the elementwise part do not depend on the recurrence: fetch them first
    float r[NT], z[NT], nv[NT], ghn[NT], hprev[NT], dy[NT], carry[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      r[n] = z[n] = nv[n] = ghn[n] = hprev[n] = dy[n] = carry[n] = 0.f;
      if (live && t >= 0) {
        const float* rs = reserve + ((long long)t * B + row) * 4 * H + unit[n];
        r[n] = __builtin_nontemporal_load(rs); z[n] = __builtin_nontemporal_load(rs + H); nv[n] = __builtin_nontemporal_load(rs + 2 * H); ghn[n] = __builtin_nontemporal_load(rs + 3 * H);
        hprev[n] = t > 0 ? out[((long long)(t - 1) * B + row) * H + unit[n]] : h_init[(long long)row * H + unit[n]];
        const float* dyp = dY + ((long long)t * B + row) * H + unit[n];
        dy[n] = __builtin_nontemporal_load(dyp);
      }
    }
    TSTAMP(0)   // operand prefetch issue
    if (t < T - 1) {
      if constexpr (LOC) wait_count_local(cnt + (size_t)(t + 1) * CSTRIDE, G, err);
      else wait_count(cnt + (size_t)(t + 1) * CSTRIDE, G, err);
      TSTAMP(1)   // poll + barrier
#ifdef B2T_TIMING
      if (t == T - 2 && threadIdx.x == 0) g_bwd_t[1][wgid] = (unsigned)wall_clock64();   // first hand-off complete: every workgroup of the row group is running
#endif
      const float* dgh = dG + (long long)(t + 1) * B * 4 * H;
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (H16) {
        const unsigned base = ((unsigned)(t + 1) % depth) * step_bytes + (unsigned)rg * npair * 1024u + (unsigned)lane * 16u;
        u32x4 v[NCB / 2];
#pragma unroll
        for (int p = 0; p < NCB / 2; ++p) {   // pairs beyond the operand meet zero weights: clamped to real (finite) data
          const unsigned pair = (unsigned)(wave * (NCB / 2) + p);
          v[p] = load_u4<LOC ? 0 : B2T_LOAD_AUX>(ring, base + (pair < npair ? pair : npair - 1u) * 1024u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NCB / 2; ++p) {
          const bf16x4 a0 = frag_lo(v[p]), a1 = frag_hi(v[p]);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (BF16) {
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, w[n][2 * p], acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, w[n][2 * p + 1], acc[n], 0, 0, 0);
            }
          }
        }
      } else {
      float4 v[NCB];
      issue_block_loads<NCB, LOC ? 0 : B2T_LOAD_AUX>(v, dgh, m0, B, 4 * H, wave * NCB * 16, 3 * H, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NCB / 2; ++p) {
        float4 a[2];
        transpose_pair(&stage[wave][((2 * p) % NSLOT) * SLOT_F], v[2 * p], v[2 * p + 1], a[0], a[1], lane);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int ci = 2 * p + h2;
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = mfma_chunk16<BF16>(a[h2], w[n][ci], acc[n]);
        }
      }
      }
      asm volatile("s_nop 0" :: "v"(acc[0][0]));
      TSTAMP(2)   // loads + MFMA
      float s[NT];
      cross_wave_reduce<NT>(red, acc, s, wave, lane);
      TSTAMP(3)   // reduce
#pragma unroll
      for (int n = 0; n < NT; ++n) carry[n] = s[n] + dzterm[n];
    } else if (dh_last && live) {
#pragma unroll
      for (int n = 0; n < NT; ++n) carry[n] = dh_last[(long long)row * H + unit[n]];
    }
    if (t < 0) {
      if (live) {
#pragma unroll
        for (int n = 0; n < NT; ++n) dh_init[(long long)row * H + unit[n]] = carry[n];
      }
      continue;
    }
    if (live) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float d = dy[n] + carry[n];
        const float dn = d * (1.0f - z[n]);
        const float dz = d * (hprev[n] - nv[n]);
        const float dn_pre = dn * (1.0f - nv[n] * nv[n]);
        const float dz_pre = dz * z[n] * (1.0f - z[n]);
        const float dr_pre = dn_pre * ghn[n] * r[n] * (1.0f - r[n]);
        const int lr = 4 * q + wave, cj = 16 * n + j;
        gs[(0 * 16 + lr) * TPN + cj] = dr_pre;
        gs[(1 * 16 + lr) * TPN + cj] = dz_pre;
        gs[(2 * 16 + lr) * TPN + cj] = dn_pre * r[n];
        gs[(3 * 16 + lr) * TPN + cj] = dn_pre;
        dzterm[n] = d * z[n];
      }
    }
    TSTAMP(4)   // gate gradients (waits for the prefetched operands)
    __syncthreads();
    TSTAMP(5)   // stage barrier
    if constexpr (H16) {
      // the peers read the ring: wave w < 3 stores gate array w of the tile as one pair of bf16 fragments (1 KB); the counter
      // moves as soon as these are acknowledged, the fp32 tile (for the GEMMs) follows unwaited
      if (wave < 3) {
        const float4 c0 = *reinterpret_cast<const float4*>(&gs[(wave * 16 + j) * TPN + 4 * q]);
        const float4 c1 = *reinterpret_cast<const float4*>(&gs[(wave * 16 + j) * TPN + 16 + 4 * q]);
        store_u4<(LOC ? B2T_LOC_RING_AUX : 16)>(ring, ((unsigned)t % depth) * step_bytes + ((unsigned)rg * npair + (unsigned)wave * G + (unsigned)tile) * 1024u + (unsigned)lane * 16u,
                      pack_frag_pair(c0, c1));
      }
      if constexpr (LOC) publish_count_local(cnt + (size_t)t * CSTRIDE);
      else publish_count(cnt + (size_t)t * CSTRIDE);
    }
    if constexpr (H16) {
      // the fp32 tiles (4 gate arrays x NT 16-unit tiles) by waves 1-3: wave 0 polls the next step's counter, and its returning
      // atomic waits for every store the wave has in flight
      const int r2 = lane >> 2;
      if (wave > 0 && m0 + r2 < B) {
#pragma unroll
        for (int idx = 0; idx < 4 * NT; ++idx) {
          if (idx % 3 != wave - 1) continue;
          const int ga = idx / NT, c4 = 16 * (idx % NT) + (lane & 3) * 4;
          store_f4<AUX>(dG + (long long)t * B * 4 * H, (unsigned)(((long long)(m0 + r2) * 4 * H + ga * H + j0 + c4) * 4),
                       *reinterpret_cast<const float4*>(&gs[(ga * 16 + r2) * TPN + c4]));
        }
      }
    } else {   // wave w writes gate array w of the tile: 64 x 16 B write-through stores per 16-unit tile
      const int r2 = lane >> 2;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int c4 = 16 * n + (lane & 3) * 4;
        if (m0 + r2 < B)
          store_f4<AUX>(dG + (long long)t * B * 4 * H,
                       (unsigned)(((long long)(m0 + r2) * 4 * H + wave * H + j0 + c4) * 4),
                       *reinterpret_cast<const float4*>(&gs[(wave * 16 + r2) * TPN + c4]));
      }
    }
    if constexpr (!H16) {
      if constexpr (LOC) publish_count_local(cnt + (size_t)t * CSTRIDE);
      else publish_count(cnt + (size_t)t * CSTRIDE);
    }
    TSTAMP(6)   // tile store + drain + publish
   }
  }
  finish_call(sync, pset);
#ifdef B2T_TIMING
  if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 17))
    for (int i = 0; i < 7; ++i) sync[8 + (blockIdx.x ? 8 : 0) + i] = (unsigned)(tacc[i] / (unsigned long long)T);
  if (threadIdx.x == 0) g_bwd_t[2][wgid] = (unsigned)wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------------------
// backward, W_hh^T slice in LDS (round 5; "paired" sweep).  The register-resident backward sweep above holds 96 registers of
// weights + 96 of operands in flight per lane: two of its workgroups fill a CU's register file, and the 768-block weight-gradient
// GEMMs and the sweeps then exclude each other (NOTES.md R4.1).  Here ONE 512-thread workgroup per CU owns 16 dh columns of TWO
// row groups (a "pair": rows [32 pair, 32 pair + 32)): its 96 KB slice of W_hh^T sits in LDS as ready-made MFMA B fragments
// (1 KB per 16-k chunk, read back with one ds_read_b128 per lane: conflict-free), the 3H-long contraction is split over the
// EIGHT waves, and every wave uses each B fragment twice -- once per row group.  Per lane: 96 registers of operands in flight
// (the latency buffer, NOTES.md R4.10c) + 8 accumulators + transients, so that a 128-register GEMM workgroup stays resident
// next to it on every CU (LDS: 96 KB weights + 18 KB transpose slots, reused for the eight-way sum, + 10 KB gate tiles = 124 KB
// of 160; the GEMM takes 33 KB).  Placement: pair p of a sweep of XCD set s (0..3, chosen by the caller per layer) runs on XCD
// (4 p + s) & 7, one workgroup per CU, tiles by ticket order; the XCD-local hand-off as above with ONE counter per pair and step.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_role(unsigned* tickets, int npairs, unsigned G, int set, int& pair, int& tile) {
  __shared__ int role[2];
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    int r = -1;
    for (int i = 0; i < npairs; ++i) if ((unsigned)((4 * i + set) & 7) == x) r = i;
    int tk = -1;
    if (r >= 0) {
      tk = (int)__hip_atomic_fetch_add(tickets + x, 1u, RLX_AGENT);
      if (tk >= (int)G) r = -1;
    }
    role[0] = r; role[1] = tk;
  }
  __syncthreads();
  pair = role[0]; tile = role[1];
  return pair >= 0;
}

constexpr int PAIR_STAGE_F = 2 * SLOT_F;          // floats of a wave's transpose slot pair (>= 2 tiles x 4 x 64 partial sums)
constexpr int PAIR_GS_F = 2 * 4 * 16 * TP;        // staged gate-gradient tiles [2 row groups][4 arrays][16 rows][TP]
template <int NCB8> constexpr size_t pair_lds_bytes() { return ((size_t)8 * NCB8 * 64 * 4 + 8 * PAIR_STAGE_F + PAIR_GS_F) * sizeof(float); }

template <int NCB8>   // 16-wide chunks of the 3H contraction per wave (8 waves: 3H <= 128 NCB8), even
__global__ __launch_bounds__(512) void gru_persist_bwd_pair_kernel(const float* __restrict__ dY, const float* __restrict__ dh_last,
                                                                    const float* __restrict__ reserve, const float* __restrict__ out,
                                                                    const float* __restrict__ h_init, const float* __restrict__ w_hh_t,
                                                                    float* dG, float* __restrict__ dh_init, int T, int B, int H,
                                                                    unsigned* sync, int set) {
  extern __shared__ __attribute__((aligned(16))) float pair_lds[];
  float4* wl = reinterpret_cast<float4*>(pair_lds);            // [8 waves][NCB8][64 lanes]: B fragments of the W_hh^T slice
  float* stage_all = pair_lds + 8 * NCB8 * 64 * 4;             // [8 waves][PAIR_STAGE_F]
  float* gs = stage_all + 8 * PAIR_STAGE_F;                    // [2][4][16][TP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __builtin_amdgcn_s_setprio(3);
  const unsigned G = (unsigned)H / 16u;
  const int j = lane & 15, q = lane >> 4;
  unsigned* err = sync;
  const unsigned pset = __hip_atomic_load(sync + 1, RLX_AGENT) & 1u;
  {
    unsigned* other = sync + 32 + (size_t)(1u - pset) * SETW;
    const int nthr = gridDim.x * gridDim.y * 512;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x; i < SETW; i += nthr) other[i] = 0u;
  }
  const int ngrp = (B + 15) / 16;
  int pair, tile;
  if (!pair_role(sync + 32 + (size_t)pset * SETW + (SETW - 16), (ngrp + 1) / 2, G, set, pair, tile)) { finish_call(sync, pset); return; }
  const int unit = tile * 16 + j;
  const int nch = 3 * H / 16;
  float4* wmine = wl + (size_t)wave * NCB8 * 64 + lane;      // written and read by the same lane of the same wave: no barrier
#pragma unroll
  for (int ci = 0; ci < NCB8; ++ci) {
    const int c = wave * NCB8 + ci;
    wmine[ci * 64] = c < nch ? *reinterpret_cast<const float4*>(w_hh_t + (long long)unit * 3 * H + c * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float* stage = stage_all + wave * PAIR_STAGE_F;
  const int sel = wave >> 2, wq = wave & 3;                   // this thread's element: row group `sel` of the pair, row 4 q + wq, column j
  const int m0a = 32 * pair, m0b = 32 * pair + 16;
  const int m0 = sel ? m0b : m0a;
  const int row = m0 + 4 * q + wq;
  const bool live = row < B;
  unsigned* cnt = sync + 32 + (size_t)pset * SETW + (size_t)(2 * pair) * T * CSTRIDE;   // the pair's counters: those of its first row group
  float dzterm = 0.f;
  const int r8 = lane >> 3, p8 = lane & 7;

  for (int t = T - 1; t >= -1; --t) {
    float r = 0.f, z = 0.f, nv = 0.f, ghn = 0.f, hprev = 0.f, dy = 0.f, carry = 0.f;
    if (live && t >= 0) {
      const float* rs = reserve + ((long long)t * B + row) * 4 * H + unit;
      r = __builtin_nontemporal_load(rs); z = __builtin_nontemporal_load(rs + H); nv = __builtin_nontemporal_load(rs + 2 * H); ghn = __builtin_nontemporal_load(rs + 3 * H);
      hprev = t > 0 ? out[((long long)(t - 1) * B + row) * H + unit] : h_init[(long long)row * H + unit];
      dy = __builtin_nontemporal_load(dY + ((long long)t * B + row) * H + unit);
    }
    if (t < T - 1) {
      wait_count_local(cnt + (size_t)(t + 1) * CSTRIDE, G, err);
      const float* dgh = dG + (long long)(t + 1) * B * 4 * H;
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      // all operand loads of BOTH row groups go out first, in the order they are consumed (pair of chunks p: row group a rows
      // 0-7, 8-15, then row group b), line-wise as in issue_block_loads; clamped, branch-free
      float4 v[2 * NCB8];
#pragma unroll
      for (int i = 0; i < 2 * NCB8; ++i) {
        const int p = i >> 2, s2 = (i >> 1) & 1, h8 = i & 1;
        const int rr = (s2 ? m0b : m0a) + h8 * 8 + r8, col = (wave * NCB8 + 2 * p) * 16 + p8 * 4;
        const long long off = (long long)(rr < B ? rr : B - 1) * 4 * H + (col < 3 * H ? col : 3 * H - 4);
        v[i] = load_f4<0>(dgh, (unsigned)(off * 4));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NCB8 / 2; ++p) {
        const float4 b0 = wmine[(2 * p) * 64], b1 = wmine[(2 * p + 1) * 64];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float4 a0, a1;
          transpose_pair(stage, v[4 * p + 2 * s2], v[4 * p + 2 * s2 + 1], a0, a1, lane);
          acc[s2] = mfma_chunk16<false>(a0, b0, acc[s2]);
          acc[s2] = mfma_chunk16<false>(a1, b1, acc[s2]);
        }
      }
      asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]));
      // eight-way sum through the waves' own slot pairs (free now: a wave's LDS operations execute in order)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) stage[(s2 * 4 + rr) * 64 + lane] = acc[s2][rr];
      __syncthreads();
      float sum[8];
#pragma unroll
      for (int sw = 0; sw < 8; ++sw) sum[sw] = stage_all[sw * PAIR_STAGE_F + (sel * 4 + wq) * 64 + lane];
      carry = (((sum[0] + sum[1]) + (sum[2] + sum[3])) + ((sum[4] + sum[5]) + (sum[6] + sum[7]))) + dzterm;
    } else if (dh_last && live) {
      carry = dh_last[(long long)row * H + unit];
    }
    if (t < 0) {
      if (live) dh_init[(long long)row * H + unit] = carry;
      continue;
    }
    if (live) {
      const float d = dy + carry;
      const float dn = d * (1.0f - z);
      const float dz = d * (hprev - nv);
      const float dn_pre = dn * (1.0f - nv * nv);
      const float dz_pre = dz * z * (1.0f - z);
      const float dr_pre = dn_pre * ghn * r * (1.0f - r);
      const int lr = 4 * q + wq;
      float* g0 = gs + sel * 4 * 16 * TP;
      g0[(0 * 16 + lr) * TP + j] = dr_pre;
      g0[(1 * 16 + lr) * TP + j] = dz_pre;
      g0[(2 * 16 + lr) * TP + j] = dn_pre * r;
      g0[(3 * 16 + lr) * TP + j] = dn_pre;
      dzterm = d * z;
    }
    __syncthreads();   // tiles staged; every wave has read its sums (the slot pairs are free for the next step's transposes)
    {   // wave w writes gate array w & 3 of row group w >> 2: 64 x 16 B write-through stores
      const int r2 = lane >> 2, c4 = (lane & 3) * 4;
      if (m0 + r2 < B)
        store_f4<B2T_LOC_ST_AUX>(dG + (long long)t * B * 4 * H, (unsigned)(((long long)(m0 + r2) * 4 * H + wq * H + tile * 16 + c4) * 4),
                                 *reinterpret_cast<const float4*>(&gs[((sel * 4 + wq) * 16 + r2) * TP + c4]));
    }
    publish_count_local(cnt + (size_t)t * CSTRIDE);
  }
  finish_call(sync, pset);
}

#ifdef B2T_TIMING
}  // namespace b2t
extern "C" int b2t_debug_bwd_times(unsigned* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(b2t::g_bwd_t), sizeof(unsigned) * 3 * 512) == hipSuccess ? 0 : 1;
}
namespace b2t {
#endif

size_t gru_persistent_sync_bytes(int T) { (void)T; return SYNC_WORDS * sizeof(unsigned) + RING_BYTES; }   // counters, then the bf16 hand-off ring

static int cu_count() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    n = p.multiProcessorCount;
  }
  return n;
}

// One persistent workgroup per CU: co-resident workgroups of concurrent sweeps on one CU gain little
// (measured: 2/CU -> 1.3x CU throughput) while every lock-step peer group then runs at the pace
// of its slowest member.  Requesting more than half of the 160 KiB LDS makes the dispatcher place concurrent
// sweeps (other layers of the pipelined plan) on different CUs instead of stacking them.
constexpr unsigned EXCLUSIVE_LDS = 84 * 1024;
template <typename K> static void want_exclusive(K kernel) {
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)EXCLUSIVE_LDS);
    done = true;
  }
}
static unsigned exclusive_lds() {
  static int v = -1;
  // opt-in: exclusive CUs make two concurrent sweeps scale perfectly in isolation (tools/bench_sweep.py) but starve
  // next to the 3-blocks-per-CU GEMMs of the pipelined step (a sweep workgroup then waits for 84 KB of LDS while the
  // GEMM keeps refilling 34 KB slots), which stalls its lock-step peers for milliseconds.
  if (v < 0) { const char* e = getenv("B2T_SWEEP_EXCLUSIVE"); v = (e && e[0] == '1') ? 1 : 0; }
  return v ? EXCLUSIVE_LDS : 0u;
}

// (Two row groups per workgroup -- half the workgroups, shared weights -- was measured 2x slower per sweep: the
// exposed latencies are the workgroup's own load/drain, not the peers'.  Removed.)
static int check_grid(int B, int H, int T, void* sync_ws, const char* what) {
  const int gx = H / 16, gy = (B + 15) / 16;
  if (!sync_ws) { set_error("%s: sync_ws is required in persistent mode", what); return 2; }
  if ((long long)gy * T * CSTRIDE > SETW - 16) {   // (the last 16 words of a set hold the XCD tickets of the local hand-off)
    set_error("%s: %d row groups x %d steps exceed the %d hand-off counters of one call", what, gy, T, SETW);
    return 2;
  }
  const int cus = cu_count();
  if (gx * gy > cus) {
    set_error("%s: persistent sweep needs %d co-resident workgroups but the device has %d CUs (use mode 0)", what,
              gx * gy, cus);
    return 4;
  }
  return 0;
}

// The XCD-local hand-off assumes what the hardware documents for this partition mode: the workgroups of a launch are dealt to
// the 8 XCDs round-robin.  Checked once per process with a probe launch (256 workgroups must land 32 per XCD on 8 XCDs);
// if it does not hold the flag is ignored and the sweeps use the device-scope hand-off.
__global__ void xcd_probe_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }
bool gru_xcd_dispatch_ok() {
  static int ok = -1;
  if (ok >= 0) return ok == 1;
  ok = 0;
  unsigned* d = nullptr;
  unsigned h[256];
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)) != hipSuccess) { (void)hipGetLastError(); return false; }
  hipLaunchKernelGGL(xcd_probe_kernel, dim3(256), dim3(64), 0, nullptr, d);
  const bool copied = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!copied) { (void)hipGetLastError(); return false; }
  int cnt[16] = {0};
  for (int i = 0; i < 256; ++i) { if (h[i] > 7u) return false; ++cnt[h[i]]; }
  for (int x = 0; x < 8; ++x) if (cnt[x] != 32) return false;
  ok = 1;
  return true;
}

// Largest H whose sweeps may hand off inside one XCD.  512 by default; bf16 operands with 32-unit workgroups make a row group of
// H = 768 24 workgroups -- it fits an XCD's 32 CUs, alone: a second sweep of the same parity waits, partly resident, for the
// first to finish (the plan's admission edges keep a third one out) -- B2T_GRU_LOCAL_MAXH opts in.
static int local_max_h(bool bf16, bool wide) {
  static const int env = getenv("B2T_GRU_LOCAL_MAXH") ? atoi(getenv("B2T_GRU_LOCAL_MAXH")) : 512;
  return (bf16 && wide) ? std::max(512, std::min(env, 1024)) : 512;
}

// bf16 sweeps with 32-unit workgroups hand their tiles over as bf16 MFMA fragments (gru_sync.h); B2T_HANDOFF16=0 (read per call:
// the tests compare the two forms in one process) selects the fp32 tiles every consumer transposes and rounds itself
static bool ring_handoff() {
  const char* e = getenv("B2T_HANDOFF16");
  return B2T_HANDOFF16 && !(e && e[0] == '0');
}

int gru_persistent_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                       float* reserve, int T, int B, int H, void* sync_ws, hipStream_t s, bool bf16,
                       bool wide, int local) {
  int rc = check_grid(B, H, T, sync_ws, "gru_layer_fwd");
  if (rc) return rc;
  if (wide && ((H % 32) != 0 || H > (bf16 ? 768 : 512))) { set_error("gru_layer_fwd: 32-unit workgroups need H %% 32 == 0 and H <= 512 (768 with bf16 operands)"); return 2; }
  const int G = wide ? H / 32 : H / 16, gy = (B + 15) / 16;
  // (H % 32: the local hand-off reads the tiles with ordinary loads, which may be served by the CU's vector cache -- invalidated at
  //  kernel start only: a 128-byte line must not hold rows of two time steps, i.e. a step's B x H floats must be whole lines.  With
  //  H = 48, B = 5 a line held the end of step t - 1 and the start of step t, and the sweep read stale values: found in round 4,
  //  tools/r4_local_check.py; such shapes use the device-scope hand-off.)
  // XCD-local hand-off (local = layer parity, -1 = off): a row group's G workgroups must fit one XCD next to those of a second
  // sweep of the same parity (32 CUs; two workgroups per CU with 16-unit workgroups, one with 32-unit ones: G <= 32 / 16)
  const bool loc = local >= 0 && H <= local_max_h(bf16, wide) && G <= ((wide && !bf16) ? 16 : 32) && gy <= 4 && H % 32 == 0 && gru_xcd_dispatch_ok();
  const dim3 grid = loc ? dim3(8 * G, 1) : dim3(G, gy), block(256);
  const int par = loc ? (local & 1) : 0;
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);   // must start zeroed once (allocation); self-cleaning afterwards
#define B2T_FWD_GO(NCH, BF, NTT, RG)                                                                                   \
  do {                                                                                                                 \
    bool launched = false;                                                                                             \
    {                                                                                                                  \
      if (loc) {                                                                                                       \
        hipLaunchKernelGGL((gru_persist_fwd_kernel<NCH, BF, NTT, true, RG>), grid, block, 0, s, gi, w_hh, b_hh, h_init, out, reserve, T, B, \
                           H, sync, par);                                                                              \
        launched = true;                                                                                               \
      }                                                                                                                \
    }                                                                                                                  \
    if (!launched) {                                                                                                   \
      want_exclusive(gru_persist_fwd_kernel<NCH, BF, NTT, false, RG>);                                                 \
      hipLaunchKernelGGL((gru_persist_fwd_kernel<NCH, BF, NTT, false, RG>), grid, block, exclusive_lds(), s, gi, w_hh, b_hh, h_init, out, reserve, \
                         T, B, H, sync, par);                                                                          \
    }                                                                                                                  \
  } while (0)
#define B2T_LAUNCH_FWD(NCH)                                                                                            \
  do {                                                                                                                 \
    if (bf16 && wide) { if (ring_handoff()) B2T_FWD_GO(NCH, true, 2, true); else B2T_FWD_GO(NCH, true, 2, false); }    \
    else if (wide) { if constexpr (NCH <= 8) B2T_FWD_GO(NCH, false, 2, false); }                                       \
    else if (bf16) B2T_FWD_GO(NCH, true, 1, false);                                                                    \
    else B2T_FWD_GO(NCH, false, 1, false);                                                                             \
  } while (0)
  if (H <= 128) B2T_LAUNCH_FWD(2);
  else if (H <= 256) B2T_LAUNCH_FWD(4);
  else if (H <= 512) B2T_LAUNCH_FWD(8);
  else if (H <= 768) B2T_LAUNCH_FWD(12);
  else if (H <= 1024) B2T_LAUNCH_FWD(16);
  else { set_error("gru_layer_fwd: H=%d > 1024 unsupported in persistent mode", H); return 2; }
#undef B2T_FWD_GO
#undef B2T_LAUNCH_FWD
  return check_hip(hipGetLastError(), "gru_layer_fwd (persistent)");
}

int gru_persistent_fwd_fused(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                             float* reserve, const float* w_ih2, const float* b_ih2, float* gi2, int T, int B, int H,
                             void* sync_ws, hipStream_t s, int local) {
  int rc = check_grid(B, H, T, sync_ws, "gru_layer_fwd_fused");
  if (rc) return rc;
  if (H > 512) { set_error("gru_layer_fwd_fused: H=%d > 512 unsupported (the two weight slices must fit a workgroup's registers + LDS)", H); return 2; }
  const int G = H / 16, gy = (B + 15) / 16;
  const bool loc = local >= 0 && G <= 32 && gy <= 4 && H % 32 == 0 && gru_xcd_dispatch_ok();
  const dim3 grid = loc ? dim3(8 * G, 1) : dim3(G, gy), block(256);
  const int par = loc ? (local & 1) : 0;
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
#define B2T_FUSED_GO(NCH, NLDS)                                                                                        \
  do {                                                                                                                 \
    if (loc) hipLaunchKernelGGL((gru_persist_fwd_fused_kernel<NCH, NLDS, true>), grid, block, 0, s, gi, w_hh, b_hh, h_init, out, reserve, \
                                w_ih2, b_ih2, gi2, T, B, H, sync, par);                                                \
    else hipLaunchKernelGGL((gru_persist_fwd_fused_kernel<NCH, NLDS, false>), grid, block, 0, s, gi, w_hh, b_hh, h_init, out, reserve, \
                            w_ih2, b_ih2, gi2, T, B, H, sync, par);                                                    \
  } while (0)
  if (H <= 128) B2T_FUSED_GO(2, 0);
  else if (H <= 256) B2T_FUSED_GO(4, 0);
  else B2T_FUSED_GO(8, 2);
#undef B2T_FUSED_GO
  return check_hip(hipGetLastError(), "gru_layer_fwd_fused (persistent)");
}

// The paired backward sweep (W_hh^T in LDS): usable where the XCD-local hand-off is (H % 32 == 0, H <= 512, B <= 64, round-robin
// dispatch verified) -- the caller falls back to gru_persistent_bwd otherwise.  set: 0..3, the XCD set {set, set + 4}.
bool gru_persistent_bwd_pair_ok(int B, int H) {
  return H % 32 == 0 && H <= 512 && (B + 15) / 16 <= 4 && gru_xcd_dispatch_ok();
}
int gru_persistent_bwd_pair(const float* dY, const float* dh_last, const float* reserve, const float* out, const float* h_init,
                            const float* w_hh_t, float* dG, float* dh_init, int T, int B, int H, void* sync_ws, hipStream_t s, int set) {
  int rc = check_grid(B, H, T, sync_ws, "gru_layer_bwd (paired)");
  if (rc) return rc;
  if (!gru_persistent_bwd_pair_ok(B, H)) { set_error("gru_layer_bwd (paired): needs H %% 32 == 0, H <= 512, B <= 64 and round-robin XCD dispatch"); return 2; }
  const int G = H / 16;
  const dim3 grid(8 * G, 1), block(512);
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
#define B2T_PAIR_GO(NCB8)                                                                                                          \
  do {                                                                                                                             \
    static bool attr = false;                                                                                                      \
    if (!attr) {                                                                                                                   \
      rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_persist_bwd_pair_kernel<NCB8>),                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds_bytes<NCB8>()), "gru_layer_bwd (paired): LDS size"); \
      if (rc) return rc;                                                                                                           \
      attr = true;                                                                                                                 \
    }                                                                                                                              \
    hipLaunchKernelGGL((gru_persist_bwd_pair_kernel<NCB8>), grid, block, pair_lds_bytes<NCB8>(), s, dY, dh_last, reserve, out, h_init, \
                       w_hh_t, dG, dh_init, T, B, H, sync, set & 3);                                                               \
  } while (0)
  const int per_wave = (3 * H / 16 + 7) / 8;
  if (per_wave <= 2) B2T_PAIR_GO(2);
  else if (per_wave <= 4) B2T_PAIR_GO(4);
  else if (per_wave <= 6) B2T_PAIR_GO(6);
  else if (per_wave <= 8) B2T_PAIR_GO(8);
  else if (per_wave <= 10) B2T_PAIR_GO(10);
  else B2T_PAIR_GO(12);
#undef B2T_PAIR_GO
  return check_hip(hipGetLastError(), "gru_layer_bwd (paired)");
}

int gru_persistent_bwd(const float* dY, const float* dh_last, const float* reserve, const float* out,
                       const float* h_init, const float* w_hh_t, float* dG, float* dh_init, int T, int B, int H,
                       void* sync_ws, hipStream_t s, bool bf16, bool wide, int local) {
  int rc = check_grid(B, H, T, sync_ws, "gru_layer_bwd");
  if (rc) return rc;
  if (wide && ((H % 32) != 0 || H > (bf16 ? 768 : 512))) { set_error("gru_layer_bwd: 32-unit workgroups need H %% 32 == 0 and H <= 512 (768 with bf16 operands)"); return 2; }
  const int G = wide ? H / 32 : H / 16, gy = (B + 15) / 16;
  const bool loc = local >= 0 && H <= local_max_h(bf16, wide) && G <= ((wide && !bf16) ? 16 : 32) && gy <= 4 && H % 32 == 0 && gru_xcd_dispatch_ok();   // see gru_persistent_fwd
  const dim3 grid = loc ? dim3(8 * G, 1) : dim3(G, gy), block(256);
  const int par = loc ? (local & 1) : 0;
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
#define B2T_BWD_GO(NCB, BF, NTT, RG)                                                                                   \
  do {                                                                                                                 \
    bool launched = false;                                                                                             \
    {                                                                                                                  \
      if (loc) {                                                                                                       \
        hipLaunchKernelGGL((gru_persist_bwd_kernel<NCB, BF, NTT, true, RG>), grid, block, 0, s, dY, dh_last, reserve, out, h_init, w_hh_t, dG, \
                           dh_init, T, B, H, sync, par);                                                               \
        launched = true;                                                                                               \
      }                                                                                                                \
    }                                                                                                                  \
    if (!launched) {                                                                                                   \
      want_exclusive(gru_persist_bwd_kernel<NCB, BF, NTT, false, RG>);                                                 \
      hipLaunchKernelGGL((gru_persist_bwd_kernel<NCB, BF, NTT, false, RG>), grid, block, exclusive_lds(), s, dY, dh_last, reserve, out, h_init, \
                         w_hh_t, dG, dh_init, T, B, H, sync, par);                                                     \
    }                                                                                                                  \
  } while (0)
#define B2T_LAUNCH_BWD(NCB)                                                                                            \
  do {                                                                                                                 \
    if (bf16 && wide) { if (ring_handoff()) B2T_BWD_GO(NCB, true, 2, true); else B2T_BWD_GO(NCB, true, 2, false); }    \
    else if (wide) { if constexpr (NCB <= 24) B2T_BWD_GO(NCB, false, 2, false); }                                      \
    else if (bf16) B2T_BWD_GO(NCB, true, 1, false);                                                                    \
    else B2T_BWD_GO(NCB, false, 1, false);                                                                             \
  } while (0)
  if (H <= 128) B2T_LAUNCH_BWD(6);
  else if (H <= 256) B2T_LAUNCH_BWD(12);
  else if (H <= 512) B2T_LAUNCH_BWD(24);
  else if (H <= 768) B2T_LAUNCH_BWD(36);
  else if (H <= 1024) B2T_LAUNCH_BWD(48);
  else { set_error("gru_layer_bwd: H=%d > 1024 unsupported in persistent mode", H); return 2; }
#undef B2T_BWD_GO
#undef B2T_LAUNCH_BWD
  return check_hip(hipGetLastError(), "gru_layer_bwd (persistent)");
}

}  // namespace b2t

// Error word of the last persistent sweep that used sync_ws (0 = clean, 1 = a bounded spin gave up).
extern "C" int b2t_gru_sync_status(const void* sync_ws, int T, int B, int* status_host, void* stream) {
  using namespace b2t;
  B2T_REQUIRE(sync_ws && status_host, "gru_sync_status: null argument");
  (void)T; (void)B;
  const size_t off = 0;  // word 0 of the sync workspace
  int rc = check_hip(hipMemcpyAsync(status_host, reinterpret_cast<const unsigned*>(sync_ws) + off, sizeof(int),
                                    hipMemcpyDeviceToHost, as_stream(stream)), "gru_sync_status: copy");
  if (rc) return rc;
  return check_hip(hipStreamSynchronize(as_stream(stream)), "gru_sync_status: sync");
}
