// gru_pipeline.hip — software-pipelined persistent GRU layer sweeps for gfx950 (mode 3).
//
// Same hand-off protocol and data layout as gru_persistent.hip (mode 1), different schedule.  Measured anatomy of a
// mode-1 step (forward, H=512, 1 workgroup per CU, ~9000 cycles): 2000 waiting for the peers' counter, 2000 until the
// first operand bytes arrive, 3100 MFMA, 700 reduce + gates, 1150 store + drain + publish: the workgroup computes for
// 40 % of a step and waits on memory latency for the rest, and a second workgroup on the same CU does not fill the gaps
// (its operand loads queue in front of the first one's publish store; measured 1.3x, not 2x).
//
// Here ONE workgroup owns the same 16 hidden units for R (2 or 4) row groups of 16 batch rows — the row groups are
// independent recurrences that share the register-resident W_hh slice — and works through the items
// (t, r) = (0,0) (0,1) .. (0,R-1) (1,0) .. in a depth-2 pipeline:
//     iteration k:  confirm the counter of item k+1 (its poll was issued one iteration ago), issue item k+1's operand
//                   loads and item k+2's poll,  THEN run item k's MFMAs / reduce / gates / publish.
// Every memory round trip of an item (poll, operand loads) therefore overlaps the previous item's MFMAs, and the
// publish->visible->poll->load chain of a row group (about 4500 cycles) has R-1 item times to complete before its
// result is needed.  With R = 4 (B = 64) the sweep is MFMA-bound: 32 workgroups per layer instead of 128, each busy.
// All issue blocks are branch-free (clamped addresses) and fenced with sched_barrier: a conditional load makes the
// compiler drain everything (vmcnt(0)) at the join, which is what serialised the earlier "two row groups per
// workgroup" attempt.
#include <stdlib.h>
#include "gru_cell.h"
#include "gru_sync.h"
#include "gru_issue.h"

namespace b2t {

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// (at most 256 registers, i.e. two waves per SIMD: a workgroup that needs a whole SIMD to itself can only start on a CU
// the GEMMs have completely left, and the sweep cannot take its first step before all of its workgroups run)
template <int NCH, int R>   // NCH: 16-wide K chunks per wave (H <= 64*NCH); R: row groups per workgroup (even)
__global__ __launch_bounds__(256, (NCH <= 8 ? 2 : 1)) void gru_pipe_fwd_kernel(const float* __restrict__ gi,
                                                              const float* __restrict__ w_hh,
                                                              const float* __restrict__ b_hh,
                                                              const float* __restrict__ h_init, float* out,
                                                              float* __restrict__ reserve, int T, int B, int H,
                                                              unsigned* sync) {
  static_assert(R == 4, "buffer parity of an item is r & 1; the deferred publish needs R >= 3");
  __shared__ __attribute__((aligned(16))) float red[4 * 3 * 4 * 64 + 16 * TP];
  float* hs = red + 4 * 3 * 4 * 64;   // staged h tile [16 rows][TP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifndef B2T_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);   // the sweep is the critical path: its waves issue ahead of co-resident GEMM waves
#endif
  const unsigned G = gridDim.x;
  const int j = lane & 15, q = lane >> 4;
  const int j0 = blockIdx.x * 16, unit = j0 + j;
  const int rg0 = blockIdx.y * R, nrg = (B + 15) / 16;
  unsigned* err = sync;
  const unsigned pset = __hip_atomic_load(sync + 1, RLX_AGENT) & 1u;
  {
    unsigned* other = counter_set(sync, 1u - pset);
    const int nthr = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < SETW; i += nthr) other[i] = 0u;
  }
  unsigned* cset = counter_set(sync, pset);
  const int nch = H / 16;

  float4 w[3][NCH];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    const int c = KCHUNK(wave, ci, NCH);
#pragma unroll
    for (int g = 0; g < 3; ++g)
      w[g][ci] = c < nch ? *reinterpret_cast<const float4*>(w_hh + ((long long)g * H + unit) * H + c * 16 + 4 * q)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float bhr = b_hh[unit], bhz = b_hh[H + unit], bhn = b_hh[2 * H + unit];

  // per row group: my output row (for the gates) and my operand row (for the MFMA A fragment), clamped into the batch
  int orow[R], arow[R];
  float hp[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int m0 = (rg0 + r) * 16;
    const int ro = m0 + 4 * q + wave, ra = m0 + j;
    orow[r] = ro < B ? ro : B - 1;
    arow[r] = ra < B ? ra : B - 1;
    hp[r] = h_init[(long long)orow[r] * H + unit];
  }
  // byte offsets of my 16-byte operand pieces inside one [B][H] slab (chunk part; the row part is added per item)
  unsigned coff[NCH];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    const int c = KCHUNK(wave, ci, NCH);
    coff[ci] = (unsigned)(((c < nch ? c : nch - 1) * 16 + 4 * q) * 4);
  }

  f32x4 abuf[2][NCH];    // operand fragments of the item in flight (parity = r & 1)
  float gbuf[2][3];      // its gi_r, gi_z, gi_n
  unsigned pvv = 0;      // raw poll word of the item after next (in flight)
  float sv[4] = {0.f, 0.f, 0.f, 0.f};   // (r, z, n, gh_n) of the previous item, stored one item late (see below)
  float hold[4] = {0.f, 0.f, 0.f, 0.f}; // the registers those stores read from
  long long sv_off = -1;

  // prologue: item (t=0, r=0) reads h_init, no counter involved; the "poll" of item (0,1) is a dummy
  {
    const u32x4s rs0 = make_rsrc(h_init);
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) issue_load_sc1_x4(abuf[0][ci], rs0, (unsigned)arow[0] * (unsigned)H * 4u + coff[ci]);
    const float* g3 = gi + ((long long)0 * B + orow[0]) * 3 * H + unit;
    issue_load_f32(gbuf[0][0], g3); issue_load_f32(gbuf[0][1], g3 + H); issue_load_f32(gbuf[0][2], g3 + 2 * H);
    issue_poll(pvv, cset);
  }
  bool pv_need = false;
  const unsigned* pv_ptr = cset;
  drain_vm();   // registers written by the assembly loads must not be touched (or reallocated) before they land
#ifdef B2T_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PSTAMP(i) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; }
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#else
#define PSTAMP(i)
#endif

  unsigned* pub_ptr = nullptr;   // counter of the previous item (its tile is still in LDS, unpublished); wave-uniform
  int pub_m0 = 0, pub_t = 0;


  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int P = r & 1;
      // ---- next item (rn, tn) and the one after (r2, t2) --------------------------------------------------------
      const int rn = (r + 1) % R, tn = (r + 1 < R) ? t : t + 1;
      const int r2 = (r + 2) % R, t2 = (r + 2 < R) ? t : t + 1;
      // (a) this item's fragments, its gi values and the poll of the next item must have landed.  They were issued
      // most of an item ago, and so was wave 0's counter increment: a full drain.  (vmcnt(n > 0) is NOT usable to
      // wait for a load that has a store or a no-return atomic behind or in front of it: those retire out of order
      // with respect to loads, so "at most n outstanding" does not say which ones.  An earlier version did, and read
      // tiles that were not published yet -- a few elements per thousand steps off by 1e-3.)
      drain_vm();
      keep_until_here(hold[0]); keep_until_here(hold[1]); keep_until_here(hold[2]); keep_until_here(hold[3]);
      after_wait(pvv);
#pragma unroll
      for (int ci = 0; ci < NCH; ++ci) after_wait(abuf[P][ci]);
      after_wait(gbuf[P][0]); after_wait(gbuf[P][1]); after_wait(gbuf[P][2]);
      PSTAMP(0)   // wait for the prefetch
      {
        const unsigned pv = __builtin_amdgcn_readfirstlane(pvv);
        if (pv_need && pv < G) poll_until(pv_ptr, G, pv, err);
      }
      PSTAMP(1)   // blocking re-poll (normally nothing)
      // (b) the previous item's h tile goes out first (wave 0), then this item's MFMAs run with the NEXT item's loads
      // sprinkled between them.  The CU's texture path accepts the 64-byte pieces of these loads at ~16 B/clock
      // (measured: 32 KB of fragments = ~2000 cycles, whatever the cache policy or locality), and a wave that issues
      // them back to back just stalls at issue for that long; fed a couple at a time between groups of MFMAs they
      // cost nothing.  Clamped addresses: past the end they re-read the last step, results unused.
      f32x4 tv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wave == 0 && pub_ptr != nullptr) {
        const int r4 = lane >> 2, c4 = (lane & 3) * 4;
        tv[0] = hs[r4 * TP + c4]; tv[1] = hs[r4 * TP + c4 + 1]; tv[2] = hs[r4 * TP + c4 + 2]; tv[3] = hs[r4 * TP + c4 + 3];
        if (pub_m0 + r4 < B) issue_store_sc1_x4(make_rsrc(out + (long long)pub_t * B * H),
                                                (unsigned)(((long long)(pub_m0 + r4) * H + j0 + c4) * 4), tv);
      }
      const int tl = tn < T ? tn : T - 1;
      const u32x4s rsn = make_rsrc(tl > 0 ? out + (long long)(tl - 1) * B * H : h_init);
      const unsigned rown = (unsigned)arow[rn] * (unsigned)H * 4u;
      const float* g3 = gi + ((long long)tl * B + orow[rn]) * 3 * H + unit;
      float* rsv = reserve + (sv_off >= 0 ? sv_off : 0);
      const bool do_rsv = sv_off >= 0 && reserve;
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(2)   // tile store issue + address set-up

      // (c) item (t, r): recurrent product from the fragments loaded one iteration ago
      f32x4 acc[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int LPG = NCH >= 8 ? 2 : 1;   // fragment loads per MFMA group: all of them go out in the first half
#pragma unroll
      for (int ci = 0; ci < NCH; ++ci) {
#pragma unroll
        for (int k = 0; k < LPG; ++k) {
          const int li = ci * LPG + k;
          if (li < NCH) issue_load_sc1_x4(abuf[P ^ 1][li], rsn, rown + coff[li]);
        }
        if (ci * LPG >= NCH && ci * LPG < NCH + 3) issue_load_f32(gbuf[P ^ 1][ci * LPG - NCH], g3 + (long long)(ci * LPG - NCH) * H);
        if (LPG == 2 && ci * LPG + 1 >= NCH && ci * LPG + 1 < NCH + 3)
          issue_load_f32(gbuf[P ^ 1][ci * LPG + 1 - NCH], g3 + (long long)(ci * LPG + 1 - NCH) * H);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][0], w[g][ci].x, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][1], w[g][ci].y, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][2], w[g][ci].z, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][3], w[g][ci].w, acc[g], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LPG == 1) {   // small H: the gi loads did not fit in the loop above
        issue_load_f32(gbuf[P ^ 1][0], g3); issue_load_f32(gbuf[P ^ 1][1], g3 + H); issue_load_f32(gbuf[P ^ 1][2], g3 + 2 * H);
      }
      // the previous item's gate values (saved for the backward sweep; nothing in this sweep waits for them)
      hold[0] = sv[0]; hold[1] = sv[1]; hold[2] = sv[2]; hold[3] = sv[3];   // untouched until the next drain, see (a)
      if (do_rsv) {
        issue_store_f32(rsv, hold[0]); issue_store_f32(rsv + H, hold[1]); issue_store_f32(rsv + 2 * H, hold[2]);
        issue_store_f32(rsv + 3 * H, hold[3]);
      }
      asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(3)   // MFMA
      // (c2) late publish of the PREVIOUS item: its tile store went out a whole MFMA phase ago; a full drain proves the
      // write-through acknowledged (the loads issued under the MFMAs are at least half a phase old).  Then the poll of the item
      // after next: counter (rg0 + r2, t2 - 1); not needed for t2 == 0, past the end, or row groups beyond the batch
      // (the load is still issued, branch-free, on a valid word).  The counter increment goes out behind the poll.
      if (wave == 0 && pub_ptr != nullptr) drain_vm();   // the tile store AND the loads issued under the MFMAs (see (a))
      keep_until_here(tv);
      {
        pv_need = (t2 > 0) && (t2 < T) && (rg0 + r2 < nrg);
        const int tc = t2 > 0 ? (t2 < T ? t2 - 1 : T - 1) : 0;
        const int rc = (rg0 + r2 < nrg) ? rg0 + r2 : nrg - 1;
        pv_ptr = cset + (size_t)rc * T + tc;
        issue_poll(pvv, pv_ptr);
      }
      if (wave == 0 && pub_ptr != nullptr) {
        if (lane == 0) __hip_atomic_fetch_add(pub_ptr, 1u, RLX_AGENT);
      }
      float gh[3];
      cross_wave_reduce<3>(red, acc, gh, wave, lane);
      PSTAMP(4)   // late publish + poll issue + reduce
      // (d) gates, stage, store the tile (published after the next item's MFMAs)
      const int m0 = (rg0 + r) * 16;
      const bool live = (m0 + 4 * q + wave) < B;
      const float ghn = gh[2] + bhn;
      const float rr = fast_sigmoid(gbuf[P][0] + gh[0] + bhr);
      const float zz = fast_sigmoid(gbuf[P][1] + gh[1] + bhz);
      const float nn = fast_tanh(gbuf[P][2] + rr * ghn);
      const float h = (1.0f - zz) * nn + zz * hp[r];
      hs[(4 * q + wave) * TP + j] = h;
      hp[r] = h;
      sv[0] = rr; sv[1] = zz; sv[2] = nn; sv[3] = ghn;
      sv_off = live ? ((long long)t * B + (m0 + 4 * q + wave)) * 4 * H + unit : -1;
      PSTAMP(5)   // gates
      __syncthreads();   // tile staged; also fences `red` for the next item
      PSTAMP(6)   // stage barrier
      pub_ptr = (m0 < B) ? cset + (size_t)(rg0 + r) * T + t : nullptr;   // stored and published during the next item
      pub_m0 = m0; pub_t = t;
      PSTAMP(7)
    }
  }
  // the last item's tile and publish, then the last reserve values
  drain_vm();
  if (wave == 0 && pub_ptr != nullptr) {
    const int r4 = lane >> 2, c4 = (lane & 3) * 4;
    if (pub_m0 + r4 < B)
      store_f4<PAUX>(out + (long long)pub_t * B * H, (unsigned)(((long long)(pub_m0 + r4) * H + j0 + c4) * 4),
                     *reinterpret_cast<const float4*>(&hs[r4 * TP + c4]));
    drain_vm();
    if (lane == 0) __hip_atomic_fetch_add(pub_ptr, 1u, RLX_AGENT);
  }
#ifdef B2T_TIMING
  if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 17))
    for (int i = 0; i < 8; ++i) sync[8 + (blockIdx.x ? 8 : 0) + i] = (unsigned)(tacc[i] / (unsigned long long)(T * R));
#endif
  if (sv_off >= 0 && reserve) {
    float* rs = reserve + sv_off;
    rs[0] = sv[0]; rs[H] = sv[1]; rs[2 * H] = sv[2]; rs[3 * H] = sv[3];
  }
  finish_call(sync, pset);
}

// ---------------------------------------------------------------------------------------------------
// backward: items (t, r), t = T-1 .. -1.  Item (t, r) contracts dGh_{t+1} of row group r (ready when counter
// (r, t+1) reaches G) into the carry, then forms and publishes the gate gradients of step t (t >= 0) or writes the
// gradient of the initial state (t = -1).
// ---------------------------------------------------------------------------------------------------
template <int NCB, int R>
__global__ __launch_bounds__(256, 1) void gru_pipe_bwd_kernel(const float* __restrict__ dY,
                                                              const float* __restrict__ dh_last,
                                                              const float* __restrict__ reserve,
                                                              const float* __restrict__ out,
                                                              const float* __restrict__ h_init,
                                                              const float* __restrict__ w_hh_t, float* dG,
                                                              float* __restrict__ dh_init, int T, int B, int H,
                                                              unsigned* sync) {
  static_assert(R == 4, "buffer parity of an item is r & 1; the deferred publish needs R >= 3");
  __shared__ __attribute__((aligned(16))) float red[4 * 4 * 64 + 4 * 16 * TP];
  float* gs = red + 4 * 4 * 64;   // staged gate-gradient tiles [4 arrays][16 rows][TP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifndef B2T_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);   // the sweep is the critical path: its waves issue ahead of co-resident GEMM waves
#endif
  const unsigned G = gridDim.x;
  const int j = lane & 15, q = lane >> 4;
  const int j0 = blockIdx.x * 16, unit = j0 + j;
  const int rg0 = blockIdx.y * R, nrg = (B + 15) / 16;
  unsigned* err = sync;
  const unsigned pset = __hip_atomic_load(sync + 1, RLX_AGENT) & 1u;
  {
    unsigned* other = counter_set(sync, 1u - pset);
    const int nthr = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < SETW; i += nthr) other[i] = 0u;
  }
  unsigned* cset = counter_set(sync, pset);
  const int nch = 3 * H / 16;

  float4 w[NCB];
#pragma unroll
  for (int ci = 0; ci < NCB; ++ci) {
    const int c = KCHUNK(wave, ci, NCB);
    w[ci] = c < nch ? *reinterpret_cast<const float4*>(w_hh_t + (long long)unit * 3 * H + c * 16 + 4 * q)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int orow[R], arow[R];
  float dzterm[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int m0 = (rg0 + r) * 16;
    const int ro = m0 + 4 * q + wave, ra = m0 + j;
    orow[r] = ro < B ? ro : B - 1;
    arow[r] = ra < B ? ra : B - 1;
    dzterm[r] = 0.f;
  }
  unsigned coff[NCB];
#pragma unroll
  for (int ci = 0; ci < NCB; ++ci) {
    const int c = KCHUNK(wave, ci, NCB);
    coff[ci] = (unsigned)(((c < nch ? c : nch - 1) * 16 + 4 * q) * 4);
  }

  f32x4 abuf[2][NCB];    // dGh_{t+1} fragments of the item in flight
  float ebuf[2][6];      // its elementwise operands: r, z, n, gh_n, h_{t-1}, dY_t
  unsigned pvv = 0;

  // Elementwise operands of item (t, r) do not depend on the recurrence: they are prefetched with the fragments.
  auto issue_elem = [&](float (&e)[6], int t, int ro) {
    const int tt = t >= 0 ? t : 0;
    const float* rs = reserve + ((long long)tt * B + ro) * 4 * H + unit;
    issue_load_f32(e[0], rs); issue_load_f32(e[1], rs + H); issue_load_f32(e[2], rs + 2 * H); issue_load_f32(e[3], rs + 3 * H);
    const float* hsrc = tt > 0 ? out + ((long long)(tt - 1) * B + ro) * H : h_init + (long long)ro * H;
    issue_load_f32(e[4], hsrc + unit);
    issue_load_f32(e[5], dY + ((long long)tt * B + ro) * H + unit);
  };

  // prologue: item (T-1, 0) has no contraction (carry = dh_last); fragments are loaded from a valid slab and unused
  {
    const u32x4s rs0 = make_rsrc(dG + (long long)(T - 1) * B * 4 * H);
#pragma unroll
    for (int ci = 0; ci < NCB; ++ci)
      issue_load_sc1_x4(abuf[0][ci], rs0, (unsigned)arow[0] * (unsigned)(4 * H) * 4u + coff[ci]);
    issue_elem(ebuf[0], T - 1, orow[0]);
    issue_poll(pvv, cset);
  }
  bool pv_need = false;
  const unsigned* pv_ptr = cset;
  drain_vm();   // registers written by the assembly loads must not be touched (or reallocated) before they land

  for (int t = T - 1; t >= -1; --t) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int P = r & 1;
      const int rn = (r + 1) % R, tn = (r + 1 < R) ? t : t - 1;
      const int r2 = (r + 2) % R, t2 = (r + 2 < R) ? t : t - 1;
      // (a) this item's fragments and operands and the next poll were drained at the end of the previous item (its
      // publish, or the explicit drain of the final t = -1 items); only thread 0's counter increment may be in flight
      after_wait(pvv);
#pragma unroll
      for (int ci = 0; ci < NCB; ++ci) after_wait(abuf[P][ci]);
#pragma unroll
      for (int i = 0; i < 6; ++i) after_wait(ebuf[P][i]);
      {
        const unsigned pv = __builtin_amdgcn_readfirstlane(pvv);
        if (pv_need && pv < G) poll_until(pv_ptr, G, pv, err);
      }
      // (b) prefetch item (tn, rn): it contracts slab tn + 1 (for tn = T-1, or past the end, a valid slab is read
      // and ignored); then the poll of item (t2, r2): counter (r2, t2 + 1) unless t2 == T-1 or past the end.
      {
        const int ts = (tn + 1 <= T - 1) ? (tn + 1 >= 0 ? tn + 1 : 0) : T - 1;
        const u32x4s rs = make_rsrc(dG + (long long)ts * B * 4 * H);
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci)
          issue_load_sc1_x4(abuf[P ^ 1][ci], rs, (unsigned)arow[rn] * (unsigned)(4 * H) * 4u + coff[ci]);
        issue_elem(ebuf[P ^ 1], tn, orow[rn]);
        pv_need = (t2 < T - 1) && (t2 >= -1) && (rg0 + r2 < nrg);
        const int tc = (t2 + 1 <= T - 1) ? (t2 + 1 >= 0 ? t2 + 1 : 0) : T - 1;
        const int rc = (rg0 + r2 < nrg) ? rg0 + r2 : nrg - 1;
        pv_ptr = cset + (size_t)rc * T + tc;
        issue_poll(pvv, pv_ptr);
      }
      __builtin_amdgcn_sched_barrier(0);

      const int m0 = (rg0 + r) * 16;
      const int row = m0 + 4 * q + wave;
      const bool live = row < B;
      float carry;
      {
        // two accumulators halve the dependent-MFMA chain; their sum is formed before the cross-wave reduction
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci) {
          acc[ci & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][0], w[ci].x, acc[ci & 1], 0, 0, 0);
          acc[ci & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][1], w[ci].y, acc[ci & 1], 0, 0, 0);
          acc[ci & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][2], w[ci].z, acc[ci & 1], 0, 0, 0);
          acc[ci & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[P][ci][3], w[ci].w, acc[ci & 1], 0, 0, 0);
        }
        f32x4 sum[1];
        sum[0] = acc[0] + acc[1];
        float s1[1];
        cross_wave_reduce<1>(red, sum, s1, wave, lane);
        carry = s1[0] + dzterm[r];
      }
      if (t == T - 1) carry = (dh_last && live) ? dh_last[(long long)row * H + unit] : 0.f;
      if (t < 0) {
        if (live) dh_init[(long long)row * H + unit] = carry;
        drain_vm();
        __syncthreads();   // keeps `red` of this item apart from the next item's partials
        continue;
      }
      {
        const float rr = ebuf[P][0], zz = ebuf[P][1], nn = ebuf[P][2], ghn = ebuf[P][3], hprev = ebuf[P][4];
        const float d = ebuf[P][5] + carry;
        const float dn = d * (1.0f - zz);
        const float dz = d * (hprev - nn);
        const float dn_pre = dn * (1.0f - nn * nn);
        const float dz_pre = dz * zz * (1.0f - zz);
        const float dr_pre = dn_pre * ghn * rr * (1.0f - rr);
        const int lr = 4 * q + wave;
        gs[(0 * 16 + lr) * TP + j] = dr_pre;
        gs[(1 * 16 + lr) * TP + j] = dz_pre;
        gs[(2 * 16 + lr) * TP + j] = dn_pre * rr;
        gs[(3 * 16 + lr) * TP + j] = dn_pre;
        dzterm[r] = d * zz;
      }
      __syncthreads();
      {   // wave w writes gate array w of the tile: 64 x 16 B write-through stores
        const int r2s = lane >> 2, c4 = (lane & 3) * 4;
        if (m0 + r2s < B)
          store_f4<PAUX>(dG + (long long)t * B * 4 * H,
                         (unsigned)(((long long)(m0 + r2s) * 4 * H + wave * H + j0 + c4) * 4),
                         *reinterpret_cast<const float4*>(&gs[(wave * 16 + r2s) * TP + c4]));
      }
      drain_vm();   // the prefetch went out a whole MFMA phase ago: this waits for the tile's write-through only
      __syncthreads();
      if (threadIdx.x == 0 && m0 < B) __hip_atomic_fetch_add(cset + (size_t)(rg0 + r) * T + t, 1u, RLX_AGENT);
    }
  }
  wait_vm<0>();
  finish_call(sync, pset);
}

// Row groups per workgroup: always 4.  The publish of an item is deferred by one item, so the item that consumes
// it must be at least two items later: R >= 3 (with R = 2 a workgroup would wait for its own, not yet issued, counter
// increment).  Row groups beyond the batch are processed as dead items (no polls, no stores).
static int pipe_rows(int B) { (void)B; return 4; }

// B2T_PIPE_EXCLUSIVE_KB=n: request n KB of (unused) dynamic LDS per workgroup so that no second sweep workgroup fits on
// the CU (GEMM workgroups, 34 KB each, still do when n <= 92).
static unsigned pipe_extra_lds() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("B2T_PIPE_EXCLUSIVE_KB"); v = e ? atoi(e) * 1024 : 0; }
  return (unsigned)v;
}
template <typename K> static void pipe_allow_lds(K kernel) {
  // (every call: instances with the same signature share this function, a flag here would cover only the first one)
  if (pipe_extra_lds() > 0)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe_extra_lds());
}

static int pipe_check(int B, int H, int T, void* sync_ws, const char* what) {
  const int nrg = (B + 15) / 16;
  if (!sync_ws) { set_error("%s: sync_ws is required in persistent mode", what); return 2; }
  if ((long long)nrg * T > SETW) {
    set_error("%s: %d row groups x %d steps exceed the %d hand-off counters of one call", what, nrg, T, SETW);
    return 2;
  }
  if (nrg < 3) { set_error("%s: the pipelined sweep pays off from 3 row groups (B > 32); use mode 1", what); return 4; }
  return 0;
}

int gru_pipeline_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                     float* reserve, int T, int B, int H, void* sync_ws, hipStream_t s) {
  int rc = pipe_check(B, H, T, sync_ws, "gru_layer_fwd");
  if (rc) return rc;
  const int R = pipe_rows(B), nrg = (B + 15) / 16;
  const dim3 grid(H / 16, (nrg + R - 1) / R), block(256);
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
#define B2T_LAUNCH(NCH)                                                                                                \
  do {                                                                                                                 \
    pipe_allow_lds(gru_pipe_fwd_kernel<NCH, 4>);                                                                       \
    hipLaunchKernelGGL((gru_pipe_fwd_kernel<NCH, 4>), grid, block, pipe_extra_lds(), s, gi, w_hh, b_hh, h_init, out, reserve, T, B, H, sync); \
  } while (0)
  if (H <= 128) B2T_LAUNCH(2);
  else if (H <= 256) B2T_LAUNCH(4);
  else if (H <= 512) B2T_LAUNCH(8);
  else if (H <= 768) B2T_LAUNCH(12);
  else { set_error("gru_layer_fwd: H=%d > 768 unsupported in pipelined mode", H); return 4; }
#undef B2T_LAUNCH
  return check_hip(hipGetLastError(), "gru_layer_fwd (pipelined)");
}

int gru_pipeline_bwd(const float* dY, const float* dh_last, const float* reserve, const float* out,
                     const float* h_init, const float* w_hh_t, float* dG, float* dh_init, int T, int B, int H,
                     void* sync_ws, hipStream_t s) {
  int rc = pipe_check(B, H, T, sync_ws, "gru_layer_bwd");
  if (rc) return rc;
  const int R = pipe_rows(B), nrg = (B + 15) / 16;
  const dim3 grid(H / 16, (nrg + R - 1) / R), block(256);
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
#define B2T_LAUNCH(NCB)                                                                                                \
  do {                                                                                                                 \
    pipe_allow_lds(gru_pipe_bwd_kernel<NCB, 4>);                                                                       \
    hipLaunchKernelGGL((gru_pipe_bwd_kernel<NCB, 4>), grid, block, pipe_extra_lds(), s, dY, dh_last, reserve, out, h_init, w_hh_t, dG, \
                       dh_init, T, B, H, sync);                                                                        \
  } while (0)
  if (H <= 128) B2T_LAUNCH(6);
  else if (H <= 256) B2T_LAUNCH(12);
  else if (H <= 512) B2T_LAUNCH(24);
  else { set_error("gru_layer_bwd: H=%d > 512 unsupported in pipelined mode", H); return 4; }
#undef B2T_LAUNCH
  return check_hip(hipGetLastError(), "gru_layer_bwd (pipelined)");
}

}  // namespace b2t
