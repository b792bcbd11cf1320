// gru_stack.hip — the whole GRU stack as ONE persistent launch per direction (mode 4).
//
// What the measurements of the per-layer sweeps said: a sweep workgroup computes for ~40 % of a step and
// waits on memory round trips for the rest; a second workgroup on the CU recovers little of that; concurrent kernels
// are placed independently, so five sweeps plus the projection GEMMs crowd some CUs and every sweep runs at the pace
// of its most crowded CU (10-14 us per step instead of 3.7-4.0); and a chunk of a layer cannot start before the
// projection GEMM of that chunk has run (launch + GEMM + two event hops between any two dependent sweeps).
//
// Here the L layers are ONE grid: workgroup (slice, layer, row block) owns 16 hidden units of one layer for FOUR row
// groups of 16 batch rows, with 8 waves:
//   waves 0-3  hold the W_hh slice (3 x 16 rows x H, 96 registers per lane) and form gh = h_{t-1} W_hh^T,
//   waves 4-7  hold the W_ih slice and form gi = x_t W_ih^T from the tile the layer BELOW has just published
//              (layer 0 reads gi from the projection GEMM that ran before the launch),
// so that no GEMM and no launch sits between two dependent time steps anywhere: layer l trails layer l-1 by one or
// two items.  The workgroup works through the items (t, r) in a depth-2 software pipeline (the schedule of
// gru_pipeline.hip: the next item's operand loads and the poll of the one after fly under the current item's MFMAs, the
// publish of an item is deferred into the next one), it is MFMA-bound, and it asks for enough LDS that nothing else
// fits on its CU.  (H/16) * L workgroups (160 at H = 512, L = 5) must be resident at once.
//
// Hand-off: as in gru_persistent.hip (sc1 write-through tile stores, drain, one agent-scope counter per (layer, row
// group, step)); the inter-layer dropout (nn.GRU, rnn_model.py:70) is applied by the PRODUCER, which publishes the
// kept/scaled copy of its tile next to the plain one (same Philox stream as b2t_dropout_f32).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "gru_cell.h"
#include "gru_sync.h"
#include "gru_issue.h"

namespace b2t {

constexpr int STACK_MAXL = B2T_STACK_MAX_LAYERS;
constexpr int STACK_LDS_FLOATS = 3 * (4 * 3 * 4 * 64) + 5 * 16 * TP + 8 * 4 * SLOT_F;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

struct StackFwd {
  const float* gi0;
  const float* w_hh[STACK_MAXL];
  const float* w_ih[STACK_MAXL];
  const float* b_hh[STACK_MAXL];
  const float* b_ih[STACK_MAXL];
  float* out[STACK_MAXL];
  float* outd[STACK_MAXL];
  float* reserve[STACK_MAXL];
  unsigned long long seed[STACK_MAXL];
  float drop_p, drop_scale;
  int T, B, H;
};

// LDS-only barrier (no vmcnt drain: the prefetched loads stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NCH: 16-wide K chunks per wave (H <= 64*NCH); EXACT: H == 64*NCH, the chunk offsets are instruction immediates
//
// Schedule of one iteration k (item k = (t, r) for the recurrent waves; the projection waves work ONE ITEM AHEAD):
//   recurrent waves 0-3:  MFMAs of item k ...................... | partials -> LDS | B1 | reduce + gates(k) -> tiles | B2
//   projection waves 4-7: first 3/8 of the MFMAs of item k+1 ... |                  B1 | rest of the MFMAs, partials  | B2
// so the two waves of a SIMD share the MFMA pipe before B1 and the gate phase (LDS reads, exp, rcp: ~2000 cycles with
// no MFMA in it) runs under the projection's remaining MFMAs.  gi partials are double buffered (written in iteration
// k-1 for item k).  Both groups prefetch the operands of their next item under their MFMAs and poll two items ahead.
template <int NCH, bool EXACT>
__global__ __launch_bounds__(512, 1) void gru_stack_fwd_kernel(const StackFwd A, unsigned* sync) {
  constexpr int R = 4;
  constexpr int NAP = NCH >= 8 ? 1 : 0;   // projection chunk PAIRS contracted before B1 (what fits the recurrent waves' gaps)
  constexpr int PART = 4 * 3 * 4 * 64;
  extern __shared__ __attribute__((aligned(16))) float red[];   // STACK_LDS_FLOATS in use + padding (see the launcher)
  float* redr = red;               // recurrent partials [4 waves][3 gates][4][64]
  float* redp = red + PART;        // projection partials, [2] of the same
  float* hs = red + 3 * PART;      // staged tiles [h, r, z, n, gh_n][16 rows][TP]
  float* tslot = hs + 5 * 16 * TP + (threadIdx.x >> 6) * 4 * SLOT_F;   // this wave's two transpose slot pairs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: everything selected by it stays scalar
  const int half = wave >> 2, kw = wave & 3;
  const bool proj = half == 1;
  const int l = blockIdx.y;
  const int T = A.T, B = A.B, H = A.H;
  if (proj) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3);   // the recurrent waves are the critical path
  const unsigned G = gridDim.x;
  const int j = lane & 15, q = lane >> 4;
  const int j0 = blockIdx.x * 16, unit = j0 + j;
  const int rg0 = blockIdx.z * R, nrg = (B + 15) / 16;
  unsigned* err = sync;
  const unsigned pset = __hip_atomic_load(sync + 1, RLX_AGENT) & 1u;
  {
    unsigned* other = counter_set(sync, 1u - pset);
    const int nthr = gridDim.x * gridDim.y * gridDim.z * 512;
    for (int i = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 512 + threadIdx.x; i < SETW; i += nthr)
      other[i] = 0u;
  }
  unsigned* cset = counter_set(sync, pset);
  const int nch = H / 16;
  const bool feeds = proj && l > 0;        // this wave contracts the lower layer's output
  const bool works = !proj || l > 0;       // layer 0's projection was done by a GEMM: its waves 4-7 only store the reserve
  const int ldep = feeds ? l - 1 : l;      // layer whose counters this wave polls
  const int ioff = proj ? 1 : 0;           // the projection waves run one item ahead

  const float* wsrc = feeds ? A.w_ih[l] : A.w_hh[l];
  float4 w[3][NCH];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    const int c = KCHUNK(kw, ci, NCH);
#pragma unroll
    for (int g = 0; g < 3; ++g)
      w[g][ci] = (works && c < nch) ? *reinterpret_cast<const float4*>(wsrc + ((long long)g * H + unit) * H + c * 16 + 4 * q)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float* bh = A.b_hh[l];
  float br = bh[unit], bz = bh[H + unit], bin = 0.f;   // r, z: b_hh + b_ih (layer 0: b_ih is inside gi0)
  const float bhn = bh[2 * H + unit];
  if (l > 0) { const float* bi = A.b_ih[l]; br += bi[unit]; bz += bi[H + unit]; bin = bi[2 * H + unit]; }
  const bool first = l == 0;
  const unsigned long long seed_l = A.seed[l];
  const float drop_p = A.drop_p, drop_scale = A.drop_scale;

  float* outp = A.out[l];                   // [T+1][B][H]: slab 0 = initial state, slab t+1 = h_t
  float* outdp = A.outd[l];
  const bool has_d = outdp != outp;
  const float* opbase = feeds ? A.outd[l - 1] : outp;   // operand of item (t, r): slab t + half
  float* resv = A.reserve[l];
  const unsigned* cdep = cset + (size_t)ldep * nrg * T;
  unsigned* cown = cset + (size_t)l * nrg * T;

  // my output row (gates) and my operand row (MFMA A fragment) in row group r, clamped into the batch; recomputed
  // where needed (registers are the scarce resource of this kernel)
  auto orow_of = [&](int r, int q_) { const int ro = (rg0 + r) * 16 + 4 * q_ + kw; return ro < B ? ro : B - 1; };
  auto arow_of = [&](int r, int j_) { const int ra = (rg0 + r) * 16 + j_; return ra < B ? ra : B - 1; };
  auto orow = [&](int r) { const int ro = (rg0 + r) * 16 + 4 * q + kw; return ro < B ? ro : B - 1; };
  auto arow = [&](int r) { const int ra = (rg0 + r) * 16 + j; return ra < B ? ra : B - 1; };
  float hp[R];
#pragma unroll
  for (int r = 0; r < R; ++r) hp[r] = outp[(long long)orow(r) * H + unit];
  // Operand loads are issued LINE BY LINE (instruction i: rows 8 (i & 1) + lane / 8 of the row group, the 32 floats
  // from column kw * 16 NCH + 32 (i >> 1), 8 consecutive lanes per 128-byte line) and transposed into MFMA fragments
  // through a per-wave LDS slot pair when they are consumed (gru_sync.h: the fragment-shaped load runs the texture
  // addresser at 16 B/clock/CU, 64 KB per item here = 4096 cycles during which every store and poll of the workgroup
  // queues behind it; measured: the reserve store stalled 2700 cycles at issue).
  static_assert(NCH % 2 == 0, "chunks are loaded in pairs");
  unsigned coff[EXACT ? 1 : NCH / 2];   // byte offset of my 16-byte piece of column block i >> 1 inside a row
  if constexpr (EXACT) {
    coff[0] = (unsigned)((kw * NCH * 16 + 4 * (lane & 7)) * 4);
  } else {
#pragma unroll
    for (int pb = 0; pb < NCH / 2; ++pb) {
      const int col = kw * NCH * 16 + pb * 32 + 4 * (lane & 7);
      coff[pb] = (unsigned)((col < H ? col : H - 4) * 4);   // beyond the operand: any valid column (zero weights)
    }
  }
  // byte offset of (row 8 hi + lane / 8 of row group r, clamped into the batch) in a [B][H] slab
  auto lrow_of = [&](int r, int hi, int ln_) { const int ra = (rg0 + r) * 16 + 8 * hi + (ln_ >> 3); return (unsigned)(ra < B ? ra : B - 1) * (unsigned)H * 4u; };
  auto issue_frag = [&](f32x4& dst, const u32x4s& rs, unsigned rowoff_lo, unsigned rowoff_hi, auto i_c) {
    constexpr int i = decltype(i_c)::value;
    const unsigned ro = (i & 1) ? rowoff_hi : rowoff_lo;
    if constexpr (EXACT) issue_load_sc1_x4_imm<(i >> 1) * 128>(dst, rs, ro + coff[0]);
    else issue_load_sc1_x4(dst, rs, ro + coff[i >> 1]);
  };
  // instructions (2p, 2p+1) of an item -> the A fragments of chunks 2p and 2p+1 (slot pair sp of this wave)
  auto transpose_pair_x = [&](int sp, const f32x4& v0, const f32x4& v1, f32x4& a0, f32x4& a1, int ln_) {
    float* slot = tslot + sp * 2 * SLOT_F;
    const int r8 = ln_ >> 3, p8 = ln_ & 7;
    *reinterpret_cast<f32x4*>(&slot[r8 * 36 + p8 * 4]) = v0;
    *reinterpret_cast<f32x4*>(&slot[SLOT_F + r8 * 36 + p8 * 4]) = v1;
    const int j_ = ln_ & 15, q_ = ln_ >> 4;
    const float* sr = slot + (j_ >> 3) * SLOT_F + (j_ & 7) * 36 + 4 * q_;
    a0 = *reinterpret_cast<const f32x4*>(sr);
    a1 = *reinterpret_cast<const f32x4*>(sr + 16);
  };

  // counter this wave needs before it may load the operands of item (t, r): its own layer's (r, t-1) for the
  // recurrent waves, the lower layer's (r, t) for the projection waves
  auto poll_for = [&](int t, int r, bool& need) -> const unsigned* {
    const int rg = rg0 + r;
    need = (t < T) && (rg < nrg) && (proj ? l > 0 : t > 0);
    int tc = t - 1 + half;
    tc = tc < 0 ? 0 : (tc > T - 1 ? T - 1 : tc);
    return cdep + (size_t)(rg < nrg ? rg : nrg - 1) * T + tc;
  };
  // operand slab of item (t, .): clamped past the end (the loads are issued, the results unused)
  auto slab_of = [&](int t) { const int tl = t < T ? t : T - 1; return opbase + (long long)(tl + half) * B * H; };

  f32x4 abuf[2][NCH];
  float gbuf[2][3];
  unsigned pvv = 0;
  bool pv_need = false;
  const unsigned* pv_ptr = cset;

  auto mfma_chunk = [&](f32x4 (&acc)[3], const f32x4& a, auto ci_c) {
    constexpr int ci = decltype(ci_c)::value;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], w[g][ci].x, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], w[g][ci].y, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], w[g][ci].z, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], w[g][ci].w, acc[g], 0, 0, 0);
    }
  };
  auto put_partials = [&](float* dst, const f32x4 (&acc)[3]) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dst[((kw * 3 + g) * 4 + rr) * 64 + lane] = acc[g][rr];
  };

  // ---- prologue -------------------------------------------------------------------------------------------------
  if (proj && first) {   // layer 0: the projection "partials" are the GEMM's gi (slot kw of partial wave 0), zeros elsewhere
    const f32x4 zero[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    put_partials(redp, zero);
    put_partials(redp + PART, zero);
  }
  __syncthreads();
  if (proj && first) {
    const float* g3 = A.gi0 + (long long)orow(0) * 3 * H + unit;   // item (0, 0)
#pragma unroll
    for (int g = 0; g < 3; ++g) redp[((0 * 3 + g) * 4 + kw) * 64 + lane] = g3[(long long)g * H];
  }
  if (feeds) {   // gi of item (0, 0), blocking
    bool need0;
    const unsigned* p0 = poll_for(0, 0, need0);
    if (need0) poll_until(p0, G, poll_once(p0), err);
    const u32x4s rs0 = make_rsrc(slab_of(0));
    static_for<NCH>([&](auto ci) { issue_frag(abuf[1][ci], rs0, lrow_of(0, 0, lane), lrow_of(0, 1, lane), ci); });
    drain_vm();
    static_for<NCH>([&](auto ci) { after_wait(abuf[1][ci]); });
    f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    static_for<NCH / 2>([&](auto p_c) {
      constexpr int pp = decltype(p_c)::value;
      f32x4 a0, a1;
      transpose_pair_x(pp & 1, abuf[1][2 * pp], abuf[1][2 * pp + 1], a0, a1, lane);
      mfma_chunk(acc, a0, std::integral_constant<int, 2 * pp>{});
      mfma_chunk(acc, a1, std::integral_constant<int, 2 * pp + 1>{});
    });
    put_partials(redp, acc);
  }
  {   // operands of this wave's first loop item, (0, ioff), and the poll of the one after
    bool need0;
    const unsigned* p0 = poll_for(0, ioff, need0);
    if (need0) poll_until(p0, G, poll_once(p0), err);
    const u32x4s rs0 = make_rsrc(slab_of(0));
    static_for<NCH>([&](auto ci) { issue_frag(abuf[0][ci], rs0, lrow_of(ioff, 0, lane), lrow_of(ioff, 1, lane), ci); });
    {   // (layer 0, projection waves: gi of item (0, 1); everyone else: a valid address, unused)
      const u32x4s rg = make_rsrc(A.gi0);
      const unsigned go = (unsigned)(((long long)orow(ioff) * 3 * H + unit) * 4);
      issue_load_buf_f32(gbuf[0][0], rg, go); issue_load_buf_f32(gbuf[0][1], rg, go + (unsigned)H * 4u);
      issue_load_buf_f32(gbuf[0][2], rg, go + (unsigned)H * 8u);
    }
    pv_ptr = poll_for(0, 1 + ioff, pv_need);
    issue_poll(pvv, pv_ptr);
  }
  drain_vm();
  __syncthreads();

#ifdef B2T_TIMING
#ifndef B2T_TIMING_LAYER
#define B2T_TIMING_LAYER 2
#endif
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define SSTAMP(i) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tprev; tprev = now_; }
#else
#define SSTAMP(i)
#endif
  unsigned* pub_ptr = nullptr;
  int pub_m0 = 0, pub_t = 0;
  f32x4 tv = f32x4{0.f, 0.f, 0.f, 0.f};   // data of the assembly-issued tile store: untouched until the drain behind it

  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int P = r & 1;
      // an opaque copy of the lane id: everything derived from it is recomputed per item instead of being hoisted out
      // of the loop as dozens of loop-invariant address registers (which spilled)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int jj = ln & 15, qq = ln >> 4;
      // (a) everything issued an item ago has landed (full drain: see gru_pipeline.hip on vmcnt(n > 0))
      drain_vm();
      keep_until_here(tv);
      after_wait(pvv);
#pragma unroll
      for (int ci = 0; ci < NCH; ++ci) after_wait(abuf[P][ci]);
      after_wait(gbuf[P][0]); after_wait(gbuf[P][1]); after_wait(gbuf[P][2]);
      SSTAMP(0)
      {
        const unsigned pv = __builtin_amdgcn_readfirstlane(pvv);
        if (pv_need && pv < G) poll_until(pv_ptr, G, pv, err);
      }
      SSTAMP(1)
      // (b) the previous item's tiles go out as 64 x 16 B per array: wave 0 the h tile, wave 1 its dropped copy, waves
      // 4-7 the four saved arrays (r, z, n, gh_n) of the reserve
      const bool publisher = pub_ptr != nullptr && (wave == 0 || (wave == 1 && has_d));
      if (pub_ptr != nullptr && (wave == 0 || (wave == 1 && has_d) || (proj && resv != nullptr))) {
        const int r4 = ln >> 2, c4 = (ln & 3) * 4;
        const float* tile = hs + (proj ? (1 + kw) * 16 * TP : 0);
        tv[0] = tile[r4 * TP + c4]; tv[1] = tile[r4 * TP + c4 + 1]; tv[2] = tile[r4 * TP + c4 + 2]; tv[3] = tile[r4 * TP + c4 + 3];
        if (wave == 1) {
          const long long e = ((long long)pub_t * B + (pub_m0 + r4)) * H + j0 + c4;   // element index from slab 1
          const float4 u = Philox::uniform4(seed_l, (uint64_t)(e >> 2), 2u);
          tv[0] = u.x >= drop_p ? tv[0] * drop_scale : 0.f;
          tv[1] = u.y >= drop_p ? tv[1] * drop_scale : 0.f;
          tv[2] = u.z >= drop_p ? tv[2] * drop_scale : 0.f;
          tv[3] = u.w >= drop_p ? tv[3] * drop_scale : 0.f;
        }
        if (pub_m0 + r4 < B) {
          if (proj) issue_store_x4(make_rsrc(resv + (long long)pub_t * B * 4 * H),
                                   (unsigned)(((long long)(pub_m0 + r4) * 4 * H + kw * H + j0 + c4) * 4), tv);
          else issue_store_sc1_x4(make_rsrc((wave == 1 ? outdp : outp) + (long long)(pub_t + 1) * B * H),
                                  (unsigned)(((long long)(pub_m0 + r4) * H + j0 + c4) * 4), tv);
        }
      }

      f32x4 acc[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int LPG = NCH >= 8 ? 2 : 1;   // next-item fragment loads per MFMA group: all go out in the first half
      const int m0 = (rg0 + r) * 16;
      // This wave's next item (operands prefetched now) and the one after (polled now); the projection waves are one
      // item ahead.  NOTE: every assembly statement with an OUTPUT register (loads, polls) sits in code common to both
      // wave groups: defined inside `if (proj) .. else ..` the two definitions meet in a phi, and a copy the register
      // allocator places there would read the register while the load is still in flight.
      const int rn = proj ? ((r + 2) & 3) : ((r + 1) & 3), tn = t + (proj ? ((r + 2) >> 2) : ((r + 1) >> 2));
      const int r2 = proj ? ((r + 3) & 3) : ((r + 2) & 3), t2 = t + (proj ? ((r + 3) >> 2) : ((r + 2) >> 2));
      const u32x4s rsn = make_rsrc(slab_of(tn));
      const unsigned rown_lo = lrow_of(rn, 0, ln), rown_hi = lrow_of(rn, 1, ln);
      const u32x4s rgi = make_rsrc(A.gi0 + (long long)(tn < T ? tn : T - 1) * B * 3 * H);
      const unsigned goff = (unsigned)(((long long)orow_of(rn, qq) * 3 * H + (j0 + jj)) * 4);
      if (proj && first) {
        // layer 0: the GEMM's gi of item k+1 (loaded an item ago) goes where the gate threads expect projection partials
        float* dst = redp + (P ^ 1) * PART;
#pragma unroll
        for (int g = 0; g < 3; ++g) dst[((0 * 3 + g) * 4 + kw) * 64 + ln] = gbuf[P][g];
      }
      __builtin_amdgcn_sched_barrier(0);
      SSTAMP(2)
      f32x4 fr[2][2];   // fragments of the pair being contracted and of the next one (transposed one pair ahead)
      if (works) transpose_pair_x(0, abuf[P][0], abuf[P][1], fr[0][0], fr[0][1], ln);
      static_for<NCH / 2>([&](auto p_c) {
        constexpr int pp = decltype(p_c)::value;
        if constexpr (pp == NAP) {
          __builtin_amdgcn_sched_barrier(0);
          if (proj) lds_barrier();   // B1 of the projection waves
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (pp + 1 < NCH / 2)
          if (works) transpose_pair_x((pp + 1) & 1, abuf[P][2 * pp + 2], abuf[P][2 * pp + 3], fr[(pp + 1) & 1][0], fr[(pp + 1) & 1][1], ln);
        static_for<2 * LPG>([&](auto k_c) {
          constexpr int li = pp * 2 * LPG + decltype(k_c)::value;
          if constexpr (li < NCH) issue_frag(abuf[P ^ 1][li], rsn, rown_lo, rown_hi, std::integral_constant<int, li>{});
        });
        if constexpr (pp == NCH / 2 - 1) {   // (only layer 0's projection waves use these)
          issue_load_buf_f32(gbuf[P ^ 1][0], rgi, goff); issue_load_buf_f32(gbuf[P ^ 1][1], rgi, goff + (unsigned)H * 4u);
          issue_load_buf_f32(gbuf[P ^ 1][2], rgi, goff + (unsigned)H * 8u);
        }
        if (works) {
          mfma_chunk(acc, fr[pp & 1][0], std::integral_constant<int, 2 * pp>{});
          mfma_chunk(acc, fr[pp & 1][1], std::integral_constant<int, 2 * pp + 1>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));
      __builtin_amdgcn_sched_barrier(0);
      SSTAMP(3)
      // the tile stores are a whole MFMA phase old: drain them (the publish follows B1, which joins the two storing
      // waves); then the poll of the item after next
      if (publisher) drain_vm();
      keep_until_here(tv);
      pv_ptr = poll_for(t2, r2, pv_need);
      issue_poll(pvv, pv_ptr);
      SSTAMP(4)
      if (proj) {
        if (works) put_partials(redp + (P ^ 1) * PART, acc);
        SSTAMP(5)
        lds_barrier();   // B2
      } else {
        put_partials(redr, acc);
        lds_barrier();   // B1
        SSTAMP(5)
        if (wave == 0 && pub_ptr != nullptr && ln == 0) __hip_atomic_fetch_add(pub_ptr, 1u, RLX_AGENT);
        float gh[3], gx[3];
        const float* rp = redp + P * PART;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const float s0 = redr[((0 * 3 + g) * 4 + kw) * 64 + ln], s1 = redr[((1 * 3 + g) * 4 + kw) * 64 + ln];
          const float s2 = redr[((2 * 3 + g) * 4 + kw) * 64 + ln], s3 = redr[((3 * 3 + g) * 4 + kw) * 64 + ln];
          gh[g] = (s0 + s1) + (s2 + s3);
          const float x0 = rp[((0 * 3 + g) * 4 + kw) * 64 + ln], x1 = rp[((1 * 3 + g) * 4 + kw) * 64 + ln];
          const float x2 = rp[((2 * 3 + g) * 4 + kw) * 64 + ln], x3 = rp[((3 * 3 + g) * 4 + kw) * 64 + ln];
          gx[g] = (x0 + x1) + (x2 + x3);
        }
        const float ghn = gh[2] + bhn;
        const float rr = fast_sigmoid(gx[0] + gh[0] + br);
        const float zz = fast_sigmoid(gx[1] + gh[1] + bz);
        const float nn = fast_tanh(gx[2] + bin + rr * ghn);
        const float h = (1.0f - zz) * nn + zz * hp[r];
        const int ti = (4 * qq + kw) * TP + jj;
        hs[ti] = h; hs[16 * TP + ti] = rr; hs[2 * 16 * TP + ti] = zz; hs[3 * 16 * TP + ti] = nn; hs[4 * 16 * TP + ti] = ghn;
        hp[r] = h;
        SSTAMP(6)
        lds_barrier();   // B2: tiles staged; also fences the partial buffers for the next item
      }
      SSTAMP(7)
      pub_ptr = (m0 < B) ? cown + (size_t)(rg0 + r) * T + t : nullptr;
      pub_m0 = m0; pub_t = t;
    }
  }
  // the last item's tiles and publish
  drain_vm();
  if (pub_ptr != nullptr && (wave == 0 || (wave == 1 && has_d) || (proj && resv != nullptr))) {
    const int r4 = lane >> 2, c4 = (lane & 3) * 4;
    if (pub_m0 + r4 < B) {
      float4 v = *reinterpret_cast<const float4*>(&hs[(proj ? (1 + kw) * 16 * TP : 0) + r4 * TP + c4]);
      if (wave == 1) {
        const long long e = ((long long)pub_t * B + (pub_m0 + r4)) * H + j0 + c4;
        const float4 u = Philox::uniform4(seed_l, (uint64_t)(e >> 2), 2u);
        v.x = u.x >= drop_p ? v.x * drop_scale : 0.f; v.y = u.y >= drop_p ? v.y * drop_scale : 0.f;
        v.z = u.z >= drop_p ? v.z * drop_scale : 0.f; v.w = u.w >= drop_p ? v.w * drop_scale : 0.f;
      }
      if (proj) store_f4<0>(resv + (long long)pub_t * B * 4 * H, (unsigned)(((long long)(pub_m0 + r4) * 4 * H + kw * H + j0 + c4) * 4), v);
      else store_f4<PAUX>((wave == 1 ? outdp : outp) + (long long)(pub_t + 1) * B * H, (unsigned)(((long long)(pub_m0 + r4) * H + j0 + c4) * 4), v);
    }
    drain_vm();
  }
  __syncthreads();
  if (wave == 0 && pub_ptr != nullptr && lane == 0) __hip_atomic_fetch_add(pub_ptr, 1u, RLX_AGENT);
#ifdef B2T_TIMING
  if (lane == 0 && kw == 0 && blockIdx.x == 0 && blockIdx.z == 0 && (int)blockIdx.y == B2T_TIMING_LAYER)
    for (int i = 0; i < 8; ++i) sync[8 + half * 8 + i] = (unsigned)(tacc[i] / (unsigned long long)(T * R));
#endif
  finish_call(sync, pset);
}

// Dynamic LDS per workgroup: what the kernel uses (80 KB) padded so that nothing else (a GEMM workgroup needs 34 KB) fits
// on its CU: an MFMA-bound workgroup must not share its SIMDs.  B2T_STACK_LDS_KB overrides (0 = only what is used).
static unsigned stack_extra_lds() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("B2T_STACK_LDS_KB"); v = (e ? atoi(e) : 132) * 1024; }
  const unsigned need = (unsigned)(STACK_LDS_FLOATS * sizeof(float));
  return (unsigned)v > need ? (unsigned)v : need;
}

}  // namespace b2t

using namespace b2t;

extern "C" int b2t_gru_stack_fwd_f32(const b2t_gru_stack_t* d, void* sync_ws, void* stream) {
  B2T_REQUIRE(d && sync_ws, "gru_stack_fwd: null descriptor / sync workspace");
  const int T = d->T, B = d->B, H = d->H, L = d->L;
  B2T_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 16) == 0 && L >= 1, "gru_stack_fwd: bad shape T=%d B=%d H=%d L=%d", T, B, H, L);
  const int nrg = (B + 15) / 16, nz = (nrg + 3) / 4;
  if (L > STACK_MAXL || (H > 256 && H != 512) || (long long)(H / 16) * L * nz > 240 || (long long)L * nrg * T > SETW) {
    set_error("gru_stack_fwd: shape not covered (L=%d H=%d B=%d T=%d): needs L <= %d, H <= 256 or H = 512, (H/16)*L*ceil(B/64) <= 240 resident "
              "workgroups and L*ceil(B/16)*T <= %d counters", L, H, B, T, STACK_MAXL, SETW);
    return 4;
  }
  B2T_REQUIRE(d->gi0 != nullptr, "gru_stack_fwd: gi0 missing");
  B2T_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "gru_stack_fwd: dropout p=%f out of range", (double)d->drop_p);
  StackFwd A;
  memset(&A, 0, sizeof(A));
  A.gi0 = d->gi0; A.T = T; A.B = B; A.H = H;
  A.drop_p = d->drop_p; A.drop_scale = 1.0f / (1.0f - d->drop_p);
  for (int l = 0; l < L; ++l) {
    B2T_REQUIRE(d->w_hh[l] && d->b_hh[l] && d->out[l] && (l == 0 || (d->w_ih[l] && d->b_ih[l])), "gru_stack_fwd: layer %d pointers missing", l);
    A.w_hh[l] = d->w_hh[l]; A.w_ih[l] = d->w_ih[l]; A.b_hh[l] = d->b_hh[l]; A.b_ih[l] = d->b_ih[l];
    A.out[l] = d->out[l]; A.outd[l] = (d->out_drop[l] && d->drop_p > 0.f) ? d->out_drop[l] : d->out[l];
    A.reserve[l] = d->reserve[l]; A.seed[l] = d->drop_seed[l];
  }
  const dim3 grid(H / 16, L, nz), block(512);
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
  hipStream_t s = as_stream(stream);
  const unsigned dyn = stack_extra_lds();
#define B2T_LAUNCH(NCH, EX)                                                                                                \
  do {                                                                                                                 \
    static bool raised = false;                                                                                        \
    if (!raised) {                                                                                          \
      int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_stack_fwd_kernel<NCH, EX>),                  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn), "gru_stack_fwd: LDS limit"); \
      if (rc) return rc;                                                                                               \
      raised = true;                                                                                                   \
    }                                                                                                                  \
    hipLaunchKernelGGL((gru_stack_fwd_kernel<NCH, EX>), grid, block, dyn, s, A, sync);                                     \
  } while (0)
  if (H <= 128) B2T_LAUNCH(2, false);
  else if (H <= 256) B2T_LAUNCH(4, false);
  else B2T_LAUNCH(8, true);
#undef B2T_LAUNCH
  return check_hip(hipGetLastError(), "gru_stack_fwd");
}
