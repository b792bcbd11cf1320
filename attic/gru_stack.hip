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
constexpr int STG_F = 6 * 4 * TP;   // a gate wave's staged rows: [h, dropped h, r, z, n, gh_n][4 rows][TP]
constexpr int STACK_LDS_FLOATS = 2 * 8 * (4 * 4 * 64) + 8 * 4 * SLOT_F + 4 * STG_F;

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
  const float* mask[STACK_MAXL];   // [T][B][H] keep/scale factors of layer l's output, or null
  int T, B, H;
};

// LDS-only barrier (no vmcnt drain: the prefetched loads stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NC: 16-wide K chunks per wave and operand (H <= 128*NC, NC even); EXACT: H == 128*NC (chunk offsets are immediates)
//
// All 8 waves are alike: wave w contracts K slice w (16*NC columns) of BOTH operands -- h_{t-1} with its W_hh piece,
// x_t (the lower layer's output) with its W_ih piece -- so the two waves of a SIMD keep the MFMA pipe busy for each
// other's LDS transposes and waits.  Iteration k:
//   (a)  drain: operands of item k, polls of item k+1 have landed (prefetched an iteration ago)
//   waves 0-3: publish item k-2 | reduce the 8 partial sets of item k-1, gates, stores of their own rows | MFMAs of item k |
//   waves 4-7: MFMAs of item k (alone on the pipe while waves 0-3 form the gates) ..........................................|
//   all:  the loads of item k+1 between the MFMAs, polls of item k+2, partials of item k -> LDS (double buffered)      |B1|
// ONE barrier per item.  The gate phase of an item runs under the other wave's MFMAs of the next item; a gate wave
// stages and stores the four rows it owns itself (no cross-wave tile, no second barrier).
template <int NC, bool EXACT, bool BF16>
__global__ __launch_bounds__(512, 1) void gru_stack_fwd_kernel(const StackFwd A, unsigned* sync) {
  constexpr int R = 4;
  constexpr int NL = 2 * NC;            // operand load instructions per item: [0, NC) h, [NC, 2 NC) x
  constexpr int NP = NC;                // chunk pairs per item: [0, NC/2) h, [NC/2, NC) x
  constexpr int PART = 4 * 4 * 64;      // one wave's partials: [r, z (h and x parts summed), n from h, n from x][4][64]
  extern __shared__ __attribute__((aligned(16))) float red[];   // STACK_LDS_FLOATS in use + padding (see the launcher)
  float* tslot = red + 2 * 8 * PART + (threadIdx.x >> 6) * 4 * SLOT_F;   // this wave's two transpose slot pairs
  float* stg = red + 2 * 8 * PART + 8 * 4 * SLOT_F + (threadIdx.x >> 6) * STG_F;   // gate waves: own rows [6 arrays][4 rows][TP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: everything selected by it stays scalar
  const bool gatew = wave < 4;          // waves 0-3 also form the gates (thread = row 4 q + wave, unit j)
  const int l = blockIdx.y;
  const int T = A.T, B = A.B, H = A.H;
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
  const bool first = l == 0;            // layer 0: gi comes from the GEMM, nothing to project

  // weights: chunk ci of this wave = columns [wave*16*NC + 16 ci, +16) of W_hh (wh) and W_ih (wx)
  typename WFrag<BF16>::type wh[3][NC], wx[3][NC];
#pragma unroll
  for (int ci = 0; ci < NC; ++ci) {
    const int col = (wave * NC + ci) * 16 + 4 * q;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      wh[g][ci] = make_wfrag<BF16>(col < H ? *reinterpret_cast<const float4*>(A.w_hh[l] + ((long long)g * H + unit) * H + col) : make_float4(0.f, 0.f, 0.f, 0.f));
      wx[g][ci] = make_wfrag<BF16>((!first && col < H) ? *reinterpret_cast<const float4*>(A.w_ih[l] + ((long long)g * H + unit) * H + col)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
  const float* bh = A.b_hh[l];
  float br = bh[unit], bz = bh[H + unit], bin = 0.f;   // r, z: b_hh + b_ih (layer 0: b_ih is inside gi0)
  const float bhn = bh[2 * H + unit];
  if (!first) { const float* bi = A.b_ih[l]; br += bi[unit]; bz += bi[H + unit]; bin = bi[2 * H + unit]; }
  const float* maskp = A.mask[l];

  float* outp = A.out[l];                   // [T+1][B][H]: slab 0 = initial state, slab t+1 = h_t
  float* outdp = A.outd[l];
  const bool has_d = maskp != nullptr;
  const float* xbase = first ? outp : A.outd[l - 1];   // x_t = slab t+1 of the lower layer's (dropped) output
  float* resv = A.reserve[l];
  unsigned* cown = cset + (size_t)l * nrg * T;
  const unsigned* clow = cset + (size_t)(first ? 0 : l - 1) * nrg * T;

  // h_{t-1} of my gate element, per row group (gate waves)
  float hp[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ro = (rg0 + r) * 16 + 4 * q + (wave & 3);
    hp[r] = outp[(long long)(ro < B ? ro : B - 1) * H + unit];
  }

  // Operand loads are issued LINE BY LINE (instruction i of an operand: rows 8 (i & 1) + lane / 8 of the row group, the
  // 32 floats from column wave*16*NC + 32 (i >> 1), 8 consecutive lanes per 128-byte line) and transposed into MFMA
  // fragments through a per-wave LDS slot pair when they are consumed (gru_sync.h: the fragment-shaped load runs the
  // texture addresser at 16 B/clock/CU -- 64 KB per item here = 4096 cycles during which every store and poll of the
  // workgroup queues behind it; measured: the reserve store stalled 2700 cycles at issue).
  static_assert(NC % 2 == 0, "chunks are loaded in pairs");
  unsigned coff[EXACT ? 1 : NC / 2];   // byte offset of my 16-byte piece of column block pb inside a row
  if constexpr (EXACT) {
    coff[0] = (unsigned)((wave * NC * 16 + 4 * (lane & 7)) * 4);
  } else {
#pragma unroll
    for (int pb = 0; pb < NC / 2; ++pb) {
      const int col = wave * NC * 16 + pb * 32 + 4 * (lane & 7);
      coff[pb] = (unsigned)((col < H ? col : H - 4) * 4);   // beyond the operand: any valid column (zero weights)
    }
  }
  // byte offset of (row 8 hi + lane / 8 of row group r, clamped into the batch) in a [B][H] slab
  auto lrow_of = [&](int r, int hi, int ln_) { const int ra = (rg0 + r) * 16 + 8 * hi + (ln_ >> 3); return (unsigned)(ra < B ? ra : B - 1) * (unsigned)H * 4u; };
  auto issue_frag = [&](f32x4& dst, const u32x4s& rs, unsigned rowoff_lo, unsigned rowoff_hi, auto i_c) {
    constexpr int i = decltype(i_c)::value;   // index within the operand
    const unsigned ro = (i & 1) ? rowoff_hi : rowoff_lo;
    if constexpr (EXACT) issue_load_sc1_x4_imm<(i >> 1) * 128>(dst, rs, ro + coff[0]);
    else issue_load_sc1_x4(dst, rs, ro + coff[i >> 1]);
  };
  // instructions (2p, 2p+1) of an operand -> the A fragments of its chunks 2p and 2p+1 (slot pair sp of this wave)
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

  // counters an item (t, r) waits for: own layer (r, t-1) [t > 0], lower layer (r, t) [l > 0]
  auto poll_own = [&](int t, int r, bool& need) -> const unsigned* {
    const int rg = rg0 + r;
    need = (t < T) && (rg < nrg) && (t > 0);
    const int tc = t - 1 < 0 ? 0 : (t - 1 > T - 1 ? T - 1 : t - 1);
    return cown + (size_t)(rg < nrg ? rg : nrg - 1) * T + tc;
  };
  auto poll_low = [&](int t, int r, bool& need) -> const unsigned* {
    const int rg = rg0 + r;
    need = (t < T) && (rg < nrg) && !first;
    return clow + (size_t)(rg < nrg ? rg : nrg - 1) * T + (t < T ? t : T - 1);
  };
  // operand slabs of item (t, .), clamped past the end (the loads are issued, the results unused)
  auto hslab = [&](int t) { return outp + (long long)(t < T ? t : T - 1) * B * H; };
  auto xslab = [&](int t) { return xbase + (long long)((t < T ? t : T - 1) + 1) * B * H; };

  f32x4 abuf[2][NL];
  float gbuf[3];
  f32x4 mk = f32x4{1.f, 1.f, 1.f, 1.f};   // dropout factors of my piece of the previous item's tile (prefetched)
  unsigned pvh = 0, pvx = 0;
  bool need_h = false, need_x = false;
  const unsigned *ptr_h = cset, *ptr_x = cset;

  // acc: [0] r, [1] z (the h and the x products share an accumulator), [2] n from h, [3] n from x.  k outer, gate
  // inner: three independent accumulators between two MFMAs on the same one.
  auto mfma_chunk = [&](f32x4 (&acc)[4], const f32x4& a, const typename WFrag<BF16>::type (&w)[3][NC], auto ci_c, auto nidx_c) {
    constexpr int ci = decltype(ci_c)::value;
    constexpr int ni = decltype(nidx_c)::value;   // 2: h operand, 3: x operand
    if constexpr (BF16) {   // one 16x16x16 bf16 MFMA per gate: the fragment layout is the fp32 path's
      const bf16x4 ab = to_bf16x4(make_float4(a[0], a[1], a[2], a[3]));
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, w[0][ci], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, w[1][ci], acc[1], 0, 0, 0);
      acc[ni] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, w[2][ci], acc[ni], 0, 0, 0);
    } else {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], w[0][ci].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], w[1][ci].x, acc[1], 0, 0, 0);
      acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], w[2][ci].x, acc[ni], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], w[0][ci].y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], w[1][ci].y, acc[1], 0, 0, 0);
      acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], w[2][ci].y, acc[ni], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], w[0][ci].z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], w[1][ci].z, acc[1], 0, 0, 0);
      acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], w[2][ci].z, acc[ni], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], w[0][ci].w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], w[1][ci].w, acc[1], 0, 0, 0);
      acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], w[2][ci].w, acc[ni], 0, 0, 0);
    }
  };

  // ---- prologue: operands of item (0, 0), polls of item (0, 1) ----------------------------------------------------
  {
    bool n0;
    const unsigned* p0 = poll_low(0, 0, n0);
    if (n0) poll_until(p0, G, poll_once(p0), err);
    const u32x4s rh = make_rsrc(hslab(0)), rx = make_rsrc(xslab(0));
    static_for<NC>([&](auto i) { issue_frag(abuf[0][i], rh, lrow_of(0, 0, lane), lrow_of(0, 1, lane), i); });
    static_for<NC>([&](auto i) { issue_frag(abuf[0][NC + i], rx, lrow_of(0, 0, lane), lrow_of(0, 1, lane), i); });
    ptr_h = poll_own(0, 1, need_h); issue_poll(pvh, ptr_h);
    ptr_x = poll_low(0, 1, need_x); issue_poll(pvx, ptr_x);
    const u32x4s rg = make_rsrc(A.gi0);
    issue_load_buf_f32(gbuf[0], rg, 0u); issue_load_buf_f32(gbuf[1], rg, 0u); issue_load_buf_f32(gbuf[2], rg, 0u);
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
  unsigned* pub_ptr = nullptr;     // counter of the item whose tiles have been stored but not published yet
  f32x4 tv = f32x4{0.f, 0.f, 0.f, 0.f}, tv2 = f32x4{0.f, 0.f, 0.f, 0.f};   // data of the assembly-issued stores: untouched until the drain behind them

  // t runs to T inclusive: the extra round only finishes the last items (gates, stores, publish); its MFMAs run on
  // clamped operands and are never used
  for (int t = 0; t <= T; ++t) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int P = r & 1;
      // an opaque copy of the lane id: everything derived from it is recomputed per item instead of being hoisted out
      // of the loop as dozens of loop-invariant address registers (which spilled)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int jj = ln & 15, qq = ln >> 4;
      const int rn = (r + 1) & 3, tn = t + ((r + 1) >> 2);      // next item (operands prefetched now)
      const int r2 = (r + 2) & 3, t2 = t + ((r + 2) >> 2);      // the one after (polled now)
      const int rp = (r + 3) & 3, tp = t - 1 + ((r + 3) >> 2);  // previous item (gates now)
      const bool gvalid = tp >= 0 && tp < T && (rg0 + rp) < nrg;
      // (a) everything issued an iteration ago has landed (full drain: see gru_pipeline.hip on vmcnt(n > 0))
      drain_vm();
      keep_until_here(tv); keep_until_here(tv2);
      after_wait(pvh); after_wait(pvx);
#pragma unroll
      for (int i = 0; i < NL; ++i) after_wait(abuf[P][i]);
      after_wait(gbuf[0]); after_wait(gbuf[1]); after_wait(gbuf[2]); after_wait(mk);
      SSTAMP(0)
      {
        const unsigned vh = __builtin_amdgcn_readfirstlane(pvh), vx = __builtin_amdgcn_readfirstlane(pvx);
        if (need_h && vh < G) poll_until(ptr_h, G, vh, err);
        if (need_x && vx < G) poll_until(ptr_x, G, vx, err);
      }
      SSTAMP(1)
      const u32x4s rsh = make_rsrc(hslab(tn)), rsx = make_rsrc(xslab(tn));
      const unsigned rown_lo = lrow_of(rn, 0, ln), rown_hi = lrow_of(rn, 1, ln);
      const int m0p = (rg0 + rp) * 16;
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 fr[2][2];   // fragments of the pair being contracted and of the next one (transposed one pair ahead)
      transpose_pair_x(0, abuf[P][0], abuf[P][1], fr[0][0], fr[0][1], ln);
      __builtin_amdgcn_sched_barrier(0);

      if (gatew) {
        // ---- gates of the previous item: 8 partial sets -> r, z, n, h -> this wave's 4 rows ------------------------
#ifndef B2T_STACK_NO_PRIO
        __builtin_amdgcn_s_setprio(3);   // non-MFMA work next to the other wave's MFMA stream gets ~1 issue slot in 16 cycles
#endif
        if (wave == 0 && pub_ptr != nullptr && ln == 0) __hip_atomic_fetch_add(pub_ptr, 1u, RLX_AGENT);   // item k-2
        const float* rd = red + (P ^ 1) * 8 * PART;   // partials of item k-1
        float sm[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < 8; w8 += 2) {
            a0 += rd[w8 * PART + (g * 4 + wave) * 64 + ln];
            a1 += rd[(w8 + 1) * PART + (g * 4 + wave) * 64 + ln];
          }
          sm[g] = a0 + a1;
        }
        // layer 0: x parts are zero (no W_ih here), the GEMM's gi (b_ih folded in) takes their place
        const float xr = first ? gbuf[0] : 0.f, xz = first ? gbuf[1] : 0.f, xn = first ? gbuf[2] : sm[3] + bin;
        const float ghn = sm[2] + bhn;
        const float rr = fast_sigmoid(sm[0] + xr + br);
        const float zz = fast_sigmoid(sm[1] + xz + bz);
        const float nn = fast_tanh(xn + rr * ghn);
        const float h = (1.0f - zz) * nn + zz * hp[rp];
        if (gvalid) hp[rp] = h;
        // my element is (row 4 qq + wave, unit jj): stage the wave's 4 rows x 16 units of each array and write them as
        // 16-byte pieces -- lanes 16 a .. 16 a + 15 take array a: first (write-through, the hand-off payload) h and its
        // dropped copy, then (ordinary stores) r, z, n, gh_n of the reserve
        const int ti = qq * TP + jj;
        stg[ti] = h; stg[2 * 4 * TP + ti] = rr; stg[3 * 4 * TP + ti] = zz; stg[4 * 4 * TP + ti] = nn; stg[5 * 4 * TP + ti] = ghn;
        {
          const int arr = ln >> 4, lr = (ln >> 2) & 3, c4 = (ln & 3) * 4;
          const int row = m0p + 4 * lr + wave;
          const bool rowok = gvalid && row < B;
          const float* sp = stg + lr * TP + c4;
          tv = *reinterpret_cast<const f32x4*>(sp);   // h (all lanes; lanes >= 32 do not store it)
          tv2 = *reinterpret_cast<const f32x4*>(sp + (2 + arr) * 4 * TP);
          if (has_d && arr == 1) tv = tv * mk;   // nn.GRU inter-layer dropout: factors 0 or 1/(1-p), made before the launch
          const long long eo = (long long)(rowok ? row : 0) * H + j0 + c4;
          float* p1 = (arr == 1 ? outdp : outp) + (long long)(tp + 1) * B * H + eo;
          if (rowok && (arr == 0 || (arr == 1 && has_d))) issue_store_sc1_x4_ptr(p1, tv);
          if (rowok && resv != nullptr)
            issue_store_x4_ptr(resv + ((long long)tp * B + row) * 4 * H + arr * H + j0 + c4, tv2);
        }
#ifndef B2T_STACK_NO_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        SSTAMP(2)
      }
      SSTAMP(3)
      static_for<NP>([&](auto p_c) {
        constexpr int pp = decltype(p_c)::value;
        constexpr bool isx = pp >= NC / 2;
        constexpr int po = isx ? pp - NC / 2 : pp;            // pair index within the operand
        if constexpr (pp + 1 < NP) {
          constexpr int pn = pp + 1;
          constexpr int base = (pn >= NC / 2) ? NC + 2 * (pn - NC / 2) : 2 * pn;
          transpose_pair_x(pn & 1, abuf[P][base], abuf[P][base + 1], fr[pn & 1][0], fr[pn & 1][1], ln);
        }
        // the next item's loads all go out in the first half of the MFMA stream (4 per pair), so that the drain in front
        // of the publish does not wait for fresh loads
        if constexpr (pp < NP / 2) {
          static_for<4>([&](auto k_c) {
            constexpr int li = 4 * pp + decltype(k_c)::value;      // 0 .. 2 NC - 1: [0, NC) h, [NC, 2 NC) x
            if constexpr (li < NC) issue_frag(abuf[P ^ 1][li], rsh, rown_lo, rown_hi, std::integral_constant<int, li>{});
            else if constexpr (li < NL) issue_frag(abuf[P ^ 1][li], rsx, rown_lo, rown_hi, std::integral_constant<int, li - NC>{});
          });
        }
#ifdef B2T_EXPERIMENT_B_IDLE   // timing experiment only (wrong results): waves 4-7 issue no MFMAs
        if (gatew)
#endif
        if constexpr (!isx) {
          mfma_chunk(acc, fr[pp & 1][0], wh, std::integral_constant<int, 2 * po>{}, std::integral_constant<int, 2>{});
          mfma_chunk(acc, fr[pp & 1][1], wh, std::integral_constant<int, 2 * po + 1>{}, std::integral_constant<int, 2>{});
        } else {
          mfma_chunk(acc, fr[pp & 1][0], wx, std::integral_constant<int, 2 * po>{}, std::integral_constant<int, 3>{});
          mfma_chunk(acc, fr[pp & 1][1], wx, std::integral_constant<int, 2 * po + 1>{}, std::integral_constant<int, 3>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
      __builtin_amdgcn_sched_barrier(0);
      SSTAMP(4)
      // the gate waves' stores are a whole MFMA phase old: drain them (the publish follows B1, which joins the four)
      if (gatew && gvalid) drain_vm();
      keep_until_here(tv); keep_until_here(tv2);
      pub_ptr = gvalid ? cown + (size_t)(rg0 + rp) * T + tp : nullptr;
      {   // layer 0: the GEMM's gi of THIS item, for the gates an iteration from now; polls of the item after next
        const int tq = t < T ? t : T - 1;
        const int ro = (rg0 + r) * 16 + 4 * qq + (wave & 3);
        const u32x4s rg = make_rsrc(A.gi0 + (long long)tq * B * 3 * H);
        const unsigned go = (unsigned)(((long long)(ro < B ? ro : B - 1) * 3 * H + (j0 + jj)) * 4);
        issue_load_buf_f32(gbuf[0], rg, go); issue_load_buf_f32(gbuf[1], rg, go + (unsigned)H * 4u);
        issue_load_buf_f32(gbuf[2], rg, go + (unsigned)H * 8u);
        {   // dropout factors of my 16-byte piece of THIS item's tile (lanes 16-31 of the gate waves use them)
          const int rw = (rg0 + r) * 16 + 4 * ((ln >> 2) & 3) + (wave & 3);
          const float* mp = (has_d ? maskp : A.gi0) + (has_d ? ((long long)tq * B + (rw < B ? rw : B - 1)) * H + j0 + (ln & 3) * 4 : 0);
          issue_load_x4_ptr(mk, mp);
        }
        ptr_h = poll_own(t2, r2, need_h); issue_poll(pvh, ptr_h);
        ptr_x = poll_low(t2, r2, need_x); issue_poll(pvx, ptr_x);
      }
      SSTAMP(5)
      {
        float* dst = red + P * 8 * PART + wave * PART;   // partials of item k: buffer k & 1
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) dst[(g * 4 + rr) * 64 + ln] = acc[g][rr];
      }
      SSTAMP(6)
      lds_barrier();   // B1: the partials of item k are complete
      SSTAMP(7)
    }
  }
  drain_vm();
  if (wave == 0 && pub_ptr != nullptr && lane == 0) __hip_atomic_fetch_add(pub_ptr, 1u, RLX_AGENT);
#ifdef B2T_TIMING
  if (lane == 0 && (wave & 3) == 0 && blockIdx.x == 0 && blockIdx.z == 0 && (int)blockIdx.y == B2T_TIMING_LAYER)
    for (int i = 0; i < 8; ++i) sync[8 + (wave >> 2) * 8 + i] = (unsigned)(tacc[i] / (unsigned long long)((T + 1) * R));
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
  StackFwd A;
  memset(&A, 0, sizeof(A));
  A.gi0 = d->gi0; A.T = T; A.B = B; A.H = H;
  for (int l = 0; l < L; ++l) {
    B2T_REQUIRE(d->w_hh[l] && d->b_hh[l] && d->out[l] && (l == 0 || (d->w_ih[l] && d->b_ih[l])), "gru_stack_fwd: layer %d pointers missing", l);
    A.w_hh[l] = d->w_hh[l]; A.w_ih[l] = d->w_ih[l]; A.b_hh[l] = d->b_hh[l]; A.b_ih[l] = d->b_ih[l];
    B2T_REQUIRE((d->drop_mask[l] == nullptr) == (d->out_drop[l] == nullptr), "gru_stack_fwd: layer %d: drop_mask and out_drop go together", l);
    A.out[l] = d->out[l]; A.outd[l] = d->out_drop[l] ? d->out_drop[l] : d->out[l];
    A.reserve[l] = d->reserve[l]; A.mask[l] = d->drop_mask[l];
  }
  const dim3 grid(H / 16, L, nz), block(512);
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
  hipStream_t s = as_stream(stream);
  const unsigned dyn = stack_extra_lds();
#define B2T_LAUNCH2(NCH, EX, BF)                                                                                                \
  do {                                                                                                                 \
    static bool raised = false;                                                                                        \
    if (!raised) {                                                                                          \
      int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_stack_fwd_kernel<NCH, EX, BF>),                  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn), "gru_stack_fwd: LDS limit"); \
      if (rc) return rc;                                                                                               \
      raised = true;                                                                                                   \
    }                                                                                                                  \
    hipLaunchKernelGGL((gru_stack_fwd_kernel<NCH, EX, BF>), grid, block, dyn, s, A, sync);                                     \
  } while (0)
#define B2T_LAUNCH(NCH, EX) do { if (d->bf16) B2T_LAUNCH2(NCH, EX, true); else B2T_LAUNCH2(NCH, EX, false); } while (0)
  if (H <= 256) B2T_LAUNCH(2, false);
  else B2T_LAUNCH(4, true);
#undef B2T_LAUNCH
#undef B2T_LAUNCH2
  return check_hip(hipGetLastError(), "gru_stack_fwd");
}
