// stream.hip — the streaming frame of the decoder as ONE kernel (BASELINE configs[4]: frame-by-frame GRU with carried state).
// OPT-IN (B2T_STREAM_FUSED=1 / b2t_ops.STREAM["fused"]); the default streaming path stays the executor's launch sequence.
// GRUDecoder.forward(x, day_idx, states, return_state=True) for a handful of patch frames (model_training/rnn_model.py:88-134,
// called once per 80 ms frame by the online decoder, evaluate_model_helpers.py:87-115): day layer -> patch -> L GRU layers, one
// time step each -> head.  Through the executor that is ~12 dependent launches, 163 us of device time per frame at the shipped
// shape (H = 768, patch 14 x 512, 32 rows); the work is one pass over 130 MB of weights.
//
// One persistent launch, one workgroup per CU, phases separated by grid barriers:
//   D      day layer: u[b][t][:] = softsign(x[b][t][:] W_day[b] + b_day[b])      tile = (sentence, 16 bins, 128 columns)
//   (t, l) GRU layer l at patch frame t: a workgroup owns 5 hidden units = 15 rows of W_ih and W_hh (r, z, n of each unit), streams
//          them ONCE against all rows of the layer input and of h_{t-1} (v_mfma_f32_16x16x4_f32: the weight rows are the M side,
//          the batch rows the N side; the concatenated K split over the 8 waves, partial tiles summed through LDS) and applies
//          the gate math for its units, so a layer step is one phase and one barrier
//   H      head: logits[b][t][:] = y_top[t][b][:] W_out^T + b_out
// Data that crosses workgroups is written through to memory (sc1 stores); u is read back with ordinary loads after an
// agent-scope acquire (every workgroup reads all of it: it should come from the XCD's L2), the layer states with sc1 loads
// (the eight L2s are not coherent with each other: MI355X_MICROARCH, per-XCD L2).
// Arithmetic is the executor path's: exact fp32 MFMA products, the same gate formulas (gru_cell.h): the two agree to summation
// order (tests/test_gpu_parity.py::test_stream_forward_equals_executor).
// MEASURED (attic/stream_probe.py, phase stamps of workgroup 0): 151-159 us per frame against the executor's 163 --
// day layer 11, layer 0 47 (66 MB of W_ih), layers 1-4 13.6 each (14 MB each), head 11, barriers 4 each -- i.e. the single launch
// removes the launches and NOT the time: a phase is 3-4 dependent memory round trips of ~3 us (all 154 busy CUs burst their
// 16-rows-x-64-byte fragment loads at once) plus the barrier.  What the stamps found on the way: (1) all 8 waves issuing the
// acquire fence = 2048 L2 invalidates per barrier, +30 us per phase; (2) 256 workgroups adding to and polling one word: 7-14 us
// per barrier, two-level counters: 4; (3) `cur = nxt` register rotation waits for the loads it should leave in flight. Left as
// an opt-in because a grid-barrier kernel needs every workgroup resident (a second such launch on another stream can starve it
// until the bounded spins expire) and a 5-10 % gain does not pay for that; what would pay is a weight layout that a wave can
// stream in whole lines (rows interleaved 16-wise at load time), which the parameter arena's PyTorch layout is not.
#include "common.h"
#include "gru_cell.h"

namespace b2t {
namespace {

constexpr int NW = 8;                 // waves per workgroup
constexpr int NTHR = NW * 64;
constexpr int UW = 5;                 // hidden units per layer tile (3 * 5 = 15 of the MFMA tile's 16 rows)
constexpr unsigned SPIN_MAX = 1u << 22;
#ifndef B2T_STREAM_SC1
#define B2T_STREAM_SC1 16   // 16: layer states read with sc1 loads, barriers without invalidate; 0: ordinary loads behind an acquire
#endif
constexpr int XA = B2T_STREAM_SC1;

struct StreamArgs {
  b2t_model_t m;
  int B, T, Tp;
  const float* x; const int32_t* day; const float* states;
  float* logits; float* hidden;
  float* u;        // [B][T][F]
  float* S;        // [L][2][B][H]
  float* ytop;     // [Tp][B][H]
  unsigned* sync;  // b2t_stream_sync_bytes: [1] finished workgroups, [2] sticky error, barrier counters (SW_*), phase stamps of workgroup 0
};

__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All workgroups arrive; write-through stores issued before are drained here, i.e. in memory when anybody leaves.  ACQ: what
// follows reads them with ORDINARY loads, so stale lines must go: one agent-scope acquire per workgroup (buffer_inv sc1 acts
// on the CU's vector cache and the XCD's L2, not on a wave; all 8 waves issuing it -- 2048 L2 invalidates per barrier --
// made every phase 30 us longer).  Without ACQ the readers use sc1 loads.
// Two levels: 256 workgroups adding to and polling ONE word saturate it (a word serves ~100 operations per microsecond: the
// flat barrier took 7-14 us).  Workgroup b arrives on the counter of group b % 8 (its own 128-byte line); the group's last
// arriver arrives on the top counter, waits for the 8 groups and releases its group's flag, which the others poll.
// Counters are monotonic within a launch (`phase` = 1, 2, ..); the last workgroup of the launch clears them.
constexpr int SW_GRP = 32, SW_FLAG = 32 * 9, SW_TOP = 32 * 17, SW_STAMP = 32 * 18, SW_WORDS = 32 * 21;   // word offsets; [1] finished, [2] error
template <bool ACQ>
__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned phase) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 7u, G = gridDim.x;
    const unsigned members = (G - g + 7u) >> 3, ngroups = G < 8u ? G : 8u;
    unsigned* err = sync + 2;
    unsigned spins = 0;
    const unsigned old = __hip_atomic_fetch_add(sync + SW_GRP + 32 * g, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == members * phase) {
      __hip_atomic_fetch_add(sync + SW_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(sync + SW_TOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ngroups * phase) {
        if ((++spins & 63u) == 0u) {
          if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (spins > SPIN_MAX) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
      }
      __hip_atomic_store(sync + SW_FLAG + 32 * g, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(sync + SW_FLAG + 32 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {
        if ((++spins & 63u) == 0u) {
          if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (spins > SPIN_MAX) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));

// 16 bytes of an activation row.  AUX = 16: sc1, an agent-scope load that is never served from a line another XCD's store
// made stale (the layer states cross workgroups through memory); AUX = 0: an ordinary cached load.
template <int AUX>
__device__ __forceinline__ float4 ld_act(const float* base_uniform, unsigned byte_off) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, AUX);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}

// One wave's share of TWO products that share the batch rows on the N side (a layer step's W_ih . in and W_hh . h, or one
// product with an empty second segment): virtual 16-wide k groups [g0, g1) of the concatenation (segment 0: G0 groups, then
// segment 1); acc0 / acc1 take the groups of their segment.
//   acc[nb] += sum_k A[row j][k] * Bm[row 16 nb + j][k]
// a0 / a1: this lane's weight-row pointers (lane = (j = lane & 15, q = lane >> 4)); b0 / b1: wave-uniform bases of the two
// activation matrices with this lane's row offsets bo0 / bo1 (bytes).  A lane loads 4 consecutive k of its row (k = 16 g + 4 q ..)
// for A and for every B tile, which feeds 4 MFMAs whose k sets are {16 g + 4 q + i : q}: the same for both operands, and the sum
// over k does not care about the order.  The next trip's loads are in flight while this trip's MFMAs issue.  NOTHING branches
// around a load: group indices are clamped and the segment is chosen by selects (a branch between loads makes the compiler
// drain the memory counter at the join: the first version, which branched on the segment, took 13 us for 1536 k -- five
// serialised round trips).
#define B2T_SP_LOAD(gidx, AV, BV, SLOT)                                                                    \
  {                                                                                                        \
    const int g_ = min((gidx), g1 - 1);                                                                    \
    const bool s0_ = g_ < G0;                                                                              \
    const int kk_ = 16 * (s0_ ? g_ : g_ - G0) + q4;                                                        \
    AV = *reinterpret_cast<const float4*>((s0_ ? a0 : a1) + kk_);                                          \
    const float* bb_ = s0_ ? b0 : b1;                                                                      \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                      \
      BV[nb][SLOT] = ld_act<AUX>(bb_, (s0_ ? bo0[nb] : bo1[nb]) + 4u * (unsigned)kk_);                     \
  }
#define B2T_SP_MMA(gidx, AV, BV, SLOT)                                                                     \
  if ((gidx) < g1) {                              /* wave-uniform: the tail trip holds fewer than GR groups */ \
    if ((gidx) < G0) {                                                                                     \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                  \
        acc0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.x, BV[nb][SLOT].x, acc0[nb], 0, 0, 0);          \
        acc0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.y, BV[nb][SLOT].y, acc0[nb], 0, 0, 0);          \
        acc0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.z, BV[nb][SLOT].z, acc0[nb], 0, 0, 0);          \
        acc0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.w, BV[nb][SLOT].w, acc0[nb], 0, 0, 0);          \
      }                                                                                                    \
    } else {                                                                                               \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                  \
        acc1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.x, BV[nb][SLOT].x, acc1[nb], 0, 0, 0);          \
        acc1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.y, BV[nb][SLOT].y, acc1[nb], 0, 0, 0);          \
        acc1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.z, BV[nb][SLOT].z, acc1[nb], 0, 0, 0);          \
        acc1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.w, BV[nb][SLOT].w, acc1[nb], 0, 0, 0);          \
      }                                                                                                    \
    }                                                                                                      \
  }
// Two register sets X / Y alternate WITHOUT copies: `cur = nxt` at the end of a trip needs the values, i.e. waits for the
// loads it was meant to leave in flight (the first version did that: every trip cost a full memory round trip).
template <int NB, int AUX>
__device__ __forceinline__ void wave_product2(const float* a0, const float* a1, const float* b0, const float* b1,
                                              const unsigned (&bo0)[NB], const unsigned (&bo1)[NB], int G0, int g0, int g1,
                                              f32x4 (&acc0)[NB], f32x4 (&acc1)[NB]) {
  if (g0 >= g1) return;
  constexpr int GR = NB >= 4 ? 2 : 4;             // groups per trip (two sets of GR (1 + NB) float4 registers are live)
  const int q4 = 4 * ((int)(threadIdx.x & 63) >> 4);
  float4 ax[GR], ay[GR], bx[NB][GR], by[NB][GR];
#pragma unroll
  for (int g = 0; g < GR; ++g) B2T_SP_LOAD(g0 + g, ax[g], bx, g)
  for (int gb = g0; gb < g1; gb += 2 * GR) {
#pragma unroll
    for (int g = 0; g < GR; ++g) B2T_SP_LOAD(gb + GR + g, ay[g], by, g)
#pragma unroll
    for (int g = 0; g < GR; ++g) B2T_SP_MMA(gb + g, ax[g], bx, g)
#pragma unroll
    for (int g = 0; g < GR; ++g) B2T_SP_LOAD(gb + 2 * GR + g, ax[g], bx, g)
#pragma unroll
    for (int g = 0; g < GR; ++g) B2T_SP_MMA(gb + GR + g, ay[g], by, g)
  }
}
#undef B2T_SP_MMA
#undef B2T_SP_LOAD

// Sum the waves' partial tiles through LDS: -> out[nb][16 r + c].  red: NW * NB * 256 floats.  Two barriers.
template <int NB>
__device__ __forceinline__ void wg_reduce(const f32x4 (&acc)[NB], float* red, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * NB + nb) * 256 + (4 * q + i) * 16 + j] = acc[nb][i];
  __syncthreads();
  for (int e = threadIdx.x; e < NB * 256; e += NTHR) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * NB * 256 + e];     // fixed order: deterministic
    out[e] = s;
  }
  __syncthreads();
}

// The workgroup's 16 x (16 NB) tiles of  A0 . B0^T  and  A1 . B1^T  (K0, K1 multiples of 16; K1 = 0: one product), the
// concatenated K split evenly over the waves: -> out0 / out1 (LDS).
template <int NB, int AUX>
__device__ __forceinline__ void wg_product2(const float* a0, const float* a1, const float* b0, const float* b1,
                                            const unsigned (&bo0)[NB], const unsigned (&bo1)[NB], int K0, int K1, float* red,
                                            float* out0, float* out1) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably wave-uniform: the group range selects buffer bases
  const int G0 = K0 >> 4, groups = G0 + (K1 >> 4);
  const int g0 = (int)((long long)groups * wave / NW), g1 = (int)((long long)groups * (wave + 1) / NW);
  f32x4 acc0[NB], acc1[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { acc0[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  wave_product2<NB, AUX>(a0, a1, b0, b1, bo0, bo1, G0, g0, g1, acc0, acc1);
  wg_reduce<NB>(acc0, red, out0);
  if (K1 > 0) wg_reduce<NB>(acc1, red, out1);
}

template <int NB>
__global__ void __launch_bounds__(NTHR) stream_forward_kernel(StreamArgs a) {
  __shared__ float red[NW * NB * 256];
  __shared__ float o_ih[NB * 256], o_hh[NB * 256];
  const int F = a.m.F, H = a.m.H, L = a.m.L, C = a.m.C, B = a.B, T = a.T, Tp = a.Tp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const unsigned G = gridDim.x;
  unsigned nbar = 0;
  int nstamp = 0;
  auto stamp = [&]() { if (blockIdx.x == 0 && threadIdx.x == 0 && nstamp < 64) a.sync[SW_STAMP + nstamp] = (unsigned)wall_clock64(); ++nstamp; };
  stamp();

  // ---- D: day layer.  tile = (b, 16 bins, 128 columns); wave w owns 16 of the columns, full K ---------------------------------
  {
    const int mt = (T + 15) / 16, ct = (F + 127) / 128;
    for (int tile = blockIdx.x; tile < B * mt * ct; tile += G) {
      const int b = tile / (mt * ct), rem = tile % (mt * ct), t0 = 16 * (rem / ct), n0 = 128 * (rem % ct) + 16 * wave;
      if (n0 >= F) continue;
      const int d = a.day[b];
      const float* W = a.m.day_w + (long long)d * a.m.day_w_stride;
      const float* xr = a.x + ((long long)b * T + min(t0 + j, T - 1)) * F;
      const int n = min(n0 + j, F - 1);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      // 64 k per trip, the next trip's operands in flight under this trip's MFMAs (F is a multiple of 16; group starts clamped)
      float4 xa[4], xn[4]; float wv[4][4], wn[4][4];
      const float* wcol = W + n;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int kk = min(16 * g, F - 16) + 4 * q;
        xa[g] = *reinterpret_cast<const float4*>(xr + kk);
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[g][i] = wcol[(long long)(kk + i) * F];
      }
      for (int kb = 0; kb < F; kb += 64) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kk = min(kb + 64 + 16 * g, F - 16) + 4 * q;
          xn[g] = *reinterpret_cast<const float4*>(xr + kk);
#pragma unroll
          for (int i = 0; i < 4; ++i) wn[g][i] = wcol[(long long)(kk + i) * F];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (kb + 16 * g < F) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].x, wv[g][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].y, wv[g][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].z, wv[g][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].w, wv[g][3], acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          xa[g] = xn[g];
#pragma unroll
          for (int i = 0; i < 4; ++i) wv[g][i] = wn[g][i];
        }
      }
      if (n0 + j < F) {
        const float bias = a.m.day_b[(long long)d * a.m.day_b_stride + n0 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = t0 + 4 * q + i;
          if (t < T) {
            const float v = acc[i] + bias;
            st_wt(a.u + ((long long)b * T + t) * F + n0 + j, v / (1.0f + fabsf(v)));
          }
        }
      }
    }
  }
  stamp();
  grid_barrier<true>(a.sync, ++nbar);      // u is read with ordinary loads (every workgroup reads all of it: L2 hits)
  stamp();

  // ---- (t, l): one GRU step of layer l -----------------------------------------------------------------------------------------
  const int ntile = (H + UW - 1) / UW;
  for (int t = 0; t < Tp; ++t) {
    for (int l = 0; l < L; ++l) {
      const int Kin = l == 0 ? (a.m.patch > 0 ? a.m.patch * F : F) : H;
      // layer input rows and the previous state (rows of `states`, or h0 for every row, at t = 0)
      const float* in_base; long long in_ld;
      if (l == 0) { in_base = a.u + (long long)t * (a.m.patch > 0 ? a.m.stride : 1) * F; in_ld = (long long)T * F; }
      else { in_base = a.S + (((long long)(l - 1) * 2 + ((t + 1) & 1)) * B) * H; in_ld = H; }
      const float* hp_base; long long hp_ld;
      if (t == 0) { hp_base = a.states ? a.states + (long long)l * B * H : a.m.h0; hp_ld = a.states ? H : 0; }
      else { hp_base = a.S + (((long long)l * 2 + (t & 1)) * B) * H; hp_ld = H; }
      float* h_new = a.S + (((long long)l * 2 + ((t + 1) & 1)) * B) * H;
      for (int tile = blockIdx.x; tile < ntile; tile += G) {
        const int u0 = tile * UW;
        const int r = min(j, 3 * UW - 1);                       // MFMA row j -> gate r / UW of unit u0 + r % UW (row 15: a copy, ignored)
        const int wrow = (r / UW) * H + min(u0 + r % UW, H - 1);
        unsigned bi[NB], bh[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int b = min(16 * nb + j, B - 1);
          bi[nb] = (unsigned)(b * in_ld * 4);
          bh[nb] = (unsigned)(b * hp_ld * 4);
        }
        const float* wi = a.m.w_ih[l] + (long long)wrow * Kin;
        const float* wh = a.m.w_hh[l] + (long long)wrow * H;
        // the input is u (ordinary loads, after the acquire) for layer 0, the layer below's new state (sc1) above; h_{t-1} is an
        // input of the call at t = 0 (ordinary), a state written one step ago (sc1) later
        if (l == 0 && t == 0) wg_product2<NB, 0>(wi, wh, in_base, hp_base, bi, bh, Kin, H, red, o_ih, o_hh);
        else if (l == 0) {                         // mixed cache policies (an immediate of the load): two products
          wg_product2<NB, 0>(wi, wi, in_base, in_base, bi, bi, Kin, 0, red, o_ih, o_ih);
          wg_product2<NB, XA>(wh, wh, hp_base, hp_base, bh, bh, H, 0, red, o_hh, o_hh);
        } else wg_product2<NB, XA>(wi, wh, in_base, hp_base, bi, bh, Kin, H, red, o_ih, o_hh);   // (sc1 on `states` at t = 0 is harmless)
        for (int e = threadIdx.x; e < UW * 16 * NB; e += NTHR) {
          const int i = e / (16 * NB), b = e % (16 * NB), unit = u0 + i;
          if (unit < H && b < B) {
            const int nb = b >> 4, c = b & 15;
            const float* bih = a.m.b_ih[l]; const float* bhh = a.m.b_hh[l];
            const float gir = o_ih[nb * 256 + (0 * UW + i) * 16 + c] + bih[unit];
            const float giz = o_ih[nb * 256 + (1 * UW + i) * 16 + c] + bih[H + unit];
            const float gin = o_ih[nb * 256 + (2 * UW + i) * 16 + c] + bih[2 * H + unit];
            const float ghr = o_hh[nb * 256 + (0 * UW + i) * 16 + c] + bhh[unit];
            const float ghz = o_hh[nb * 256 + (1 * UW + i) * 16 + c] + bhh[H + unit];
            const float ghn = o_hh[nb * 256 + (2 * UW + i) * 16 + c] + bhh[2 * H + unit];
            const float rg = sigmoidf_(gir + ghr), zg = sigmoidf_(giz + ghz);
            const float ng = tanhf(gin + rg * ghn);
            const float hprev = t == 0 ? hp_base[b * hp_ld + unit] : __hip_atomic_load(hp_base + b * hp_ld + unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float hv = (1.0f - zg) * ng + zg * hprev;
            st_wt(h_new + (long long)b * H + unit, hv);
            if (l == L - 1) st_wt(a.ytop + ((long long)t * B + b) * H + unit, hv);
            if (t == Tp - 1) a.hidden[((long long)l * B + b) * H + unit] = hv;
          }
        }
        __syncthreads();
      }
      stamp();
      grid_barrier<XA == 0>(a.sync, ++nbar);   // the states and y_top are read with sc1 loads
      stamp();
    }
  }

  // ---- H: head.  rows (t, b) of the top layer's outputs against the C rows of W_out -----------------------------------------------
  {
    const int M = Tp * B, ct = (C + 15) / 16, mt = (M + 16 * NB - 1) / (16 * NB);
    for (int tile = blockIdx.x; tile < ct * mt; tile += G) {
      const int c0 = 16 * (tile % ct), m0 = 16 * NB * (tile / ct);
      unsigned br[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) br[nb] = (unsigned)((long long)min(m0 + 16 * nb + j, M - 1) * H * 4);
      const float* wo = a.m.out_w + (long long)min(c0 + j, C - 1) * H;
      wg_product2<NB, XA>(wo, wo, a.ytop, a.ytop, br, br, H, 0, red, o_ih, o_hh);
      for (int e = threadIdx.x; e < NB * 256; e += NTHR) {
        const int nb = e >> 8, c = c0 + ((e & 255) >> 4), m = m0 + 16 * nb + (e & 15);
        if (c < C && m < M) {
          const int t = m / B, b = m % B;
          a.logits[((long long)b * Tp + t) * C + c] = o_ih[e] + a.m.out_b[c];
        }
      }
      __syncthreads();
    }
  }

  stamp();
  // ---- end of call: a timed-out barrier poisons the outputs; the last workgroup re-arms the counters ---------------------------
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0 && __hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) a.logits[0] = __builtin_nanf("");
    const unsigned done = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == G - 1u) {
      for (int g = 0; g < 8; ++g) {
        __hip_atomic_store(a.sync + SW_GRP + 32 * g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sync + SW_FLAG + 32 * g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(a.sync + SW_TOP, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int n_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    n = p.multiProcessorCount;
  }
  return n;
}

}  // namespace
}  // namespace b2t

using namespace b2t;

static int out_frames(const b2t_model_t* m, int T) { return m->patch > 0 ? (T < m->patch ? 0 : (T - m->patch) / m->stride + 1) : T; }

extern "C" int b2t_stream_supported(const b2t_model_t* m, int B, int T) {
  if (!m || B <= 0 || B > 64 || T <= 0) return 0;
  const int Tp = out_frames(m, T);
  if (Tp <= 0 || Tp > 8) return 0;                               // a handful of frames: every (t, l) step is a grid barrier
  if (m->F % 16 || m->H % 16 || m->L <= 0 || m->L > B2T_MAX_LAYERS || m->C <= 0) return 0;
  return 1;
}

extern "C" size_t b2t_stream_sync_bytes(void) { return sizeof(unsigned) * SW_WORDS; }

extern "C" size_t b2t_stream_ws_bytes(const b2t_model_t* m, int B, int T) {
  if (!b2t_stream_supported(m, B, T)) return 0;
  const size_t Tp = (size_t)out_frames(m, T);
  return sizeof(float) * ((size_t)B * T * m->F + (size_t)m->L * 2 * B * m->H + Tp * B * m->H) + 256;
}

extern "C" int b2t_stream_forward_f32(const b2t_model_t* m, int B, int T, const float* x, const int32_t* day_idx, const float* states,
                                      float* logits, float* hidden, void* ws, size_t ws_bytes, void* sync, void* stream) {
  B2T_REQUIRE(m && x && day_idx && logits && hidden && ws && sync, "stream_forward: null buffer");
  B2T_REQUIRE(b2t_stream_supported(m, B, T), "stream_forward: unsupported shape (B <= 64, <= 8 output frames, F and H multiples of 16)");
  B2T_REQUIRE(ws_bytes >= b2t_stream_ws_bytes(m, B, T), "stream_forward: workspace too small");
  B2T_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, "stream_forward: buffers must be 16-byte aligned");
  for (int l = 0; l < m->L; ++l)
    B2T_REQUIRE((reinterpret_cast<uintptr_t>(m->w_ih[l]) & 15) == 0 && (reinterpret_cast<uintptr_t>(m->w_hh[l]) & 15) == 0, "stream_forward: weights must be 16-byte aligned");
  B2T_REQUIRE(!states || (reinterpret_cast<uintptr_t>(states) & 15) == 0, "stream_forward: states must be 16-byte aligned");
  B2T_REQUIRE((reinterpret_cast<uintptr_t>(m->h0) & 15) == 0 && (reinterpret_cast<uintptr_t>(m->out_w) & 15) == 0, "stream_forward: h0 / out.weight must be 16-byte aligned");
  const int G = n_cus();
  B2T_REQUIRE(G > 0, "stream_forward: no device");
  StreamArgs a;
  a.m = *m; a.B = B; a.T = T; a.Tp = out_frames(m, T);
  a.x = x; a.day = day_idx; a.states = states; a.logits = logits; a.hidden = hidden;
  float* w = reinterpret_cast<float*>(ws);
  a.u = w; w += (size_t)B * T * m->F;
  a.S = w; w += (size_t)m->L * 2 * B * m->H;
  a.ytop = w;
  a.sync = reinterpret_cast<unsigned*>(sync);
  hipStream_t s = as_stream(stream);
  if (B <= 16) hipLaunchKernelGGL(stream_forward_kernel<1>, dim3(G), dim3(NTHR), 0, s, a);
  else if (B <= 32) hipLaunchKernelGGL(stream_forward_kernel<2>, dim3(G), dim3(NTHR), 0, s, a);
  else hipLaunchKernelGGL(stream_forward_kernel<4>, dim3(G), dim3(NTHR), 0, s, a);
  B2T_CHECK_LAUNCH("stream_forward_kernel");
  return 0;
}
