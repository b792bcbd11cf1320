// gemm_bf16p.hip — the amp-mode GEMM (use_amp / autocast(bfloat16), rnn_trainer.py:535) in two passes:
//   1. PACK: each operand is read once through the descriptor's generalized addressing (row maps, gaps, either storage
//      order) and written as a dense, k-contiguous bf16 matrix [rows padded to 128][K padded to 64] (round-to-nearest-even,
//      zero fill) into caller-provided scratch;
//   2. GEMM: C = A_p . B_p^T on 128x128x64 tiles, v_mfma_f32_32x32x16_bf16, 16-byte global loads that go to LDS unconverted,
//      register prefetch two k-tiles ahead (two register sets: one tile of MFMA work is shorter than an L2 round trip under
//      load), double-buffered LDS (144-byte rows: a fragment is one conflict-free 16-byte read), one barrier per k-tile, fp32
//      accumulation and the fp32 epilogue of gemm_bf16.hip (bias, Softsign and its backward, accumulate, row-mapped C,
//      split-K slabs).
// Same numerics contract as b2t_gemm_bf16_f32 (operands rounded to bf16, fp32 accumulate, fp32 out).  Why two passes: the
// one-pass kernel converts fp32 operands on their way into LDS and reaches 315 TF/s at 4096^3; with packed operands the
// same tile shape runs at 700-840 TF/s (tools/ubench/gemm_bf16p.hip), and a pack pass is bandwidth-bound and small next
// to it.  Z-batched GEMMs (the day layer): every matrix of the batch packed, grid y of the tile kernel (round 5).
// Round 5: a second tile kernel, 256 x 256 x 64 with 8 waves (gemm_bf16p_kernel256, below), takes the products whose 256-tiles fill
// the chip evenly (the shipped shape's layer-0 weight and input gradients: 660 -> 880 TF/s, bit-identical results).
#include "common.h"
#include "gemm_args.h"

namespace b2t {
namespace {

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

constexpr int PM = 128, PN = 128, PK = 64, PPITCH = PK + 8;   // block tile; LDS row pitch in bf16 elements (144 B)

__device__ __forceinline__ unsigned pk2(float lo, float hi) {
  using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

struct PackArgs {
  const float* P; __bf16* out;
  int rows, rows_pad, K, Kp;
  long long s0, s1; int div;     // row map of the source (rows if k-contiguous, k if row-contiguous)
  int brk, gap;                  // contiguous index i >= brk reads from i + gap (A operand only)
  float* sum; long long sum_ks;  // row-contiguous source only (or null): sum[(k / 64) * sum_ks + r] = fp32 sum of the tile's 64 k
  // k-contiguous DENSE source only (s0 == K, no row map, no gap; round 5): the inter-layer dropout of nn.GRU folded into the pack --
  // element (r, k) is flat element elem0 + r * K + k of the tensor b2t_dropout_f32 masks (same Philox draw, same 1 / (1 - p) scale:
  // bit-identical to dropout-then-pack), and the dropped fp32 values are also written to `dup` (the backward pass reads them)
  float drop_p, drop_scale; unsigned long long drop_seed; long long drop_elem0; float* dup;
  // Z-batched operands (round 5: the day layer's per-sentence GEMMs): matrix z reads from P + (zmap ? zmap[z] : z) * src_sz and is
  // written at out + z * rows_pad * Kp (grid y of pack_kc, grid z of pack_mc); no sums / dropout with Z > 1
  long long src_sz; const int* zmap;
};

// k-contiguous source: element (r, k) at P + rowoff(r) + k (+ gap for k >= brk).  One thread = 8 consecutive k of one row.
__global__ __launch_bounds__(256) void pack_kc_kernel(PackArgs a) {
  const int g8 = a.Kp >> 3;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)a.rows_pad * g8) return;
  const int r = (int)(i / g8), k = (int)(i % g8) * 8;
  const int z = blockIdx.y;
  a.P += (long long)(a.zmap ? a.zmap[z] : z) * a.src_sz;
  a.out += (long long)z * a.rows_pad * a.Kp;
  uint4 o = make_uint4(0u, 0u, 0u, 0u);
  if (r < a.rows && k < a.K) {
    const float* p = a.P + rowoff(r, a.s0, a.s1, a.div) + k + ((a.brk > 0 && k >= a.brk) ? a.gap : 0);
    if (k + 8 <= a.K) {
      float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
      if (a.drop_p > 0.f) {     // (K % 8 == 0 on this path: every lane takes the full-vector branch)
        const long long e = a.drop_elem0 + (long long)r * a.K + k;
        const float4 u0 = Philox::uniform4(a.drop_seed, (uint64_t)(e >> 2), 2u), u1 = Philox::uniform4(a.drop_seed, (uint64_t)(e >> 2) + 1u, 2u);
        v0.x = u0.x >= a.drop_p ? v0.x * a.drop_scale : 0.f; v0.y = u0.y >= a.drop_p ? v0.y * a.drop_scale : 0.f;
        v0.z = u0.z >= a.drop_p ? v0.z * a.drop_scale : 0.f; v0.w = u0.w >= a.drop_p ? v0.w * a.drop_scale : 0.f;
        v1.x = u1.x >= a.drop_p ? v1.x * a.drop_scale : 0.f; v1.y = u1.y >= a.drop_p ? v1.y * a.drop_scale : 0.f;
        v1.z = u1.z >= a.drop_p ? v1.z * a.drop_scale : 0.f; v1.w = u1.w >= a.drop_p ? v1.w * a.drop_scale : 0.f;
        if (a.dup) { float* q = a.dup + (long long)r * a.K + k; *reinterpret_cast<float4*>(q) = v0; *reinterpret_cast<float4*>(q + 4) = v1; }
      }
      o.x = pk2(v0.x, v0.y); o.y = pk2(v0.z, v0.w); o.z = pk2(v1.x, v1.y); o.w = pk2(v1.z, v1.w);
    } else {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = k + j < a.K ? p[j] : 0.f;
      o.x = pk2(v[0], v[1]); o.y = pk2(v[2], v[3]); o.z = pk2(v[4], v[5]); o.w = pk2(v[6], v[7]);
    }
  }
  *reinterpret_cast<uint4*>(a.out + (long long)r * a.Kp + k) = o;
}

// row-contiguous source (k-major storage): element (r, k) at P + rowoff(k) + r (+ gap for r >= brk).  A block transposes a
// 64 (r) x 64 (k) tile through LDS: float4 loads along r, 16-byte bf16 stores along k.
__global__ __launch_bounds__(256) void pack_mc_kernel(PackArgs a) {
  __shared__ float t[64][65];
  const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64, tid = threadIdx.x;
  a.P += (long long)(a.zmap ? a.zmap[blockIdx.z] : (int)blockIdx.z) * a.src_sz;
  a.out += (long long)blockIdx.z * a.rows_pad * a.Kp;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it;          // 1024 float4: k = idx / 16, r4 = (idx % 16) * 4
    const int k = k0 + (idx >> 4), r = r0 + (idx & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < a.K && r < a.rows) {
      const float* p = a.P + rowoff(k, a.s0, a.s1, a.div) + r + ((a.brk > 0 && r >= a.brk) ? a.gap : 0);
      if (r + 4 <= a.rows) v = *reinterpret_cast<const float4*>(p);
      else { v.x = p[0]; if (r + 1 < a.rows) v.y = p[1]; if (r + 2 < a.rows) v.z = p[2]; }
    }
    float* d = &t[idx >> 4][(idx & 15) * 4];
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  if (a.sum && tid < 64 && r0 + tid < a.rows) {   // by-product: the tile's column sums (bias gradients of a GRU layer), from the fp32 values
    float sacc = 0.f;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) sacc += t[k][tid];
    a.sum[(long long)blockIdx.y * a.sum_ks + r0 + tid] = sacc;
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + 256 * it;          // 512 stores: r = idx % 64, k8 = idx / 64
    const int r = idx & 63, k8 = (idx >> 6) * 8;
    uint4 o;
    o.x = pk2(t[k8][r], t[k8 + 1][r]); o.y = pk2(t[k8 + 2][r], t[k8 + 3][r]);
    o.z = pk2(t[k8 + 4][r], t[k8 + 5][r]); o.w = pk2(t[k8 + 6][r], t[k8 + 7][r]);
    if (r0 + r < a.rows_pad && k0 + k8 < a.Kp) *reinterpret_cast<uint4*>(a.out + (long long)(r0 + r) * a.Kp + k0 + k8) = o;
  }
}

#define B2T_PFETCH(R, k0)                                                                                                 \
  R##a0 = *reinterpret_cast<const uint4*>(ag + (k0)); R##a1 = *reinterpret_cast<const uint4*>(ag + 32ll * Kp + (k0));      \
  R##a2 = *reinterpret_cast<const uint4*>(ag + 64ll * Kp + (k0)); R##a3 = *reinterpret_cast<const uint4*>(ag + 96ll * Kp + (k0)); \
  R##b0 = *reinterpret_cast<const uint4*>(bg + (k0)); R##b1 = *reinterpret_cast<const uint4*>(bg + 32ll * Kp + (k0));      \
  R##b2 = *reinterpret_cast<const uint4*>(bg + 64ll * Kp + (k0)); R##b3 = *reinterpret_cast<const uint4*>(bg + 96ll * Kp + (k0));
#define B2T_PSTASH(R, buf)                                                                                                \
  { __bf16* ad = As + (buf) * PM * PPITCH + (tid >> 3) * PPITCH + (tid & 7) * 8;                                          \
    __bf16* bd = Bs + (buf) * PM * PPITCH + (tid >> 3) * PPITCH + (tid & 7) * 8;                                          \
    *reinterpret_cast<uint4*>(ad) = R##a0; *reinterpret_cast<uint4*>(ad + 32 * PPITCH) = R##a1;                           \
    *reinterpret_cast<uint4*>(ad + 64 * PPITCH) = R##a2; *reinterpret_cast<uint4*>(ad + 96 * PPITCH) = R##a3;             \
    *reinterpret_cast<uint4*>(bd) = R##b0; *reinterpret_cast<uint4*>(bd + 32 * PPITCH) = R##b1;                           \
    *reinterpret_cast<uint4*>(bd + 64 * PPITCH) = R##b2; *reinterpret_cast<uint4*>(bd + 96 * PPITCH) = R##b3; }
// tile kt out of LDS buffer CUR; the loads of tile kt + 2 go to register set RL, tile kt + 1 (set RS, requested a tile ago) is stashed
#define B2T_PTILE(CUR, RL, RS)                                                                                            \
    {                                                                                                                     \
      const int k2 = (kt + 2 < nk ? kt + 2 : nk - 1) * PK;                                                                \
      B2T_PFETCH(RL, k2)                                                                                                  \
      const __bf16* a0p = As + (CUR) * PM * PPITCH + (wm * 64 + li) * PPITCH + 8 * lk;                                    \
      const __bf16* a1p = a0p + 32 * PPITCH;                                                                              \
      const __bf16* b0p = Bs + (CUR) * PM * PPITCH + (wn * 64 + li) * PPITCH + 8 * lk;                                    \
      const __bf16* b1p = b0p + 32 * PPITCH;                                                                              \
      _Pragma("unroll")                                                                                                   \
      for (int kk = 0; kk < PK; kk += 16) {                                                                               \
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(a0p + kk), a1 = *reinterpret_cast<const bf16x8*>(a1p + kk);    \
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b0p + kk), b1 = *reinterpret_cast<const bf16x8*>(b1p + kk);    \
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc00, 0, 0, 0);                                          \
        acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc01, 0, 0, 0);                                          \
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc10, 0, 0, 0);                                          \
        acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc11, 0, 0, 0);                                          \
      }                                                                                                                   \
      B2T_PSTASH(RS, (CUR) ^ 1)                                                                                           \
      __syncthreads();                                                                                                    \
    }

// (plain variables for the prefetch registers, no lambdas over arrays: the array form was demoted to scratch memory by the
// compiler -- 230 TF/s instead of 840)
__global__ __launch_bounds__(256, 2) void gemm_bf16p_kernel(GemmArgs g, const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, int Kp,
                                                            long long a_zs, long long b_zs) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 2 * PM * PPITCH];
  __bf16* As = smem; __bf16* Bs = smem + 2 * PM * PPITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  int ks = blockIdx.z;
  int m0, n0;
  {   // XCD-aware block order (see gemm.hip): a K slice's tiles on one XCD when the slice count allows, else tiles dealt to the XCDs
    const int gx = (g.N + PN - 1) / PN, nwg = gridDim.x;
    int tile;
    if (g.ks_xcd) {
      const int lin = blockIdx.x + nwg * blockIdx.z, xcd = lin & 7, j = lin >> 3;
      ks = xcd + 8 * (j / nwg); tile = j % nwg;
    } else {
      const int b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
      tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    }
    // Round 5: an XCD's workgroups run ~64 CONSECUTIVE tiles at a time in step along k.  Numbered row-major, 64 tiles of a wide GEMM
    // (the shipped shape's layer 0: 56 tile columns) are one tile row: 1 A panel + 56 B panels fetched per k step -- 2 GB of L2 misses
    // per 258-GFLOP GEMM, 400-470 TF/s.  Numbered down groups of tile_gm rows (8 x 8 blocks) they share 8 + 8 panels.
    if (g.tile_gm > 1) {
      const int gy = nwg / gx, per_group = g.tile_gm * gx;
      const int grp = tile / per_group, first_m = grp * g.tile_gm;
      const int gm_eff = min(g.tile_gm, gy - first_m), r = tile - grp * per_group;
      m0 = (first_m + r % gm_eff) * PM; n0 = (r / gm_eff) * PN;
    } else {
      m0 = (tile / gx) * PM; n0 = (tile % gx) * PN;
    }
  }
  const int z = blockIdx.y;                        // Z-batched: matrix z of the packed operands, of C, of ep_aux; bias by b_zmap
  const float* bias = (g.bias && ks == 0) ? g.bias + (long long)(g.b_zmap ? g.b_zmap[z] : z) * g.bias_sz : nullptr;
  float* C = g.C + (long long)z * g.c_sz + (long long)ks * g.c_ks;
  const float* ep_aux = g.ep_aux ? g.ep_aux + (long long)z * g.c_sz : nullptr;
  const int kb = ks * g.kchunk, ke = min(Kp, kb + g.kchunk), nk = (ke - kb) / PK;
  const __bf16* ag = Ap + z * a_zs + (long long)(m0 + (tid >> 3)) * Kp + (tid & 7) * 8 + kb;
  const __bf16* bg = Bp + z * b_zs + (long long)(n0 + (tid >> 3)) * Kp + (tid & 7) * 8 + kb;
  uint4 r0a0, r0a1, r0a2, r0a3, r0b0, r0b1, r0b2, r0b3;   // two register sets (as in gemm_bf16p_kernel256): tile t + 2 is requested while
  uint4 r1a0, r1a1, r1a2, r1a3, r1b0, r1b1, r1b2, r1b3;   // tile t is multiplied
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  const int lk = lane >> 5, li = lane & 31;
  if (nk > 0) {
    B2T_PFETCH(r0, 0) B2T_PSTASH(r0, 0)
    B2T_PFETCH(r0, (nk > 1 ? 1 : 0) * PK)
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      B2T_PTILE(0, r1, r0)
      ++kt;
      B2T_PTILE(1, r0, r1)
      --kt;
    }
    if (kt < nk) { B2T_PTILE(0, r1, r0) }
  }
  // epilogue (gemm_bf16.hip): C/D layout of the 32x32 MFMAs: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const f32x16 acc[2][2] = {{acc00, acc01}, {acc10, acc11}};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        if (row < g.M) {
          float v = acc[i][j][e] + bv;
          if (g.epilogue == 1) v = v / (1.0f + fabsf(v));
          const long long coff = rowoff(row, g.c_s0, g.c_s1, g.c_div) + col;
          if (g.epilogue == 2) { const float a = 1.0f - fabsf(ep_aux[coff]); v *= a * a; }   // softsign backward
          float* p = C + coff;
          if (g.accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}


// ---- 256 x 256 x 64 tiles, 8 waves (2 x 4), a wave owns 128 x 64 (round 5) -----------------------------------------------------
// Why: in the 128 x 128 kernel a wave's 64 x 64 sub-tile reads 4 fragments from LDS per 4 MFMAs; with two workgroups per CU that is
// 256 LDS cycles per k16 step next to 256 MFMA cycles per SIMD -- the LDS port is as busy as the matrix cores, which is the
// 450-650 TF/s the chip-filling GEMMs of the shipped shape's layer 0 run at.  A 128 x 64 wave tile reads 6 fragments per 8 MFMAs
// (0.75 of the MFMA time), and a 256-row block fetches each operand panel half as often.  Same structure otherwise: register
// prefetch of the next k tile, double-buffered LDS (144-byte rows), one barrier per k tile; one workgroup (144 KB of LDS) per CU,
// two waves per SIMD.  Used when the tile count fills the chip evenly (gemm_bf16p_run); operand rows beyond the packed matrix
// (padded to 128) are clamped -- those output rows are never stored.
constexpr int QM = 256, QN = 256;
constexpr size_t Q_LDS = (size_t)2 * (QM + QN) * PPITCH * sizeof(__bf16);

#define B2T_QFETCH(R, k0)                                                                                                 \
  R##a0 = *reinterpret_cast<const uint4*>(Ap + ao0 + (k0)); R##a1 = *reinterpret_cast<const uint4*>(Ap + ao1 + (k0));      \
  R##a2 = *reinterpret_cast<const uint4*>(Ap + ao2 + (k0)); R##a3 = *reinterpret_cast<const uint4*>(Ap + ao3 + (k0));      \
  R##b0 = *reinterpret_cast<const uint4*>(Bp + bo0 + (k0)); R##b1 = *reinterpret_cast<const uint4*>(Bp + bo1 + (k0));      \
  R##b2 = *reinterpret_cast<const uint4*>(Bp + bo2 + (k0)); R##b3 = *reinterpret_cast<const uint4*>(Bp + bo3 + (k0));
#define B2T_QSTASH(R, buf)                                                                                                \
  { __bf16* ad = As + (buf) * QM * PPITCH + (tid >> 3) * PPITCH + (tid & 7) * 8;                                          \
    __bf16* bd = Bs + (buf) * QN * PPITCH + (tid >> 3) * PPITCH + (tid & 7) * 8;                                          \
    *reinterpret_cast<uint4*>(ad) = R##a0; *reinterpret_cast<uint4*>(ad + 64 * PPITCH) = R##a1;                           \
    *reinterpret_cast<uint4*>(ad + 128 * PPITCH) = R##a2; *reinterpret_cast<uint4*>(ad + 192 * PPITCH) = R##a3;           \
    *reinterpret_cast<uint4*>(bd) = R##b0; *reinterpret_cast<uint4*>(bd + 64 * PPITCH) = R##b1;                           \
    *reinterpret_cast<uint4*>(bd + 128 * PPITCH) = R##b2; *reinterpret_cast<uint4*>(bd + 192 * PPITCH) = R##b3; }

__global__ __launch_bounds__(512) void gemm_bf16p_kernel256(GemmArgs g, const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, int Kp, int Mp, int Np) {
  extern __shared__ __attribute__((aligned(16))) __bf16 qsmem[];
  __bf16* As = qsmem; __bf16* Bs = qsmem + 2 * QM * PPITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  int ks = blockIdx.z;
  int m0, n0;
  {
    const int gx = (Np + QN - 1) / QN, nwg = gridDim.x;
    int tile;
    if (g.ks_xcd) {
      const int lin = blockIdx.x + nwg * blockIdx.z, xcd = lin & 7, j = lin >> 3;
      ks = xcd + 8 * (j / nwg); tile = j % nwg;
    } else {
      const int b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
      tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    }
    if (g.tile_gm > 1) {
      const int gy = nwg / gx, per_group = g.tile_gm * gx;
      const int grp = tile / per_group, first_m = grp * g.tile_gm;
      const int gm_eff = min(g.tile_gm, gy - first_m), r = tile - grp * per_group;
      m0 = (first_m + r % gm_eff) * QM; n0 = (r / gm_eff) * QN;
    } else {
      m0 = (tile / gx) * QM; n0 = (tile % gx) * QN;
    }
  }
  const float* bias = (g.bias && ks == 0) ? g.bias : nullptr;
  float* C = g.C + (long long)ks * g.c_ks;
  const int kb = ks * g.kchunk, ke = min(Kp, kb + g.kchunk), nk = (ke - kb) / PK;
  const int r0 = tid >> 3, kc = (tid & 7) * 8 + kb;
  const long long ao0 = (long long)min(m0 + r0, Mp - 1) * Kp + kc, ao1 = (long long)min(m0 + r0 + 64, Mp - 1) * Kp + kc;
  const long long ao2 = (long long)min(m0 + r0 + 128, Mp - 1) * Kp + kc, ao3 = (long long)min(m0 + r0 + 192, Mp - 1) * Kp + kc;
  const long long bo0 = (long long)min(n0 + r0, Np - 1) * Kp + kc, bo1 = (long long)min(n0 + r0 + 64, Np - 1) * Kp + kc;
  const long long bo2 = (long long)min(n0 + r0 + 128, Np - 1) * Kp + kc, bo3 = (long long)min(n0 + r0 + 192, Np - 1) * Kp + kc;
  uint4 r0a0, r0a1, r0a2, r0a3, r0b0, r0b1, r0b2, r0b3;   // two register sets: the global loads of tile t + 2 are issued while tile t is
  uint4 r1a0, r1a1, r1a2, r1a3, r1b0, r1b1, r1b2, r1b3;   // multiplied and land 1.5 tiles later (one set: half a tile, less than an L2 round trip under load)
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int lk = lane >> 5, li = lane & 31;
  if (nk > 0) {
    // One k tile: [MFMAs of k16 step 0 on fragments read behind the PREVIOUS barrier] [step 1] [stash of the next tile: its global
    // loads were issued at the top] [step 2] [fragment reads of step 3] BARRIER [fragment reads of the next tile's step 0]
    // [MFMAs of step 3].  The stash, the barrier and the first LDS reads of a tile all sit under MFMA work of the same wave.
    // Hazards: the stash of tile t+1 goes to the buffer tile t-1 was read from -- every wave's last read of t-1 is before barrier
    // t-1, the stash is behind it; tile t+1 is read behind barrier t, which is behind every wave's stash.
    B2T_QFETCH(r0, 0) B2T_QSTASH(r0, 0)
    B2T_QFETCH(r0, (nk > 1 ? 1 : 0) * PK)                                  // tile 1, stashed in the middle of tile 0
    __syncthreads();
    const __bf16* abase = As + (wm * 128 + li) * PPITCH + 8 * lk;
    const __bf16* bbase = Bs + (wn * 64 + li) * PPITCH + 8 * lk;
    bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(abase), fa1 = *reinterpret_cast<const bf16x8*>(abase + 32 * PPITCH);
    bf16x8 fa2 = *reinterpret_cast<const bf16x8*>(abase + 64 * PPITCH), fa3 = *reinterpret_cast<const bf16x8*>(abase + 96 * PPITCH);
    bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(bbase), fb1 = *reinterpret_cast<const bf16x8*>(bbase + 32 * PPITCH);
#define B2T_QMMA(a0, a1, a2, a3, b0, b1)                                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);                            \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);                            \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);                            \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);                            \
    acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[2][0], 0, 0, 0);                            \
    acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[2][1], 0, 0, 0);                            \
    acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b0, acc[3][0], 0, 0, 0);                            \
    acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[3][1], 0, 0, 0);
#define B2T_QFRAGS(ap_, bp_, kk)                                                                               \
    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>((ap_) + (kk)), x1 = *reinterpret_cast<const bf16x8*>((ap_) + 32 * PPITCH + (kk)); \
    const bf16x8 x2 = *reinterpret_cast<const bf16x8*>((ap_) + 64 * PPITCH + (kk)), x3 = *reinterpret_cast<const bf16x8*>((ap_) + 96 * PPITCH + (kk)); \
    const bf16x8 y0 = *reinterpret_cast<const bf16x8*>((bp_) + (kk)), y1 = *reinterpret_cast<const bf16x8*>((bp_) + 32 * PPITCH + (kk));
// tile kt out of LDS buffer CUR: loads of tile kt + 2 into set RL, tile kt + 1 (set RS, loaded a tile ago) stashed into the other buffer
#define B2T_QTILE(CUR, RL, RS)                                                                                 \
    {                                                                                                          \
      const int k2 = (kt + 2 < nk ? kt + 2 : nk - 1) * PK;                                                     \
      B2T_QFETCH(RL, k2)                                                                                       \
      const __bf16* ap = abase + (CUR) * QM * PPITCH;                                                          \
      const __bf16* bp = bbase + (CUR) * QN * PPITCH;                                                          \
      B2T_QMMA(fa0, fa1, fa2, fa3, fb0, fb1)                                                                   \
      { B2T_QFRAGS(ap, bp, 16) B2T_QMMA(x0, x1, x2, x3, y0, y1) }                                              \
      B2T_QSTASH(RS, (CUR) ^ 1)                                                                                \
      { B2T_QFRAGS(ap, bp, 32) B2T_QMMA(x0, x1, x2, x3, y0, y1) }                                              \
      {                                                                                                        \
        B2T_QFRAGS(ap, bp, 48)                                                                                 \
        __syncthreads();                                                                                       \
        const __bf16* an = abase + ((CUR) ^ 1) * QM * PPITCH;                                                  \
        const __bf16* bn = bbase + ((CUR) ^ 1) * QN * PPITCH;                                                  \
        fa0 = *reinterpret_cast<const bf16x8*>(an); fa1 = *reinterpret_cast<const bf16x8*>(an + 32 * PPITCH);  \
        fa2 = *reinterpret_cast<const bf16x8*>(an + 64 * PPITCH); fa3 = *reinterpret_cast<const bf16x8*>(an + 96 * PPITCH); \
        fb0 = *reinterpret_cast<const bf16x8*>(bn); fb1 = *reinterpret_cast<const bf16x8*>(bn + 32 * PPITCH);  \
        B2T_QMMA(x0, x1, x2, x3, y0, y1)                                                                       \
      }                                                                                                        \
    }
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      B2T_QTILE(0, r1, r0)
      ++kt;
      B2T_QTILE(1, r0, r1)
      --kt;
    }
    if (kt < nk) { B2T_QTILE(0, r1, r0) }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        if (row < g.M) {
          float v = acc[i][j][e] + bv;
          if (g.epilogue == 1) v = v / (1.0f + fabsf(v));
          const long long coff = rowoff(row, g.c_s0, g.c_s1, g.c_div) + col;
          if (g.epilogue == 2) { const float a = 1.0f - fabsf(g.ep_aux[coff]); v *= a * a; }
          float* p = C + coff;
          if (g.accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}

int pad_to(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace
}  // namespace b2t

extern "C" size_t b2t_gemm_bf16p_ws_bytes(int M, int N, int K) {
  using namespace b2t;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const size_t Kp = (size_t)pad_to(K, PK);
  return ((size_t)pad_to(M, PM) + (size_t)pad_to(N, PN)) * Kp * sizeof(__bf16) + 512;
}
// Z-batched (d->Z > 1): every matrix of the batch is packed, Z times the operands of one
extern "C" size_t b2t_gemm_bf16p_ws_bytes_z(int M, int N, int K, int Z) {
  using namespace b2t;
  if (M <= 0 || N <= 0 || K <= 0 || Z <= 0) return 0;
  const size_t Kp = (size_t)pad_to(K, PK);
  return (size_t)Z * ((size_t)pad_to(M, PM) + (size_t)pad_to(N, PN)) * Kp * sizeof(__bf16) + 512;
}

// ---- internal (csrc/exec.cpp): operands packed ahead of the GEMM, or by someone else ----------------------------------------
// gemm_bf16p_pack: one operand of the GEMM `d` describes (which = 0: A, 1: B) into `out` (gemm_bf16p_operand_bytes); the weights of a
// pass are packed ONCE (they change once per step, and a pipelined pass multiplies by them once per time chunk), and the A pack of the
// forward's inter-layer projections carries nn.GRU's dropout (PackDrop).  gemm_bf16p_run: the GEMM with either operand pre-packed
// (null: packed here into `ws` as b2t_gemm_bf16p_f32 does).
namespace b2t {

size_t gemm_bf16p_operand_bytes(int rows, int K) {
  return (size_t)pad_to(rows, PM) * (size_t)pad_to(K, PK) * sizeof(__bf16);
}

static void launch_pack(const float* P, __bf16* out, int rows, int rows_pad, int K, int Kp, bool kc, long long s0, long long s1, int div, int brk,
                        int gap, float* sum, long long sum_ks, const PackDrop* drop, hipStream_t s, int Z = 1, long long src_sz = 0,
                        const int* zmap = nullptr) {
  PackArgs a{P, out, rows, rows_pad, K, Kp, s0, s1, div, brk, gap, sum, sum_ks, 0.f, 1.f, 0ull, 0ll, nullptr, src_sz, zmap};
  if (drop && drop->p > 0.f) { a.drop_p = drop->p; a.drop_scale = 1.0f / (1.0f - drop->p); a.drop_seed = drop->seed; a.drop_elem0 = drop->elem0; a.dup = drop->dup; }
  if (kc) {
    const long long items = (long long)rows_pad * (Kp / 8);
    hipLaunchKernelGGL(pack_kc_kernel, dim3((unsigned)((items + 255) / 256), Z), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(pack_mc_kernel, dim3(rows_pad / 64, Kp / 64, Z), dim3(256), 0, s, a);
  }
}

int gemm_bf16p_pack(const b2t_gemm_desc* d, int which, void* out, hipStream_t s, const PackDrop* drop) {
  B2T_REQUIRE(d && out && (which == 0 || which == 1), "gemm_bf16p_pack: bad arguments");
  B2T_REQUIRE(((uintptr_t)out & 255) == 0, "gemm_bf16p_pack: the packed operand must be 256-byte aligned");
  const int Kp = pad_to(d->K, PK);
  const int Z = d->Z > 1 ? d->Z : 1;
  B2T_REQUIRE(Z == 1 || ((!drop || drop->p <= 0.f) && d->a_sum == nullptr && d->a_brk == 0), "gemm_bf16p_pack: Z-batched operands carry no dropout / sums / gap");
  if (which == 0) {
    B2T_REQUIRE(!drop || drop->p <= 0.f || (d->a_kcontig && d->a_s0 == d->K && d->a_div == 0 && d->a_brk == 0 && d->K % 8 == 0 && (drop->elem0 % 4) == 0),
                "gemm_bf16p_pack: dropout goes with a dense k-contiguous A (row stride = K, K %% 8 == 0)");
    B2T_REQUIRE(d->a_brk == 0 || d->a_brk % 8 == 0, "gemm_bf16p_pack: a_brk must be a multiple of 8");
    launch_pack(d->A, reinterpret_cast<__bf16*>(out), d->M, pad_to(d->M, PM), d->K, Kp, d->a_kcontig != 0, d->a_s0, d->a_s1, d->a_div, d->a_brk, d->a_gap,
                d->a_sum, d->a_sum_ks, drop, s, Z, d->a_sz, nullptr);
  } else {
    B2T_REQUIRE(!drop || drop->p <= 0.f, "gemm_bf16p_pack: dropout is for the A operand");
    launch_pack(d->B, reinterpret_cast<__bf16*>(out), d->N, pad_to(d->N, PN), d->K, Kp, d->b_kcontig != 0, d->b_s0, d->b_s1, d->b_div, 0, 0, nullptr, 0, nullptr, s,
                Z, d->b_sz, d->b_zmap);
  }
  B2T_CHECK_LAUNCH("gemm_bf16p_pack");
  return 0;
}

int gemm_bf16p_run(const b2t_gemm_desc* d, const void* Ap_pre, const void* Bp_pre, void* ws, size_t ws_bytes, hipStream_t s, const PackDrop* dropA) {
  B2T_REQUIRE(d != nullptr && (ws != nullptr || (Ap_pre && Bp_pre)), "b2t_gemm_bf16p_f32: null descriptor / workspace");
  B2T_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->Z >= 1 && (d->Z > 1 || d->b_zmap == nullptr), "b2t_gemm_bf16p_f32: bad shape M=%d N=%d K=%d Z=%d",
              d->M, d->N, d->K, d->Z);
  const int Z = d->Z;
  B2T_REQUIRE(Z == 1 || (d->splitk <= 1 && (d->a_sz % 4) == 0 && (d->b_sz % 4) == 0 && Z <= 65535), "b2t_gemm_bf16p_f32: Z-batched: no split-K, z strides multiples of 4, Z <= 65535");
  B2T_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0 && ((uintptr_t)ws & 255) == 0, "b2t_gemm_bf16p_f32: A/B must be 16-byte, ws 256-byte aligned");
  B2T_REQUIRE((d->a_s0 % 4) == 0 && (d->a_s1 % 4) == 0 && (d->b_s0 % 4) == 0 && (d->b_s1 % 4) == 0,
              "b2t_gemm_bf16p_f32: A/B strides must be multiples of 4 elements");
  const int Mp = pad_to(d->M, PM), Np = pad_to(d->N, PN), Kp = pad_to(d->K, PK);
  const size_t needA = Ap_pre ? 0 : (size_t)Z * Mp * Kp * sizeof(__bf16), needB = Bp_pre ? 0 : (size_t)Z * Np * Kp * sizeof(__bf16);
  B2T_REQUIRE(ws_bytes >= needA + needB, "b2t_gemm_bf16p_f32: workspace of %zu bytes, need %zu", ws_bytes, needA + needB);
  GemmArgs g;
  B2T_REQUIRE(d->a_sum == nullptr || !d->a_kcontig, "b2t_gemm_bf16p_f32: a_sum goes with an m-contiguous A (a_kcontig = 0)");
  B2T_REQUIRE(d->a_sum == nullptr || !Ap_pre, "b2t_gemm_bf16p_f32: a_sum is a by-product of packing A");
  { int rc = fill_gemm_args(d, g, PK, PM, "b2t_gemm_bf16p_f32"); if (rc) return rc; }
  B2T_REQUIRE(d->a_brk == 0 || d->a_brk % 8 == 0, "b2t_gemm_bf16p_f32: a_brk must be a multiple of 8");
  __bf16* Aw = reinterpret_cast<__bf16*>(ws);
  __bf16* Bw = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(ws) + needA);
  if (!Ap_pre) { int rc = gemm_bf16p_pack(d, 0, Aw, s, dropA); if (rc) return rc; }
  if (!Bp_pre) { int rc = gemm_bf16p_pack(d, 1, Bw, s, nullptr); if (rc) return rc; }
  const __bf16* Ap = Ap_pre ? reinterpret_cast<const __bf16*>(Ap_pre) : Aw;
  const __bf16* Bp = Bp_pre ? reinterpret_cast<const __bf16*>(Bp_pre) : Bw;
  g.kchunk = pad_to((Kp + g.splitk - 1) / g.splitk, PK);
  dim3 grid((Np / PN) * (Mp / PM), Z, g.splitk), block(256);
  { static const bool off = getenv("B2T_GEMM_KS_XCD") && atoi(getenv("B2T_GEMM_KS_XCD")) == 0; g.ks_xcd = !off && g.splitk >= 8 && (g.splitk & 7) == 0; }
  {   // grouped tile order for wide GEMMs (B2T_GEMM_GM: 0 = row-major always, n = groups of n tile rows wherever there are > 16 tile columns)
    static const int gm_env = getenv("B2T_GEMM_GM") ? atoi(getenv("B2T_GEMM_GM")) : 8;
    g.tile_gm = (gm_env > 1 && Np / PN > 16 && Mp / PM >= 2) ? gm_env : 1;
  }
  {   // the 256 x 256 kernel where its tiles fill the chip evenly (one workgroup per CU): B2T_GEMM_256 = 0 never, 1 (default) by this rule, 2 whenever there are >= 64 tiles
    const char* e = getenv("B2T_GEMM_256");
    const int mode = e ? atoi(e) : 1;
    const long long t256 = (long long)((Mp + QM - 1) / QM) * ((Np + QN - 1) / QN) * g.splitk;
    const long long rounds = (t256 + 255) / 256;
    const bool fills = t256 >= 200 && t256 * 100 >= rounds * 256 * 80;
    if (Z == 1 && (mode == 2 ? t256 >= 64 : (mode == 1 && fills))) {
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel256), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q_LDS);
      B2T_REQUIRE(attr == hipSuccess, "b2t_gemm_bf16p_f32: %zu bytes of LDS refused", Q_LDS);
      g.tile_gm = (g.tile_gm > 1 && (Np + QN - 1) / QN > 8 && (Mp + QM - 1) / QM >= 2) ? 4 : 1;
      dim3 grid2(((Np + QN - 1) / QN) * ((Mp + QM - 1) / QM), 1, g.splitk);
      hipLaunchKernelGGL(gemm_bf16p_kernel256, grid2, dim3(512), Q_LDS, s, g, Ap, Bp, Kp, Mp, Np);
      B2T_CHECK_LAUNCH("b2t_gemm_bf16p_f32 (256)");
      return 0;
    }
  }
  hipLaunchKernelGGL(gemm_bf16p_kernel, grid, block, 0, s, g, Ap, Bp, Kp, (long long)Mp * Kp, (long long)Np * Kp);
  B2T_CHECK_LAUNCH("b2t_gemm_bf16p_f32");
  return 0;
}

}  // namespace b2t

extern "C" int b2t_gemm_bf16p_f32(const b2t_gemm_desc* d, void* ws, size_t ws_bytes, void* stream) {
  using namespace b2t;
  B2T_REQUIRE(d != nullptr && ws != nullptr, "b2t_gemm_bf16p_f32: null descriptor / workspace");
  B2T_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "b2t_gemm_bf16p_f32: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  B2T_REQUIRE(d->Z >= 1 && ws_bytes >= b2t_gemm_bf16p_ws_bytes_z(d->M, d->N, d->K, d->Z), "b2t_gemm_bf16p_f32: workspace of %zu bytes, need %zu", ws_bytes,
              b2t_gemm_bf16p_ws_bytes_z(d->M, d->N, d->K, d->Z > 0 ? d->Z : 1));
  return gemm_bf16p_run(d, nullptr, nullptr, ws, ws_bytes, as_stream(stream), nullptr);
}
