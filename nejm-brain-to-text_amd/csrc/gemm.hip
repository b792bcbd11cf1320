// gemm.hip — exact-fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32).
//
// C[z][m][n] (+)= sum_k A(z,m,k) * B(z,n,k) (+ bias[n]) -> optional softsign.
// 128x128xBK block tile, 256 threads = 4 wave64 in a 2x2 grid, each wave owns a 64x64 sub-tile =
// 2x2 MFMA 32x32 accumulators (64 accumulator VGPRs).  Operands are staged through LDS k-major
// ([k][m]) so an MFMA fragment read is one conflict-free ds_read_b32 per operand (lanes 0-31 read 32
// consecutive floats); fragment reads run one k-step ahead of the MFMAs that consume them.
// Global->register prefetch of tile k+1 overlaps the MFMAs of tile k; LDS is double buffered (one barrier
// per k-tile).  Full k-tiles take a branch-free load path (out-of-range rows/columns are clamped to valid
// addresses: they only feed accumulator rows/columns that are never stored); only a partial last k-tile
// takes the predicated path.  The f32-input MFMA is bit-for-bit a k-ordered fmaf chain (no reduced-
// precision path exists on gfx950), so results match an fp32 CPU GEMM to summation-order roundoff.
#include "common.h"
#include "gemm_args.h"

namespace b2t {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));

#ifndef B2T_GEMM_BK
#define B2T_GEMM_BK 16
#endif
#ifndef B2T_GEMM_OCC
#define B2T_GEMM_OCC 3
#endif
constexpr int BM = 128, BN = 128, BKT = B2T_GEMM_BK;
constexpr int NLD = BKT / 8;       // float4 loads per thread per operand tile (128 x BKT floats / 256 threads)
constexpr int TPR = BKT / 4;       // k-contiguous mode: threads per tile row
constexpr int RPP = 256 / TPR;     // k-contiguous mode: rows covered per pass
// LDS row pitch (floats).  k-contiguous operands are transposed on the way in (4 scalar ds_write_b32 per
// float4): an odd pitch spreads the lanes that share a row over distinct banks.  m-contiguous operands are
// stored as float4 rows and need a 16-byte aligned pitch.
template <bool KC> struct Pitch { static constexpr int v = KC ? 129 : 132; };
constexpr int PITCH_MAX = 132;

// Load one operand tile slice owned by this thread into 2 float4 registers.
// KC (k contiguous): tile rows are the M/N index (128 of them), 16 k per row -> thread (row=t/4+64r, k4=t%4)
// MC (m contiguous): tile rows are k (16 of them), 128 m per row        -> thread (krow=t/32+8r, m4=t%32)
// ---- fast path: a full k-tile, no predicates --------------------------------------------------------
// KC (k contiguous): thread (row = tid/TPR + RPP*r, k4 = tid%TPR); roff[r] = offset of its (clamped) row.
// MC (m contiguous): thread (krow = tid/32 + 8*r, m4 = tid%32);    mcl = clamped, 4-aligned column offset.
template <bool KC>
__device__ __forceinline__ void load_full(const float* __restrict__ P, const long long (&roff)[NLD], int mcl, int k0,
                                          long long s0, long long s1, int div, float4 (&v)[NLD], int tid) {
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    if constexpr (KC) {
      v[r] = *reinterpret_cast<const float4*>(P + roff[r] + k0 + (tid % TPR) * 4);
    } else {
      v[r] = *reinterpret_cast<const float4*>(P + rowoff(k0 + (tid >> 5) + 8 * r, s0, s1, div) + mcl);
    }
  }
}

// ---- slow path: partial last k-tile (zero-filled beyond K, rows/columns beyond the extent zero) ------
template <bool KC>
__device__ __forceinline__ void load_tail(const float* __restrict__ P, const long long (&roff)[NLD], int ext_m,
                                          int base_m, int k0, int Kend, long long s0, long long s1, int div,
                                          float4 (&v)[NLD], int tid) {
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KC) {
      const int row = base_m + tid / TPR + RPP * r;
      const int k = k0 + (tid % TPR) * 4;
      if (row < ext_m) {
        const float* p = P + roff[r] + k;
        if (k < Kend) o.x = p[0];
        if (k + 1 < Kend) o.y = p[1];
        if (k + 2 < Kend) o.z = p[2];
        if (k + 3 < Kend) o.w = p[3];
      }
    } else {
      const int k = k0 + (tid >> 5) + 8 * r;
      const int m = base_m + (tid & 31) * 4;
      if (k < Kend) {
        const float* p = P + rowoff(k, s0, s1, div) + m;
        if (m < ext_m) o.x = p[0];
        if (m + 1 < ext_m) o.y = p[1];
        if (m + 2 < ext_m) o.z = p[2];
        if (m + 3 < ext_m) o.w = p[3];
      }
    }
    v[r] = o;
  }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float4 (&v)[NLD], int tid) {
  constexpr int PITCH = Pitch<KC>::v;
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    if constexpr (KC) {
      const int row = tid / TPR + RPP * r;
      const int k = (tid % TPR) * 4;
      S[(k + 0) * PITCH + row] = v[r].x;
      S[(k + 1) * PITCH + row] = v[r].y;
      S[(k + 2) * PITCH + row] = v[r].z;
      S[(k + 3) * PITCH + row] = v[r].w;
    } else {
      const int k = (tid >> 5) + 8 * r;
      const int m = (tid & 31) * 4;
      *reinterpret_cast<float4*>(&S[k * PITCH + m]) = v[r];
    }
  }
}

// 16-byte device-scope (sc1) load: never served from a stale line of this XCD's L2 (slab tiles written by other XCDs)
__device__ __forceinline__ float4 load_f4_sc1(const float* base_uniform, unsigned byte_off) {
  using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, B2T_GEMM_OCC) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BKT * PITCH_MAX];
  constexpr int PA = Pitch<AKC>::v, PB = Pitch<BKC>::v;
  float* As = smem;
  float* Bs = smem + 2 * BKT * PITCH_MAX;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int z = blockIdx.z / g.splitk;
  int ks = blockIdx.z - z * g.splitk;
  // XCD-aware tile order: the dispatcher sends workgroup b (linear id, x fastest) to XCD b % 8 (private L2 each).
  int m0, n0;
  {
    const int gx = (g.N + BN - 1) / BN, nwg = gridDim.x;
    int tile;
    if (g.ks_xcd) {
      // split-K with a multiple of 8 slices (Z = 1): ALL tiles of a K slice on one XCD.  The slice's rows of A and B are
      // then fetched by one L2 only and shared there by the tiles that walk them in step (with tiles dealt to the XCDs
      // instead, every XCD fetches every slice of B, and a slice of A goes to two of them).
      const int lin = blockIdx.x + nwg * blockIdx.z, xcd = lin & 7, j = lin >> 3;
      ks = xcd + 8 * (j / nwg); tile = j % nwg; z = 0;
    } else {
      // every XCD walks a CONTIGUOUS range of tiles in row-major order: the n-tiles that share an A row-panel hit the
      // same L2 instead of fetching the panel once per XCD.  Bijective for any tile count (cdna_hip_programming.md T1).
      const int b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
      tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    }
    m0 = (tile / gx) * BM; n0 = (tile % gx) * BN;
  }

  const float* A = g.A + (long long)z * g.a_sz;
  const int zb = g.b_zmap ? g.b_zmap[z] : z;
  const float* B = g.B + (long long)zb * g.b_sz;
  const float* bias = (g.bias && ks == 0) ? g.bias + (long long)zb * g.bias_sz : nullptr;
  float* C = g.C + (long long)z * g.c_sz + (long long)ks * g.c_ks;

  // per-thread row offsets (k-contiguous operands) / clamped column offsets (m-contiguous operands)
  long long roffA[NLD], roffB[NLD];
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    const int ra_ = m0 + tid / TPR + RPP * r, rb_ = n0 + tid / TPR + RPP * r;
    roffA[r] = AKC ? rowoff(ra_ < g.M ? ra_ : g.M - 1, g.a_s0, g.a_s1, g.a_div) : 0;
    roffB[r] = BKC ? rowoff(rb_ < g.N ? rb_ : g.N - 1, g.b_s0, g.b_s1, g.b_div) : 0;
  }
  // m-contiguous: a float4 that starts inside the (4-padded) row stays inside the row pitch; one that starts
  // beyond it is redirected to the last in-range float4.
  const int mclA = min(m0 + (tid & 31) * 4, ((g.M + 3) & ~3) - 4);
  const int mclB = min(n0 + (tid & 31) * 4, ((g.N + 3) & ~3) - 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kb = ks * g.kchunk;                      // split-K: this block reduces k in [kb, K)
  const int K = min(g.K, kb + g.kchunk);
  const int nk = (K - kb + BKT - 1) / BKT;
  const int nfull = (K - kb) / BKT;                  // tiles that need no k predicate
  float4 ra[NLD], rb[NLD];

  auto fetch = [&](int kt) {
    const int k0 = kb + kt * BKT;
    // gap in A's contiguous index (tile-uniform: a_brk is a multiple of the tile extent): shift the base pointer
    const float* Ag = A + ((g.a_brk > 0 && (AKC ? k0 : m0) >= g.a_brk) ? g.a_gap : 0);
    if (kt < nfull) {
      load_full<AKC>(Ag, roffA, mclA, k0, g.a_s0, g.a_s1, g.a_div, ra, tid);
      load_full<BKC>(B, roffB, mclB, k0, g.b_s0, g.b_s1, g.b_div, rb, tid);
    } else {
      load_tail<AKC>(Ag, roffA, g.M, m0, k0, K, g.a_s0, g.a_s1, g.a_div, ra, tid);
      load_tail<BKC>(B, roffB, g.N, n0, k0, K, g.b_s0, g.b_s1, g.b_div, rb, tid);
    }
  };
  // optional by-product (m-contiguous A, first column tile only): sums over k of the A tile columns this thread stages
  const bool do_sum = !AKC && g.a_sum != nullptr && n0 == 0;
  float4 asum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add_a = [&]() {
    if constexpr (!AKC) {
      if (do_sum) {
#pragma unroll
        for (int r = 0; r < NLD; ++r) { asum.x += ra[r].x; asum.y += ra[r].y; asum.z += ra[r].z; asum.w += ra[r].w; }
      }
    }
  };
  fetch(0);
  add_a();
  store_tile<AKC>(As, ra, tid);
  store_tile<BKC>(Bs, rb, tid);
  __syncthreads();

  const int lk = lane >> 5, li = lane & 31;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) fetch(kt + 1);
    const float* as = As + cur * BKT * PA + wm * 64 + li + lk * PA;
    const float* bs = Bs + cur * BKT * PB + wn * 64 + li + lk * PB;
    // fragment reads run one k-step ahead of the MFMAs
    float a0 = as[0], a1 = as[32], b0 = bs[0], b1 = bs[32];
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 2) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (kk + 2 < BKT) {
        na0 = as[(kk + 2) * PA]; na1 = as[(kk + 2) * PA + 32];
        nb0 = bs[(kk + 2) * PB]; nb1 = bs[(kk + 2) * PB + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (more) {
      add_a();
      store_tile<AKC>(As + (cur ^ 1) * BKT * PA, ra, tid);
      store_tile<BKC>(Bs + (cur ^ 1) * BKT * PB, rb, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  if constexpr (!AKC) {
    if (do_sum) {   // (uniform per workgroup; the k loop ended with a barrier, so the tile buffers are free)
      float4* red = reinterpret_cast<float4*>(smem);
      red[tid] = asum;                              // [k row group tid >> 5][m4 = tid & 31]
      __syncthreads();
      if (tid < 32) {
        float4 t = red[tid];
#pragma unroll
        for (int gq = 1; gq < 8; ++gq) { const float4 u = red[gq * 32 + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        float* o = g.a_sum + ((long long)z * g.splitk + ks) * g.a_sum_ks;
        const int m = m0 + tid * 4;
        if (m < g.M) o[m] = t.x;
        if (m + 1 < g.M) o[m + 1] = t.y;
        if (m + 2 < g.M) o[m + 2] = t.z;
        if (m + 3 < g.M) o[m + 3] = t.w;
      }
    }
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
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
          if (g.epilogue == 2) { const float a = 1.0f - fabsf(g.ep_aux[(long long)z * g.c_sz + coff]); v *= a * a; }   // softsign backward
          float* p = C + coff;
          if (g.accumulate) v += *p;
          if (g.ks_cnt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // slab tile: written through (read by another XCD's workgroup)
          else *p = v;
        }
      }
    }
  }
  // In-kernel slab reduction (split-K): the last slice workgroup of this tile to get here sums the tile's slabs in slice
  // order -- the order the separate reduction pass uses, so the result is bit-identical and does not depend on arrival.
  if (g.ks_cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    const int gx = (g.N + BN - 1) / BN, tile_id = (m0 / BM) * gx + n0 / BN;
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(g.ks_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == (unsigned)g.splitk - 1u;
      if (last) __hip_atomic_store(g.ks_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = last;
    }
    __syncthreads();
    if (flag[0]) {
      for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
        const int r = m0 + idx / (BN / 4), cc = n0 + (idx % (BN / 4)) * 4;
        if (r >= g.M || cc >= g.N) continue;                 // (N % 4 == 0: a float4 is inside the row or beyond it)
        const long long e = (long long)r * g.N + cc;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.ks_acc) acc = *reinterpret_cast<const float4*>(g.ks_out + e);
        for (int sl = 0; sl < g.splitk; ++sl) {
          const float4 v = load_f4_sc1(g.C + (long long)sl * g.c_ks, (unsigned)(e * 4));
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(g.ks_out + e) = acc;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Skinny products (M <= 64 rows: one streamed frame of a few dozen utterances, evaluate_model_helpers.py:87-115 /
// language-model streaming): C[M][N] = A[M][K] . B[N][K]^T (+ bias) is bound by streaming B (the weights) once, and a
// 128-row tile wastes most of its MFMA rows and, with split-K slabs, three launches.  Here a workgroup owns 16 columns of
// C: its 8 waves split K, every lane loads 16 bytes of its weight row and of each 16-row block of A per 16-k chunk
// straight from global memory into MFMA operands (no LDS staging: A is L2-resident, B is read exactly once), and the
// eight partial tiles are summed through LDS.  Layer 0 of the shipped shape (32 x 7168 x 2304): ~150 us as split-K -> ~30.
template <int MT, bool BKC>   // 16-row blocks of A; B k-contiguous ([N][K]) or n-contiguous ([K][N]: the day weights)
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GemmArgs g) {
  __shared__ float red[8][MT][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int z = blockIdx.y, zb = g.b_zmap ? g.b_zmap[z] : z;             // batched form: the day layer's per-sentence products
  g.A += (long long)z * g.a_sz; g.B += (long long)zb * g.b_sz; g.C += (long long)z * g.c_sz;
  if (g.bias) g.bias += (long long)zb * g.bias_sz;
  const int ncol = n0 + j < g.N ? n0 + j : g.N - 1;                       // clamped: columns beyond N are never stored
  const float* brow = BKC ? g.B + rowoff(ncol, g.b_s0, g.b_s1, g.b_div) + 4 * q : g.B + ncol;
  const float* arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = 16 * mt + j;
    arow[mt] = g.A + rowoff(r < g.M ? r : g.M - 1, g.a_s0, g.a_s1, g.a_div) + 4 * q;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // K in chunks of 16, dealt to the waves round-robin in groups of 4 chunks (64 k: one 256-byte run of a weight row)
  const int nchunk = g.K / 16;
  for (int c0 = wave * 4; c0 < nchunk; c0 += 32) {
    float4 bw[4], av[4][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = (c0 + u < nchunk ? c0 + u : c0) * 16;                 // the tail re-reads a valid chunk and is masked below
      if constexpr (BKC) {
        bw[u] = *reinterpret_cast<const float4*>(brow + k);
      } else {   // element (n, k) at B + rowoff(k) + n: four row-strided loads, 16 lanes (n) contiguous each
        const int kq = k + 4 * q;
        bw[u] = make_float4(brow[rowoff(kq, g.b_s0, g.b_s1, g.b_div)], brow[rowoff(kq + 1, g.b_s0, g.b_s1, g.b_div)],
                            brow[rowoff(kq + 2, g.b_s0, g.b_s1, g.b_div)], brow[rowoff(kq + 3, g.b_s0, g.b_s1, g.b_div)]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[u][mt] = *reinterpret_cast<const float4*>(arow[mt] + k);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c0 + u < nchunk) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt].x, bw[u].x, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt].y, bw[u].y, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt].z, bw[u].z, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt].w, bw[u].w, acc[mt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][mt][e][lane] = acc[mt][e];
  __syncthreads();
  // D layout of v_mfma_f32_16x16x4_f32: lane (j, q), element e -> row 4 q + e, column j.  Thread t sums one (mt, e, lane).
  for (int idx = tid; idx < MT * 4 * 64; idx += 512) {
    const int l = idx & 63, e = (idx >> 6) & 3, mt = idx >> 8;
    const float v0 = (red[0][mt][e][l] + red[1][mt][e][l]) + (red[2][mt][e][l] + red[3][mt][e][l]);
    const float v1 = (red[4][mt][e][l] + red[5][mt][e][l]) + (red[6][mt][e][l] + red[7][mt][e][l]);
    const int row = 16 * mt + 4 * (l >> 4) + e, col = n0 + (l & 15);
    if (row < g.M && col < g.N) {
      float v = v0 + v1 + (g.bias ? g.bias[col] : 0.f);
      if (g.epilogue == 1) v = v / (1.0f + fabsf(v));
      float* p = g.C + rowoff(row, g.c_s0, g.c_s1, g.c_div) + col;
      if (g.accumulate) v += *p;
      *p = v;
    }
  }
}

}  // namespace b2t

extern "C" int b2t_gemm_f32(const b2t_gemm_desc* d, void* stream) {
  using namespace b2t;
  B2T_REQUIRE(d != nullptr, "b2t_gemm_f32: null descriptor");
  B2T_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->Z > 0, "b2t_gemm_f32: bad shape M=%d N=%d K=%d Z=%d",
              d->M, d->N, d->K, d->Z);
  B2T_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "b2t_gemm_f32: A/B must be 16-byte aligned");
  B2T_REQUIRE((d->a_s0 % 4) == 0 && (d->a_s1 % 4) == 0 && (d->a_sz % 4) == 0 && (d->b_s0 % 4) == 0 &&
                  (d->b_s1 % 4) == 0 && (d->b_sz % 4) == 0,
              "b2t_gemm_f32: A/B strides must be multiples of 4 elements");
  B2T_REQUIRE(d->a_sum == nullptr || !d->a_kcontig, "b2t_gemm_f32: a_sum goes with an m-contiguous A (a_kcontig = 0)");
  B2T_REQUIRE(d->ks_counters == nullptr || (d->splitk > 1 && d->Z == 1 && (d->N % 4) == 0 && d->ks_out != nullptr && ((uintptr_t)d->ks_out & 15) == 0 &&
                                            (long long)d->M * d->N * 4 < (1ll << 31)),
              "b2t_gemm_f32: the in-kernel slab reduction needs splitk > 1, Z = 1, N %% 4 == 0 and a 16-byte aligned ks_out");
  GemmArgs g;
  { int rc = fill_gemm_args(d, g, BKT, BM, "b2t_gemm_f32"); if (rc) return rc; }
  dim3 grid(((d->N + BN - 1) / BN) * ((d->M + BM - 1) / BM), 1, d->Z * g.splitk), block(256);
  { static const bool off = getenv("B2T_GEMM_KS_XCD") && atoi(getenv("B2T_GEMM_KS_XCD")) == 0; g.ks_xcd = !off && d->Z == 1 && g.splitk >= 8 && (g.splitk & 7) == 0; }
  hipStream_t s = as_stream(stream);
  if (d->M <= 64 && d->splitk >= 0 && g.splitk == 1 && d->a_kcontig && (d->K % 16) == 0 && d->a_brk == 0 && d->epilogue != 2) {
    // a few rows against a wide weight matrix: stream the weights once (gemm_skinny_kernel)
    const dim3 sg((d->N + 15) / 16, d->Z), sb(512);
#define B2T_SKINNY(MT)                                                                                  \
    do {                                                                                                \
      if (d->b_kcontig) hipLaunchKernelGGL((gemm_skinny_kernel<MT, true>), sg, sb, 0, s, g);              \
      else hipLaunchKernelGGL((gemm_skinny_kernel<MT, false>), sg, sb, 0, s, g);                          \
    } while (0)
    if (d->M <= 16) B2T_SKINNY(1);
    else if (d->M <= 32) B2T_SKINNY(2);
    else if (d->M <= 48) B2T_SKINNY(3);
    else B2T_SKINNY(4);
#undef B2T_SKINNY
    B2T_CHECK_LAUNCH("b2t_gemm_f32 (skinny)");
    return 0;
  }
  if (d->a_kcontig && d->b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, g);
  else if (d->a_kcontig && !d->b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, g);
  else if (!d->a_kcontig && d->b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, g);
  B2T_CHECK_LAUNCH("b2t_gemm_f32");
  return 0;
}
