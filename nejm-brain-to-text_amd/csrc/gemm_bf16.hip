// gemm_bf16.hip — the GEMM of gemm.hip with bf16 matrix-core operands (the reference's `use_amp` regime:
// model_training/rnn_args.yaml `use_amp: true`, autocast(dtype=torch.bfloat16) at rnn_trainer.py:535,698).
//
// Same descriptor, same fp32 tensors in memory: C[z][m][n] (+)= sum_k bf16(A(z,m,k)) * bf16(B(z,n,k)) (+ bias[n]) with
// fp32 accumulation and fp32 output.  Operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their
// way from global memory into LDS; nothing is stored in bf16 in HBM (the reference's autocast also rounds the OUTPUT of
// every matmul to bf16: this path keeps it in fp32, i.e. it is at least as accurate).
// 128x128x32 block tile, 256 threads = 2x2 waves x (2x2) 32x32 accumulators, v_mfma_f32_32x32x16_bf16 (a lane holds
// row lane%32, 8 consecutive k from 8*(lane/32)): tiles are staged [row][k] with an 80-byte row pitch so that a
// fragment is ONE 16-byte LDS read and 8 consecutive lanes touch all 32 banks.  k-contiguous operands convert a float4
// to one 8-byte LDS store; m-contiguous operands load a 4 x 4 block, transpose it in registers and store four 8-byte
// pieces into interleaved rows (lds_row).  Register prefetch of
// the next k-tile, double-buffered LDS, one barrier per k-tile, XCD-aware tile order: as in gemm.hip.
// The matrix time per k is 1/16 of the fp32 kernel's; with fp32 operands in memory the kernel is bound by operand
// traffic (32 KB per 128x128x32 step), not by MFMA.
#include "common.h"
#include "gemm_args.h"

namespace b2t {

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

constexpr int HBM_ = 128, HBN = 128, HBK = 32;
constexpr int HP = 40;            // LDS row pitch in bf16 elements (80 B)
constexpr int HNL = 4;            // float4 loads per thread per operand tile (128 x 32 floats / 256 threads)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

// LDS row of tile row r.  m-contiguous operands are stored by threads that own 4 CONSECUTIVE rows each (a float4 along
// m), i.e. for a given element of the float4 the 32 lanes of a half-wave hit rows 4 apart: with rows in natural order
// that is 320 bytes = 16 banks apart -- two banks for 32 lanes.  Interleaving the rows (r -> (r % 4) * 32 + r / 4)
// makes those 32 rows consecutive (80 bytes apart: 8 distinct bank groups); fragment reads of 32 consecutive logical
// rows then fall into 4 runs of 8 consecutive LDS rows, still conflict-free.
template <bool KC> __device__ __forceinline__ int lds_row(int r) { return KC ? r : ((r & 3) * 32 + (r >> 2)); }

// One operand tile slice per thread.
// KC (k contiguous): thread (row = tid/8 + 32 r, k4 = (tid%8)*4), r = 0..3.
// MC (m contiguous): thread (m4 = (tid%32)*4, k = 4*(tid/32) + r), r = 0..3: a 4 (k) x 4 (m) block, transposed in registers
//                    into four 8-byte LDS stores (4 consecutive k of one m each).
template <bool KC>
__device__ __forceinline__ void hload(const float* __restrict__ P, const long long (&roff)[HNL], int mcl, int ext_m, int base_m,
                                      int k0, int Kend, bool full, long long s0, long long s1, int div, float4 (&v)[HNL], int tid) {
#pragma unroll
  for (int r = 0; r < HNL; ++r) {
    if constexpr (KC) {
      const int k = k0 + (tid & 7) * 4;
      if (full) {
        v[r] = *reinterpret_cast<const float4*>(P + roff[r] + k);
      } else {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const int row = base_m + (tid >> 3) + 32 * r;
        if (row < ext_m) {
          const float* p = P + roff[r] + k;
          if (k < Kend) o.x = p[0];
          if (k + 1 < Kend) o.y = p[1];
          if (k + 2 < Kend) o.z = p[2];
          if (k + 3 < Kend) o.w = p[3];
        }
        v[r] = o;
      }
    } else {
      const int k = k0 + 4 * (tid >> 5) + r;
      if (full) {
        v[r] = *reinterpret_cast<const float4*>(P + rowoff(k, s0, s1, div) + mcl);
      } else {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const int m = base_m + (tid & 31) * 4;
        if (k < Kend) {
          const float* p = P + rowoff(k, s0, s1, div) + m;
          if (m < ext_m) o.x = p[0];
          if (m + 1 < ext_m) o.y = p[1];
          if (m + 2 < ext_m) o.z = p[2];
          if (m + 3 < ext_m) o.w = p[3];
        }
        v[r] = o;
      }
    }
  }
}

template <bool KC>
__device__ __forceinline__ void hstore(__bf16* __restrict__ S, const float4 (&v)[HNL], int tid) {
  if constexpr (KC) {
#pragma unroll
    for (int r = 0; r < HNL; ++r) {
      const int row = (tid >> 3) + 32 * r, k = (tid & 7) * 4;
      uint2 w;
      w.x = pack_bf16(v[r].x, v[r].y); w.y = pack_bf16(v[r].z, v[r].w);
      *reinterpret_cast<uint2*>(&S[row * HP + k]) = w;
    }
  } else {
    const int k = 4 * (tid >> 5), l32 = tid & 31;   // rows m = 4 l32 + i live in LDS rows i * 32 + l32
    uint2 w;
    w.x = pack_bf16(v[0].x, v[1].x); w.y = pack_bf16(v[2].x, v[3].x);
    *reinterpret_cast<uint2*>(&S[(0 * 32 + l32) * HP + k]) = w;
    w.x = pack_bf16(v[0].y, v[1].y); w.y = pack_bf16(v[2].y, v[3].y);
    *reinterpret_cast<uint2*>(&S[(1 * 32 + l32) * HP + k]) = w;
    w.x = pack_bf16(v[0].z, v[1].z); w.y = pack_bf16(v[2].z, v[3].z);
    *reinterpret_cast<uint2*>(&S[(2 * 32 + l32) * HP + k]) = w;
    w.x = pack_bf16(v[0].w, v[1].w); w.y = pack_bf16(v[2].w, v[3].w);
    *reinterpret_cast<uint2*>(&S[(3 * 32 + l32) * HP + k]) = w;
  }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 3) void gemm_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 2 * HBM_ * HP];
  __bf16* As = smem;
  __bf16* Bs = smem + 2 * HBM_ * HP;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z / g.splitk;
  const int ks = blockIdx.z - z * g.splitk;
  int m0, n0;
  {   // XCD-aware tile order (see gemm.hip)
    const int gx = (g.N + HBN - 1) / HBN, nwg = gridDim.x;
    const int b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    m0 = (tile / gx) * HBM_; n0 = (tile % gx) * HBN;
  }
  const float* A = g.A + (long long)z * g.a_sz;
  const int zb = g.b_zmap ? g.b_zmap[z] : z;
  const float* B = g.B + (long long)zb * g.b_sz;
  const float* bias = (g.bias && ks == 0) ? g.bias + (long long)zb * g.bias_sz : nullptr;
  float* C = g.C + (long long)z * g.c_sz + (long long)ks * g.c_ks;

  long long roffA[HNL], roffB[HNL];
#pragma unroll
  for (int r = 0; r < HNL; ++r) {
    const int ra_ = m0 + (tid >> 3) + 32 * r, rb_ = n0 + (tid >> 3) + 32 * r;
    roffA[r] = AKC ? rowoff(ra_ < g.M ? ra_ : g.M - 1, g.a_s0, g.a_s1, g.a_div) : 0;
    roffB[r] = BKC ? rowoff(rb_ < g.N ? rb_ : g.N - 1, g.b_s0, g.b_s1, g.b_div) : 0;
  }
  const int mclA = min(m0 + (tid & 31) * 4, ((g.M + 3) & ~3) - 4);
  const int mclB = min(n0 + (tid & 31) * 4, ((g.N + 3) & ~3) - 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kb = ks * g.kchunk;
  const int K = min(g.K, kb + g.kchunk);
  const int nk = (K - kb + HBK - 1) / HBK;
  const int nfull = (K - kb) / HBK;
  // a full tile may use the clamped fast path only if the m / n extent needs no zero fill either: rows beyond the
  // extent are clamped (their products land in accumulator rows that are never stored), so only k matters
  float4 ra[HNL], rb[HNL];
  auto fetch = [&](int kt) {
    const int k0 = kb + kt * HBK;
    const float* Ag = A + ((g.a_brk > 0 && (AKC ? k0 : m0) >= g.a_brk) ? g.a_gap : 0);
    const bool full = kt < nfull;
    hload<AKC>(Ag, roffA, mclA, g.M, m0, k0, K, full, g.a_s0, g.a_s1, g.a_div, ra, tid);
    hload<BKC>(B, roffB, mclB, g.N, n0, k0, K, full, g.b_s0, g.b_s1, g.b_div, rb, tid);
  };
  fetch(0);
  hstore<AKC>(As, ra, tid);
  hstore<BKC>(Bs, rb, tid);
  __syncthreads();

  const int lk = lane >> 5, li = lane & 31;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) fetch(kt + 1);
    const __bf16* as0 = As + cur * HBM_ * HP + lds_row<AKC>(wm * 64 + li) * HP + 8 * lk;
    const __bf16* as1 = As + cur * HBM_ * HP + lds_row<AKC>(wm * 64 + 32 + li) * HP + 8 * lk;
    const __bf16* bs0 = Bs + cur * HBM_ * HP + lds_row<BKC>(wn * 64 + li) * HP + 8 * lk;
    const __bf16* bs1 = Bs + cur * HBM_ * HP + lds_row<BKC>(wn * 64 + 32 + li) * HP + 8 * lk;
#pragma unroll
    for (int kk = 0; kk < HBK; kk += 16) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(as0 + kk), a1 = *reinterpret_cast<const bf16x8*>(as1 + kk);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bs0 + kk), b1 = *reinterpret_cast<const bf16x8*>(bs1 + kk);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      hstore<AKC>(As + (cur ^ 1) * HBM_ * HP, ra, tid);
      hstore<BKC>(Bs + (cur ^ 1) * HBM_ * HP, rb, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout of the 32x32 MFMAs: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
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
          *p = v;
        }
      }
    }
  }
}

}  // namespace b2t

extern "C" int b2t_gemm_bf16_f32(const b2t_gemm_desc* d, void* stream) {
  using namespace b2t;
  B2T_REQUIRE(d != nullptr, "b2t_gemm_bf16_f32: null descriptor");
  B2T_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->Z > 0, "b2t_gemm_bf16_f32: bad shape M=%d N=%d K=%d Z=%d",
              d->M, d->N, d->K, d->Z);
  B2T_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "b2t_gemm_bf16_f32: A/B must be 16-byte aligned");
  B2T_REQUIRE((d->a_s0 % 4) == 0 && (d->a_s1 % 4) == 0 && (d->a_sz % 4) == 0 && (d->b_s0 % 4) == 0 &&
                  (d->b_s1 % 4) == 0 && (d->b_sz % 4) == 0,
              "b2t_gemm_bf16_f32: A/B strides must be multiples of 4 elements");
  GemmArgs g;
  B2T_REQUIRE(d->a_sum == nullptr, "b2t_gemm_bf16_f32: a_sum is a by-product of the fp32 tile kernel only");
  { int rc = fill_gemm_args(d, g, HBK, HBM_, "b2t_gemm_bf16_f32"); if (rc) return rc; }
  dim3 grid(((d->N + HBN - 1) / HBN) * ((d->M + HBM_ - 1) / HBM_), 1, d->Z * g.splitk), block(256);
  hipStream_t s = as_stream(stream);
  if (d->a_kcontig && d->b_kcontig) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, s, g);
  else if (d->a_kcontig && !d->b_kcontig) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, s, g);
  else if (!d->a_kcontig && d->b_kcontig) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, s, g);
  B2T_CHECK_LAUNCH("b2t_gemm_bf16_f32");
  return 0;
}
