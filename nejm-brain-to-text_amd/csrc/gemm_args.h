// gemm_args.h — kernel-side argument block shared by the GEMM kernels (gemm.hip: exact fp32; gemm_bf16.hip: bf16 operands).
#pragma once
#include "common.h"

namespace b2t {

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;
  long long a_s0, a_s1; int a_div; long long a_sz;
  long long b_s0, b_s1; int b_div; long long b_sz;
  long long c_s0, c_s1; int c_div; long long c_sz;
  const int* b_zmap; long long bias_sz;
  int epilogue; int accumulate;
  int splitk; int kchunk; long long c_ks;
  int a_brk; int a_gap;   // contiguous index i of A (k if a_kcontig, else m): i >= a_brk reads from i + a_gap
  const float* ep_aux;    // epilogue 2: u, laid out like C
  float* a_sum; long long a_sum_ks;   // fp32 tile kernel, m-contiguous A: per-slice sums over k of A[k][m]
  unsigned* ks_cnt; float* ks_out; int ks_acc;   // fp32 tile kernel, split-K: in-kernel slab reduction by a tile's last slice workgroup
  int ks_xcd;             // split-K: 1 = a K slice's tiles all run on one XCD (slices dealt to the XCDs), 0 = tiles dealt to the XCDs
  int tile_gm;            // packed bf16 kernel: tiles are numbered down groups of tile_gm tile rows (1 / 0 = row-major); see gemm_bf16p.hip
};

__device__ __forceinline__ long long rowoff(int i, long long s0, long long s1, int div) {
  return div > 0 ? (long long)(i / div) * s1 + (long long)(i % div) * s0 : (long long)i * s0;
}

// Descriptor -> kernel arguments.  bk / bm: the kernel's tile extents along k and m (split-K chunks are multiples of bk,
// a gap in A's contiguous index must fall on a tile boundary).
inline int fill_gemm_args(const b2t_gemm_desc* d, GemmArgs& g, int bk, int bm, const char* name) {
  g.A = d->A; g.B = d->B; g.C = d->C; g.bias = d->bias;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.a_s0 = d->a_s0; g.a_s1 = d->a_s1; g.a_div = d->a_div; g.a_sz = d->a_sz;
  g.b_s0 = d->b_s0; g.b_s1 = d->b_s1; g.b_div = d->b_div; g.b_sz = d->b_sz;
  g.c_s0 = d->c_s0; g.c_s1 = d->c_s1; g.c_div = d->c_div; g.c_sz = d->c_sz;
  g.b_zmap = d->b_zmap; g.bias_sz = d->bias_sz; g.epilogue = d->epilogue; g.accumulate = d->accumulate;
  g.splitk = d->splitk > 1 ? d->splitk : 1;
  B2T_REQUIRE(g.splitk == 1 || (d->epilogue == 0 && d->accumulate == 0),
              "%s: split-K slabs cannot carry an epilogue/accumulate (reduce them with b2t_colsum_f32)", name);
  g.kchunk = ((d->K + g.splitk - 1) / g.splitk + bk - 1) / bk * bk;
  g.c_ks = d->c_ks;
  g.a_brk = d->a_brk; g.a_gap = d->a_gap;
  g.ep_aux = d->ep_aux;
  g.a_sum = d->a_sum; g.a_sum_ks = d->a_sum_ks;
  g.ks_xcd = 0; g.tile_gm = 1;
  g.ks_cnt = d->ks_counters; g.ks_out = d->ks_out; g.ks_acc = d->ks_accumulate;
  B2T_REQUIRE(d->epilogue != 2 || d->ep_aux != nullptr, "%s: epilogue 2 needs ep_aux", name);
  B2T_REQUIRE(d->a_brk == 0 || (d->a_brk > 0 && d->a_gap % 4 == 0 &&
                                (d->a_kcontig ? d->a_brk % bk == 0 : (d->a_brk % bm == 0 && d->M % bm == 0))),
              "%s: a_brk must be a multiple of the tile extent (%d along k, %d along m with M %% %d == 0), a_gap of 4", name, bk, bm, bm);
  return 0;
}

// gemm_bf16p.hip, internal (see there): operands of the two-pass bf16 GEMM packed ahead of it
struct PackDrop { float p; unsigned long long seed; long long elem0; float* dup; };
size_t gemm_bf16p_operand_bytes(int rows, int K);
int gemm_bf16p_pack(const b2t_gemm_desc* d, int which, void* out, hipStream_t s, const PackDrop* drop = nullptr);
int gemm_bf16p_run(const b2t_gemm_desc* d, const void* Ap_pre, const void* Bp_pre, void* ws, size_t ws_bytes, hipStream_t s, const PackDrop* dropA = nullptr);

}  // namespace b2t
