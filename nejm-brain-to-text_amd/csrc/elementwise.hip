// elementwise.hip — HBM-bound passes of the hot path: fused augmentation + Gaussian smoothing,
// softsign backward, column sums, patch fold, dropout, transpose, per-day gradient reduction.
// All kernels read/write 16 B per lane along the contiguous (feature) axis.
#include <algorithm>
#include "common.h"

namespace b2t {

struct Taps { float v[33]; };

__device__ __forceinline__ float4 f4_fma(float s, float4 a, float4 acc) {
  acc.x = fmaf(s, a.x, acc.x); acc.y = fmaf(s, a.y, acc.y);
  acc.z = fmaf(s, a.z, acc.z); acc.w = fmaf(s, a.w, acc.w);
  return acc;
}

// Noisy input sample at position `tin` of the cut sequence (zero outside [0,Tc)).
__device__ __forceinline__ float4 noisy_fetch(const float* __restrict__ x, int b, int tin, int f, int T, int F,
                                              int Tc, int cut, float ws, float os, uint64_t seed,
                                              const float* __restrict__ wn, float4 offv) {
  if (tin < 0 || tin >= Tc) return make_float4(0.f, 0.f, 0.f, 0.f);
  const long long e = ((long long)b * T + (tin + cut)) * F + f;
  float4 v = *reinterpret_cast<const float4*>(x + e);
  if (ws > 0.f) {
    float4 n = wn ? *reinterpret_cast<const float4*>(wn + e) : Philox::normal4(seed, (uint64_t)(e >> 2), 0u);
    v.x = fmaf(ws, n.x, v.x); v.y = fmaf(ws, n.y, v.y); v.z = fmaf(ws, n.z, v.z); v.w = fmaf(ws, n.w, v.w);
  }
  v.x += offv.x; v.y += offv.y; v.z += offv.z; v.w += offv.w;
  return v;
}

constexpr int TCH = 32;  // outputs per thread along T

// Sliding-window form: each noisy input is generated once per chunk (+ (NT-1)/TCH halo).
template <int NT>
__global__ __launch_bounds__(128) void augment_smooth_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int T, int F, int Tc, int T_out, int cut, int left,
                                                             float ws, float os, uint64_t seed,
                                                             const float* __restrict__ wn,
                                                             const float* __restrict__ on, Taps taps) {
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TCH;
  for (int f = threadIdx.x * 4; f < F; f += blockDim.x * 4) {
    float4 offv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (os > 0.f) {
      float4 n = on ? *reinterpret_cast<const float4*>(on + (long long)b * F + f)
                    : Philox::normal4(seed, (uint64_t)(((long long)b * F + f) >> 2), 1u);
      offv = make_float4(os * n.x, os * n.y, os * n.z, os * n.w);
    }
    float4 win[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) win[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < TCH + NT - 1; ++s) {
      const int tin = t0 - left + s;
      float4 v = noisy_fetch(x, b, tin, f, T, F, Tc, cut, ws, os, seed, wn, offv);
#pragma unroll
      for (int j = 0; j < NT - 1; ++j) win[j] = win[j + 1];
      win[NT - 1] = v;
      const int t = t0 + s - (NT - 1);
      if (s >= NT - 1 && t < T_out) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc = f4_fma(taps.v[j], win[j], acc);
        *reinterpret_cast<float4*>(y + ((long long)b * T_out + t) * F + f) = acc;
      }
    }
  }
}

// Generic tap count: direct form (regenerates the counter-based noise per tap; rarely used).
__global__ __launch_bounds__(128) void augment_smooth_generic_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                     int T, int F, int Tc, int T_out, int cut, int left,
                                                                     float ws, float os, uint64_t seed,
                                                                     const float* __restrict__ wn,
                                                                     const float* __restrict__ on, Taps taps, int nt) {
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TCH;
  for (int f = threadIdx.x * 4; f < F; f += blockDim.x * 4) {
    float4 offv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (os > 0.f) {
      float4 n = on ? *reinterpret_cast<const float4*>(on + (long long)b * F + f)
                    : Philox::normal4(seed, (uint64_t)(((long long)b * F + f) >> 2), 1u);
      offv = make_float4(os * n.x, os * n.y, os * n.z, os * n.w);
    }
    for (int t = t0; t < t0 + TCH && t < T_out; ++t) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < nt; ++j)
        acc = f4_fma(taps.v[j], noisy_fetch(x, b, t + j - left, f, T, F, Tc, cut, ws, os, seed, wn, offv), acc);
      *reinterpret_cast<float4*>(y + ((long long)b * T_out + t) * F + f) = acc;
    }
  }
}

__global__ void softsign_bwd_kernel(const float* __restrict__ u, float* __restrict__ du, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(u)[i];
    float4 d = reinterpret_cast<float4*>(du)[i];
    float s;
    s = 1.f - fabsf(a.x); d.x *= s * s;
    s = 1.f - fabsf(a.y); d.y *= s * s;
    s = 1.f - fabsf(a.z); d.z *= s * s;
    s = 1.f - fabsf(a.w); d.w *= s * s;
    reinterpret_cast<float4*>(du)[i] = d;
  }
}

// Column sums, two deterministic stages: block (cx, ry) sums rows [ry*RPB, ...) of 64 columns (256 with 16-byte loads).
// Rows per block: 512 for long matrices, 64 for short ones (a function of `rows` alone: it sizes the workspace) -- the day
// layer's bias gradient is 64 sentences x [500 rows x 512]: with 512-row blocks that was 512 workgroups of 125 dependent 4-byte
// loads per thread, 98 us for 65 MB on the step's tail; with 64-row blocks, float4 columns and all of a thread's 16 loads in flight
// it is a bandwidth-bound pass.
constexpr int CS_RPB = 512, CS_RPB_SHORT = 64;
__host__ __device__ inline int cs_rpb(long long rows) { return rows <= 4096 ? CS_RPB_SHORT : CS_RPB; }
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long long rows, int cols,
                                                             long long ld, float* __restrict__ part, long long x_sz, int rpb) {
  __shared__ float red[4][64];
  x += (long long)blockIdx.z * x_sz;
  part += (long long)blockIdx.z * gridDim.y * cols;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rpb;
  float s = 0.f;
  if (c < cols)
    for (long long r = r0 + w; r < r0 + rpb && r < rows; r += 4) s += x[r * ld + c];
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < cols)
    part[(long long)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// the same sums (same rows per wave, same order) with four columns per lane: cols % 4 == 0, ld % 4 == 0, 16-byte aligned slabs
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ x, long long rows, int cols,
                                                              long long ld, float* __restrict__ part, long long x_sz) {
  __shared__ float4 red[4][64];
  x += (long long)blockIdx.z * x_sz;
  part += (long long)blockIdx.z * gridDim.y * cols;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const long long r0 = (long long)blockIdx.y * CS_RPB_SHORT;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols) {
    float4 v[CS_RPB_SHORT / 4];
#pragma unroll
    for (int k = 0; k < CS_RPB_SHORT / 4; ++k) {
      const long long r = r0 + w + 4 * k;
      v[k] = r < rows ? *reinterpret_cast<const float4*>(x + r * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < CS_RPB_SHORT / 4; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < cols) {
    const float4 a = red[0][lane], b = red[1][lane], d = red[2][lane], e = red[3][lane];
    *reinterpret_cast<float4*>(part + (long long)blockIdx.y * cols + c) =
        make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int nparts, int cols, float* __restrict__ out,
                                    int accumulate, long long out_sz) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  part += (long long)blockIdx.y * nparts * cols;
  out += (long long)blockIdx.y * out_sz;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(long long)p * cols + c];
  out[c] = accumulate ? out[c] + s : s;
}

using f32x4 = float __attribute__((ext_vector_type(4)));
// out[i] (+)= sum_s slab[s][i]  — deterministic split-K reduction, float4 per lane, slabs summed in order
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int nslab, long long n4, float* __restrict__ out,
                                   int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 acc = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // eight slab loads in flight, added in slab order (the order is part of the contract); the slabs are read once: non-temporal
    int s = 0;
    for (; s + 8 <= nslab; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slab + (long long)(s + k) * n4 * 4) + i);
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc.x += v[k][0]; acc.y += v[k][1]; acc.z += v[k][2]; acc.w += v[k][3]; }
    }
    for (; s < nslab; ++s) {
      const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slab + (long long)s * n4 * 4) + i);
      acc.x += v[0]; acc.y += v[1]; acc.z += v[2]; acc.w += v[3];
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}

__global__ void patch_fold_kernel(const float* __restrict__ dv, float* __restrict__ du, int T, int F, int Tp,
                                  int patch, int stride) {
  // du[b,t,f] = sum_{p,k: p*stride+k=t} dv[b,p,k*F+f]
  const int b = blockIdx.z, t = blockIdx.y;
  const int f = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (f >= F) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int p_hi = t / stride; if (p_hi > Tp - 1) p_hi = Tp - 1;
  for (int p = p_hi; p >= 0; --p) {
    const int k = t - p * stride;
    if (k >= patch) break;
    float4 v = *reinterpret_cast<const float4*>(dv + (((long long)b * Tp + p) * patch + k) * F + f);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(du + ((long long)b * T + t) * F + f) = acc;
}

// patch_fold + the backward of the input dropout + the Softsign backward in one pass (the three ran back to back on the step's
// tail: 3 x 65 MB read and written): du = fold(dv) -> * dropout mask (the forward's Philox stream: element index of [B][T][F]) ->
// * (1 - |u|)^2, each element through the same operations in the same order as the three kernels.
__global__ void patch_fold_day_bwd_kernel(const float* __restrict__ dv, const float* __restrict__ u, float* __restrict__ du, int T, int F,
                                          int Tp, int patch, int stride, float p, float scale, uint64_t seed) {
  const int b = blockIdx.z, t = blockIdx.y;
  const int f = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (f >= F) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int p_hi = t / stride; if (p_hi > Tp - 1) p_hi = Tp - 1;
  for (int q = p_hi; q >= 0; --q) {
    const int k = t - q * stride;
    if (k >= patch) break;
    float4 v = *reinterpret_cast<const float4*>(dv + (((long long)b * Tp + q) * patch + k) * F + f);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const long long e = ((long long)b * T + t) * F + f;
  if (p > 0.f) {
    const float4 r = Philox::uniform4(seed, (uint64_t)(e / 4), 2u);
    acc.x = r.x >= p ? acc.x * scale : 0.f; acc.y = r.y >= p ? acc.y * scale : 0.f;
    acc.z = r.z >= p ? acc.z * scale : 0.f; acc.w = r.w >= p ? acc.w * scale : 0.f;
  }
  const float4 a = *reinterpret_cast<const float4*>(u + e);
  float s;
  s = 1.f - fabsf(a.x); acc.x *= s * s;
  s = 1.f - fabsf(a.y); acc.y *= s * s;
  s = 1.f - fabsf(a.z); acc.z *= s * s;
  s = 1.f - fabsf(a.w); acc.w *= s * s;
  *reinterpret_cast<float4*>(du + e) = acc;
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n4, float p, float scale,
                               uint64_t seed, long long idx0) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 u = Philox::uniform4(seed, (uint64_t)(idx0 + i), 2u);
    v.x = u.x >= p ? v.x * scale : 0.f; v.y = u.y >= p ? v.y * scale : 0.f;
    v.z = u.z >= p ? v.z * scale : 0.f; v.w = u.w >= p ? v.w * scale : 0.f;
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

__global__ void dropout_mask_kernel(float* __restrict__ y, long long n4, float p, float scale, uint64_t seed, long long idx0) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 u = Philox::uniform4(seed, (uint64_t)(idx0 + i), 2u);
    reinterpret_cast<float4*>(y)[i] = make_float4(u.x >= p ? scale : 0.f, u.y >= p ? scale : 0.f, u.z >= p ? scale : 0.f,
                                                  u.w >= p ? scale : 0.f);
  }
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (r0 + i < rows && c < cols) tile[i][threadIdx.x] = in[(long long)(r0 + i) * cols + c];
  __syncthreads();
  int r = r0 + threadIdx.x, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (c0 + i < cols && r < rows) out[(long long)(c0 + i) * rows + r] = tile[threadIdx.x][i];
}

// out[slot_of_first_sample_with_that_day] : out[d][i] = sum_{b: day[b]==d} slab[b][i], for days present.
// One block column per i-chunk; blockIdx.y = b; only the FIRST sample of each day does the sum (deterministic order).
__global__ void day_reduce_kernel(const float* __restrict__ slab, const int* __restrict__ day, int B, long long n,
                                  float* __restrict__ out, long long out_stride) {
  const int b = blockIdx.y;
  const int d = day[b];
  for (int j = 0; j < b; ++j) if (day[j] == d) return;  // not the first of its day
  // the day's samples, listed once per block: the sum below then issues four loads at a time instead of one dependent load per
  // sample behind a branch (same order of additions: bit-identical; 36 -> ~12 us on the step's tail at 16 samples per day)
  __shared__ int list[1024];
  __shared__ int cnt_s;
  if (threadIdx.x == 0) {
    int c = 0;
    for (int j = b; j < B && c < 1024; ++j) if (day[j] == d) list[c++] = j;
    cnt_s = c;
  }
  __syncthreads();
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const int cnt = cnt_s;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 4 <= cnt; k += 4) {
    const float4 v0 = *reinterpret_cast<const float4*>(slab + (long long)list[k] * n + i);
    const float4 v1 = *reinterpret_cast<const float4*>(slab + (long long)list[k + 1] * n + i);
    const float4 v2 = *reinterpret_cast<const float4*>(slab + (long long)list[k + 2] * n + i);
    const float4 v3 = *reinterpret_cast<const float4*>(slab + (long long)list[k + 3] * n + i);
    acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
    acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
    acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
    acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
  }
  for (; k < cnt; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(slab + (long long)list[k] * n + i);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (cnt == 1024)                                          // (more than 1024 samples of one day in a batch: the rest, one by one)
    for (int j = list[1023] + 1; j < B; ++j) {
      if (day[j] != d) continue;
      const float4 v = *reinterpret_cast<const float4*>(slab + (long long)j * n + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  *reinterpret_cast<float4*>(out + (long long)d * out_stride + i) = acc;
}

}  // namespace b2t

using namespace b2t;

extern "C" int b2t_augment_smooth_f32(const float* x, float* y, int B, int T, int F, int cut, float white_std,
                                      float offset_std, uint64_t seed, const float* white_noise,
                                      const float* offset_noise, const float* taps_host, int ntaps, int padding_mode,
                                      void* stream) {
  B2T_REQUIRE(B > 0 && T > 0 && F > 0 && (F % 4) == 0, "augment_smooth: bad shape B=%d T=%d F=%d (F%%4 must be 0)", B, T, F);
  B2T_REQUIRE(ntaps >= 1 && ntaps <= 33 && taps_host, "augment_smooth: ntaps=%d out of [1,33]", ntaps);
  B2T_REQUIRE(cut >= 0 && cut < T, "augment_smooth: cut=%d out of range", cut);
  B2T_REQUIRE(padding_mode == 0 || padding_mode == 1, "augment_smooth: padding_mode must be 0 (same) or 1 (valid)");
  const int Tc = T - cut;
  const int T_out = padding_mode == 0 ? Tc : Tc - ntaps + 1;
  B2T_REQUIRE(T_out > 0, "augment_smooth: sequence (T=%d, cut=%d) shorter than kernel (%d taps)", T, cut, ntaps);
  const int left = padding_mode == 0 ? (ntaps - 1) / 2 : 0;
  Taps tp; memset(&tp, 0, sizeof(tp));
  for (int i = 0; i < ntaps; ++i) tp.v[i] = taps_host[i];
  dim3 grid((T_out + TCH - 1) / TCH, B), block(128);
  hipStream_t s = as_stream(stream);
  if (ntaps == 9)
    hipLaunchKernelGGL((augment_smooth_kernel<9>), grid, block, 0, s, x, y, T, F, Tc, T_out, cut, left, white_std,
                       offset_std, seed, white_noise, offset_noise, tp);
  else
    hipLaunchKernelGGL(augment_smooth_generic_kernel, grid, block, 0, s, x, y, T, F, Tc, T_out, cut, left, white_std,
                       offset_std, seed, white_noise, offset_noise, tp, ntaps);
  B2T_CHECK_LAUNCH("b2t_augment_smooth_f32");
  return 0;
}

extern "C" int b2t_softsign_bwd_f32(const float* u, float* du, long long n, void* stream) {
  B2T_REQUIRE(n > 0 && (n % 4) == 0, "softsign_bwd: n=%lld must be a positive multiple of 4", n);
  long long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(softsign_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), u, du, n4);
  B2T_CHECK_LAUNCH("b2t_softsign_bwd_f32");
  return 0;
}

// rnn_trainer.py:532 / :705: adjusted_lens = ((n_time_steps - patch_size) / patch_stride + 1).to(torch.int32) -- torch divides in
// fp32 and truncates; six small torch kernels on the path between the head GEMM and the CTC, one launch here
__global__ void adjusted_lens_kernel(const void* n_time, int is64, int B, int patch, int stride, int32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const long long n = is64 ? static_cast<const long long*>(n_time)[i] : (long long)static_cast<const int32_t*>(n_time)[i];
  out[i] = patch > 0 ? (int32_t)((float)(n - patch) / (float)stride + 1.0f) : (int32_t)n;
}
extern "C" int b2t_adjusted_lens_i32(const void* n_time_steps, int is_int64, int B, int patch_size, int patch_stride, int32_t* out,
                                     void* stream) {
  B2T_REQUIRE(n_time_steps && out && B > 0, "adjusted_lens: null argument / empty batch");
  B2T_REQUIRE(patch_size == 0 || patch_stride > 0, "adjusted_lens: patch_size %d needs patch_stride > 0", patch_size);
  hipLaunchKernelGGL(adjusted_lens_kernel, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), n_time_steps, is_int64, B,
                     patch_size, patch_stride, out);
  B2T_CHECK_LAUNCH("b2t_adjusted_lens_i32");
  return 0;
}

extern "C" size_t b2t_colsum_ws_bytes(long long rows, int cols) {
  const int rpb = b2t::cs_rpb(rows);
  long long nparts = (rows + rpb - 1) / rpb;
  return (size_t)(nparts * cols * sizeof(float));
}

extern "C" int b2t_colsum_f32(const float* x, long long rows, int cols, long long ld, float* out, int accumulate,
                              float* ws, int Z, long long x_sz, long long out_sz, void* stream) {
  B2T_REQUIRE(rows > 0 && cols > 0 && ws && Z > 0, "colsum: bad args rows=%lld cols=%d Z=%d", rows, cols, Z);
  const int rpb = cs_rpb(rows);
  int nparts = (int)((rows + rpb - 1) / rpb);
  hipStream_t s = as_stream(stream);
  const bool vec4 = rpb == CS_RPB_SHORT && (cols % 4) == 0 && (ld % 4) == 0 && (x_sz % 4) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 15) == 0;
  if (vec4) hipLaunchKernelGGL(colsum_partial4_kernel, dim3((cols + 255) / 256, nparts, Z), dim3(256), 0, s, x, rows, cols, ld, ws, x_sz);
  else hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 63) / 64, nparts, Z), dim3(256), 0, s, x, rows, cols, ld, ws, x_sz, rpb);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 255) / 256, Z), dim3(256), 0, s, ws, nparts, cols, out, accumulate,
                     out_sz);
  B2T_CHECK_LAUNCH("b2t_colsum_f32");
  return 0;
}

extern "C" int b2t_slab_reduce_f32(const float* slab, int nslab, long long n, float* out, int accumulate, void* stream) {
  B2T_REQUIRE(slab && out && nslab > 0 && n > 0 && (n % 4) == 0 && ((uintptr_t)out & 15) == 0,
              "slab_reduce: n=%lld must be a positive multiple of 4 and out 16-byte aligned", n);
  const long long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), slab, nslab, n4, out, accumulate);
  B2T_CHECK_LAUNCH("b2t_slab_reduce_f32");
  return 0;
}

extern "C" int b2t_patch_fold_f32(const float* dv, float* du, int B, int T, int F, int Tp, int patch, int stride,
                                  void* stream) {
  B2T_REQUIRE(B > 0 && T > 0 && F > 0 && (F % 4) == 0 && patch > 0 && stride > 0, "patch_fold: bad args");
  dim3 block(128), grid((F / 4 + 127) / 128, T, B);
  hipLaunchKernelGGL(patch_fold_kernel, grid, block, 0, as_stream(stream), dv, du, T, F, Tp, patch, stride);
  B2T_CHECK_LAUNCH("b2t_patch_fold_f32");
  return 0;
}

extern "C" int b2t_patch_fold_day_bwd_f32(const float* dv, const float* u, float* du, int B, int T, int F, int Tp, int patch, int stride,
                                         float drop_p, uint64_t seed, void* stream) {
  B2T_REQUIRE(B > 0 && T > 0 && F > 0 && (F % 4) == 0 && patch > 0 && stride > 0 && dv && u && du && drop_p >= 0.f && drop_p < 1.f,
              "patch_fold_day_bwd: bad args");
  dim3 block(128), grid((F / 4 + 127) / 128, T, B);
  hipLaunchKernelGGL(patch_fold_day_bwd_kernel, grid, block, 0, as_stream(stream), dv, u, du, T, F, Tp, patch, stride, drop_p,
                     1.0f / (1.0f - drop_p), seed);
  B2T_CHECK_LAUNCH("b2t_patch_fold_day_bwd_f32");
  return 0;
}

extern "C" int b2t_dropout_f32(const float* x, float* y, long long n, float p, uint64_t seed, long long elem0,
                               void* stream) {
  B2T_REQUIRE(n > 0 && (n % 4) == 0 && (elem0 % 4) == 0 && p >= 0.f && p < 1.f, "dropout: bad args n=%lld p=%f", n, (double)p);
  long long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, n4, p, 1.0f / (1.0f - p), seed, elem0 / 4);
  B2T_CHECK_LAUNCH("b2t_dropout_f32");
  return 0;
}

// out[b][t][0:W] = t < lens[b] ? flat[row_off[b] + t][0:W] : 0   (4-byte elements; W % 4 == 0 takes 16-byte accesses)
__global__ void batch_gather_kernel(const uint32_t* __restrict__ flat, const long long* __restrict__ row_off,
                                    const int32_t* __restrict__ lens, uint32_t* __restrict__ out, int T_out, int W) {
  const int b = blockIdx.y;
  const long long src = row_off[b] * (long long)W;
  int n = lens[b];
  if (n > T_out) n = T_out;
  const long long live = (long long)n * W, total = (long long)T_out * W;
  uint32_t* o = out + (long long)b * total;
  if ((W & 3) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(flat + src);
    uint4* o4 = reinterpret_cast<uint4*>(o);
    const long long live4 = live >> 2, total4 = total >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x)
      o4[i] = i < live4 ? s4[i] : make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
      o[i] = i < live ? flat[src + i] : 0u;
  }
}

extern "C" int b2t_batch_gather_b32(const void* flat, const int64_t* row_off, const int32_t* lens, void* out, int B, int T_out,
                                    int W, void* stream) {
  B2T_REQUIRE(flat && row_off && lens && out && B > 0 && T_out > 0 && W > 0, "batch_gather: bad args B=%d T=%d W=%d", B, T_out, W);
  B2T_REQUIRE((W & 3) != 0 || ((((uintptr_t)flat) | ((uintptr_t)out)) & 15) == 0, "batch_gather: 16-byte alignment needed when W %% 4 == 0");
  const long long work = ((long long)T_out * W + 3) / 4;
  int bx = (int)((work + 255) / 256); if (bx > 64) bx = 64; if (bx < 1) bx = 1;
  hipLaunchKernelGGL(batch_gather_kernel, dim3(bx, B), dim3(256), 0, as_stream(stream), static_cast<const uint32_t*>(flat),
                     reinterpret_cast<const long long*>(row_off), lens, static_cast<uint32_t*>(out), T_out, W);
  B2T_CHECK_LAUNCH("b2t_batch_gather_b32");
  return 0;
}

extern "C" int b2t_dropout_mask_f32(float* y, long long n, float p, uint64_t seed, long long elem0, void* stream) {
  B2T_REQUIRE(y && n > 0 && (n % 4) == 0 && (elem0 % 4) == 0 && p >= 0.f && p < 1.f, "dropout_mask: bad args n=%lld p=%f", n, (double)p);
  long long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), y, n4, p, 1.0f / (1.0f - p), seed, elem0 / 4);
  B2T_CHECK_LAUNCH("b2t_dropout_mask_f32");
  return 0;
}

// y[o][i][k] += sum_{j <= i} w[o][j][k]: the reference's random-walk augmentation, features += cumsum(noise, dim=axis)
// (rnn_trainer.py:464-465).  One thread per (o, k) line, summed in index order like torch.cumsum on the CPU.
__global__ void cumsum_add_kernel(const float* __restrict__ w, float* __restrict__ y, long long outer, int n, long long inner) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= outer * inner) return;
  const long long o = id / inner, k = id % inner;
  const long long base = o * n * inner + k;
  float acc = 0.f;
  for (int i = 0; i < n; ++i) {
    acc += w[base + (long long)i * inner];
    y[base + (long long)i * inner] += acc;
  }
}
extern "C" int b2t_cumsum_add_f32(const float* w, float* y, long long outer, int n, long long inner, void* stream) {
  B2T_REQUIRE(w && y && outer > 0 && n > 0 && inner > 0, "cumsum_add: bad args");
  const long long tot = outer * inner;
  hipLaunchKernelGGL(cumsum_add_kernel, dim3((unsigned)((tot + 127) / 128)), dim3(128), 0, as_stream(stream), w, y, outer, n, inner);
  B2T_CHECK_LAUNCH("b2t_cumsum_add_f32");
  return 0;
}

__global__ void broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long long)rows * n) dst[i] = src[i % n];
}
extern "C" int b2t_broadcast_rows_f32(const float* src, float* dst, int rows, int n, void* stream) {
  B2T_REQUIRE(src && dst && rows > 0 && n > 0, "broadcast_rows: bad args");
  const long long tot = (long long)rows * n;
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), src, dst, rows, n);
  B2T_CHECK_LAUNCH("b2t_broadcast_rows_f32");
  return 0;
}

extern "C" int b2t_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
  B2T_REQUIRE(rows > 0 && cols > 0, "transpose: bad shape");
  dim3 block(32, 8), grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, as_stream(stream), in, out, rows, cols);
  B2T_CHECK_LAUNCH("b2t_transpose_f32");
  return 0;
}

extern "C" int b2t_day_reduce_f32(const float* slab, const int32_t* day_idx, int B, long long n, float* out,
                                  long long out_stride, void* stream) {
  B2T_REQUIRE(B > 0 && n > 0 && (n % 4) == 0, "day_reduce: bad args");
  dim3 block(256), grid((unsigned)((n / 4 + 255) / 256), B);
  hipLaunchKernelGGL(day_reduce_kernel, grid, block, 0, as_stream(stream), slab, day_idx, B, n, out, out_stride);
  B2T_CHECK_LAUNCH("b2t_day_reduce_f32");
  return 0;
}

// ---- up to four device-to-device copies in ONE launch (the static input / output buffers of a replayed streaming graph,
//      rnn_model._graph_forward: a copy per tensor costs a launch each, ~10 us of host time apiece) ----------------------------
namespace b2t {
struct CopySegs { const uint32_t* src[4]; uint32_t* dst[4]; long long words[4]; int n; };
__global__ void copy_segments_kernel(CopySegs c) {
  const long long stride = (long long)gridDim.x * blockDim.x, i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < c.n; ++k)
    for (long long i = i0; i < c.words[k]; i += stride) c.dst[k][i] = c.src[k][i];
}
}  // namespace b2t

extern "C" int b2t_copy_segments_b32(const void* const* src, void* const* dst, const long long* words, int n, void* stream) {
  using namespace b2t;
  B2T_REQUIRE(src && dst && words && n >= 1 && n <= 4, "copy_segments: 1..4 segments");
  CopySegs c;
  long long most = 0;
  for (int k = 0; k < 4; ++k) {
    c.src[k] = k < n ? static_cast<const uint32_t*>(src[k]) : nullptr;
    c.dst[k] = k < n ? static_cast<uint32_t*>(dst[k]) : nullptr;
    c.words[k] = k < n ? words[k] : 0;
    B2T_REQUIRE(k >= n || (c.src[k] && c.dst[k] && c.words[k] >= 0), "copy_segments: null segment %d", k);
    most = std::max(most, c.words[k]);
  }
  c.n = n;
  const int blocks = (int)std::min<long long>(1024, std::max<long long>(1, (most + 255) / 256));
  hipLaunchKernelGGL(copy_segments_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), c);
  B2T_CHECK_LAUNCH("b2t_copy_segments_b32");
  return 0;
}

// The same copies with the segment list read AT EXECUTION TIME from a table in pinned host memory: table = {n, src[4], dst[4],
// words[4]} (13 x int64).  Captured into a streaming call's hipGraph in front of and behind the model pass, it lets one replay
// take that call's input tensors and fill that call's fresh output tensors -- the host only rewrites the table.
namespace b2t {
__global__ void copy_indirect_kernel(const long long* __restrict__ table) {
  const int n = (int)table[0];
  const long long stride = (long long)gridDim.x * blockDim.x, i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < n && k < 4; ++k) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(table[1 + k]);
    uint32_t* dst = reinterpret_cast<uint32_t*>(table[5 + k]);
    const long long words = table[9 + k];
    for (long long i = i0; i < words; i += stride) dst[i] = src[i];
  }
}
}  // namespace b2t

extern "C" int b2t_copy_indirect_b32(const long long* table, int blocks, void* stream) {
  using namespace b2t;
  B2T_REQUIRE(table && blocks >= 1 && blocks <= 4096, "copy_indirect: bad arguments");
  hipLaunchKernelGGL(copy_indirect_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), table);
  B2T_CHECK_LAUNCH("b2t_copy_indirect_b32");
  return 0;
}
