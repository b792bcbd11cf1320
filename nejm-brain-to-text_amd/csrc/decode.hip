// decode.hip — greedy CTC decode + Levenshtein distance (validation PER path), LM-decoder prologue.
#include "common.h"

namespace b2t {

// One workgroup per sentence: per-frame argmax (first maximum wins, like torch.argmax), then
// unique_consecutive + blank removal (model_training/rnn_trainer.py:725-728) via a block scan.
__global__ __launch_bounds__(256) void greedy_kernel(const float* __restrict__ logits, const int32_t* __restrict__ lens,
                                                     int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                     int32_t* __restrict__ argmax_out, int T, int C) {
  extern __shared__ int sm[];  // am[T], flag-scan scratch [256]
  int* am = sm;
  int* wsum = sm + T;
  const int b = blockIdx.x, tid = threadIdx.x;
  int L = lens[b]; if (L > T) L = T; if (L < 0) L = 0;
  for (int t = tid; t < T; t += blockDim.x) {
    const float* r = logits + ((long long)b * T + t) * C;
    float best = r[0]; int bi = 0;
    for (int k = 1; k < C; ++k) { const float v = r[k]; if (v > best) { best = v; bi = k; } }
    am[t] = bi;
    if (argmax_out) argmax_out[(long long)b * T + t] = bi;
  }
  __syncthreads();
  // chunked exclusive scan of keep flags: each thread owns a contiguous run of frames
  const int per = (L + blockDim.x - 1) / blockDim.x;
  const int t0 = tid * per, t1 = min(L, t0 + per);
  int cnt = 0;
  for (int t = t0; t < t1; ++t) cnt += (am[t] != 0 && (t == 0 || am[t] != am[t - 1]));
  wsum[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < (int)blockDim.x; ++i) { const int c = wsum[i]; wsum[i] = acc; acc += c; }
    out_len[b] = acc;
  }
  __syncthreads();
  int pos = wsum[tid];
  for (int t = t0; t < t1; ++t)
    if (am[t] != 0 && (t == 0 || am[t] != am[t - 1])) out_ids[(long long)b * T + pos++] = am[t];
}

// Levenshtein distance, one wave per sentence pair.  Row recurrence
//   cur[j] = min(prev[j]+1, cur[j-1]+1, prev[j-1]+(a_i!=b_j))
// is evaluated as tmp[j] = min(prev[j]+1, prev[j-1]+cost) followed by a prefix-min of (tmp[k]-k) (+j),
// which removes the serial cur[j-1] dependency: wave-wide shuffle scan per 64-column strip.
__global__ __launch_bounds__(64) void edit_distance_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ a_len,
                                                           int La_max, const int32_t* __restrict__ bq,
                                                           const int32_t* __restrict__ b_len, int Lb_max,
                                                           int32_t* __restrict__ dist) {
  extern __shared__ int rows[];  // prev[Lb_max+1], cur[Lb_max+1]
  const int s = blockIdx.x, lane = threadIdx.x;
  int la = a_len[s]; if (la > La_max) la = La_max;
  int lb = b_len[s]; if (lb > Lb_max) lb = Lb_max;
  const int32_t* A = a + (long long)s * La_max;
  const int32_t* Bq = bq + (long long)s * Lb_max;
  int* prev = rows; int* cur = rows + (Lb_max + 1);
  for (int j = lane; j <= lb; j += 64) prev[j] = j;
  __syncthreads();
  for (int i = 1; i <= la; ++i) {
    const int ai = A[i - 1];
    int carry = i;  // cur[0] - 0
    if (lane == 0) cur[0] = i;
    for (int j0 = 1; j0 <= lb; j0 += 64) {
      const int j = j0 + lane;
      int v = 0x3fffffff;
      if (j <= lb) v = min(prev[j] + 1, prev[j - 1] + (ai != Bq[j - 1])) - j;
      // inclusive prefix-min over the strip
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (lane >= o) v = min(v, u);
      }
      v = min(v, carry);
      if (j <= lb) cur[j] = v + j;
      carry = __shfl(v, 63);
    }
    __syncthreads();
    int* tsw = prev; prev = cur; cur = tsw;
  }
  if (lane == 0) dist[s] = prev[lb];
}

__global__ void lm_prologue_kernel(const float* __restrict__ logits, const float* __restrict__ log_priors,
                                   float blank_penalty, float* __restrict__ logp, int rows, int C) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* x = logits + (long long)r * C;
  float m = x[0];
  for (int k = 1; k < C; ++k) m = fmaxf(m, x[k]);
  float s = 0.f;
  for (int k = 0; k < C; ++k) s += expf(x[k] - m);
  const float lse = m + logf(s);
  for (int k = 0; k < C; ++k) {
    float v = x[k] - lse - (log_priors ? log_priors[(long long)r * C + k] : 0.f);
    if (k == 0) v -= blank_penalty;
    logp[(long long)r * C + k] = v;
  }
}

}  // namespace b2t

using namespace b2t;

extern "C" int b2t_greedy_decode_f32(const float* logits, const int32_t* lens, int32_t* out_ids, int32_t* out_len,
                                     int32_t* argmax_out, int B, int T, int C, void* stream) {
  B2T_REQUIRE(B > 0 && T > 0 && C > 0, "greedy_decode: bad shape");
  size_t smem = sizeof(int) * ((size_t)T + 256);
  B2T_REQUIRE(smem <= 64 * 1024, "greedy_decode: T=%d too long for one workgroup's LDS", T);
  hipLaunchKernelGGL(greedy_kernel, dim3(B), dim3(256), smem, as_stream(stream), logits, lens, out_ids, out_len,
                     argmax_out, T, C);
  B2T_CHECK_LAUNCH("b2t_greedy_decode_f32");
  return 0;
}

extern "C" int b2t_edit_distance_i32(const int32_t* a, const int32_t* a_len, int La_max, const int32_t* b,
                                     const int32_t* b_len, int Lb_max, int32_t* dist, int B, void* stream) {
  B2T_REQUIRE(B > 0 && La_max >= 0 && Lb_max >= 0, "edit_distance: bad shape");
  const int la = La_max > 0 ? La_max : 1, lb = Lb_max > 0 ? Lb_max : 1;   // the kernel's row pitch: size the LDS from the clamped value
  size_t smem = sizeof(int) * 2 * ((size_t)lb + 1);
  B2T_REQUIRE(smem <= 64 * 1024, "edit_distance: Lb_max=%d too long", Lb_max);
  hipLaunchKernelGGL(edit_distance_kernel, dim3(B), dim3(64), smem, as_stream(stream), a, a_len, la, b, b_len, lb, dist);
  B2T_CHECK_LAUNCH("b2t_edit_distance_i32");
  return 0;
}

extern "C" int b2t_lm_prologue_f32(const float* logits, const float* log_priors, float blank_penalty, float* logp,
                                   int rows, int C, void* stream) {
  B2T_REQUIRE(rows > 0 && C > 0, "lm_prologue: bad shape");
  hipLaunchKernelGGL(lm_prologue_kernel, dim3((rows + 127) / 128), dim3(128), 0, as_stream(stream), logits, log_priors,
                     blank_penalty, logp, rows, C);
  B2T_CHECK_LAUNCH("b2t_lm_prologue_f32");
  return 0;
}
