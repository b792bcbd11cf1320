// ctc.hip — fused log-softmax + CTC loss (alpha recursion) + gradient wrt logits (beta recursion).
//
// One workgroup per sentence; thread s owns extended-label state s (e = [0,l1,0,l2,...,lS,0]).
// alpha_{t-1}/beta_{t+1} live in LDS (double buffered, one barrier per time step); alpha_t is also
// written to the HBM scratch so the beta pass can form alpha_t + beta_t.  The per-class log-sum
// LSE_{s:e_s=k}(alpha+beta) is evaluated deterministically: blank by wave 0 (one state per lane +
// shuffle tree), every other class by one thread of wave 1 walking that class's position list.
// Formulas follow torch.nn.CTCLoss(blank=0, reduction='none', zero_infinity=False) as called at
// model_training/rnn_trainer.py:242,538-545.
#include "common.h"

namespace b2t {

__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) m = 0.f;
  return logf(expf(a - m) + expf(b - m) + expf(c - m)) + m;
}
__device__ __forceinline__ float lse2(float a, float b) {
  float m = fmaxf(a, b);
  if (m == -INFINITY) m = 0.f;
  return logf(expf(a - m) + expf(b - m)) + m;
}

__global__ __launch_bounds__(1024) void ctc_kernel(const float* __restrict__ logits, const int32_t* __restrict__ targets,
                                                   const int32_t* __restrict__ in_len,
                                                   const int32_t* __restrict__ tgt_len, float* __restrict__ loss,
                                                   float* __restrict__ alpha_ws, float* __restrict__ dlogits, int T,
                                                   int C, int S_max, int ldd, float grad_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int LxMax = 2 * S_max + 1;
  float* lse = reinterpret_cast<float*>(smem_raw);  // [T]
  float* abuf = lse + T;                             // [2][LxMax]
  float* ab = abuf + 2 * LxMax;                      // [LxMax]
  float* lcab = ab + LxMax;                          // [C]
  float* nllp = lcab + C;                            // [1]
  int* ext = reinterpret_cast<int*>(nllp + 1);       // [LxMax]
  int* order = ext + LxMax;                          // [LxMax] positions grouped by class
  int* cstart = order + LxMax;                       // [C+1]

  int Tb = in_len[b];
  int Sb = tgt_len[b];
  if (Tb > T) Tb = T;
  if (Sb > S_max) Sb = S_max;
  const int Lx = 2 * Sb + 1;
  const float* lg = logits + (long long)b * T * C;
  const int32_t* tg = targets + (long long)b * S_max;
  float* aw = alpha_ws + (long long)b * T * LxMax;

  // ---- prologue: log-sum-exp per frame, extended labels, class position lists -------------------
  for (int t = tid; t < Tb; t += nthr) {
    const float* r = lg + (long long)t * C;
    float m = r[0];
    for (int k = 1; k < C; ++k) m = fmaxf(m, r[k]);
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += expf(r[k] - m);
    lse[t] = m + logf(s);
  }
  for (int s = tid; s < Lx; s += nthr) ext[s] = (s & 1) ? tg[s >> 1] : 0;
  __syncthreads();
  if (tid < C) {  // count non-blank positions of class tid (blank handled separately)
    int cnt = 0;
    if (tid > 0)
      for (int s = 1; s < Lx; s += 2) cnt += (ext[s] == tid);
    lcab[tid] = __int_as_float(cnt);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < C; ++k) { cstart[k] = acc; acc += __float_as_int(lcab[k]); }
    cstart[C] = acc;
  }
  __syncthreads();
  if (tid > 0 && tid < C) {
    int p = cstart[tid];
    for (int s = 1; s < Lx; s += 2)
      if (ext[s] == tid) order[p++] = s;
  }
  __syncthreads();

  const int s = tid;
  const bool live = s < Lx;
  const int es = live ? ext[s] : 0;
  const bool skip_in = live && s >= 2 && es != 0 && es != ext[s - 2];       // alpha: from s-2
  const bool skip_out = live && s + 2 < Lx && ext[s + 2] != 0 && ext[s + 2] != es;  // beta: to s+2

  if (Tb <= 0) {  // no frames: infeasible (zero_infinity=False -> inf)
    if (tid == 0) loss[b] = INFINITY;
    if (dlogits)
      for (long long i = tid; i < (long long)T * ldd; i += nthr) dlogits[(long long)b * T * ldd + i] = 0.f;
    return;
  }

  // ---- alpha pass ----------------------------------------------------------------------------------
  int cur = 0;
  if (live) {
    float a0 = -INFINITY;
    if (s == 0) a0 = lg[0] - lse[0];
    else if (s == 1) a0 = lg[es] - lse[0];
    abuf[s] = a0;
    aw[s] = a0;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* ap = abuf + cur * LxMax;
    float* an = abuf + (cur ^ 1) * LxMax;
    if (live) {
      const float la1 = ap[s];
      const float la2 = s >= 1 ? ap[s - 1] : -INFINITY;
      const float la3 = skip_in ? ap[s - 2] : -INFINITY;
      const float v = lse3(la1, la2, la3) + (lg[(long long)t * C + es] - lse[t]);
      an[s] = v;
      aw[(long long)t * LxMax + s] = v;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (tid == 0) {
    const float* ap = abuf + cur * LxMax;
    const float ll = Lx > 1 ? lse2(ap[Lx - 1], ap[Lx - 2]) : ap[0];
    nllp[0] = -ll;
    loss[b] = -ll;
  }
  __syncthreads();
  if (!dlogits) return;
  const float nll = nllp[0];
  float* dl = dlogits + (long long)b * T * ldd;

  // ---- beta pass + gradient -------------------------------------------------------------------------
  cur = 0;
  for (int t = Tb - 1; t >= 0; --t) {
    const float* bp = abuf + cur * LxMax;
    float* bn = abuf + (cur ^ 1) * LxMax;
    const float lpt = live ? (lg[(long long)t * C + es] - lse[t]) : 0.f;
    if (live) {
      float v;
      if (t == Tb - 1) {
        v = (s == Lx - 1 || s == Lx - 2) ? lpt : -INFINITY;
      } else {
        const float lb1 = bp[s];
        const float lb2 = s + 1 < Lx ? bp[s + 1] : -INFINITY;
        const float lb3 = skip_out ? bp[s + 2] : -INFINITY;
        v = lse3(lb1, lb2, lb3) + lpt;
      }
      bn[s] = v;
      ab[s] = v + aw[(long long)t * LxMax + s];
    }
    __syncthreads();
    cur ^= 1;
    // per-class log-sum of alpha+beta
    const int wave = tid >> 6, lane = tid & 63;
    if (wave == 0) {  // blank: even positions
      float m = -INFINITY;
      for (int p = 2 * lane; p < Lx; p += 128) m = fmaxf(m, ab[p]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      const float ms = (m == -INFINITY) ? 0.f : m;
      float sum = 0.f;
      for (int p = 2 * lane; p < Lx; p += 128) sum += expf(ab[p] - ms);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (lane == 0) lcab[0] = logf(sum) + ms;
    } else if (wave == 1 && lane + 1 < C) {
      const int k = lane + 1;
      const int p0 = cstart[k], p1 = cstart[k + 1];
      float m = -INFINITY;
      for (int p = p0; p < p1; ++p) m = fmaxf(m, ab[order[p]]);
      const float ms = (m == -INFINITY) ? 0.f : m;
      float sum = 0.f;
      for (int p = p0; p < p1; ++p) sum += expf(ab[order[p]] - ms);
      lcab[k] = (p1 > p0) ? logf(sum) + ms : -INFINITY;
    }
    __syncthreads();
    if (tid < ldd) {
      float g = 0.f;
      if (tid < C) {
        const float lp = lg[(long long)t * C + tid] - lse[t];
        g = grad_scale * (expf(lp) - expf(lcab[tid] + nll - lp));
      }
      dl[(long long)t * ldd + tid] = g;
    }
    // the next iteration's first barrier orders lcab/ab reuse
  }
  // rows beyond the input length: exactly zero
  for (long long i = (long long)Tb * ldd + tid; i < (long long)T * ldd; i += nthr) dl[i] = 0.f;
}

}  // namespace b2t

using namespace b2t;

extern "C" int b2t_ctc_loss_f32(const float* logits, const int32_t* targets, const int32_t* in_len,
                                const int32_t* tgt_len, float* loss, float* alpha_ws, float* dlogits, int B, int T,
                                int C, int S_max, int ldd, float grad_scale, void* stream) {
  B2T_REQUIRE(B > 0 && T > 0 && C > 1 && S_max >= 0, "ctc_loss: bad shape B=%d T=%d C=%d S_max=%d", B, T, C, S_max);
  const int LxMax = 2 * S_max + 1;
  B2T_REQUIRE(LxMax <= 1024, "ctc_loss: 2*S_max+1 = %d exceeds 1024 extended states", LxMax);
  B2T_REQUIRE(C <= 64 && (!dlogits || (ldd >= C && ldd <= 128)), "ctc_loss: C=%d (<=64) / ldd=%d (C..128) unsupported", C, ldd);
  B2T_REQUIRE(alpha_ws != nullptr, "ctc_loss: alpha_ws required");
  int threads = ((LxMax + 63) / 64) * 64;
  if (threads < 128) threads = 128;
  size_t smem = sizeof(float) * ((size_t)T + 3 * (size_t)LxMax + C + 1) + sizeof(int) * (2 * (size_t)LxMax + C + 1) + 16;
  B2T_REQUIRE(smem <= 160 * 1024 - 64, "ctc_loss: T=%d needs %zu bytes of LDS (>160 KiB)", T, smem);
  hipStream_t s = as_stream(stream);
  if (smem > 64 * 1024) {
    int rc = check_hip(hipFuncSetAttribute((const void*)ctc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                       "ctc_loss: raise dynamic LDS limit");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(threads), smem, s, logits, targets, in_len, tgt_len, loss, alpha_ws,
                     dlogits, T, C, S_max, ldd, grad_scale);
  B2T_CHECK_LAUNCH("b2t_ctc_loss_f32");
  return 0;
}
