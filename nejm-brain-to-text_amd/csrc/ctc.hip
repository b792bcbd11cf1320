// ctc.hip — fused log-softmax + CTC loss (alpha recursion) + gradient wrt logits (beta recursion).
//
// One workgroup per sentence; thread s owns extended-label state s (e = [0,l1,0,l2,...,lS,0]); the workgroup's last
// wave owns no state and forms the gradient rows.  The sentence's [T][C] log-probabilities are staged in LDS (when they
// fit), alpha_{t-1}/beta_{t+1} live in LDS (double buffered, ONE LDS-only barrier per time step); alpha_t is also
// written to the HBM scratch and read back by the beta pass two steps ahead of its use.  The gradient needs, per
// frame and class, the sum of the posteriors exp(alpha+beta-log p-log P) of that class's states: the state threads
// drop them into class-grouped slots, the gradient wave adds each class in a fixed order (deterministic) while the
// recursion already runs the next frame.  (Round-1 history at B=64, T=500, S=60: 1.25 ms with a global gather, a
// full __syncthreads and log-space class sums inside every step; 0.6 ms now; 0.41 ms is the recursion alone.)
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

// Barrier for LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL operations (vmcnt(0)) so
// that they are visible to the workgroup; inside the recursions that put the alpha store / gradient store / alpha
// prefetch round trip (a microsecond) on every one of the 2T dependent steps.  Nothing global is communicated between
// threads inside the loops (alpha rows are re-read by the thread that wrote them, after a full barrier).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(1024) void ctc_kernel(const float* __restrict__ logits, const int32_t* __restrict__ targets,
                                                   const int32_t* __restrict__ in_len,
                                                   const int32_t* __restrict__ tgt_len, float* __restrict__ loss,
                                                   float* __restrict__ alpha_ws, float* __restrict__ dlogits, int T,
                                                   int C, int S_max, int ldd, float grad_scale, int lp_in_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int LxMax = 2 * S_max + 1;
  float* lse = reinterpret_cast<float*>(smem_raw);  // [T]
  float* abuf = lse + T;                             // [2][LxMax]
  float* ab = abuf + 2 * LxMax;                      // [2][LxMax] posteriors, grouped by class, double buffered
  float* lcab = ab + 2 * LxMax;                      // [C]
  float* nllp = lcab + C;                            // [1]
  float* bpart = nllp + 1;                           // [64] partial sums of the blank posteriors
  int* ext = reinterpret_cast<int*>(bpart + 64);     // [LxMax]
  int* order = ext + LxMax;                          // [LxMax] positions grouped by class
  int* cstart = order + LxMax;                       // [C+1]
  // [T][C] log-probabilities of this sentence (when they fit): the two recursions are 2T dependent steps, and a global
  // gather per step (first touch of that logits row: ~2000 cycles) was 80 % of the kernel's time
  float* lpl = reinterpret_cast<float*>(cstart + C + 1 + ((C + 1) & 1));

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
  if (lp_in_lds)
    for (int i = tid; i < Tb * C; i += nthr) lpl[i] = lg[i] - lse[i / C];
  // log p_t(k): from LDS when staged, else from the logits in HBM
#define LP(t, k) (lp_in_lds ? lpl[(t) * C + (k)] : (lg[(long long)(t) * C + (k)] - lse[(t)]))
  if (tid < C) {  // positions per class; blank (class 0) owns the even positions
    int cnt = 0;
    if (tid > 0) {
      for (int s = 1; s < Lx; s += 2) cnt += (ext[s] == tid);
    } else {
      cnt = (Lx + 1) / 2;
    }
    lcab[tid] = __int_as_float(cnt);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < C; ++k) { cstart[k] = acc; acc += __float_as_int(lcab[k]); }
    cstart[C] = acc;
  }
  __syncthreads();
  // order[s] = slot of state s in the class-grouped posterior array (class k occupies [cstart[k], cstart[k+1]), its
  // states in increasing s), so that the per-class sums read contiguous ranges in a fixed order
  if (tid < C) {
    int p = cstart[tid];
    if (tid > 0) {
      for (int s = 1; s < Lx; s += 2)
        if (ext[s] == tid) order[s] = p++;
    } else {
      for (int s = 0; s < Lx; s += 2) order[s] = p++;
    }
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
    if (s == 0) a0 = LP(0, 0);
    else if (s == 1) a0 = LP(0, es);
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
      const float v = lse3(la1, la2, la3) + LP(t, es);
      an[s] = v;
      aw[(long long)t * LxMax + s] = v;
    }
    lds_barrier();
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
  // The state threads run the beta recursion and leave the posterior of (t, s),
  //     exp(alpha_t(s) + beta_t(s) - log p_t(e_s) - log P(l|x))  <= 1,
  // in the slot of its class-grouped buffer (plain probabilities: no max / log-sum-exp needed downstream; infeasible
  // sentences give -inf + inf = NaN as in the log-space form).  The LAST wave of the workgroup owns no states: it
  // turns step t's posteriors into the gradient row while the state threads are already on step t-1 (one barrier per
  // step; the posterior buffer is double buffered).  All sums run in a fixed order: deterministic.
  const int nstate = nthr - 64;
  const bool gradwave = tid >= nstate;
  const int gt = tid - nstate;
  const int myslot = live ? order[s] : 0;
  // list bounds of the gradient wave's lanes (fixed for the sentence)
  const int gk = (gradwave && gt < C) ? gt : 0;
  const int bl0 = cstart[0] + (gt & 7), bl1 = cstart[1];           // blank slots of lane gt < 8: every 8th
  const int cl0 = cstart[gk], cl1 = (gk >= 1) ? cstart[gk + 1] : cstart[gk];   // class gt >= 1 (empty for lane 0)
  cur = 0;
  // alpha_t comes back from HBM: fetched two steps ahead of its use (an L2/HBM round trip is several steps long)
  float aw0 = live ? aw[(long long)(Tb - 1) * LxMax + s] : 0.f;
  float aw1 = (live && Tb >= 2) ? aw[(long long)(Tb - 2) * LxMax + s] : 0.f;
  for (int t = Tb - 1; t >= 0; --t) {
    const float* bp = abuf + cur * LxMax;
    float* bn = abuf + (cur ^ 1) * LxMax;
    float* pb = ab + (t & 1) * LxMax;
    if (live) {
      const float aw2 = t >= 2 ? aw[(long long)(t - 2) * LxMax + s] : 0.f;
      const float lpt = LP(t, es);
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
      pb[myslot] = expf(v + aw0 + nll - lpt);
      aw0 = aw1; aw1 = aw2;
    }
    lds_barrier();
    cur ^= 1;
    if (gradwave) {
      // blank (the longest list, Sb + 1 states): 8 lanes take every 8th slot, lane 0 adds the 8 partial sums;
      // the other classes: lane k walks its (short) list.
      // The reads of a list are issued 8 at a time (clamped index, masked value) and then added in list order: a
      // loop of dependent "read, wait, add" steps costs an LDS round trip per element.
      auto ordered_sum = [&](int p0, int p1, int stride) {
        float acc = 0.f;
        for (int base = p0; base < p1; base += 8 * stride) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int p = base + i * stride;
            v[i] = pb[p < p1 ? p : p0];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) acc += (base + i * stride < p1) ? v[i] : 0.f;
        }
        return acc;
      };
      const float lpk = (gt < C) ? LP(t, gt) : 0.f;                      // issued early, used after the sums
      float sum = 0.f;
      if (gt < 8) bpart[gt] = ordered_sum(bl0, bl1, 8);                // blank: 8 lanes take every 8th slot ...
      if (gt >= 1 && gt < C) sum = ordered_sum(cl0, cl1, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // one wave, lock-step: bpart is complete
      if (gt == 0) sum = ((bpart[0] + bpart[1]) + (bpart[2] + bpart[3])) + ((bpart[4] + bpart[5]) + (bpart[6] + bpart[7]));
      for (int k = gt; k < ldd; k += 64) {
        float g = 0.f;
        if (k < C) g = grad_scale * (expf(lpk) - sum);                   // (k < C <= 64: k == gt)
        dl[(long long)t * ldd + k] = g;
      }
    }
  }
  // rows beyond the input length: exactly zero
  for (long long i = (long long)Tb * ldd + tid; i < (long long)T * ldd; i += nthr) dl[i] = 0.f;
#undef LP
}

}  // namespace b2t

using namespace b2t;

extern "C" int b2t_ctc_loss_f32(const float* logits, const int32_t* targets, const int32_t* in_len,
                                const int32_t* tgt_len, float* loss, float* alpha_ws, float* dlogits, int B, int T,
                                int C, int S_max, int ldd, float grad_scale, void* stream) {
  B2T_REQUIRE(B > 0 && T > 0 && C > 1 && S_max >= 0, "ctc_loss: bad shape B=%d T=%d C=%d S_max=%d", B, T, C, S_max);
  const int LxMax = 2 * S_max + 1;
  B2T_REQUIRE(LxMax <= 960, "ctc_loss: 2*S_max+1 = %d exceeds 960 extended states", LxMax);
  B2T_REQUIRE(C <= 64 && (!dlogits || (ldd >= C && ldd <= 128)), "ctc_loss: C=%d (<=64) / ldd=%d (C..128) unsupported", C, ldd);
  B2T_REQUIRE(alpha_ws != nullptr, "ctc_loss: alpha_ws required");
  int threads = ((LxMax + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  threads += 64;   // the last wave owns no states: it forms the gradient rows (see the beta pass)
  size_t smem = sizeof(float) * ((size_t)T + 4 * (size_t)LxMax + C + 1 + 64) + sizeof(int) * (2 * (size_t)LxMax + C + 2) + 16;
  B2T_REQUIRE(smem <= 160 * 1024 - 64, "ctc_loss: T=%d needs %zu bytes of LDS (>160 KiB)", T, smem);
  const size_t lp_bytes = sizeof(float) * (size_t)T * C;
  const int lp_in_lds = smem + lp_bytes <= 150 * 1024;   // T=500, C=41: 82 KB
  if (lp_in_lds) smem += lp_bytes;
  hipStream_t s = as_stream(stream);
  if (smem > 64 * 1024) {
    int rc = check_hip(hipFuncSetAttribute((const void*)ctc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                       "ctc_loss: raise dynamic LDS limit");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(threads), smem, s, logits, targets, in_len, tgt_len, loss, alpha_ws,
                     dlogits, T, C, S_max, ldd, grad_scale, lp_in_lds);
  B2T_CHECK_LAUNCH("b2t_ctc_loss_f32");
  return 0;
}
