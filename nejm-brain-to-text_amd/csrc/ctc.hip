// ctc.hip — fused log-softmax + CTC loss + gradient wrt logits for gfx950.
//
// Formulas follow torch.nn.CTCLoss(blank=0, reduction='none', zero_infinity=False) as called at
// model_training/rnn_trainer.py:242,538-545.  Extended label sequence e = [0,l1,0,l2,...,lS,0].
//
// Two launches:
//   ctc_recursion_kernel  grid (B, 2).  Workgroup (b, 0) runs the alpha recursion of sentence b and writes the loss,
//                         workgroup (b, 1) the beta recursion -- concurrently: neither needs the other.  Thread s owns
//                         state s; the sentence's [T][C] log-probabilities are staged in LDS (when they fit),
//                         alpha_{t-1} / beta_{t+1} are double buffered in LDS with ONE LDS-only barrier per step, every
//                         row also goes to HBM (alpha_ws / beta_ws).
//   ctc_grad_kernel       all frames in parallel (one wave per frame): the posterior of (t, s),
//                         exp(alpha_t(s) + beta_t(s) - log p_t(e_s) - log P(l|x)) <= 1, is summed per class in a fixed
//                         order (deterministic) and turned into the softmax-backward row; exactly 0 beyond input_lengths.
// Round-1 history at B=64, T=500, S=60 (the chip is otherwise idle during this step): 1.25 ms as one kernel with a
// global gather, a full __syncthreads and log-space class sums inside each of the 2T dependent steps; 0.63 ms with the
// log-probabilities in LDS, LDS-only barriers and a dedicated gradient wave; this version splits the two recursions
// over two workgroups and takes the gradient off the serial path altogether.
#include <stdlib.h>
#include "common.h"

namespace b2t {

// log(e^a + e^b + e^c) on the chain of T dependent steps: raw v_exp_f32 / v_log_f32 (base 2, ~1 ulp).  The sum lies in [1, 3]
// and every exponent argument is <= 0, so none of what __expf / __logf add around the instructions (denormal rescaling by
// ldexp, a four-operation exact multiply by ln 2) is needed: 18 VALU operations per state instead of 33 -- the recursions are
// bound by exactly this instruction stream (one wave, K states per lane, no memory on the chain).
__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) m = 0.f;
  const float L2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const float sum = __builtin_amdgcn_exp2f((a - m) * L2E) + __builtin_amdgcn_exp2f((b - m) * L2E) + __builtin_amdgcn_exp2f((c - m) * L2E);
  return __builtin_amdgcn_logf(sum) * LN2 + m;
}
__device__ __forceinline__ float lse2(float a, float b) {
  float m = fmaxf(a, b);
  if (m == -INFINITY) m = 0.f;
  return __logf(__expf(a - m) + __expf(b - m)) + m;
}

// Barrier for LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL operations (vmcnt(0)); inside
// the recursions that would put the alpha / beta row store round trip on every one of the T dependent steps.  Nothing
// global is communicated between threads inside the loops.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the recursions inside ONE wave (2 S + 1 <= 64 K states, K = 2 / 4 / 8 consecutive states per lane) ---------------------
// The T dependent steps were 0.38 us each with a thread per state: three LDS reads of the previous row, the log-sum-exp, an
// LDS write and a workgroup barrier between the two waves.  With lane l holding states K l .. K l + K - 1 in registers the
// previous row never leaves the register file: a state's left (alpha) / right (beta) neighbours are the lane's own values or
// the neighbouring lane's edge values, fetched with two whole-wave DPP shifts (v_mov_b32 wave_shr:1 / wave_shl:1; the lane
// without a neighbour keeps -inf) -- no LDS traffic on the chain, no barrier, no wait for the row stores.  The emission
// log-probabilities of step t + 1 are read from LDS while step t computes.
__device__ __forceinline__ float from_lane_below(float v) {   // lane l <- lane l - 1; lane 0 <- -inf
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)0xff800000u, (int)__float_as_uint(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float v) {   // lane l <- lane l + 1; lane 63 <- -inf
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)0xff800000u, (int)__float_as_uint(v), 0x130, 0xf, 0xf, false));
}

// LPL: the emission log-probabilities are staged in LDS -- a COMPILE-time split: with the choice made inside the loop the
// compiler has to assume pending global loads at the join and drains the memory counter there, i.e. waits for the previous
// row's store on every step (the loop then runs at the store round trip, ~800 cycles per step, whatever else is removed).
template <int K, bool LPL>
__device__ __forceinline__ void ctc_wave_recursion(bool is_beta, const float* __restrict__ lg, const float* lse, const float* lpl,
                                                   const int* ext, float* abuf, float* __restrict__ rows,
                                                   float* __restrict__ loss_b, int Tb, int C, int Lx, int LxMax) {
  const int lane = threadIdx.x;            // wave 0 only
  const int s0 = K * lane;
  int es[K]; bool live[K], skip[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int s = s0 + k;
    live[k] = s < Lx;
    es[k] = live[k] ? ext[s] : 0;
    skip[k] = !is_beta ? (live[k] && s >= 2 && es[k] != 0 && es[k] != ext[s - 2])
                       : (live[k] && s + 2 < Lx && ext[s + 2] != 0 && ext[s + 2] != es[k]);
  }
  auto lp_of = [&](int t, int k) { return LPL ? lpl[t * C + es[k]] : (lg[(long long)t * C + es[k]] - lse[t]); };
  float p[K], lpn[K];
  if (!is_beta) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int s = s0 + k;
      p[k] = (live[k] && s <= 1) ? lp_of(0, k) : -INFINITY;
      if (live[k]) rows[s] = p[k];
      lpn[k] = Tb > 1 ? lp_of(1, k) : 0.f;
    }
    // unrolled: a row's store reads its registers when it EXECUTES, so a register may not be rewritten before the store has
    // left (the compiler waits on the memory counter for that); four steps' values in different registers leave the stores
    // three steps to drain instead of putting their round trip on every step
#pragma unroll 4
    for (int t = 1; t < Tb; ++t) {
      float lpc[K];
#pragma unroll
      for (int k = 0; k < K; ++k) { lpc[k] = lpn[k]; lpn[k] = lp_of(min(t + 1, Tb - 1), k); }
      const float up1 = from_lane_below(p[K - 1]), up2 = from_lane_below(p[K - 2]);
      float v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float a2 = k >= 1 ? p[k - 1] : up1;
        const float a3 = k >= 2 ? p[k - 2] : (k == 1 ? up1 : up2);
        v[k] = live[k] ? lse3(p[k], a2, skip[k] ? a3 : -INFINITY) + lpc[k] : -INFINITY;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) { p[k] = v[k]; if (live[k]) rows[(long long)t * LxMax + s0 + k] = v[k]; }
    }
    // -log(alpha_T(Lx-1) + alpha_T(Lx-2)): through LDS (once)
#pragma unroll
    for (int k = 0; k < K; ++k) if (live[k]) abuf[s0 + k] = p[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *loss_b = -(Lx > 1 ? lse2(abuf[Lx - 1], abuf[Lx - 2]) : abuf[0]);
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int s = s0 + k;
      p[k] = (live[k] && (s == Lx - 1 || s == Lx - 2)) ? lp_of(Tb - 1, k) : -INFINITY;
      if (live[k]) rows[(long long)(Tb - 1) * LxMax + s] = p[k];
      lpn[k] = Tb > 1 ? lp_of(Tb - 2, k) : 0.f;
    }
#pragma unroll 4
    for (int t = Tb - 2; t >= 0; --t) {
      float lpc[K];
#pragma unroll
      for (int k = 0; k < K; ++k) { lpc[k] = lpn[k]; lpn[k] = lp_of(max(t - 1, 0), k); }
      const float dn1 = from_lane_above(p[0]), dn2 = from_lane_above(p[1]);
      float v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float b2 = k + 1 < K ? p[k + 1] : dn1;
        const float b3 = k + 2 < K ? p[k + 2] : (k + 1 < K ? dn1 : dn2);
        // a state past the end never feeds a live one: its own value stays -inf
        v[k] = live[k] ? lse3(p[k], b2, skip[k] ? b3 : -INFINITY) + lpc[k] : -INFINITY;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) { p[k] = v[k]; if (live[k]) rows[(long long)t * LxMax + s0 + k] = v[k]; }
    }
  }
}

__global__ __launch_bounds__(1024) void ctc_recursion_kernel(const float* __restrict__ logits,
                                                             const int32_t* __restrict__ targets,
                                                             const int32_t* __restrict__ in_len,
                                                             const int32_t* __restrict__ tgt_len,
                                                             float* __restrict__ loss, float* __restrict__ alpha_ws,
                                                             float* __restrict__ beta_ws, int T, int C, int S_max,
                                                             int lp_in_lds, int wave_k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x;
  const bool is_beta = blockIdx.y == 1;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int LxMax = 2 * S_max + 1;
  float* lse = reinterpret_cast<float*>(smem_raw);  // [T]
  float* abuf = lse + T;                             // [2][LxMax]
  int* ext = reinterpret_cast<int*>(abuf + 2 * LxMax);   // [LxMax]
  float* lpl = reinterpret_cast<float*>(ext + LxMax + (LxMax & 1));   // [T][C] when lp_in_lds

  int Tb = in_len[b];
  int Sb = tgt_len[b];
  if (Tb > T) Tb = T;
  if (Sb > S_max) Sb = S_max;
  const int Lx = 2 * Sb + 1;
  const float* lg = logits + (long long)b * T * C;
  const int32_t* tg = targets + (long long)b * S_max;
  float* rows = (is_beta ? beta_ws : alpha_ws) + (long long)b * T * LxMax;

  if (Tb <= 0) {  // no frames: infeasible (zero_infinity=False -> inf)
    if (tid == 0 && !is_beta) loss[b] = INFINITY;
    return;
  }
  // ---- prologue: log-sum-exp per frame, extended labels, staged log-probabilities ----------------
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
  __syncthreads();
#define LP(t, k) (lp_in_lds ? lpl[(t) * C + (k)] : (lg[(long long)(t) * C + (k)] - lse[(t)]))

  if (wave_k) {   // the whole recursion in wave 0, K states per lane (no barrier below: the other waves are done)
    if (tid >= 64) return;
#define B2T_CTC_WAVE_CALL(KK)                                                                                              \
    { if (lp_in_lds) ctc_wave_recursion<KK, true>(is_beta, lg, lse, lpl, ext, abuf, rows, loss + b, Tb, C, Lx, LxMax);           \
      else ctc_wave_recursion<KK, false>(is_beta, lg, lse, lpl, ext, abuf, rows, loss + b, Tb, C, Lx, LxMax); }
    if (wave_k == 2) B2T_CTC_WAVE_CALL(2)
    else if (wave_k == 4) B2T_CTC_WAVE_CALL(4)
    else B2T_CTC_WAVE_CALL(8)
#undef B2T_CTC_WAVE_CALL
    return;
  }
  const int s = tid;
  const bool live = s < Lx;
  const int es = live ? ext[s] : 0;
  const bool skip_in = live && s >= 2 && es != 0 && es != ext[s - 2];       // alpha: from s-2
  const bool skip_out = live && s + 2 < Lx && ext[s + 2] != 0 && ext[s + 2] != es;  // beta: to s+2

  int cur = 0;
  if (!is_beta) {
    // ---- alpha --------------------------------------------------------------------------------------
    if (live) {
      float a0 = -INFINITY;
      if (s == 0) a0 = LP(0, 0);
      else if (s == 1) a0 = LP(0, es);
      abuf[s] = a0;
      rows[s] = a0;
    }
    lds_barrier();
    for (int t = 1; t < Tb; ++t) {
      const float* ap = abuf + cur * LxMax;
      float* an = abuf + (cur ^ 1) * LxMax;
      if (live) {
        const float la1 = ap[s];
        const float la2 = s >= 1 ? ap[s - 1] : -INFINITY;
        const float la3 = skip_in ? ap[s - 2] : -INFINITY;
        const float v = lse3(la1, la2, la3) + LP(t, es);
        an[s] = v;
        rows[(long long)t * LxMax + s] = v;
      }
      lds_barrier();
      cur ^= 1;
    }
    if (tid == 0) {
      const float* ap = abuf + cur * LxMax;
      loss[b] = -(Lx > 1 ? lse2(ap[Lx - 1], ap[Lx - 2]) : ap[0]);
    }
  } else {
    // ---- beta (includes the emission at t, like alpha) --------------------------------------------------
    for (int t = Tb - 1; t >= 0; --t) {
      const float* bp = abuf + cur * LxMax;
      float* bn = abuf + (cur ^ 1) * LxMax;
      if (live) {
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
        rows[(long long)t * LxMax + s] = v;
      }
      lds_barrier();
      cur ^= 1;
    }
  }
#undef LP
}

// One wave per frame; workgroup (b, j) covers frames j*FPB .. of sentence b.
constexpr int GRAD_WAVES = 4;
__global__ __launch_bounds__(64 * GRAD_WAVES) void ctc_grad_kernel(const float* __restrict__ logits,
                                                                   const int32_t* __restrict__ targets,
                                                                   const int32_t* __restrict__ in_len,
                                                                   const int32_t* __restrict__ tgt_len,
                                                                   const float* __restrict__ loss,
                                                                   const float* __restrict__ alpha_ws,
                                                                   const float* __restrict__ beta_ws,
                                                                   float* __restrict__ dlogits, int T, int C, int S_max,
                                                                   int ldd, float grad_scale, int frames_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LxMax = 2 * S_max + 1;
  int* ext = reinterpret_cast<int*>(smem_raw);            // [LxMax]
  int* plist = ext + LxMax;                               // [LxMax] states grouped by class, increasing s inside a class
  int* cstart = plist + LxMax;                            // [C+1]
  int* ccount = cstart + C + 1;                           // [C]
  float* lprow = reinterpret_cast<float*>(ccount + C);    // [GRAD_WAVES][64]
  float* post = lprow + GRAD_WAVES * 64;                  // [GRAD_WAVES][LxMax]

  int Tb = in_len[b];
  int Sb = tgt_len[b];
  if (Tb > T) Tb = T;
  if (Tb < 0) Tb = 0;
  if (Sb > S_max) Sb = S_max;
  const int Lx = 2 * Sb + 1;
  const float* lg = logits + (long long)b * T * C;
  const int32_t* tg = targets + (long long)b * S_max;
  float* dl = dlogits + (long long)b * T * ldd;
  const int f0 = blockIdx.y * frames_per_block, f1 = min(T, f0 + frames_per_block);

  for (int s = tid; s < Lx; s += blockDim.x) ext[s] = (s & 1) ? tg[s >> 1] : 0;
  __syncthreads();
  if (tid < C) {
    int cnt = 0;
    if (tid > 0) { for (int s = 1; s < Lx; s += 2) cnt += (ext[s] == tid); } else { cnt = (Lx + 1) / 2; }
    ccount[tid] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < C; ++k) { cstart[k] = acc; acc += ccount[k]; }
    cstart[C] = acc;
  }
  __syncthreads();
  if (tid < C) {
    int p = cstart[tid];
    if (tid > 0) { for (int s = 1; s < Lx; s += 2) if (ext[s] == tid) plist[p++] = s; }
    else { for (int s = 0; s < Lx; s += 2) plist[p++] = s; }
  }
  __syncthreads();
  const float nll = loss[b];
  float* mypost = post + wave * LxMax;
  float* mylp = lprow + wave * 64;
  const int p0 = lane < C ? cstart[lane] : 0, p1 = lane < C ? cstart[lane + 1] : 0;

  for (int t = f0 + wave; t < f1; t += GRAD_WAVES) {
    if (t >= Tb) {   // rows beyond the input length: exactly zero
      for (int k = lane; k < ldd; k += 64) dl[(long long)t * ldd + k] = 0.f;
      continue;
    }
    // log-softmax of the frame (one class per lane, wave reductions)
    const float x = lane < C ? lg[(long long)t * C + lane] : -INFINITY;
    float m = x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float e = lane < C ? expf(x - m) : 0.f;
    // fixed-order sum (the tree of a butterfly is the same for every lane): deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    const float lpk = x - (m + logf(e));
    mylp[lane] = lpk;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave, lock-step
    const float* ar = alpha_ws + ((long long)b * T + t) * LxMax;
    const float* br = beta_ws + ((long long)b * T + t) * LxMax;
    for (int s = lane; s < Lx; s += 64) mypost[s] = expf(ar[s] + br[s] + nll - mylp[ext[s]]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int base = p0; base < p1; base += 8) {          // reads 8 at a time (clamped), added in list order
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = mypost[plist[base + i < p1 ? base + i : p0]];
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += (base + i < p1) ? v[i] : 0.f;
    }
    for (int k = lane; k < ldd; k += 64) dl[(long long)t * ldd + k] = k < C ? grad_scale * (expf(lpk) - sum) : 0.f;
  }
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
  if (threads < 64) threads = 64;
  size_t smem = sizeof(float) * ((size_t)T + 2 * (size_t)LxMax) + sizeof(int) * ((size_t)LxMax + 2) + 16;
  B2T_REQUIRE(smem <= 160 * 1024 - 64, "ctc_loss: T=%d needs %zu bytes of LDS (>160 KiB)", T, smem);
  const size_t lp_bytes = sizeof(float) * (size_t)T * C;
  const int lp_in_lds = smem + lp_bytes <= 150 * 1024;   // T=500, C=41: 82 KB
  if (lp_in_lds) smem += lp_bytes;
  hipStream_t s = as_stream(stream);
  if (smem > 64 * 1024) {
    int rc = check_hip(hipFuncSetAttribute((const void*)ctc_recursion_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                       "ctc_loss: raise dynamic LDS limit");
    if (rc) return rc;
  }
  float* beta_ws = alpha_ws + (size_t)B * T * LxMax;   // second half of the caller's scratch (only touched with dlogits)
  // <= 512 extended states: the recursion runs inside one wave with 2 / 4 / 8 states per lane (B2T_CTC_WAVE=0: a thread per state)
  const char* env_wave = getenv("B2T_CTC_WAVE");          // read per call: the tests switch it
  const bool wave_ok = !(env_wave && atoi(env_wave) == 0);
  const int wave_k = !wave_ok ? 0 : LxMax <= 128 ? 2 : LxMax <= 256 ? 4 : LxMax <= 512 ? 8 : 0;
  if (wave_k && threads < 256) threads = 256;   // the prologue (log-sum-exp per frame, staging) still uses the whole workgroup
  hipLaunchKernelGGL(ctc_recursion_kernel, dim3(B, dlogits ? 2 : 1), dim3(threads), smem, s, logits, targets, in_len, tgt_len,
                     loss, alpha_ws, beta_ws, T, C, S_max, lp_in_lds, wave_k);
  B2T_CHECK_LAUNCH("b2t_ctc_loss_f32 (recursions)");
  if (dlogits) {
    const int fpb = 64;   // frames per workgroup (16 per wave)
    const size_t gsm = sizeof(int) * (2 * (size_t)LxMax + 2 * (size_t)C + 1) + sizeof(float) * ((size_t)GRAD_WAVES * 64 + (size_t)GRAD_WAVES * LxMax) + 16;
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(B, (T + fpb - 1) / fpb), dim3(64 * GRAD_WAVES), gsm, s, logits, targets, in_len,
                       tgt_len, loss, alpha_ws, beta_ws, dlogits, T, C, S_max, ldd, grad_scale, fpb);
    B2T_CHECK_LAUNCH("b2t_ctc_loss_f32 (gradient)");
  }
  return 0;
}
