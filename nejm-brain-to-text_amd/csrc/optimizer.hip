// optimizer.hip — clip_grad_norm_ + AdamW over a flat parameter arena (one launch each, no host sync).
//
// The host lays every parameter out in ONE fp32 arena, each tensor ("segment") padded to a multiple
// of 1024 floats, so a 256-thread block (float4 per lane) maps to exactly one 1024-float chunk and
// chunk2seg[chunk] names its tensor.  Device-resident tables per segment:
//   seg_group : 0 = bias group, 1 = day group, 2 = other   (model_training/rnn_trainer.py:267-269)
//   seg_day   : day index of a day-layer tensor, -1 otherwise
//   seg_step  : the tensor's own AdamW step count (torch keeps `step` per parameter)
//   active    : 1 if the tensor has a gradient this step.  Day tensors of sessions absent from the
//               batch have grad None in the reference (rnn_trainer.py:514 zero_grad) and are skipped
//               by clip_grad_norm_ and AdamW; their step counters do not advance.
#include "common.h"

namespace b2t {

constexpr int CHUNK = 1024;

__global__ void opt_prepare_kernel(const int32_t* __restrict__ day_idx, int B, const int32_t* __restrict__ seg_day,
                                   int nseg, int32_t* __restrict__ active) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const int d = seg_day[s];
  int a = 1;
  if (d == -2) {          // frozen tensor (requires_grad = False: rnn_trainer.py:249-254)
    a = 0;
  } else if (d >= 0) {
    a = 0;
    for (int b = 0; b < B; ++b) a |= (day_idx[b] == d);
  }
  active[s] = a;
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, const int* __restrict__ chunk2seg,
                                                            const int* __restrict__ active, float* __restrict__ part) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.f;
  if (active[chunk2seg[c]]) {
    const float4 v = reinterpret_cast<const float4*>(g + (long long)c * CHUNK)[threadIdx.x];
    s = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out4 = {sum g^2, norm, clip_coef, status}; also advances the per-tensor step counters of active tensors.
// status: 0 OK, 1 a persistent sweep's error word is set (hand-off timeout: gradients invalid), 2 non-finite norm
// (clip_grad_norm_(error_if_nonfinite=True), rnn_trainer.py:551-555).  Sticky; a bad step advances nothing.
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ part, int n, float max_norm,
                                                           float* __restrict__ out3, const int* __restrict__ active,
                                                           int* __restrict__ seg_step, int nseg,
                                                           const unsigned* __restrict__ err_words, int n_err,
                                                           long long err_stride) {
  __shared__ double red[16];
  __shared__ int bad_s, err_s;
  if (threadIdx.x == 0) err_s = 0;
  __syncthreads();
  // the sweeps' sticky error words: one lane each (10 dependent device-scope loads by one thread before)
  for (int i = threadIdx.x; i < n_err; i += blockDim.x)
    if (__hip_atomic_load(const_cast<unsigned*>(err_words) + (long long)i * err_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicOr(&err_s, 1);
  // four independent chains per thread: the loads of a round are in flight together (one chain: 42 dependent rounds of load + add
  // for the 43 k partials of the shipped model, 57 us on the step's tail)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const int bd = blockDim.x;
  int i = threadIdx.x;
  for (; i + 3 * bd < n; i += 4 * bd) {
    const float a = part[i], b = part[i + bd], c = part[i + 2 * bd], d = part[i + 3 * bd];
    s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
  }
  for (; i < n; i += bd) s0 += (double)part[i];
  double s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    const float sumsq = (float)t;
    const float norm = sqrtf(sumsq);
    float coef = 1.0f;
    if (max_norm > 0.f) coef = fminf(1.0f, max_norm / (norm + 1e-6f));
    float status = out3[3];
    if (!(status > 0.f)) {
      status = err_s ? 1.f : 0.f;
      if (status == 0.f && !(fabsf(norm) <= 3.0e38f)) status = 2.f;   // inf or nan
    }
    out3[0] = sumsq; out3[1] = norm; out3[2] = coef; out3[3] = status;
    bad_s = status != 0.f;
  }
  __syncthreads();
  if (seg_step && !bad_s)
    for (int i = threadIdx.x; i < nseg; i += blockDim.x) seg_step[i] += (active[i] != 0);
}

// Data parallel: the status word is MAX-reduced over the ranks between the norm and the update, so the step counters
// may only advance once the GLOBAL status is known (b2t_grad_norm_clip_f32 is then called with seg_step = NULL).
__global__ void opt_advance_kernel(const int* __restrict__ active, int* __restrict__ seg_step, int nseg,
                                   const float* __restrict__ out4) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg || out4[3] != 0.f) return;
  seg_step[s] += (active[s] != 0);
}

struct GroupHyp { float lr[3]; float wd[3]; };

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const int* __restrict__ chunk2seg,
                                                    const int* __restrict__ active, const int* __restrict__ seg_group,
                                                    const int* __restrict__ seg_step, const float* __restrict__ clip3,
                                                    int apply_clip, GroupHyp hyp, double beta1, double beta2, float eps) {
  const int c = blockIdx.x;
  const int seg = chunk2seg[c];
  if (!active[seg]) return;
  if (clip3 && clip3[3] != 0.f) return;   // invalid gradients (hand-off timeout / non-finite norm): never applied
  const int grp = seg_group[seg];
  const float lr = hyp.lr[grp], wd = hyp.wd[grp];
  const int k = seg_step[seg];  // already advanced for this step (1-based)
  const float bc1 = (float)(1.0 - pow(beta1, (double)k));
  const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)k));
  const float coef = (clip3 && apply_clip) ? clip3[2] : 1.0f;
  const float b1 = (float)beta1, b2 = (float)beta2;
  const long long i = (long long)c * (CHUNK / 4) + threadIdx.x;
  float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<float4*>(g)[i];
  float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
  const float decay = 1.0f - lr * wd, step = lr / bc1, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#define B2T_ADAM1(P, G, M, V)                         \
  {                                                   \
    const float gr = G * coef;                        \
    P = P * decay;                                    \
    M = b1 * M + omb1 * gr;                           \
    V = b2 * V + omb2 * gr * gr;                      \
    P = P - step * (M / (sqrtf(V) / bc2s + eps));     \
    G = gr;                                           \
  }
  B2T_ADAM1(pp.x, gg.x, mm.x, vv.x)
  B2T_ADAM1(pp.y, gg.y, mm.y, vv.y)
  B2T_ADAM1(pp.z, gg.z, mm.z, vv.z)
  B2T_ADAM1(pp.w, gg.w, mm.w, vv.w)
#undef B2T_ADAM1
  reinterpret_cast<float4*>(p)[i] = pp;
  reinterpret_cast<float4*>(g)[i] = gg;  // clipped gradient, like clip_grad_norm_ leaves it
  reinterpret_cast<float4*>(m)[i] = mm;
  reinterpret_cast<float4*>(v)[i] = vv;
}

}  // namespace b2t

using namespace b2t;

extern "C" int b2t_opt_prepare(const int32_t* day_idx, int B, const int32_t* seg_day, int nseg, int32_t* active,
                               void* stream) {
  B2T_REQUIRE(B > 0 && nseg > 0, "opt_prepare: bad args");
  hipLaunchKernelGGL(opt_prepare_kernel, dim3((nseg + 127) / 128), dim3(128), 0, as_stream(stream), day_idx, B, seg_day,
                     nseg, active);
  B2T_CHECK_LAUNCH("b2t_opt_prepare");
  return 0;
}

extern "C" int b2t_opt_advance(const int32_t* active, int32_t* seg_step, int nseg, const float* out4, void* stream) {
  B2T_REQUIRE(active && seg_step && out4 && nseg > 0, "opt_advance: bad args");
  hipLaunchKernelGGL(opt_advance_kernel, dim3((nseg + 127) / 128), dim3(128), 0, as_stream(stream), active, seg_step, nseg,
                     out4);
  B2T_CHECK_LAUNCH("b2t_opt_advance");
  return 0;
}

extern "C" int b2t_grad_norm_clip_f32(const float* grads, const int32_t* chunk2seg, const int32_t* active, int nchunks,
                                      float max_norm, float* partial_ws, float* out3, int32_t* seg_step, int nseg,
                                      const uint32_t* err_words, int n_err, long long err_stride, void* stream) {
  B2T_REQUIRE(nchunks > 0 && partial_ws && out3, "grad_norm_clip: bad args");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nchunks), dim3(256), 0, s, grads, chunk2seg, active, partial_ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, s, partial_ws, nchunks, max_norm, out3, active,
                     seg_step, nseg, err_words, err_words ? n_err : 0, err_stride);
  B2T_CHECK_LAUNCH("b2t_grad_norm_clip_f32");
  return 0;
}

extern "C" int b2t_adamw_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int32_t* chunk2seg,
                             const int32_t* active, const int32_t* seg_group, const int32_t* seg_step, int nchunks,
                             const float* clip3, int apply_clip, const float* lr3_host, const float* wd3_host,
                             double beta1, double beta2, float eps, void* stream) {
  B2T_REQUIRE(nchunks > 0 && lr3_host && wd3_host, "adamw: bad args");
  GroupHyp h;
  for (int i = 0; i < 3; ++i) { h.lr[i] = lr3_host[i]; h.wd[i] = wd3_host[i]; }
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, as_stream(stream), params, grads, exp_avg, exp_avg_sq,
                     chunk2seg, active, seg_group, seg_step, clip3, apply_clip, h, beta1, beta2, eps);
  B2T_CHECK_LAUNCH("b2t_adamw_f32");
  return 0;
}
