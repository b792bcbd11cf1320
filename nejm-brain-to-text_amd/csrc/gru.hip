// gru.hip — GRU layer sweeps (torch.nn.GRU semantics, gate order r,z,n) for gfx950.
//
// Work decomposition (both modes): workgroup (js, mb) owns hidden units [16*js, 16*js+16) and batch
// rows [16*mb, 16*mb+16).  The recurrent product h_{t-1} W_hh^T for that 16x16 patch (x3 gates) is
// computed with v_mfma_f32_16x16x4_f32 (exact fp32): A = h rows (lane i=l&15, k-group q=l>>4),
// B = W_hh rows of the owned units; the K range is split over the 4 waves of the workgroup in
// 16-wide chunks and reduced through LDS, after which each thread owns exactly one (row, unit)
// output and the gate non-linearities are lane-local.
//
//   mode 0 (step-launch): one launch per time step; the kernel boundary orders h_t between steps.
//   mode 1 (persistent):  see gru_persistent.hip — W_hh slices stay in registers for all T steps and
//                         h_t is handed between workgroups with agent-scope flags.
#include "common.h"
#include "gru_cell.h"
#include "gru_sync.h"

namespace b2t {

// ---------------------------------------------------------------------------------------------------
// forward, one time step
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ gi_t, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_hh,
                                                           const float* __restrict__ h_prev, float* __restrict__ out_t,
                                                           float* __restrict__ res_t, int B, int H) {
  __shared__ float red[4 * 3 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int j = lane & 15, q = lane >> 4;
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nch = H / 16;
  const int arow = m0 + j;  // A-operand row of this lane
  for (int c = wave; c < nch; c += 4) {
    const int kb = c * 16 + 4 * q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (arow < B) a = *reinterpret_cast<const float4*>(h_prev + (long long)arow * H + kb);
    float4 w[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) w[g] = *reinterpret_cast<const float4*>(w_hh + ((long long)g * H + j0 + j) * H + kb);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[g].x, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[g].y, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[g].z, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[g].w, acc[g], 0, 0, 0);
    }
  }
  float gh[3];
  cross_wave_reduce<3>(red, acc, gh, wave, lane);
  const int row = m0 + 4 * q + wave, unit = j0 + j;
  if (row < B) {
    const float hp = h_prev[(long long)row * H + unit];
    gru_gate_fwd(gi_t + (long long)row * 3 * H, b_hh, gh, hp, unit, H, out_t + (long long)row * H,
                 res_t ? res_t + (long long)row * 4 * H : nullptr);
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, one time step.  Launch index t runs T-1 ... 0, then t = -1 (carry only -> dh_init).
//   phase A (t < T-1): carry = dzterm + dGh_{t+1} . W_hh[:, slice]      (K = 3H)
//   phase B (t >= 0) : gate gradients at t for the owned slice -> dG[t], dzterm
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ dh_last,
                                                           const float* __restrict__ reserve,
                                                           const float* __restrict__ out,
                                                           const float* __restrict__ h_init,
                                                           const float* __restrict__ w_hh_t, float* __restrict__ dG,
                                                           float* __restrict__ dh_init, float* __restrict__ dzterm,
                                                           int t, int T, int B, int H) {
  __shared__ float red[4 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int j = lane & 15, q = lane >> 4;
  const int row = m0 + 4 * q + wave, unit = j0 + j;
  float carry = 0.f;
  if (t < T - 1) {
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    const float* dgh = dG + (long long)(t + 1) * B * 4 * H;  // dGh_{t+1}: cols [0,3H) of the 4H row
    const int nch = 3 * H / 16;
    const int arow = m0 + j;
    for (int c = wave; c < nch; c += 4) {
      const int kb = c * 16 + 4 * q;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (arow < B) a = *reinterpret_cast<const float4*>(dgh + (long long)arow * 4 * H + kb);
      const float4 w = *reinterpret_cast<const float4*>(w_hh_t + (long long)(j0 + j) * 3 * H + kb);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[0], 0, 0, 0);
    }
    float s[1];
    cross_wave_reduce<1>(red, acc, s, wave, lane);
    if (row < B) carry = s[0] + dzterm[(long long)row * H + unit];
  } else if (dh_last && row < B) {
    carry = dh_last[(long long)row * H + unit];
  }
  if (row >= B) return;
  if (t < 0) {
    dh_init[(long long)row * H + unit] = carry;
    return;
  }
  const float* rs = reserve + ((long long)t * B + row) * 4 * H;
  const float hp = t > 0 ? out[((long long)(t - 1) * B + row) * H + unit] : h_init[(long long)row * H + unit];
  const float d = dY[((long long)t * B + row) * H + unit] + carry;
  float dz_term;
  gru_gate_bwd(rs, hp, d, unit, H, dG + ((long long)t * B + row) * 4 * H, &dz_term);
  dzterm[(long long)row * H + unit] = dz_term;
}

}  // namespace b2t

using namespace b2t;

extern "C" size_t b2t_gru_sync_bytes(int T) { return b2t::gru_persistent_sync_bytes(T); }
extern "C" size_t b2t_gru_ws_bytes(int T, int B, int H) { (void)B; (void)H; return b2t::gru_persistent_sync_bytes(T); }

extern "C" int b2t_gru_layer_fwd_f32(const float* gi, const float* w_hh, const float* b_hh, const float* h_init,
                                     float* out, float* reserve, float* h_last, int T, int B, int H, int mode,
                                     void* sync_ws, void* stream) {
  B2T_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 16) == 0, "gru_layer_fwd: bad shape T=%d B=%d H=%d (H%%16 must be 0)", T, B, H);
  hipStream_t s = as_stream(stream);
  const bool bf16 = (mode & B2T_GRU_BF16) != 0;   // bf16 operands of the recurrent product (persistent mode 1 only)
  const bool wide = (mode & B2T_GRU_WIDE) != 0;   // 32 hidden units per workgroup (with bf16 operands)
  const int local = (mode & B2T_GRU_LOCAL) ? ((mode & B2T_GRU_PARITY) ? 1 : 0) : -1;   // XCD-local hand-off, layer parity
  mode &= ~(B2T_GRU_BF16 | B2T_GRU_WIDE | B2T_GRU_LOCAL | B2T_GRU_PARITY);
  B2T_REQUIRE(mode == 0 || mode == 1, "gru_layer_fwd: unknown mode %d", mode);
  B2T_REQUIRE(!bf16 || mode == 1, "gru_layer_fwd: B2T_GRU_BF16 goes with mode 1");
  if (mode == 1) {
    int rc = gru_persistent_fwd(gi, w_hh, b_hh, h_init, out, reserve, T, B, H, sync_ws, s, bf16, wide, local);
    if (rc) return rc;
  } else {
    dim3 grid(H / 16, (B + 15) / 16), block(256);
    for (int t = 0; t < T; ++t) {
      const float* hp = t == 0 ? h_init : out + (long long)(t - 1) * B * H;
      hipLaunchKernelGGL(gru_step_fwd_kernel, grid, block, 0, s, gi + (long long)t * B * 3 * H, w_hh, b_hh, hp,
                         out + (long long)t * B * H, reserve ? reserve + (long long)t * B * 4 * H : nullptr, B, H);
    }
    B2T_CHECK_LAUNCH("b2t_gru_layer_fwd_f32");
  }
  if (h_last) {
    int rc = check_hip(hipMemcpyAsync(h_last, out + (long long)(T - 1) * B * H, sizeof(float) * B * H,
                                      hipMemcpyDeviceToDevice, s), "gru_layer_fwd: h_last copy");
    if (rc) return rc;
  }
  return 0;
}

extern "C" int b2t_gru_layer_fwd_fused_f32(const float* gi, const float* w_hh, const float* b_hh, const float* h_init,
                                           float* out, float* reserve, float* h_last, const float* w_ih_next,
                                           const float* b_ih_next, float* gi_next, int T, int B, int H, int mode,
                                           void* sync_ws, void* stream) {
  B2T_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 16) == 0 && H <= 512, "gru_layer_fwd_fused: bad shape T=%d B=%d H=%d (H%%16 must be 0, H <= 512)", T, B, H);
  B2T_REQUIRE(gi && w_hh && b_hh && h_init && out && w_ih_next && b_ih_next && gi_next, "gru_layer_fwd_fused: null argument");
  hipStream_t s = as_stream(stream);
  const int local = (mode & B2T_GRU_LOCAL) ? ((mode & B2T_GRU_PARITY) ? 1 : 0) : -1;
  mode &= ~(B2T_GRU_LOCAL | B2T_GRU_PARITY);
  B2T_REQUIRE(mode == 1, "gru_layer_fwd_fused: persistent exact-fp32 sweeps only (mode 1, optionally | B2T_GRU_LOCAL | B2T_GRU_PARITY), got %d", mode);
  int rc = gru_persistent_fwd_fused(gi, w_hh, b_hh, h_init, out, reserve, w_ih_next, b_ih_next, gi_next, T, B, H, sync_ws, s, local);
  if (rc) return rc;
  if (h_last)
    return check_hip(hipMemcpyAsync(h_last, out + (long long)(T - 1) * B * H, sizeof(float) * B * H, hipMemcpyDeviceToDevice, s),
                     "gru_layer_fwd_fused: h_last copy");
  return 0;
}

extern "C" int b2t_gru_layer_bwd_f32(const float* dY, const float* dh_last, const float* reserve, const float* out,
                                     const float* h_init, const float* w_hh_t, float* dG, float* dh_init,
                                     float* carry_ws, int T, int B, int H, int mode, void* sync_ws, void* stream) {
  B2T_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 16) == 0, "gru_layer_bwd: bad shape T=%d B=%d H=%d", T, B, H);
  hipStream_t s = as_stream(stream);
  const bool bf16 = (mode & B2T_GRU_BF16) != 0;
  const bool wide = (mode & B2T_GRU_WIDE) != 0;
  const int local = (mode & B2T_GRU_LOCAL) ? ((mode & B2T_GRU_PARITY) ? 1 : 0) : -1;
  const bool paired = (mode & B2T_GRU_PAIRED) != 0;
  const int set = (mode >> B2T_GRU_SET_SHIFT) & 3;
  mode &= ~(B2T_GRU_BF16 | B2T_GRU_WIDE | B2T_GRU_LOCAL | B2T_GRU_PARITY | B2T_GRU_PAIRED | (3 << B2T_GRU_SET_SHIFT));
  B2T_REQUIRE(mode == 0 || mode == 1, "gru_layer_bwd: unknown mode %d", mode);
  B2T_REQUIRE(!bf16 || mode == 1, "gru_layer_bwd: B2T_GRU_BF16 goes with mode 1");
  B2T_REQUIRE(!paired || (mode == 1 && !bf16 && !wide), "gru_layer_bwd: B2T_GRU_PAIRED goes with mode 1, exact fp32");
  if (paired && gru_persistent_bwd_pair_ok(B, H))     // (shapes it does not serve take the register-resident sweep below)
    return gru_persistent_bwd_pair(dY, dh_last, reserve, out, h_init, w_hh_t, dG, dh_init, T, B, H, sync_ws, s, set);
  if (mode == 1)
    return gru_persistent_bwd(dY, dh_last, reserve, out, h_init, w_hh_t, dG, dh_init, T, B, H, sync_ws, s, bf16, wide, local);
  B2T_REQUIRE(carry_ws != nullptr, "gru_layer_bwd: carry_ws is required in mode 0 ([B][H] floats)");
  dim3 grid(H / 16, (B + 15) / 16), block(256);
  for (int t = T - 1; t >= -1; --t)
    hipLaunchKernelGGL(gru_step_bwd_kernel, grid, block, 0, s, dY, dh_last, reserve, out, h_init, w_hh_t, dG, dh_init,
                       carry_ws, t, T, B, H);
  B2T_CHECK_LAUNCH("b2t_gru_layer_bwd_f32");
  return 0;
}
