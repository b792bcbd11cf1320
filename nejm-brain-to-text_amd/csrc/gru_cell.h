// gru_cell.h — device pieces shared by the step-launch and persistent GRU sweeps.
#pragma once
#include "common.h"

namespace b2t {

using f32x4 = float __attribute__((ext_vector_type(4)));

// Sum the 4 waves' partial 16x16 MFMA accumulators through LDS.  After the call thread (wave, lane)
// owns output row 4*(lane>>4) + wave, column lane&15 of each of the NG tiles.
// red must hold 4*NG*4*64 floats.  Contains one __syncthreads(); callers looping over steps must
// place another barrier before the next call reuses `red`.
template <int NG>
__device__ __forceinline__ void cross_wave_reduce(float* red, const f32x4 (&acc)[NG], float (&out)[NG], int wave,
                                                  int lane) {
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * NG + g) * 4 + r) * 64 + lane] = acc[g][r];
  __syncthreads();
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const float s0 = red[((0 * NG + g) * 4 + wave) * 64 + lane];
    const float s1 = red[((1 * NG + g) * 4 + wave) * 64 + lane];
    const float s2 = red[((2 * NG + g) * 4 + wave) * 64 + lane];
    const float s3 = red[((3 * NG + g) * 4 + wave) * 64 + lane];
    out[g] = (s0 + s1) + (s2 + s3);
  }
}

// Gate math of one (row, unit): gi row pointer [3H], recurrent sums gh[3] (without bias).
// r = s(gi_r+gh_r), z = s(gi_z+gh_z), n = tanh(gi_n + r*gh_n), h = (1-z) n + z h_prev.
__device__ __forceinline__ float gru_gate_fwd(const float* __restrict__ gi_row, const float* __restrict__ b_hh,
                                              const float (&gh)[3], float hp, int unit, int H,
                                              float* __restrict__ out_row, float* __restrict__ res_row) {
  const float ghr = gh[0] + b_hh[unit];
  const float ghz = gh[1] + b_hh[H + unit];
  const float ghn = gh[2] + b_hh[2 * H + unit];
  const float r = sigmoidf_(gi_row[unit] + ghr);
  const float z = sigmoidf_(gi_row[H + unit] + ghz);
  const float n = tanhf(gi_row[2 * H + unit] + r * ghn);
  const float h = (1.0f - z) * n + z * hp;
  out_row[unit] = h;
  if (res_row) {
    res_row[unit] = r;
    res_row[H + unit] = z;
    res_row[2 * H + unit] = n;
    res_row[3 * H + unit] = ghn;
  }
  return h;
}

// Gate gradients of one (row, unit).  rs = reserve row [4H] = (r,z,n,gh_n).
// Writes dG row [4H] = (dr_pre, dz_pre, dn_pre*r, dn_pre); *dz_term = d*z (the direct path into dh_{t-1}).
__device__ __forceinline__ void gru_gate_bwd(const float* __restrict__ rs, float hp, float d, int unit, int H,
                                             float* __restrict__ dg_row, float* dz_term) {
  const float r = rs[unit], z = rs[H + unit], n = rs[2 * H + unit], ghn = rs[3 * H + unit];
  const float dn = d * (1.0f - z);
  const float dz = d * (hp - n);
  const float dn_pre = dn * (1.0f - n * n);
  const float dz_pre = dz * z * (1.0f - z);
  const float dr_pre = dn_pre * ghn * r * (1.0f - r);
  dg_row[unit] = dr_pre;
  dg_row[H + unit] = dz_pre;
  dg_row[2 * H + unit] = dn_pre * r;
  dg_row[3 * H + unit] = dn_pre;
  *dz_term = d * z;
}

// persistent sweeps (gru_persistent.hip)
size_t gru_persistent_sync_bytes(int T);
int gru_persistent_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                       float* reserve, int T, int B, int H, void* sync_ws, hipStream_t s, bool bf16 = false,
                       bool wide = false, int local = -1);
int gru_persistent_fwd_fused(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                             float* reserve, const float* w_ih2, const float* b_ih2, float* gi2, int T, int B, int H,
                             void* sync_ws, hipStream_t s, int local = -1);
int gru_persistent_bwd(const float* dY, const float* dh_last, const float* reserve, const float* out,
                       const float* h_init, const float* w_hh_t, float* dG, float* dh_init, int T, int B, int H,
                       void* sync_ws, hipStream_t s, bool bf16 = false, bool wide = false, int local = -1);

// the layer wavefront (gru_wave.hip, round 6): all L sweeps of a direction in ONE launch, layer l + 1 a step or two behind layer l
struct WaveFwdArgs {
  int L, T, B, H;
  const float* gi0;                                        // [T][B][3H]: layer 0's input projection (b_ih[0] included)
  const float* w_hh[B2T_MAX_LAYERS]; const float* b_hh[B2T_MAX_LAYERS];
  const float* w_ih[B2T_MAX_LAYERS]; const float* b_ih[B2T_MAX_LAYERS];   // layers >= 1 ([3H][H], [3H])
  const float* h_init[B2T_MAX_LAYERS];                     // [B][H]
  float* out[B2T_MAX_LAYERS];                              // [T][B][H]
  float* outd[B2T_MAX_LAYERS];                             // dropout(out[l]) for the layer above's weight gradient (drop_p > 0, l < L - 1)
  float* reserve[B2T_MAX_LAYERS];                          // [T][B][4H] = (r, z, n, gh_n), or null
  char* ring[B2T_MAX_LAYERS]; char* ringd[B2T_MAX_LAYERS]; // fragment rings: slot 0..T of h / dropped h
  unsigned* cnt;                                           // 16 ticket words + [L][2][row groups][T + 1], zeroed by the launcher
  unsigned* tickets;                                       // (set by the launcher: the first 16 words of cnt)
  unsigned* timing;                                        // timing build only: [L][8] cycles per step and phase
  unsigned* err;                                           // sticky error word (a bounded spin gave up)
  float drop_p, drop_scale; unsigned long long seed[B2T_MAX_LAYERS]; long long elem0;
  int flags;                                               // bit 0: sc1 fragment loads (A/B knob)
};
struct WaveBwdArgs {
  int L, T, B, H;
  const float* dY_top;                                     // [T][B][H]: gradient wrt the top layer's outputs
  const float* dh_last[B2T_MAX_LAYERS];                    // per layer [B][H] or null: gradient wrt the state after the last step (a later chunk's dh_init)
  float* dh_init[B2T_MAX_LAYERS];                          // per layer [B][H]
  const float* w_hh_t[B2T_MAX_LAYERS];                     // [H][3H] = W_hh^T
  const float* w_ih_t[B2T_MAX_LAYERS];                     // [H][3H] = W_ih^T of layers >= 1
  const float* h_init[B2T_MAX_LAYERS]; const float* out[B2T_MAX_LAYERS]; const float* reserve[B2T_MAX_LAYERS];
  float* dG[B2T_MAX_LAYERS];                               // [T][B][4H] = (dr, dz, dn r, dn)
  char* ring[B2T_MAX_LAYERS];                              // fragment rings: slot t, 4 arrays
  char* ringx[B2T_MAX_LAYERS];                             // the written-through copy the layer below reads in the local form
  unsigned* cnt;                                           // 16 ticket words + 64 progress words + [L][2][row groups][T]
  unsigned* tickets;
  unsigned* prog;                                          // non-null (any value; the launcher points it into cnt): publish progress for gated consumers
  unsigned* timing;
  unsigned* err;
  float drop_p, drop_scale; unsigned long long seed[B2T_MAX_LAYERS]; long long elem0;
  int flags;                                               // bit 0: sc1 fragment loads (A/B knob)
};
bool gru_wave_ok(int L, int T, int B, int H, const char** why);
bool gru_wave_local(int L, int H);
bool gru_wave_ks(int L, int T, int B, int H);      // the K-split form serves this shape on this device
bool gru_xcd_dispatch_ok();
size_t gru_wave_ring_bytes_fwd(int T, int B, int H);
size_t gru_wave_ring_bytes_bwd(int T, int B, int H);
size_t gru_wave_cnt_words_fwd(int L, int T, int B);
size_t gru_wave_cnt_words_bwd(int L, int T, int B);
int gru_wave_fwd(const WaveFwdArgs& a, hipStream_t s);
int gru_wave_bwd(const WaveBwdArgs& a, hipStream_t s);   // a.flags bit 1: counters already cleared (gru_wave_bwd_clear); bit 2: dG written through (readers while the sweep runs); bit 3: K-split form, w_hh_t / w_ih_t are the untransposed [3H][H] matrices
int gru_wave_bwd_clear(unsigned* cnt, int L, int T, int B, hipStream_t s);
int gru_wave_gate(unsigned* cnt, int layer, int t0, int T, int B, int H, unsigned* err, hipStream_t s);

bool gru_persistent_bwd_pair_ok(int B, int H);
int gru_persistent_bwd_pair(const float* dY, const float* dh_last, const float* reserve, const float* out, const float* h_init,
                            const float* w_hh_t, float* dG, float* dh_init, int T, int B, int H, void* sync_ws, hipStream_t s, int set);

}  // namespace b2t
