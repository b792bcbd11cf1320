// gru_persistent.hip — persistent (single-launch) GRU layer sweeps for gfx950.
//
// Same decomposition as the step-launch kernels in gru.hip — workgroup (js, mb) owns hidden units
// [16js,16js+16) x batch rows [16mb,16mb+16) — but ONE launch runs all T steps:
//   * the workgroup's slice of W_hh (forward: rows of the owned units; backward: columns, read from the
//     transposed copy) is loaded ONCE into VGPRs as MFMA B-operands and stays there for the whole sweep
//     (H=512: 96 VGPRs per lane per wave; the K range is split over the 4 waves);
//   * h_t (forward) / dGh_t (backward) is exchanged between the G = H/16 workgroups of a row group
//     through HBM/L2 with the write-through hand-off of cdna_hip_programming.md §6 Guideline 16 (R1):
//     producers store the payload with sc1 (write-through) stores, every storing wave drains vmcnt(0),
//     barrier, one lane bumps an agent-scope counter; consumers poll that ONE word relaxed and then read
//     the payload with sc1 loads (no L1 hit possible, no fence needed).  Counters are zeroed by a
//     memset node ahead of every launch; every spin is bounded and reports through an error word.
//   * row groups are independent recurrences, so the 4 groups of a B=64 batch progress independently.
// All workgroups must be co-resident (grid <= CU count); the entry point checks that.
#include "gru_cell.h"

namespace b2t {

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned SPIN_LIMIT = 4u << 20;  // ~seconds; a healthy hand-off takes microseconds

// Thread 0 polls until *p >= target (or the error word is set / the spin limit is hit), then barrier.
__device__ __forceinline__ void wait_count(unsigned* p, unsigned target, unsigned* err) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, RLX_AGENT) < target) {
      ++spins;
      if ((spins & 255u) == 0u) {
        if (__hip_atomic_load(err, RLX_AGENT) != 0u) break;
        if (spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, RLX_AGENT); break; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// All waves have issued their sc1 payload stores: drain, barrier, one lane publishes.
__device__ __forceinline__ void publish_count(unsigned* p) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, 1u, RLX_AGENT);
}

// 16-byte sc1 (L1-bypassing) load through a buffer descriptor based at a wave-uniform pointer.
__device__ __forceinline__ float4 load_sc1_f4(const float* base_uniform, unsigned byte_off) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, /*aux: sc1*/ 16);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}

__device__ __forceinline__ void store_sc1(float* p, float v) { __hip_atomic_store(p, v, RLX_AGENT); }

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <int NCH, int MT>  // NCH: 16-wide K chunks per wave (H <= 64*NCH); MT: row groups per workgroup
__global__ __launch_bounds__(256, 1) void gru_persist_fwd_kernel(const float* __restrict__ gi,
                                                                 const float* __restrict__ w_hh,
                                                                 const float* __restrict__ b_hh,
                                                                 const float* __restrict__ h_init, float* out,
                                                                 float* __restrict__ reserve, int T, int B, int H,
                                                                 unsigned* sync) {
  __shared__ float red[4 * 3 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * 16;
  const unsigned G = gridDim.x;
  const int j = lane & 15, q = lane >> 4;
  const int unit = j0 + j;
  unsigned* err = sync + (size_t)gridDim.y * MT * T;
  const int nch = H / 16;

  float4 w[3][NCH];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    const int c = wave + 4 * ci;
#pragma unroll
    for (int g = 0; g < 3; ++g)
      w[g][ci] = c < nch ? *reinterpret_cast<const float4*>(w_hh + ((long long)g * H + unit) * H + c * 16 + 4 * q)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float bhr = b_hh[unit], bhz = b_hh[H + unit], bhn = b_hh[2 * H + unit];
  float hpv[MT];
#pragma unroll
  for (int rr = 0; rr < MT; ++rr) {
    const int row = (blockIdx.y * MT + rr) * 16 + 4 * q + wave;
    hpv[rr] = row < B ? h_init[(long long)row * H + unit] : 0.f;
  }

  for (int t = 0; t < T; ++t) {
#pragma unroll
   for (int rr = 0; rr < MT; ++rr) {
    // Row groups are independent recurrences: while the peers' h_{t-1} of group r is in flight,
    // this workgroup is busy with the other groups (the hand-off latency hides behind their MFMAs).
    const int rg = blockIdx.y * MT + rr;
    const int m0 = rg * 16;
    if (m0 >= B) continue;
    const int row = m0 + 4 * q + wave, arow = m0 + j;
    const bool live = row < B;
    unsigned* cnt = sync + (size_t)rg * T;
    float hp = hpv[rr];
    float gir = 0.f, giz = 0.f, gin = 0.f;
    if (live) {
      const float* g3 = gi + ((long long)t * B + row) * 3 * H + unit;
      gir = g3[0]; giz = g3[H]; gin = g3[2 * H];
    }
    const float* hsrc = h_init;
    if (t > 0) {
      wait_count(cnt + (t - 1), G, err);
      hsrc = out + (long long)(t - 1) * B * H;
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a[NCH];
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int c = wave + 4 * ci;
      a[ci] = (c < nch && arow < B) ? load_sc1_f4(hsrc, (unsigned)(((long long)arow * H + c * 16 + 4 * q) * 4))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].x, w[g][ci].x, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].y, w[g][ci].y, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].z, w[g][ci].z, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].w, w[g][ci].w, acc[g], 0, 0, 0);
      }
    }
    float gh[3];
    cross_wave_reduce<3>(red, acc, gh, wave, lane);
    if (live) {
      const float ghn = gh[2] + bhn;
      const float r = sigmoidf_(gir + gh[0] + bhr);
      const float z = sigmoidf_(giz + gh[1] + bhz);
      const float n = tanhf(gin + r * ghn);
      const float h = (1.0f - z) * n + z * hp;
      store_sc1(out + ((long long)t * B + row) * H + unit, h);
      if (reserve) {
        float* rs = reserve + ((long long)t * B + row) * 4 * H + unit;
        rs[0] = r; rs[H] = z; rs[2 * H] = n; rs[3 * H] = ghn;
      }
      hp = h;
    }
    hpv[rr] = hp;
    publish_count(cnt + t);  // also fences `red` for the next iteration
   }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward.  Workgroup (ks, mb) owns dh columns [16ks,16ks+16): each step it (A) contracts the full
// dGh_{t+1} row block with its register-resident W_hh[:, slice] and (B) forms the gate gradients of
// step t for its slice, publishing them as dG[t] for the other workgroups of the row group.
// ---------------------------------------------------------------------------------------------------
template <int NCB, int MT>  // NCB: 16-wide chunks of the 3H contraction per wave (3H <= 64*NCB)
__global__ __launch_bounds__(256, 1) void gru_persist_bwd_kernel(const float* __restrict__ dY,
                                                                 const float* __restrict__ dh_last,
                                                                 const float* __restrict__ reserve,
                                                                 const float* __restrict__ out,
                                                                 const float* __restrict__ h_init,
                                                                 const float* __restrict__ w_hh_t, float* dG,
                                                                 float* __restrict__ dh_init, int T, int B, int H,
                                                                 unsigned* sync) {
  __shared__ float red[4 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * 16;
  const unsigned G = gridDim.x;
  const int j = lane & 15, q = lane >> 4;
  const int unit = j0 + j;
  unsigned* err = sync + (size_t)gridDim.y * MT * T;
  const int nch = 3 * H / 16;

  float4 w[NCB];
#pragma unroll
  for (int ci = 0; ci < NCB; ++ci) {
    const int c = wave + 4 * ci;
    w[ci] = c < nch ? *reinterpret_cast<const float4*>(w_hh_t + (long long)unit * 3 * H + c * 16 + 4 * q)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float dzv[MT];
#pragma unroll
  for (int rr = 0; rr < MT; ++rr) dzv[rr] = 0.f;

  for (int t = T - 1; t >= -1; --t) {
#pragma unroll
   for (int rr = 0; rr < MT; ++rr) {
    const int rg = blockIdx.y * MT + rr;
    const int m0 = rg * 16;
    if (m0 >= B) continue;
    const int row = m0 + 4 * q + wave, arow = m0 + j;
    const bool live = row < B;
    unsigned* cnt = sync + (size_t)rg * T;
    float dzterm = dzv[rr];
    // operands of the elementwise part do not depend on the recurrence: fetch them first
    float r = 0.f, z = 0.f, n = 0.f, ghn = 0.f, hprev = 0.f, dy = 0.f;
    if (live && t >= 0) {
      const float* rs = reserve + ((long long)t * B + row) * 4 * H + unit;
      r = rs[0]; z = rs[H]; n = rs[2 * H]; ghn = rs[3 * H];
      hprev = t > 0 ? out[((long long)(t - 1) * B + row) * H + unit] : h_init[(long long)row * H + unit];
      dy = dY[((long long)t * B + row) * H + unit];
    }
    float carry = 0.f;
    if (t < T - 1) {
      wait_count(cnt + (t + 1), G, err);
      const float* dgh = dG + (long long)(t + 1) * B * 4 * H;
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      float4 a[NCB];
#pragma unroll
      for (int ci = 0; ci < NCB; ++ci) {
        const int c = wave + 4 * ci;
        a[ci] = (c < nch && arow < B) ? load_sc1_f4(dgh, (unsigned)(((long long)arow * 4 * H + c * 16 + 4 * q) * 4))
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int ci = 0; ci < NCB; ++ci) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].x, w[ci].x, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].y, w[ci].y, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].z, w[ci].z, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ci].w, w[ci].w, acc[0], 0, 0, 0);
      }
      float s[1];
      cross_wave_reduce<1>(red, acc, s, wave, lane);
      carry = s[0] + dzterm;
    } else if (dh_last && live) {
      carry = dh_last[(long long)row * H + unit];
    }
    if (t < 0) {
      if (live) dh_init[(long long)row * H + unit] = carry;
      continue;
    }
    if (live) {
      const float d = dy + carry;
      const float dn = d * (1.0f - z);
      const float dz = d * (hprev - n);
      const float dn_pre = dn * (1.0f - n * n);
      const float dz_pre = dz * z * (1.0f - z);
      const float dr_pre = dn_pre * ghn * r * (1.0f - r);
      float* dg = dG + ((long long)t * B + row) * 4 * H + unit;
      store_sc1(dg, dr_pre);
      store_sc1(dg + H, dz_pre);
      store_sc1(dg + 2 * H, dn_pre * r);
      store_sc1(dg + 3 * H, dn_pre);
      dzterm = d * z;
    }
    dzv[rr] = dzterm;
    publish_count(cnt + t);
   }
  }
}

size_t gru_persistent_sync_bytes(int T) { return ((size_t)T * 64 + 64) * sizeof(unsigned); }

static int cu_count() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    n = p.multiProcessorCount;
  }
  return n;
}

// Row groups per workgroup: 2 halves the workgroup count (weights are shared by the groups) and lets each
// workgroup overlap one group's hand-off latency with the other group's MFMAs.
static int pick_mt(int B) { (void)B; return 1; }  // MT=2 measured 2x slower per sweep: the exposed latencies are the workgroup's own load/drain, not the peers'

static int check_grid(int B, int H, int T, void* sync_ws, const char* what) {
  const int mt = pick_mt(B);
  const int gx = H / 16, gy = ((B + 15) / 16 + mt - 1) / mt;
  if (!sync_ws) { set_error("%s: sync_ws is required in persistent mode", what); return 2; }
  if (gy > 64) { set_error("%s: B=%d exceeds 1024 rows in persistent mode", what, B); return 2; }
  const int cus = cu_count();
  if (gx * gy > cus) {
    set_error("%s: persistent sweep needs %d co-resident workgroups but the device has %d CUs (use mode 0)", what,
              gx * gy, cus);
    return 4;
  }
  (void)T;
  return 0;
}

int gru_persistent_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                       float* reserve, int T, int B, int H, void* sync_ws, hipStream_t s) {
  int rc = check_grid(B, H, T, sync_ws, "gru_layer_fwd");
  if (rc) return rc;
  const int mt = pick_mt(B);
  dim3 grid(H / 16, ((B + 15) / 16 + mt - 1) / mt), block(256);
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
  rc = check_hip(hipMemsetAsync(sync, 0, ((size_t)grid.y * mt * T + 16) * sizeof(unsigned), s), "gru_layer_fwd: memset");
  if (rc) return rc;
#define B2T_LAUNCH_FWD(NCH)                                                                                            \
  do {                                                                                                                 \
    if (mt == 2)                                                                                                       \
      hipLaunchKernelGGL((gru_persist_fwd_kernel<NCH, 2>), grid, block, 0, s, gi, w_hh, b_hh, h_init, out, reserve, T, \
                         B, H, sync);                                                                                  \
    else                                                                                                               \
      hipLaunchKernelGGL((gru_persist_fwd_kernel<NCH, 1>), grid, block, 0, s, gi, w_hh, b_hh, h_init, out, reserve, T, \
                         B, H, sync);                                                                                  \
  } while (0)
  if (H <= 128) B2T_LAUNCH_FWD(2);
  else if (H <= 256) B2T_LAUNCH_FWD(4);
  else if (H <= 512) B2T_LAUNCH_FWD(8);
  else if (H <= 768) B2T_LAUNCH_FWD(12);
  else if (H <= 1024) B2T_LAUNCH_FWD(16);
  else { set_error("gru_layer_fwd: H=%d > 1024 unsupported in persistent mode", H); return 2; }
#undef B2T_LAUNCH_FWD
  return check_hip(hipGetLastError(), "gru_layer_fwd (persistent)");
}

int gru_persistent_bwd(const float* dY, const float* dh_last, const float* reserve, const float* out,
                       const float* h_init, const float* w_hh_t, float* dG, float* dh_init, int T, int B, int H,
                       void* sync_ws, hipStream_t s) {
  int rc = check_grid(B, H, T, sync_ws, "gru_layer_bwd");
  if (rc) return rc;
  const int mt = pick_mt(B);
  dim3 grid(H / 16, ((B + 15) / 16 + mt - 1) / mt), block(256);
  unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
  rc = check_hip(hipMemsetAsync(sync, 0, ((size_t)grid.y * mt * T + 16) * sizeof(unsigned), s), "gru_layer_bwd: memset");
  if (rc) return rc;
#define B2T_LAUNCH_BWD(NCB)                                                                                           \
  do {                                                                                                                \
    if (mt == 2)                                                                                                      \
      hipLaunchKernelGGL((gru_persist_bwd_kernel<NCB, 2>), grid, block, 0, s, dY, dh_last, reserve, out, h_init,      \
                         w_hh_t, dG, dh_init, T, B, H, sync);                                                         \
    else                                                                                                              \
      hipLaunchKernelGGL((gru_persist_bwd_kernel<NCB, 1>), grid, block, 0, s, dY, dh_last, reserve, out, h_init,      \
                         w_hh_t, dG, dh_init, T, B, H, sync);                                                         \
  } while (0)
  if (H <= 128) B2T_LAUNCH_BWD(6);
  else if (H <= 256) B2T_LAUNCH_BWD(12);
  else if (H <= 512) B2T_LAUNCH_BWD(24);
  else if (H <= 768) B2T_LAUNCH_BWD(36);
  else if (H <= 1024) B2T_LAUNCH_BWD(48);
  else { set_error("gru_layer_bwd: H=%d > 1024 unsupported in persistent mode", H); return 2; }
#undef B2T_LAUNCH_BWD
  return check_hip(hipGetLastError(), "gru_layer_bwd (persistent)");
}

}  // namespace b2t

// Error word of the last persistent sweep that used sync_ws (0 = clean, 1 = a bounded spin gave up).
extern "C" int b2t_gru_sync_status(const void* sync_ws, int T, int B, int* status_host, void* stream) {
  using namespace b2t;
  B2T_REQUIRE(sync_ws && status_host, "gru_sync_status: null argument");
  const int mt = b2t::pick_mt(B);
  const size_t off = (size_t)(((B + 15) / 16 + mt - 1) / mt) * mt * T;
  int rc = check_hip(hipMemcpyAsync(status_host, reinterpret_cast<const unsigned*>(sync_ws) + off, sizeof(int),
                                    hipMemcpyDeviceToHost, as_stream(stream)), "gru_sync_status: copy");
  if (rc) return rc;
  return check_hip(hipStreamSynchronize(as_stream(stream)), "gru_sync_status: sync");
}
