// gru_granule.hip — persistent GRU sweeps with data-tagged hand-off ("the data is the flag").
//
// Same workgroup decomposition and register-resident W_hh slices as gru_persistent.hip, but the
// inter-workgroup exchange of the recurrent vector uses 8-byte {value, tag} granules
// (cdna_hip_programming.md §6 Guideline 16, R2): a producer writes each value together with the tag of its
// time step in ONE aligned write-through store; a consumer lane re-reads the granules of its own MFMA
// A-operand with sc1 (L1-bypassing) loads until every tag matches.  There is no counter, no drain, no
// barrier and no second round trip between "data written" and "data consumed": the hand-off costs one store
// propagation plus one load round trip (~1000 cycles measured, tools/ubench/xcd_handoff.hip) instead of the
// flag protocol's store-drain + atomic + poll + payload load (~5000 cycles in gru_persistent.hip).
// Tags are unique per (call, step): tag = epoch + t + 1 with a per-call epoch, so stale granules of earlier
// calls can never match and the granule buffer needs no clearing.  Spins are bounded and report through an
// error word; correctness does not depend on workgroup placement.
#include <atomic>
#include "gru_cell.h"

namespace b2t {

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned G_SPIN_LIMIT = 1u << 20;
constexpr int GTP = 20;   // LDS pitch of the staged 16x16 tile
constexpr int GCTRL = 64; // control words in front of the granule area (word 0: error flag)

__device__ __forceinline__ u32x4 gload_sc1(const void* base_uniform, unsigned byte_off) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_uniform), 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, /*sc1*/ 16);
}
__device__ __forceinline__ void gstore_sc1(void* base_uniform, unsigned byte_off, u32x4 v) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, byte_off, 0, /*sc1*/ 16);
}

// Spin until the NCH x 4 granules of this lane's A-operand carry `tag`; returns them as floats.
// gbase: wave-uniform pointer to the granule block of the producing step ([B][ld] granules of 8 bytes).
template <int NCH>
__device__ __forceinline__ void spin_load_operand(const void* gbase, int arow, int B, int ld, int nch, int wave, int q,
                                                  unsigned tag, unsigned* err, float4 (&a)[NCH]) {
  const bool row_ok = arow < B;
  // Each chunk comes from a different producer workgroup: keep what has arrived, re-read only what has not.
  u32x4 g0[NCH], g1[NCH];
  unsigned pending = 0;   // wave-uniform bit mask of chunks still missing
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    g0[ci] = u32x4{0u, tag, 0u, tag};
    g1[ci] = u32x4{0u, tag, 0u, tag};
    if (wave + 4 * ci < nch) pending |= 1u << ci;
  }
  for (unsigned spins = 0; pending; ++spins) {
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      if ((pending >> ci) & 1u) {
        const int c = wave + 4 * ci;
        if (row_ok) {
          const unsigned off = (unsigned)(((long long)arow * ld + c * 16 + 4 * q) * 8);
          g0[ci] = gload_sc1(gbase, off);
          g1[ci] = gload_sc1(gbase, off + 16);
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      if ((pending >> ci) & 1u) {
        const bool ok = !row_ok || (g0[ci].y == tag && g0[ci].w == tag && g1[ci].y == tag && g1[ci].w == tag);
        if (__all(ok)) pending &= ~(1u << ci);
      }
    }
    if (pending && spins > G_SPIN_LIMIT) {
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, RLX_AGENT);
      break;
    }
    if (pending && (spins & 63u) == 63u && __hip_atomic_load(err, RLX_AGENT) != 0u) break;  // another wave gave up
  }
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci)
    a[ci] = make_float4(__uint_as_float(g0[ci].x), __uint_as_float(g0[ci].z), __uint_as_float(g1[ci].x),
                        __uint_as_float(g1[ci].z));
}

// Wave 0 publishes the staged 16x16 tile as 256 granules (2 x 16-byte write-through stores per lane).
__device__ __forceinline__ void publish_tile(void* gbase, const float* tile, int m0, int B, int ld, int j0, int lane,
                                             unsigned tag) {
  const int r = lane >> 2, c4 = (lane & 3) * 4;
  if (m0 + r < B) {
    const float4 v = *reinterpret_cast<const float4*>(&tile[r * GTP + c4]);
    const unsigned off = (unsigned)(((long long)(m0 + r) * ld + j0 + c4) * 8);
    gstore_sc1(gbase, off, u32x4{__float_as_uint(v.x), tag, __float_as_uint(v.y), tag});
    gstore_sc1(gbase, off + 16, u32x4{__float_as_uint(v.z), tag, __float_as_uint(v.w), tag});
  }
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256, 1) void gru_granule_fwd_kernel(const float* __restrict__ gi,
                                                                 const float* __restrict__ w_hh,
                                                                 const float* __restrict__ b_hh,
                                                                 const float* __restrict__ h_init,
                                                                 float* __restrict__ out, float* __restrict__ reserve,
                                                                 int T, int B, int H, unsigned* ctrl, unsigned epoch) {
  __shared__ __attribute__((aligned(16))) float red[4 * 3 * 4 * 64 + 16 * GTP];
  float* hs = red + 4 * 3 * 4 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int j = lane & 15, q = lane >> 4;
  const int unit = j0 + j, row = m0 + 4 * q + wave, arow = m0 + j;
  const int nch = H / 16;
  unsigned* err = ctrl;
  unsigned char* gran = reinterpret_cast<unsigned char*>(ctrl + GCTRL);   // [T][B][H] granules
  const long long gstep = (long long)B * H * 8;

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
  const bool live = row < B;
  float hp = live ? h_init[(long long)row * H + unit] : 0.f;

  for (int t = 0; t < T; ++t) {
    float gir = 0.f, giz = 0.f, gin = 0.f;
    if (live) {
      const float* g3 = gi + ((long long)t * B + row) * 3 * H + unit;
      gir = g3[0]; giz = g3[H]; gin = g3[2 * H];
    }
    float4 a[NCH];
    if (t == 0) {   // initial state: written by an earlier launch, plain loads
#pragma unroll
      for (int ci = 0; ci < NCH; ++ci) {
        const int c = wave + 4 * ci;
        a[ci] = (c < nch && arow < B) ? *reinterpret_cast<const float4*>(h_init + (long long)arow * H + c * 16 + 4 * q)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      spin_load_operand<NCH>(gran + (long long)(t - 1) * gstep, arow, B, H, nch, wave, q, epoch + (unsigned)t, err, a);
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
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
      hs[(4 * q + wave) * GTP + j] = h;
      out[((long long)t * B + row) * H + unit] = h;          // plain copy for the GEMMs / backward
      if (reserve) {
        float* rs = reserve + ((long long)t * B + row) * 4 * H + unit;
        rs[0] = r; rs[H] = z; rs[2 * H] = n; rs[3 * H] = ghn;
      }
      hp = h;
    }
    __syncthreads();   // tile staged; also fences `red` for the next step
    if (wave == 0 && t + 1 < T) publish_tile(gran + (long long)t * gstep, hs, m0, B, H, j0, lane, epoch + (unsigned)t + 1u);
  }
}

static std::atomic<unsigned> g_epoch{1};

size_t gru_granule_bytes(int T, int B, int H) { return (size_t)GCTRL * 4 + (size_t)T * B * H * 8; }

int gru_granule_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                    float* reserve, int T, int B, int H, void* ws, hipStream_t s) {
  if (!ws) { set_error("gru_layer_fwd: workspace required in granule mode"); return 2; }
  if (T > 8000) { set_error("gru_layer_fwd: T=%d exceeds 8000 steps per call in granule mode", T); return 2; }
  const int gx = H / 16, gy = (B + 15) / 16;
  int dev = 0; hipDeviceProp_t p;
  static int cus = -1;
  if (cus < 0) { cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 0; }
  if (gx * gy > cus) {
    set_error("gru_layer_fwd: granule sweep needs %d co-resident workgroups but the device has %d CUs (use mode 0)", gx * gy, cus);
    return 4;
  }
  unsigned* ctrl = reinterpret_cast<unsigned*>(ws);
  int rc = check_hip(hipMemsetAsync(ctrl, 0, GCTRL * sizeof(unsigned), s), "gru_layer_fwd: memset");
  if (rc) return rc;
  const unsigned epoch = (g_epoch.fetch_add(1) & 0x7ffffu) << 13;   // 8192 tags per call
  dim3 grid(gx, gy), block(256);
#define B2T_LAUNCH_GF(NCH) \
  hipLaunchKernelGGL((gru_granule_fwd_kernel<NCH>), grid, block, 0, s, gi, w_hh, b_hh, h_init, out, reserve, T, B, H, ctrl, epoch)
  if (H <= 128) B2T_LAUNCH_GF(2);
  else if (H <= 256) B2T_LAUNCH_GF(4);
  else if (H <= 512) B2T_LAUNCH_GF(8);
  else if (H <= 768) B2T_LAUNCH_GF(12);
  else if (H <= 1024) B2T_LAUNCH_GF(16);
  else { set_error("gru_layer_fwd: H=%d > 1024 unsupported in granule mode", H); return 2; }
#undef B2T_LAUNCH_GF
  return check_hip(hipGetLastError(), "gru_layer_fwd (granule)");
}

}  // namespace b2t
