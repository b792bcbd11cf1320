// common.h — shared helpers for libb2t_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/b2t.h"

namespace b2t {

void set_error(const char* fmt, ...);

// Translate a HIP status into the library's error convention (returns non-zero on error).
int check_hip(hipError_t e, const char* what);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define B2T_CHECK_LAUNCH(what)                                   \
  do {                                                           \
    int _rc = ::b2t::check_hip(hipGetLastError(), what);         \
    if (_rc) return _rc;                                         \
  } while (0)

#define B2T_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) { ::b2t::set_error(__VA_ARGS__); return 2; }    \
  } while (0)

// ---- Philox4x32-10 counter-based RNG (Salmon et al. 2011) --------------------------------
struct Philox {
  static __device__ __forceinline__ uint4 run(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
      const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
      ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
      key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
  }
  // 4 standard normals from one counter (Box-Muller on two uniform pairs)
  static __device__ __forceinline__ float4 normal4(uint64_t seed, uint64_t idx, uint32_t stream) {
    uint4 c = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u);
    uint2 k = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    uint4 r = run(c, k);
    const float S = 2.3283064365386963e-10f;  // 2^-32
    float u0 = ((float)r.x + 0.5f) * S, u1 = ((float)r.y + 0.5f) * S;
    float u2 = ((float)r.z + 0.5f) * S, u3 = ((float)r.w + 0.5f) * S;
    u0 = fminf(fmaxf(u0, 1e-10f), 1.0f); u2 = fminf(fmaxf(u2, 1e-10f), 1.0f);
    float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    return make_float4(ra * c0, ra * s0, rb * c1, rb * s1);
  }
  static __device__ __forceinline__ float4 uniform4(uint64_t seed, uint64_t idx, uint32_t stream) {
    uint4 c = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u);
    uint2 k = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    uint4 r = run(c, k);
    const float S = 2.3283064365386963e-10f;
    return make_float4((float)r.x * S, (float)r.y * S, (float)r.z * S, (float)r.w * S);
  }
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

}  // namespace b2t
