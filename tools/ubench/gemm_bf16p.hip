// gemm_bf16p.hip — feasibility probe: C[M][N] (fp32) = A[M][K] * B[N][K]^T with bf16 operands ALREADY PACKED dense and
// k-contiguous in memory (what a pack pass in front of the amp-mode GEMMs would produce).  128x128 block tile, BK = 64,
// 256 threads = 2x2 waves x (2x2) 32x32x16 MFMAs, 16-byte global loads straight into LDS rows of 144 bytes (64 bf16 + pad),
// register prefetch of the next k-tile, double-buffered LDS, one barrier per k-tile.
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_bf16p gemm_bf16p.hip ; run: ./gemm_bf16p [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
constexpr int BM = 128, BN = 128, BK = 64, PITCH = BK + 8;   // bf16 elements per LDS row (144 B)
constexpr int NLD = BM * BK / 8 / 256;                       // 16-byte loads per thread per operand tile (= 4)

#define FETCH(k0)                                                                                      \
  ra0 = *reinterpret_cast<const uint4*>(ag + (k0)); ra1 = *reinterpret_cast<const uint4*>(ag + 32ll * K + (k0)); \
  ra2 = *reinterpret_cast<const uint4*>(ag + 64ll * K + (k0)); ra3 = *reinterpret_cast<const uint4*>(ag + 96ll * K + (k0)); \
  rb0 = *reinterpret_cast<const uint4*>(bg + (k0)); rb1 = *reinterpret_cast<const uint4*>(bg + 32ll * K + (k0)); \
  rb2 = *reinterpret_cast<const uint4*>(bg + 64ll * K + (k0)); rb3 = *reinterpret_cast<const uint4*>(bg + 96ll * K + (k0));
#define STASH(buf)                                                                                     \
  { __bf16* ad = As + (buf) * BM * PITCH + (tid >> 3) * PITCH + (tid & 7) * 8;                         \
    __bf16* bd = Bs + (buf) * BM * PITCH + (tid >> 3) * PITCH + (tid & 7) * 8;                         \
    *reinterpret_cast<uint4*>(ad) = ra0; *reinterpret_cast<uint4*>(ad + 32 * PITCH) = ra1;             \
    *reinterpret_cast<uint4*>(ad + 64 * PITCH) = ra2; *reinterpret_cast<uint4*>(ad + 96 * PITCH) = ra3; \
    *reinterpret_cast<uint4*>(bd) = rb0; *reinterpret_cast<uint4*>(bd + 32 * PITCH) = rb1;             \
    *reinterpret_cast<uint4*>(bd + 64 * PITCH) = rb2; *reinterpret_cast<uint4*>(bd + 96 * PITCH) = rb3; }

template <int WGS>
__global__ __launch_bounds__(256, WGS) void k_gemm(const __bf16* __restrict__ A, const __bf16* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 2 * BM * PITCH];
  __bf16* As = smem; __bf16* Bs = smem + 2 * BM * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int gx = N / BN;
  int m0, n0;
  {
    const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    m0 = (tile / gx) * BM; n0 = (tile % gx) * BN;
  }
  const __bf16* ag = A + (long long)(m0 + (tid >> 3)) * K + (tid & 7) * 8;
  const __bf16* bg = B + (long long)(n0 + (tid >> 3)) * K + (tid & 7) * 8;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  FETCH(0) STASH(0)
  __syncthreads();
  const int lk = lane >> 5, li = lane & 31, nk = K / BK;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int knext = kt + 1 < nk ? (kt + 1) * BK : kt * BK;   // the last iteration re-reads its own tile: no branch around the loads
    FETCH(knext)
    const __bf16* a0p = As + cur * BM * PITCH + (wm * 64 + li) * PITCH + 8 * lk;
    const __bf16* a1p = a0p + 32 * PITCH;
    const __bf16* b0p = Bs + cur * BM * PITCH + (wn * 64 + li) * PITCH + 8 * lk;
    const __bf16* b1p = b0p + 32 * PITCH;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(a0p + kk), a1 = *reinterpret_cast<const bf16x8*>(a1p + kk);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b0p + kk), b1 = *reinterpret_cast<const bf16x8*>(b1p + kk);
      acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc11, 0, 0, 0);
    }
    STASH(cur ^ 1)
    __syncthreads();
    cur ^= 1;
  }
  f32x16 acc[2][2] = {{acc00, acc01}, {acc10, acc11}};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        C[(long long)row * N + col] = acc[i][j][e];
      }
    }
}

template <int WGS>
__global__ __launch_bounds__(256, WGS) void k_gemm2(const __bf16* __restrict__ A, const __bf16* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 2 * BM * PITCH];
  __bf16* As = smem; __bf16* Bs = smem + 2 * BM * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int gx = N / BN;
  int m0, n0;
  {
    const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    m0 = (tile / gx) * BM; n0 = (tile % gx) * BN;
  }
  // thread -> (row = tid/8 + 32 r, 16-byte piece tid%8) of a 128 x 64 tile
  const __bf16* ag = A + (long long)(m0 + (tid >> 3)) * K + (tid & 7) * 8;
  const __bf16* bg = B + (long long)(n0 + (tid >> 3)) * K + (tid & 7) * 8;
  // two register sets: the tile fetched in iteration kt is written to LDS in iteration kt + 1 (a whole iteration of latency hiding)
  uint4 ra[2][NLD], rb[2][NLD];
  auto fetch = [&](int set, int k0) {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      ra[set][r] = *reinterpret_cast<const uint4*>(ag + (long long)(32 * r) * K + k0);
      rb[set][r] = *reinterpret_cast<const uint4*>(bg + (long long)(32 * r) * K + k0);
    }
  };
  auto stash = [&](int set, int buf) {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      *reinterpret_cast<uint4*>(&As[buf * BM * PITCH + ((tid >> 3) + 32 * r) * PITCH + (tid & 7) * 8]) = ra[set][r];
      *reinterpret_cast<uint4*>(&Bs[buf * BM * PITCH + ((tid >> 3) + 32 * r) * PITCH + (tid & 7) * 8]) = rb[set][r];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int lk = lane >> 5, li = lane & 31, nk = K / BK;
  fetch(0, 0); stash(0, 0);
  if (nk > 1) fetch(1, BK);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; kt += 2) {
   // unrolled by two so that the register sets are addressed statically
#pragma unroll
   for (int h = 0; h < 2; ++h) {
    const int k_ = kt + h;
    if (k_ >= nk) break;
    const bool more = k_ + 1 < nk;
    if (k_ + 2 < nk) fetch(h, (k_ + 2) * BK);
    const __bf16* a0p = As + cur * BM * PITCH + (wm * 64 + li) * PITCH + 8 * lk;
    const __bf16* a1p = a0p + 32 * PITCH;
    const __bf16* b0p = Bs + cur * BM * PITCH + (wn * 64 + li) * PITCH + 8 * lk;
    const __bf16* b1p = b0p + 32 * PITCH;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(a0p + kk), a1 = *reinterpret_cast<const bf16x8*>(a1p + kk);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b0p + kk), b1 = *reinterpret_cast<const bf16x8*>(b1p + kk);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) stash(h ^ 1, cur ^ 1);
    __syncthreads();
    cur ^= 1;
   }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        C[(long long)row * N + col] = acc[i][j][e];
      }
    }
}

int main(int argc, char** argv) {
  const int shapes[][3] = {{4096, 4096, 4096}, {7808 + 128, 2304, 7168}, {2304, 7168, 7808 + 64 * 2}, {7936, 2304, 768}, {32000, 1536, 512}, {1536, 512, 32000}};
  for (auto& sh : shapes) {
    int M = sh[0], N = sh[1], K = sh[2];
    if (argc == 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    M = (M + 127) / 128 * 128; N = (N + 127) / 128 * 128; K = (K + 63) / 64 * 64;
    __bf16 *A, *B; float* C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<unsigned short> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f80 + (unsigned short)(rand() & 0x7f) - ((rand() & 1) << 15);   // ~ +-[1,2)
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    h.resize((size_t)N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)(rand() & 0x7f);
    hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs = 2; wgs <= 3; ++wgs) {
      auto launch = [&]() {
        dim3 grid((M / BM) * (N / BN));
        if (wgs == 2) hipLaunchKernelGGL((k_gemm<2>), grid, dim3(256), 0, 0, A, B, C, M, N, K);
        else hipLaunchKernelGGL((k_gemm<3>), grid, dim3(256), 0, 0, A, B, C, M, N, K);
      };
      launch(); launch(); hipDeviceSynchronize();
      hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      printf("M=%d N=%d K=%d occupancy target %d: %.1f us, %.1f TF/s", M, N, K, wgs, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
      auto launch2 = [&]() {
        dim3 grid((M / BM) * (N / BN));
        if (wgs == 2) hipLaunchKernelGGL((k_gemm2<2>), grid, dim3(256), 0, 0, A, B, C, M, N, K);
        else hipLaunchKernelGGL((k_gemm2<3>), grid, dim3(256), 0, 0, A, B, C, M, N, K);
      };
      launch2(); launch2(); hipDeviceSynchronize();
      hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch2(); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      printf("   | prefetch distance 2: %.1f us, %.1f TF/s\n", ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    }
    hipFree(A); hipFree(B); hipFree(C);
    if (argc == 4) break;
  }
  return 0;
}
