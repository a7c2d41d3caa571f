// Probe: fp32-accurate GEMM on the bf16 matrix cores by operand splitting (C = A * B^T, both K-contiguous).
//   v = hi + mid + lo  (three bf16, 24 significant bits);  a*b ~= hh + (hm + mh) + (hl + lh + mm)   [6 MFMAs, "x6"]
//   or hh + hm + mh                                                                                  [3 MFMAs, "x3"]
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so x6 is worth up to 2.7x and x3 up to 5.3x.
// Checks accuracy against an fp64 host reference on a sub-block and reports TFLOP/s (fp32-equivalent FLOPs 2MNK).
//   hipcc --offload-arch=gfx950 -O3 -o sb16_gemm.bin sb16_gemm.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float v) {  // round to nearest even
  unsigned u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

__global__ void split_k(const float* __restrict__ x, u16* __restrict__ hi, u16* __restrict__ mid, u16* __restrict__ lo, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const u16 h = f2bf(v);
  const float r1 = v - bf2f(h);
  const u16 m = f2bf(r1);
  const float r2 = r1 - bf2f(m);
  hi[i] = h; mid[i] = m; lo[i] = f2bf(r2);
}

// NP = number of products: 6 or 3 (or 1 = plain bf16)
template <int BM, int BN, int NP, int SB = 0, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void sb16_gemm_k(const u16* Ah, const u16* Am, const u16* Al, const u16* Bh, const u16* Bm,
                                                     const u16* Bl, float* C, int M, int N, int K) {
  constexpr int BK = 32;            // bf16 elements per K tile = 64 bytes per row
  constexpr int PITCH = 40;         // u16 per LDS row (80 bytes): conflict-free ds_read_b128
  constexpr int NPL = NP == 1 ? 1 : (NP == 3 ? 2 : 3);   // planes needed per operand
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int AJ = BM / 64, BJ = BN / 64;  // 16-byte loads per thread per plane (256 threads x 8 bf16 = 64 rows x 32 k)
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  // layout: [stage][operand A planes..., operand B planes...]
  constexpr int A_PL = BM * PITCH, B_PL = BN * PITCH, STAGE = NPL * (A_PL + B_PL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / BN;
  const unsigned tot = gridDim.x, lin = blockIdx.x, xper = tot >> 3, xrem = tot & 7, xcd = lin & 7;
  const unsigned lg = xcd * xper + (xcd < xrem ? xcd : xrem) + (lin >> 3);
  const int tile_m = lg / tiles_n, tile_n = lg - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const u16* Ap[3] = {Ah, Am, Al};
  const u16* Bp[3] = {Bh, Bm, Bl};
  const int q = tid & 3, row = tid >> 2;  // 4 x 16-byte quads per 64-byte row, 64 rows per pass
  f32x4 ra[NPL][AJ], rb[NPL][BJ];
  const int nkt = K / BK;
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int j = 0; j < AJ; ++j)
        ra[p][j] = *reinterpret_cast<const f32x4*>(Ap[p] + (size_t)(m0 + row + 64 * j) * K + kt * BK + q * 8);
#pragma unroll
      for (int j = 0; j < BJ; ++j)
        rb[p][j] = *reinterpret_cast<const f32x4*>(Bp[p] + (size_t)(n0 + row + 64 * j) * K + kt * BK + q * 8);
    }
  };
  auto lstore = [&](int st) {
    u16* s = smem + st * STAGE;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4*>(s + p * A_PL + (row + 64 * j) * PITCH + q * 8) = ra[p][j];
#pragma unroll
      for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(s + NPL * A_PL + p * B_PL + (row + 64 * j) * PITCH + q * 8) = rb[p][j];
    }
  };
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gload(0); lstore(0);
  if (nkt > 1) gload(1);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const u16* s = smem + cur * STAGE;
    if (!SB) {
      if (kt + 1 < nkt) lstore(cur ^ 1);
      if (kt + 2 < nkt) gload(kt + 2);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[NPL][TM], fb[NPL][TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[p][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(s + p * A_PL + (wm * WTM + i * 32 + li) * PITCH + ks * 16 + lh * 8));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[p][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(s + NPL * A_PL + p * B_PL + (wn * WTN + j * 32 + li) * PITCH + ks * 16 + lh * 8));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first
          if (NP == 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
          }
          if (NP >= 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (SB) {  // single LDS stage: tile kt + 1 waits in registers during the MFMAs, then replaces the stage
      if (kt + 1 < nkt) lstore(0);
      if (kt + 2 < nkt) gload(kt + 2);
      __syncthreads();
    } else {
      cur ^= 1;
    }
  }
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) {
    const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
    C[(size_t)(m0 + wm * WTM + i * 32 + rr) * N + n0 + wn * WTN + j * 32 + li] = acc[i][j][r];
  }
}

template <int BM, int BN, int NP, int SB = 0, int OCC = 1>
void run(const char* name, u16** A, u16** B, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
  if (M % BM || N % BN) return;
  constexpr int NPL = NP == 1 ? 1 : (NP == 3 ? 2 : 3);
  const size_t lds = (size_t)(SB ? 1 : 2) * NPL * (BM + BN) * 40 * 2;
  auto k = sb16_gemm_k<BM, BN, NP, SB, OCC>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("%s attr fail (lds %zu)\n", name, lds); return; }
  dim3 grid((M / BM) * (N / BN));
  const int w = (int)(80e-3 / (2.0 * M * N * K / 300e12)) + 2;
  for (int r = 0; r < w; ++r) hipLaunchKernelGGL(k, grid, dim3(256), lds, 0, A[0], A[1], A[2], B[0], B[1], B[2], C, M, N, K);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, grid, dim3(256), lds, 0, A[0], A[1], A[2], B[0], B[1], B[2], C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // accuracy on a 48 x 40 corner vs fp64
  std::vector<float> hC((size_t)64 * N);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxrel = 0, scale = 0;
  for (int i = 0; i < 48; ++i) for (int j = 0; j < 40; ++j) {
    double s = 0, sa = 0;
    for (int kk = 0; kk < K; ++kk) { const double t = (double)hA[(size_t)i * K + kk] * hB[(size_t)j * K + kk]; s += t; sa += fabs(t); }
    const double e = fabs(hC[(size_t)i * N + j] - s) / sa;   // error relative to the absolute mass of the dot product
    if (e > maxrel) maxrel = e;
    scale = sa;
  }
  printf("  %-22s grid %5d lds %6zu  %7.1f us  %7.1f TF/s (fp32-equivalent)   max |err| / sum|terms| = %.2e\n", name, grid.x, lds,
         ms / reps * 1e3, 2.0 * M * N * K * reps / (ms * 1e-3) / 1e12, maxrel);
  (void)scale;
}

int main() {
  const int shapes[][3] = {{24576, 256, 2304}, {6144, 512, 4608}, {98304, 128, 1152}, {4096, 4096, 4096}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.3f;
    for (auto& v : hB) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *C;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    u16 *A[3], *B[3];
    for (int p = 0; p < 3; ++p) { hipMalloc(&A[p], hA.size() * 2); hipMalloc(&B[p], hB.size() * 2); }
    hipLaunchKernelGGL(split_k, dim3((hA.size() + 255) / 256), dim3(256), 0, 0, dA, A[0], A[1], A[2], hA.size());
    hipLaunchKernelGGL(split_k, dim3((hB.size() + 255) / 256), dim3(256), 0, 0, dB, B[0], B[1], B[2], hB.size());
    hipDeviceSynchronize();
    printf("M=%d N=%d K=%d (%.1f GF)\n", M, N, K, 2.0 * M * N * K / 1e9);
    run<128, 128, 3>("x3 128x128 dbuf occ1", A, B, C, M, N, K, hA, hB);
    run<128, 128, 3, 1, 2>("x3 128x128 sbuf occ2", A, B, C, M, N, K, hA, hB);
    run<128, 128, 3, 1, 3>("x3 128x128 sbuf occ3", A, B, C, M, N, K, hA, hB);
    run<128, 256, 3>("x3 128x256 dbuf occ1", A, B, C, M, N, K, hA, hB);
    run<256, 128, 3>("x3 256x128 dbuf occ1", A, B, C, M, N, K, hA, hB);
    run<128, 256, 3, 1, 1>("x3 128x256 sbuf occ1", A, B, C, M, N, K, hA, hB);
    run<128, 256, 3, 1, 2>("x3 128x256 sbuf occ2", A, B, C, M, N, K, hA, hB);
    run<64, 128, 3, 1, 3>("x3 64x128 sbuf occ3", A, B, C, M, N, K, hA, hB);
    run<64, 128, 3, 0, 2>("x3 64x128 dbuf occ2", A, B, C, M, N, K, hA, hB);
    run<128, 128, 6, 1, 2>("x6 128x128 sbuf occ2", A, B, C, M, N, K, hA, hB);
    run<128, 256, 6, 1, 1>("x6 128x256 sbuf occ1", A, B, C, M, N, K, hA, hB);
    hipFree(dA); hipFree(dB); hipFree(C);
    for (int p = 0; p < 3; ++p) { hipFree(A[p]); hipFree(B[p]); }
  }
  return 0;
}
