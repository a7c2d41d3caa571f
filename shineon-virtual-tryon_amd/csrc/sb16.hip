// Split-bf16 3x3 convolution for the FROZEN VGG19 chain of the perceptual loss (opt-in, non-headline):
// fp32 operands are split v = hi + mid (+ lo), two bf16 planes each, and the product is formed on the bf16 matrix cores as
//     a * b ~= a_hi b_hi + a_hi b_mid + a_mid b_hi          (3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// The bf16 MFMA runs at 16x the rate of v_mfma_f32_32x32x2_f32, so three of them are worth up to 5.3x the exact-fp32 path.
// Dropped terms are <= 2^-16 |a||b| each and of random sign: measured (tools/probes/sb16_gemm.hip, K = 2304 ... 4608)
// max |error| / sum|terms| = 3.2e-7 ... 9.4e-7, against 1.2e-7 ... 1.7e-7 for six products (and for the fp32 MFMA chain)
// and 2e-4 for plain bf16.  Reference ops replaced: the torchvision VGG19 conv3x3 + ReLU stack behind
// models/networks/vgg.py:6-36 / loss.py:106-122 (weights frozen there, vgg.py:25-27 - which is why they can be split once).
//
// so_sb16_conv3x3: y[pix][ko] = act(sum_{r,s,c} x[pix@(r,s)][c] * w[ko][r][s][c] + bias[ko]), stride 1, pad 1, NHWC, C % 32 == 0,
// Ko % 64 == 0; x and w arrive as (hi, mid) bf16 planes; y is written in fp32 and, optionally, already split into planes for
// the next convolution.  The input gradient of such a layer is the same kernel on flipped + transposed weight planes
// (so_sb16_prep_weights(..., transpose = 1)) with the ReLU gate of the layer below applied in the epilogue.
#include "common.h"
#include "../../include/shineon_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int so_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float v) {  // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

__device__ __forceinline__ f32x4 bload16(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  const so_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
  f32x4 r;
  r[0] = __int_as_float(v[0]); r[1] = __int_as_float(v[1]); r[2] = __int_as_float(v[2]); r[3] = __int_as_float(v[3]);
  return r;
}

#define SB_OOB 0x80000000u

// x [rows][ld] fp32 (first C columns) -> hi, mid planes [rows][C] bf16
__global__ __launch_bounds__(256) void split_k(const float* __restrict__ x, int ld, int C, u16* __restrict__ hi,
                                               u16* __restrict__ mid, long long total4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // quad index over rows * C / 4
  if (i >= total4) return;
  const int cq = C / 4;
  const long long row = i / cq;
  const int c = (int)(i - row * cq) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ld + c);
  u16 h[4], m[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    h[u] = f2bf(v[u]);
    m[u] = f2bf(v[u] - bf2f(h[u]));
  }
  *reinterpret_cast<uint2*>(hi + row * C + c) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
  *reinterpret_cast<uint2*>(mid + row * C + c) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
}

// w OHWI fp32 [Ko][9][Cw] (Cw >= C: padded input channels) -> planes.  transpose = 0: [Ko][9][C];
// transpose = 1 (input-gradient weights): out[c][r'][s'][ko] = w[ko][2 - r'][2 - s'][c]   ([C][9][Ko])
__global__ __launch_bounds__(256) void prep_w_k(const float* __restrict__ w, int Ko, int C, int Cw, int transpose,
                                                u16* __restrict__ hi, u16* __restrict__ mid, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v;
  if (!transpose) {
    const int c = (int)(i % C);
    const long long t = i / C;            // ko * 9 + tap
    v = w[t * Cw + c];
  } else {
    const int ko = (int)(i % Ko);
    const long long t = i / Ko;           // c * 9 + tap'
    const int tap = (int)(t % 9), c = (int)(t / 9);
    v = w[((long long)ko * 9 + (8 - tap)) * Cw + c];
  }
  const u16 h = f2bf(v);
  hi[i] = h;
  mid[i] = f2bf(v - bf2f(h));
}

struct SbConv {
  const u16 *xh, *xm, *wh, *wm;
  const float* bias;
  const float* gate;   // optional ReLU gate [rows][Ko] (fp32): out = gate > 0 ? out : 0
  float* y;            // [rows][ldy]
  u16 *yh, *ym;        // optional output planes [rows][Ko]
  int Nb, H, W, C, Ko, ldy, relu;
  unsigned x_bytes, w_bytes;
};

template <int BM, int BN>
__global__ __launch_bounds__(256, (BM * BN <= 64 * 128) ? 2 : 1) void sb16_conv_k(const SbConv p) {
  constexpr int BK = 32, PITCH = 40, NPL = 2;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int AJ = BM / 64, BJ = BN / 64;
  static_assert(AJ >= 1 && BJ >= 1 && TM >= 1 && TN >= 1, "tiles are multiples of 64");
  constexpr int A_PL = BM * PITCH, B_PL = BN * PITCH, STAGE = NPL * (A_PL + B_PL);
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.Nb * p.H * p.W, K = 9 * p.C;
  const int tiles_n = p.Ko / BN;
  const unsigned tot = gridDim.x, lin = blockIdx.x, xper = tot >> 3, xrem = tot & 7, xcd = lin & 7;
  const unsigned lg = xcd * xper + (xcd < xrem ? xcd : xrem) + (lin >> 3);
  const int tile_m = lg / tiles_n, tile_n = lg - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const __amdgpu_buffer_rsrc_t rXh = __builtin_amdgcn_make_buffer_rsrc((void*)p.xh, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXm = __builtin_amdgcn_make_buffer_rsrc((void*)p.xm, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rWh = __builtin_amdgcn_make_buffer_rsrc((void*)p.wh, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rWm = __builtin_amdgcn_make_buffer_rsrc((void*)p.wm, 0, (int)p.w_bytes, 0x00020000);
  const int q = tid & 3, row = tid >> 2;  // 4 x 16-byte quads (8 bf16) per 64-byte k row, 64 rows per pass
  int a_org[AJ], a_h0[AJ], a_w0[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + row + 64 * j;
    a_org[j] = 0; a_h0[j] = -(1 << 28); a_w0[j] = 0;
    if (m < M) {
      const int hw = p.H * p.W;
      const int n = m / hw, rem = m - n * hw;
      const int ho = rem / p.W, wo = rem - ho * p.W;
      a_h0[j] = ho - 1; a_w0[j] = wo - 1;
      a_org[j] = ((n * p.H + ho - 1) * p.W + wo - 1) * p.C;   // may be negative; only used when in range
    }
  }
  int b_row[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) b_row[j] = (n0 + row + 64 * j) * K;
  const int nkt = K / BK;
  int u_r = 0, u_s = 0, u_c0 = 0;   // block-uniform (tap row, tap column, first channel) of the next tile to load
  f32x4 ra[NPL][AJ], rb[NPL][BJ];
  auto gload = [&](int kt) {
    const bool live = kt < nkt;
    const int r = u_r, s = u_s, c = u_c0 + q * 8;
    u_c0 += BK;
    if (u_c0 >= p.C) { u_c0 = 0; u_s += 1; if (u_s == 3) { u_s = 0; u_r += 1; } }
    const int tap_off = (r * p.W + s) * p.C + c;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int hi = a_h0[j] + r, wi = a_w0[j] + s;
      const bool ok = live & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
      const unsigned off = ok ? (unsigned)(a_org[j] + tap_off) * 2u : SB_OOB;
      ra[0][j] = bload16(rXh, off);
      ra[1][j] = bload16(rXm, off);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const unsigned off = live ? (unsigned)(b_row[j] + kt * BK + q * 8) * 2u : SB_OOB;
      rb[0][j] = bload16(rWh, off);
      rb[1][j] = bload16(rWm, off);
    }
  };
  auto lstore = [&](int st) {
    u16* s = smem + st * STAGE;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4*>(s + pl * A_PL + (row + 64 * j) * PITCH + q * 8) = ra[pl][j];
#pragma unroll
      for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(s + NPL * A_PL + pl * B_PL + (row + 64 * j) * PITCH + q * 8) = rb[pl][j];
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gload(0);
  lstore(0);
  gload(1);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const u16* s = smem + cur * STAGE;
    lstore(cur ^ 1);      // tile kt + 1 (zeros past the end)
    gload(kt + 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[NPL][TM], fb[NPL][TN];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(s + pl * A_PL + (wm * WTM + i * 32 + li) * PITCH + ks * 16 + lh * 8));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(s + NPL * A_PL + pl * B_PL + (wn * WTN + j * 32 + li) * PITCH + ks * 16 + lh * 8));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);   // mid * hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);   // hi * mid
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);   // hi * hi
        }
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + li;
      const float bv = p.bias ? p.bias[n] : 0.f;
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          float v = acc[i][j][r] + bv;
          if (p.relu) v = v > 0.f ? v : 0.f;
          if (p.gate && !(p.gate[(size_t)m * p.Ko + n] > 0.f)) v = 0.f;
          p.y[(size_t)m * p.ldy + n] = v;
          if (p.yh) {
            const u16 h = f2bf(v);
            p.yh[(size_t)m * p.Ko + n] = h;
            p.ym[(size_t)m * p.Ko + n] = f2bf(v - bf2f(h));
          }
        }
      }
    }
}

template <int BM, int BN>
int launch(const SbConv& p, hipStream_t st) {
  const size_t lds = (size_t)2 * 2 * (BM + BN) * 40 * sizeof(u16);
  auto k = sb16_conv_k<BM, BN>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int M = p.Nb * p.H * p.W;
  dim3 grid((unsigned)(so_cdiv(M, BM) * (p.Ko / BN)));
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return SO_LAUNCH_CHECK();
}

}  // namespace

extern "C" {

int so_sb16_split(const float* x, int ldx, int C, void* hi, void* mid, long long rows, void* stream) {
  if (rows <= 0) return 0;
  if ((C & 3) || (ldx & 3) || (((uintptr_t)x) & 15)) return SO_ERR_ALIGN;
  const long long total4 = rows * (C / 4);
  hipLaunchKernelGGL(split_k, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, C, (u16*)hi,
                     (u16*)mid, total4);
  return SO_LAUNCH_CHECK();
}

int so_sb16_prep_weights(const float* w_ohwi, int Ko, int C, int Cw, int transpose, void* hi, void* mid, void* stream) {
  const long long total = (long long)Ko * 9 * C;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(prep_w_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, Ko, C, Cw,
                     transpose, (u16*)hi, (u16*)mid, total);
  return SO_LAUNCH_CHECK();
}

int so_sb16_conv3x3(const void* xh, const void* xm, const void* wh, const void* wm, const float* bias, const float* gate,
                    float* y, int ldy, void* yh, void* ym, int Nb, int H, int W, int C, int Ko, int relu, void* stream) {
  if ((C % 32) || (Ko % 64)) return SO_ERR_SHAPE;
  const long long xb = (long long)Nb * H * W * C * 2, wb = (long long)Ko * 9 * C * 2;
  if (xb >= 0x7FFFFFF0LL || wb >= 0x7FFFFFF0LL) return SO_ERR_SHAPE;
  SbConv p;
  p.xh = (const u16*)xh; p.xm = (const u16*)xm; p.wh = (const u16*)wh; p.wm = (const u16*)wm;
  p.bias = bias; p.gate = gate; p.y = y; p.yh = (u16*)yh; p.ym = (u16*)ym;
  p.Nb = Nb; p.H = H; p.W = W; p.C = C; p.Ko = Ko; p.ldy = ldy; p.relu = relu;
  p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  // largest tile that still gives every CU a block (256 CUs); below that the 64x64 tile (two blocks per CU)
  const long long M = (long long)Nb * H * W;
  auto tiles = [&](int bm, int bn) { return (long long)so_cdiv(M, bm) * (Ko / bn); };
  if (Ko % 128 == 0 && tiles(128, 128) >= 256) return launch<128, 128>(p, st);
  if (Ko % 128 == 0 && tiles(64, 128) >= 256) return launch<64, 128>(p, st);
  if (Ko % 128 != 0 && tiles(128, 64) >= 256) return launch<128, 64>(p, st);
  return launch<64, 64>(p, st);
}

}  // extern "C"
