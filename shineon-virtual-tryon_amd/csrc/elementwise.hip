// HBM-bound pointwise / resampling / reduction kernels of the try-on hot path (gfx950).
// All tensors are "pixel rows x channel columns" (NHWC) with an explicit row stride, so channel slices
// of a concat buffer are valid operands.  Every kernel has a 16-byte (4-channel) path that is used when
// the channel count, the strides and the base pointers allow it, and a scalar path otherwise.
//
// Reference ops restated here (file:line in /root/reference):
//   activations            models/networks/cpvton/unet.py:132-135,201-211 ; models/networks/activation.py:4-18
//   nn.Upsample(x2,bilinear, align_corners=False)   models/networks/cpvton/unet.py:138,155,166
//   torch.cat(dim=1)       models/networks/cpvton/unet.py:198 ; models/unet_mask_model.py:69 ; util/__init__.py:64-66
//   MaxPool2d(2,2)         torchvision vgg19.features (models/networks/vgg.py:9-23)
//   L1Loss / F.l1_loss     models/networks/loss.py:110-122 ; models/warp_model.py:88 ; models/unet_mask_model.py:174-184
//   tanh/sigmoid/blend     models/unet_mask_model.py:84-133
//   Adam                   models/base_model.py:165-168
#include "common.h"
#include "../../include/shineon_hip.h"

namespace {

template <int VEC>
struct Pack {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Pack<VEC> ldp(const float* p) {
  Pack<VEC> r;
  if constexpr (VEC == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
  } else {
    r.v[0] = p[0];
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ void stp(float* p, const Pack<VEC>& r) {
  if constexpr (VEC == 4) {
    f32x4 t = {r.v[0], r.v[1], r.v[2], r.v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  } else {
    p[0] = r.v[0];
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;  // 256 CUs x 8 blocks x 4: grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ activations
template <int VEC>
__global__ __launch_bounds__(256) void act_fwd_k(const float* __restrict__ x, int ldx,
                                                 float* __restrict__ y, int ldy, unsigned rows,
                                                 unsigned C, int act, float param) {
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, cq = idx - row * CQ;
    Pack<VEC> a = ldp<VEC>(x + (size_t)row * ldx + cq * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a.v[i] = so_actf(act, a.v[i], param);
    stp<VEC>(y + (size_t)row * ldy + cq * VEC, a);
  }
}

// dx = dy * act'(x)
template <int VEC>
__global__ __launch_bounds__(256) void act_bwd_k(const float* __restrict__ x, int ldx,
                                                 const float* __restrict__ dy, int lddy,
                                                 float* __restrict__ dx, int lddx, unsigned rows,
                                                 unsigned C, int act, float param) {
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, cq = idx - row * CQ;
    const Pack<VEC> a = ldp<VEC>(x + (size_t)row * ldx + cq * VEC);
    Pack<VEC> g = ldp<VEC>(dy + (size_t)row * lddy + cq * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g.v[i] *= so_actg(act, a.v[i], param);
    stp<VEC>(dx + (size_t)row * lddx + cq * VEC, g);
  }
}

// dx = res + dy * act'(x): the gradient of a tensor that feeds BOTH a skip connection (gradient `res`) and an activation
// (gradient `dy` of the activation's output) - the U-Net block input, unet.py:187-198.  One rounding per operation, in the
// order autograd's separate kernels use (multiply, then add; no fused multiply-add), so the bits equal act_bwd_k + a + b.
template <int VEC>
__global__ __launch_bounds__(256) void act_bwd_add_k(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ dy, int lddy,
                                                     const float* __restrict__ res, int ldres,
                                                     float* __restrict__ dx, int lddx, unsigned rows,
                                                     unsigned C, int act, float param) {
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, cq = idx - row * CQ;
    const Pack<VEC> a = ldp<VEC>(x + (size_t)row * ldx + cq * VEC);
    Pack<VEC> g = ldp<VEC>(dy + (size_t)row * lddy + cq * VEC);
    const Pack<VEC> r = ldp<VEC>(res + (size_t)row * ldres + cq * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      // (hipcc contracts __fadd_rn(r, __fmul_rn(..)) into one v_fma_f32 like any a + b * c: switch contraction off lexically)
#pragma clang fp contract(off)
      const float prod = g.v[i] * so_actg(act, a.v[i], param);
      g.v[i] = r.v[i] + prod;
    }
    stp<VEC>(dx + (size_t)row * lddx + cq * VEC, g);
  }
}

// ------------------------------------------------------------------ strided 2-D copy / accumulate
// dst[row][0:Cd] = (c < Cs ? src[row][c] : 0)   (zero channel padding when Cd > Cs); mode 1: dst += src
template <int VEC>
__global__ __launch_bounds__(256) void copy2d_k(const float* __restrict__ src, int lds_, unsigned Cs,
                                                float* __restrict__ dst, int ldd, unsigned Cd,
                                                unsigned rows, int accumulate) {
  const unsigned CQ = Cd / VEC;
  const unsigned total = rows * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, cq = idx - row * CQ;
    Pack<VEC> a;
    if (cq * VEC < Cs) {
      a = ldp<VEC>(src + (size_t)row * lds_ + cq * VEC);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) a.v[i] = 0.f;
    }
    float* d = dst + (size_t)row * ldd + cq * VEC;
    if (accumulate) {
      const Pack<VEC> o = ldp<VEC>(d);
#pragma unroll
      for (int i = 0; i < VEC; ++i) a.v[i] += o.v[i];
    }
    stp<VEC>(d, a);
  }
}

// ------------------------------------------------------------------ layout changes
// NCHW (planar, contiguous) -> NHWC rows with stride ldd, written at channel offset 0 of dst.
// One block transposes a [32 pixels][C-chunk of 32] tile through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_k(const float* __restrict__ src,
                                                      float* __restrict__ dst, int ldd, int C,
                                                      int Cd, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    tile[i][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    if (p < HW && c < Cd) dst[((size_t)n * HW + p) * ldd + c] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_k(const float* __restrict__ src, int lds_,
                                                      float* __restrict__ dst, int C, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    tile[i][tx] = (c < C && p < HW) ? src[((size_t)n * HW + p) * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    if (p < HW && c < C) dst[((size_t)n * C + c) * HW + p] = tile[tx][i];
  }
}

// OHWI weights w[ko][t][c] -> IHWO wt[c][t][ko] (t = filter tap): one LDS tile transpose per (tap, 32x32 tile).
// The transposed copy lets the input-gradient GEMM read both operands k-contiguous.
__global__ __launch_bounds__(256) void ohwi_to_ihwo_k(const float* __restrict__ w, float* __restrict__ wt,
                                                      int Ko, int T, int C) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int ko = k0 + i, c = c0 + tx;
    tile[i][tx] = (ko < Ko && c < C) ? w[((size_t)ko * T + t) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, ko = k0 + tx;
    if (c < C && ko < Ko) wt[((size_t)c * T + t) * Ko + ko] = tile[tx][i];
  }
}

// ------------------------------------------------------------------ bilinear x2 upsample
// PyTorch upsample_bilinear2d, align_corners=False, scale 2:
//   src = max(0, (o + 0.5) * 0.5 - 0.5); i0 = floor(src); i1 = min(i0 + 1, in - 1); l1 = src - i0.
__device__ __forceinline__ void up_src(int o, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = ((float)o + 0.5f) * 0.5f - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

template <int VEC>
// The source may be the channel concatenation [x | x2] of two tensors (U-Net skip connection): channels < C1 come
// from x, the rest from x2 (x2 == nullptr / C1 == C: single source), so the concatenation is never materialised.
__global__ __launch_bounds__(256) void upsample2x_fwd_k(const float* __restrict__ x, int ldx,
                                                        float* __restrict__ y, int ldy, unsigned Nb,
                                                        unsigned H, unsigned W, unsigned C, int act,
                                                        float act_param, const float* __restrict__ x2, int ldx2,
                                                        unsigned C1) {
  const unsigned CQ = C / VEC, Ho = 2 * H, Wo = 2 * W;
  const unsigned total = Nb * Ho * Wo * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned cq = t % CQ; t /= CQ;
    const unsigned wo = t % Wo; t /= Wo;
    const unsigned ho = t % Ho;
    const unsigned n = t / Ho;
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    up_src((int)ho, (int)H, h0, h1, lh0, lh1);
    up_src((int)wo, (int)W, w0, w1, lw0, lw1);
    const unsigned c0 = cq * VEC;
    const bool first = c0 < C1;
    const int ld = first ? ldx : ldx2;
    const float* base = (first ? x + c0 : x2 + (c0 - C1)) + (size_t)n * H * W * ld;
    Pack<VEC> a00 = ldp<VEC>(base + ((size_t)h0 * W + w0) * ld);
    Pack<VEC> a01 = ldp<VEC>(base + ((size_t)h0 * W + w1) * ld);
    Pack<VEC> a10 = ldp<VEC>(base + ((size_t)h1 * W + w0) * ld);
    Pack<VEC> a11 = ldp<VEC>(base + ((size_t)h1 * W + w1) * ld);
    if (act != SO_ACT_NONE) {  // activation of the source pixels fused in front of the interpolation
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        a00.v[i] = so_actf(act, a00.v[i], act_param); a01.v[i] = so_actf(act, a01.v[i], act_param);
        a10.v[i] = so_actf(act, a10.v[i], act_param); a11.v[i] = so_actf(act, a11.v[i], act_param);
      }
    }
    Pack<VEC> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      o.v[i] = lh0 * (lw0 * a00.v[i] + lw1 * a01.v[i]) + lh1 * (lw0 * a10.v[i] + lw1 * a11.v[i]);
    stp<VEC>(y + ((size_t)(n * Ho + ho) * Wo + wo) * ldy + cq * VEC, o);
  }
}

// Tiled form for the large levels: a block takes UP_TH x UP_TW source pixels x UP_CC channels, stages them (+ a one-pixel
// clamped halo) through LDS with the activation applied ONCE per source element, and writes the 2*UP_TH x 2*UP_TW outputs.
// The element-per-thread form above fetches four neighbours per output (4x the output size through the vector L1) and
// evaluates the activation four times per source element (GELU: erf) - 3.1 of 8 TB/s on the 128x96 -> 256x192 level.
// Same taps, weights and expression per output as upsample2x_fwd_k: the results are bit-identical.
constexpr int UP_TH = 8, UP_TW = 8, UP_CC = 32, UP_CQ = UP_CC / 4;
__global__ __launch_bounds__(256) void upsample2x_fwd_tiled_k(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                              int ldy, unsigned H, unsigned W, unsigned tiles_w,
                                                              unsigned tiles_hw, unsigned cchunks, int act, float act_param,
                                                              const float* __restrict__ x2, int ldx2, unsigned C1) {
  __shared__ f32x4 tile[(UP_TH + 2) * (UP_TW + 2) * UP_CQ];
  unsigned b = blockIdx.x;
  const unsigned cc = b % cchunks; b /= cchunks;
  const unsigned tw = b % tiles_w;
  const unsigned th = (b % tiles_hw) / tiles_w;
  const unsigned n = b / tiles_hw;
  const int h_lo = (int)(th * UP_TH) - 1, w_lo = (int)(tw * UP_TW) - 1;
  const unsigned c0 = cc * UP_CC;
  const bool first = c0 < C1;
  const int ld = first ? ldx : ldx2;
  const float* base = (first ? x + c0 : x2 + (c0 - C1)) + (size_t)n * H * W * ld;
  for (unsigned i = threadIdx.x; i < (UP_TH + 2) * (UP_TW + 2) * UP_CQ; i += 256u) {
    const unsigned q = i % UP_CQ, pix = i / UP_CQ;
    int h = h_lo + (int)(pix / (UP_TW + 2)), w = w_lo + (int)(pix % (UP_TW + 2));
    h = h < 0 ? 0 : (h > (int)H - 1 ? (int)H - 1 : h);
    w = w < 0 ? 0 : (w > (int)W - 1 ? (int)W - 1 : w);
    f32x4 v = *reinterpret_cast<const f32x4*>(base + ((size_t)h * W + w) * ld + q * 4);
    if (act != SO_ACT_NONE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = so_actf(act, v[k], act_param);
    }
    tile[i] = v;
  }
  __syncthreads();
  const unsigned q = threadIdx.x % UP_CQ;
  const unsigned ocol = (threadIdx.x / UP_CQ) % (2 * UP_TW);
  const unsigned wo = tw * (2 * UP_TW) + ocol;
  if (wo >= 2 * W) return;
  int w0, w1;
  float lw0, lw1;
  up_src((int)wo, (int)W, w0, w1, lw0, lw1);
  w0 -= w_lo; w1 -= w_lo;
  constexpr unsigned ROWS_PER_PASS = 256 / (UP_CQ * 2 * UP_TW);
#pragma unroll 2
  for (unsigned orow = threadIdx.x / (UP_CQ * 2 * UP_TW); orow < 2 * UP_TH; orow += ROWS_PER_PASS) {
    const unsigned ho = th * (2 * UP_TH) + orow;
    if (ho >= 2 * H) break;
    int h0, h1;
    float lh0, lh1;
    up_src((int)ho, (int)H, h0, h1, lh0, lh1);
    h0 -= h_lo; h1 -= h_lo;
    const f32x4 a00 = tile[(h0 * (UP_TW + 2) + w0) * UP_CQ + q], a01 = tile[(h0 * (UP_TW + 2) + w1) * UP_CQ + q];
    const f32x4 a10 = tile[(h1 * (UP_TW + 2) + w0) * UP_CQ + q], a11 = tile[(h1 * (UP_TW + 2) + w1) * UP_CQ + q];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = lh0 * (lw0 * a00[k] + lw1 * a01[k]) + lh1 * (lw0 * a10[k] + lw1 * a11[k]);
    *reinterpret_cast<f32x4*>(y + ((size_t)(n * 2 * H + ho) * (2 * W) + wo) * ldy + c0 + q * 4) = o;
  }
}

// Gather form of the backward pass (deterministic, no atomics): each input pixel i receives weight from
// output pixels 2i-1 (.25), 2i (.75 or 1 at i==0), 2i+1 (.75 or 1 at i==in-1), 2i+2 (.25).
__device__ __forceinline__ void up_taps(int i, int in, int o[4], float w[4]) {
  o[0] = 2 * i - 1; w[0] = (i >= 1) ? 0.25f : 0.f;
  o[1] = 2 * i;     w[1] = (i == 0) ? 1.0f : 0.75f;
  o[2] = 2 * i + 1; w[2] = (i == in - 1) ? 1.0f : 0.75f;
  o[3] = 2 * i + 2; w[3] = (i + 1 <= in - 1) ? 0.25f : 0.f;
}

template <int VEC>
__global__ __launch_bounds__(256) void upsample2x_bwd_k(const float* __restrict__ dy, int lddy,
                                                        float* __restrict__ dx, int lddx,
                                                        unsigned Nb, unsigned H, unsigned W,
                                                        unsigned C, const float* __restrict__ x, int ldx, int act,
                                                        float act_param, float* __restrict__ dx2, int lddx2,
                                                        const float* __restrict__ x2, int ldx2, unsigned C1) {
  const unsigned CQ = C / VEC, Ho = 2 * H, Wo = 2 * W;
  const unsigned total = Nb * H * W * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned cq = t % CQ; t /= CQ;
    const unsigned wi = t % W; t /= W;
    const unsigned hi = t % H;
    const unsigned n = t / H;
    int oh[4], ow[4];
    float wh[4], ww[4];
    up_taps((int)hi, (int)H, oh, wh);
    up_taps((int)wi, (int)W, ow, ww);
    Pack<VEC> acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] = 0.f;
    const float* base = dy + (size_t)n * Ho * Wo * lddy + cq * VEC;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wh[a] == 0.f) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (ww[b] == 0.f) continue;
        const Pack<VEC> g = ldp<VEC>(base + ((size_t)oh[a] * Wo + ow[b]) * lddy);
        const float wgt = wh[a] * ww[b];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.v[i] += wgt * g.v[i];
      }
    }
    const unsigned c0 = cq * VEC;
    const bool first = c0 < C1;
    const size_t pix = (size_t)(n * H + hi) * W + wi;
    if (act != SO_ACT_NONE) {  // chain rule through the activation fused in front of the upsample
      const Pack<VEC> xv = ldp<VEC>(first ? x + pix * ldx + c0 : x2 + pix * ldx2 + (c0 - C1));
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.v[i] *= so_actg(act, xv.v[i], act_param);
    }
    stp<VEC>(first ? dx + pix * lddx + c0 : dx2 + pix * lddx2 + (c0 - C1), acc);
  }
}

// Tiled backward pass for the large levels: the (2*UP_TH + 2) x (2*UP_TW + 2) patch of dy a block's UP_TH x UP_TW source
// pixels gather from goes through LDS once (the element-per-thread form fetches every dy element four times); taps outside
// the image carry weight 0 and are skipped exactly as above, so the sums have the same terms in the same order.
template <int TH>
__global__ __launch_bounds__(256) void upsample2x_bwd_tiled_k(const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                              int lddx, unsigned H, unsigned W, unsigned tiles_w,
                                                              unsigned tiles_hw, unsigned cchunks,
                                                              const float* __restrict__ x, int ldx, int act, float act_param,
                                                              float* __restrict__ dx2, int lddx2,
                                                              const float* __restrict__ x2, int ldx2, unsigned C1) {
  constexpr int PH = 2 * TH + 2, PW = 2 * UP_TW + 2;
  __shared__ f32x4 patch[PH * PW * UP_CQ];
  unsigned b = blockIdx.x;
  const unsigned cc = b % cchunks; b /= cchunks;
  const unsigned tw = b % tiles_w;
  const unsigned th = (b % tiles_hw) / tiles_w;
  const unsigned n = b / tiles_hw;
  const int oh_lo = 2 * (int)(th * TH) - 1, ow_lo = 2 * (int)(tw * UP_TW) - 1;
  const unsigned c0 = cc * UP_CC, Ho = 2 * H, Wo = 2 * W;
  const float* base = dy + (size_t)n * Ho * Wo * lddy + c0;
  for (unsigned i = threadIdx.x; i < PH * PW * UP_CQ; i += 256u) {
    const unsigned q = i % UP_CQ, pix = i / UP_CQ;
    const int oh = oh_lo + (int)(pix / PW), ow = ow_lo + (int)(pix % PW);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (oh >= 0 && oh < (int)Ho && ow >= 0 && ow < (int)Wo)
      v = *reinterpret_cast<const f32x4*>(base + ((size_t)oh * Wo + ow) * lddy + q * 4);
    patch[i] = v;
  }
  __syncthreads();
  const unsigned q = threadIdx.x % UP_CQ;
  const bool first = c0 < C1;
  for (unsigned p = threadIdx.x / UP_CQ; p < TH * UP_TW; p += 256 / UP_CQ) {
    const unsigned hi = th * TH + p / UP_TW, wi = tw * UP_TW + p % UP_TW;
    if (hi >= H || wi >= W) continue;
    int oh[4], ow[4];
    float wh[4], ww[4];
    up_taps((int)hi, (int)H, oh, wh);
    up_taps((int)wi, (int)W, ow, ww);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wh[a] == 0.f) continue;
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) {
        if (ww[b2] == 0.f) continue;
        const f32x4 g = patch[((oh[a] - oh_lo) * PW + (ow[b2] - ow_lo)) * UP_CQ + q];
        const float wgt = wh[a] * ww[b2];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += wgt * g[k];
      }
    }
    const size_t pix = (size_t)(n * H + hi) * W + wi;
    const unsigned c = c0 + q * 4;
    if (act != SO_ACT_NONE) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(first ? x + pix * ldx + c : x2 + pix * ldx2 + (c - C1));
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] *= so_actg(act, xv[k], act_param);
    }
    *reinterpret_cast<f32x4*>(first ? dx + pix * lddx + c : dx2 + pix * lddx2 + (c - C1)) = acc;
  }
}

// ------------------------------------------------------------------ 2x2 max pool (stride 2)
template <int VEC>
__global__ __launch_bounds__(256) void maxpool2_fwd_k(const float* __restrict__ x, int ldx,
                                                      float* __restrict__ y, int ldy, unsigned Nb,
                                                      unsigned H, unsigned W, unsigned C) {
  const unsigned CQ = C / VEC, Ho = H / 2, Wo = W / 2;
  const unsigned total = Nb * Ho * Wo * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned cq = t % CQ; t /= CQ;
    const unsigned wo = t % Wo; t /= Wo;
    const unsigned ho = t % Ho;
    const unsigned n = t / Ho;
    const float* base = x + ((size_t)(n * H + 2 * ho) * W + 2 * wo) * ldx + cq * VEC;
    const Pack<VEC> a = ldp<VEC>(base), b = ldp<VEC>(base + ldx);
    const Pack<VEC> c = ldp<VEC>(base + (size_t)W * ldx), d = ldp<VEC>(base + (size_t)(W + 1) * ldx);
    Pack<VEC> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.v[i] = fmaxf(fmaxf(a.v[i], b.v[i]), fmaxf(c.v[i], d.v[i]));
    stp<VEC>(y + ((size_t)(n * Ho + ho) * Wo + wo) * ldy + cq * VEC, o);
  }
}

// dx[window] = dy at the first maximum in (h, w) scan order (PyTorch CPU tie rule), else 0.
template <int VEC>
__global__ __launch_bounds__(256) void maxpool2_bwd_k(const float* __restrict__ x, int ldx,
                                                      const float* __restrict__ dy, int lddy,
                                                      float* __restrict__ dx, int lddx, unsigned Nb,
                                                      unsigned H, unsigned W, unsigned C, int relu_gate) {
  const unsigned CQ = C / VEC, Ho = H / 2, Wo = W / 2;
  const unsigned total = Nb * Ho * Wo * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned cq = t % CQ; t /= CQ;
    const unsigned wo = t % Wo; t /= Wo;
    const unsigned ho = t % Ho;
    const unsigned n = t / Ho;
    const size_t p00 = ((size_t)(n * H + 2 * ho) * W + 2 * wo);
    const float* base = x + p00 * ldx + cq * VEC;
    const Pack<VEC> a = ldp<VEC>(base), b = ldp<VEC>(base + ldx);
    const Pack<VEC> c = ldp<VEC>(base + (size_t)W * ldx), d = ldp<VEC>(base + (size_t)(W + 1) * ldx);
    Pack<VEC> g = ldp<VEC>(dy + ((size_t)(n * Ho + ho) * Wo + wo) * lddy + cq * VEC);
    Pack<VEC> ga, gb, gc, gd;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      int arg = 0;
      float m = a.v[i];
      if (b.v[i] > m) { m = b.v[i]; arg = 1; }
      if (c.v[i] > m) { m = c.v[i]; arg = 2; }
      if (d.v[i] > m) { m = d.v[i]; arg = 3; }
      if (relu_gate && !(m > 0.f)) g.v[i] = 0.f;  // x is a ReLU output: the window's maximum gates the gradient
      ga.v[i] = arg == 0 ? g.v[i] : 0.f;
      gb.v[i] = arg == 1 ? g.v[i] : 0.f;
      gc.v[i] = arg == 2 ? g.v[i] : 0.f;
      gd.v[i] = arg == 3 ? g.v[i] : 0.f;
    }
    float* o = dx + p00 * lddx + cq * VEC;
    stp<VEC>(o, ga);
    stp<VEC>(o + lddx, gb);
    stp<VEC>(o + (size_t)W * lddx, gc);
    stp<VEC>(o + (size_t)(W + 1) * lddx, gd);
  }
}

// ------------------------------------------------------------------ column sums (bias gradient)
// partial[chunk][c] = sum over rows of the chunk;  then final[c] = sum over chunks (fixed order).
__global__ __launch_bounds__(256) void colsum_partial_k(const float* __restrict__ x, int ldx,
                                                        unsigned rows, unsigned C, unsigned chunk,
                                                        float* __restrict__ part) {
  __shared__ float red[256];
  const unsigned CPB = C >= 256 ? 256 : C;
  const unsigned RL = 256 / CPB;
  const unsigned tx = threadIdx.x % CPB, ty = threadIdx.x / CPB;
  const unsigned col = blockIdx.y * CPB + tx;
  const unsigned r0 = blockIdx.x * chunk;
  unsigned r1 = r0 + chunk;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  if (ty < RL && col < C)
#pragma unroll 8
    for (unsigned r = r0 + ty; r < r1; r += RL) s += x[(size_t)r * ldx + col];
  red[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && col < C) {
    float t = 0.f;
    for (unsigned l = 0; l < RL; ++l) t += red[l * CPB + tx];
    part[(size_t)blockIdx.x * C + col] = t;
  }
}

__global__ __launch_bounds__(1024) void colsum_final_k(const float* __restrict__ part, unsigned nchunk,
                                                       unsigned C, float* __restrict__ out,
                                                       int accumulate) {
  // 16 columns x 64 chunk-lanes per block, lanes merged in fixed order (deterministic)
  __shared__ float sh[1024];
  const unsigned tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const unsigned c = blockIdx.x * 16u + tx;
  float s = 0.f;
  if (c < C)
#pragma unroll 8
    for (unsigned k = ty; k < nchunk; k += 64) s += part[(size_t)k * C + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ty != 0 || c >= C) return;
  s = 0.f;
  for (unsigned l = 0; l < 64; ++l) s += sh[l * 16 + tx];
  out[c] = accumulate ? out[c] + s : s;
}

// One launch for short matrices (<= 4096 rows: attention projections, the 16x12 / 8x6 / 4x3 feature maps): a block of
// 32 channels x 32 row-lanes, lanes merged in fixed order through LDS.
__global__ __launch_bounds__(1024) void colsum_small_k(const float* __restrict__ x, int ldx, unsigned rows, unsigned C,
                                                       float* __restrict__ out, int accumulate) {
  __shared__ float sh[32][33];
  const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const unsigned col = blockIdx.x * 32u + tx;
  float s = 0.f;
  if (col < C)
#pragma unroll 8
    for (unsigned r = ty; r < rows; r += 32) s += x[(size_t)r * ldx + col];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty != 0 || col >= C) return;
  s = 0.f;
  for (unsigned l = 0; l < 32; ++l) s += sh[l][tx];
  out[col] = accumulate ? out[col] + s : s;
}

// ------------------------------------------------------------------ L1 loss (mean |a-b|)
template <int VEC>
__global__ __launch_bounds__(256) void l1_partial_k(const float* __restrict__ a, int lda,
                                                    const float* __restrict__ b, int ldb,
                                                    unsigned rows, unsigned C,
                                                    float* __restrict__ part) {
  __shared__ float red[4];
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  float s = 0.f;
#pragma unroll 4
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, c = (idx - row * CQ) * VEC;
    const Pack<VEC> av = ldp<VEC>(a + (size_t)row * lda + c), bv = ldp<VEC>(b + (size_t)row * ldb + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += fabsf(av.v[i] - bv.v[i]);
  }
  s = so_block_sum256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = scale * sum(part[0:n])   (single block, fixed order)
__global__ __launch_bounds__(256) void sum_final_k(const float* __restrict__ part, unsigned n,
                                                   float scale, float* __restrict__ out,
                                                   int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (unsigned i = threadIdx.x; i < n; i += 256) s += part[i];
  s = so_block_sum256(s, red);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + scale * s : scale * s;
}

// da = sign(a-b) * gout[0] * scale ; (optionally) db = -da ; accumulate: da += ...
template <int VEC>
__global__ __launch_bounds__(256) void l1_bwd_k(const float* __restrict__ a, int lda,
                                                const float* __restrict__ b, int ldb,
                                                const float* __restrict__ gout, float scale,
                                                float* __restrict__ da, int ldda, unsigned rows,
                                                unsigned C, int accumulate, int relu_gate) {
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  const float g = gout[0] * scale;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, c = (idx - row * CQ) * VEC;
    const Pack<VEC> av = ldp<VEC>(a + (size_t)row * lda + c), bv = ldp<VEC>(b + (size_t)row * ldb + c);
    float* o = da + (size_t)row * ldda + c;
    Pack<VEC> r;
    if (accumulate) r = ldp<VEC>(o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float d = av.v[i] - bv.v[i];
      float v = d > 0.f ? g : (d < 0.f ? -g : 0.f);
      if (relu_gate && !(av.v[i] > 0.f)) v = 0.f;  // `a` is a ReLU output: chain through the ReLU in the same pass
      r.v[i] = accumulate ? r.v[i] + v : v;
    }
    stp<VEC>(o, r);
  }
}

// ------------------------------------------------------------------ try-on composition
// o: U-Net output rows [pix][ldo] with channels (r,g,b,mask); cloth rows [pix][ldc] (3 channels).
//   p_rendered = tanh(o[0:3]); mask = sigmoid(o[3]); p_tryon = (1-mask) p_rendered + mask cloth
// (reference: models/unet_mask_model.py:84-86,126-129, single-frame case)
__global__ __launch_bounds__(256) void tryon_compose_fwd_k(const float* __restrict__ o, int ldo,
                                                           const float* __restrict__ cloth, int ldc,
                                                           float* __restrict__ rendered, int ldr,
                                                           float* __restrict__ mask, int ldm,
                                                           float* __restrict__ tryon, int ldt,
                                                           int tryon_pad, unsigned pix) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < pix; i += gridDim.x * 256u) {
    const float* op = o + (size_t)i * ldo;
    const float* cp = cloth + (size_t)i * ldc;
    const float m = so_sigmoid(op[3]);
    mask[(size_t)i * ldm] = m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r = tanhf(op[c]);
      rendered[(size_t)i * ldr + c] = r;
      tryon[(size_t)i * ldt + c] = (1.f - m) * r + m * cp[c];
    }
    for (int c = 3; c < tryon_pad; ++c) tryon[(size_t)i * ldt + c] = 0.f;
  }
}

// do[c] = (d_tryon[c] (1-m) + d_rendered[c]) (1 - r^2) ; do[3] = (sum_c d_tryon[c] (cloth[c]-r[c]) + d_mask) m (1-m)
__global__ __launch_bounds__(256) void tryon_compose_bwd_k(
    const float* __restrict__ rendered, int ldr, const float* __restrict__ mask, int ldm,
    const float* __restrict__ cloth, int ldc, const float* __restrict__ d_tryon, int lddt,
    const float* __restrict__ d_rendered, int lddr, const float* __restrict__ d_mask, int lddm,
    float* __restrict__ d_o, int lddo, unsigned pix) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < pix; i += gridDim.x * 256u) {
    const float m = mask[(size_t)i * ldm];
    float gm = d_mask ? d_mask[(size_t)i * lddm] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r = rendered[(size_t)i * ldr + c];
      const float gt = d_tryon ? d_tryon[(size_t)i * lddt + c] : 0.f;
      float gr = gt * (1.f - m);
      if (d_rendered) gr += d_rendered[(size_t)i * lddr + c];
      gm += gt * (cloth[(size_t)i * ldc + c] - r);
      d_o[(size_t)i * lddo + c] = gr * (1.f - r * r);
    }
    d_o[(size_t)i * lddo + 3] = gm * m * (1.f - m);
  }
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam defaults)
// m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
template <int VEC>
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v,
                                              unsigned n, float lr, float b1, float b2, float eps,
                                              float bc1, float bc2_sqrt, float grad_scale) {
  const unsigned nq = n / VEC;
  const float step = lr / bc1;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < nq; idx += gridDim.x * 256u) {
    Pack<VEC> pp = ldp<VEC>(p + (size_t)idx * VEC);
    const Pack<VEC> gg = ldp<VEC>(g + (size_t)idx * VEC);
    Pack<VEC> mm = ldp<VEC>(m + (size_t)idx * VEC);
    Pack<VEC> vv = ldp<VEC>(v + (size_t)idx * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float gi = gg.v[i] * grad_scale;
      mm.v[i] = b1 * mm.v[i] + (1.f - b1) * gi;
      vv.v[i] = b2 * vv.v[i] + (1.f - b2) * gi * gi;
      const float denom = sqrtf(vv.v[i]) / bc2_sqrt + eps;
      pp.v[i] -= step * (mm.v[i] / denom);
    }
    stp<VEC>(p + (size_t)idx * VEC, pp);
    stp<VEC>(m + (size_t)idx * VEC, mm);
    stp<VEC>(v + (size_t)idx * VEC, vv);
  }
}

__global__ __launch_bounds__(256) void fill_k(float* __restrict__ p, unsigned n, float val) {
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) p[idx] = val;
}

// y = a * x + b * y   (flat)
__global__ __launch_bounds__(256) void axpby_k(const float* __restrict__ x, float a,
                                               float* __restrict__ y, float b, unsigned n) {
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u)
    y[idx] = a * x[idx] + (b == 0.f ? 0.f : b * y[idx]);
}

// out = alpha[0] * a + b   (attention output gamma * o + x, sagan.py:53)
__global__ __launch_bounds__(256) void scale_add_k(const float* __restrict__ a, int lda,
                                                   const float* __restrict__ alpha,
                                                   const float* __restrict__ b, int ldb,
                                                   float* __restrict__ out, int ldo, unsigned rows,
                                                   unsigned C) {
  const unsigned total = rows * C;
  const float al = alpha[0];
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / C, c = idx - row * C;
    out[(size_t)row * ldo + c] = al * a[(size_t)row * lda + c] + b[(size_t)row * ldb + c];
  }
}

// out = (1 - m) * a + m * b with a one-channel mask broadcast over C channels
// (flow-mask and try-on-mask blends of the multi-frame path, unet_mask_model.py:118-129)
__global__ __launch_bounds__(256) void blend_fwd_k(const float* __restrict__ a, int lda,
                                                   const float* __restrict__ b, int ldb,
                                                   const float* __restrict__ m, int ldm,
                                                   float* __restrict__ out, int ldo, unsigned rows,
                                                   unsigned C) {
  const unsigned total = rows * C;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / C, c = idx - row * C;
    const float mv = m[(size_t)row * ldm];
    out[(size_t)row * ldo + c] = (1.f - mv) * a[(size_t)row * lda + c] + mv * b[(size_t)row * ldb + c];
  }
}

// da = (1-m) g ; db = m g ; dm = sum_c g (b - a)      (one thread per pixel row)
__global__ __launch_bounds__(256) void blend_bwd_k(const float* __restrict__ a, int lda,
                                                   const float* __restrict__ b, int ldb,
                                                   const float* __restrict__ m, int ldm,
                                                   const float* __restrict__ g, int ldg,
                                                   float* __restrict__ da, int ldda,
                                                   float* __restrict__ db, int lddb,
                                                   float* __restrict__ dm, int lddm, unsigned rows,
                                                   unsigned C) {
  for (unsigned row = blockIdx.x * 256u + threadIdx.x; row < rows; row += gridDim.x * 256u) {
    const float mv = m[(size_t)row * ldm];
    float s = 0.f;
    for (unsigned c = 0; c < C; ++c) {
      const float gv = g[(size_t)row * ldg + c];
      if (da) da[(size_t)row * ldda + c] = (1.f - mv) * gv;
      if (db) db[(size_t)row * lddb + c] = mv * gv;
      s += gv * (b[(size_t)row * ldb + c] - a[(size_t)row * lda + c]);
    }
    if (dm) dm[(size_t)row * lddm] = s;
  }
}

}  // namespace

#define VEC_OK2(p1, l1, p2, l2, C) (((C) & 3) == 0 && ((l1) & 3) == 0 && ((l2) & 3) == 0 && al16(p1) && al16(p2))

__global__ void counter_bump_k(unsigned* counter) { counter[0] += 1u; }

// every store of the kernels in front of this node (same stream) is complete when it starts; the release makes them visible
// system-wide before the flag moves
__global__ void signal_store_k(unsigned* flag, const unsigned* counter, int system_scope) {
  if (system_scope) {   // the waiter is the command processor (hipStreamWaitValue32)
    __threadfence_system();
    __hip_atomic_store(flag, counter[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {              // the waiter is a kernel on this device (so_stream_wait_ge mode 1): agent scope is enough
    __hip_atomic_store(flag, counter[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Bounded: besides the signal word the lane looks, every 64th poll, at a host-pinned abort word (the host sets it on any
// exception path, without needing a stream) and at the 100 MHz wall clock.  On abort / deadline it stores an error code into
// the host-pinned status word and RETURNS, so that the communication stream - and, with several ranks, every peer inside
// the collective queued behind this wait - moves on instead of hanging; the host raises at the next finish().
__global__ void wait_flag_k(const unsigned* flag, unsigned value, const unsigned* abort_word, unsigned* status,
                            unsigned long long max_ticks) {
  const unsigned long long t0 = wall_clock64();
  unsigned polls = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
    __builtin_amdgcn_s_sleep(64);
    if ((++polls & 63u) == 0u && status != nullptr) {
      unsigned code = 0u;
      if (abort_word != nullptr && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
        code = 1u;   // SO_WAIT_ABORTED
      else if ((unsigned long long)wall_clock64() - t0 > max_ticks)
        code = 2u;   // SO_WAIT_DEADLINE
      if (code) {
        __hip_atomic_store(status, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

extern "C" {

int so_act_fwd(const float* x, int ldx, float* y, int ldy, long long rows, int C, int act,
               float param, void* stream) {
  if (rows * C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (VEC_OK2(x, ldx, y, ldy, C))
    hipLaunchKernelGGL(act_fwd_k<4>, dim3(grid_for(rows * C / 4)), dim3(256), 0, st, x, ldx, y, ldy,
                       (unsigned)rows, (unsigned)C, act, param);
  else
    hipLaunchKernelGGL(act_fwd_k<1>, dim3(grid_for(rows * C)), dim3(256), 0, st, x, ldx, y, ldy,
                       (unsigned)rows, (unsigned)C, act, param);
  return SO_LAUNCH_CHECK();
}

int so_act_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx,
               long long rows, int C, int act, float param, void* stream) {
  if (rows * C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (VEC_OK2(x, ldx, dy, lddy, C) && (lddx & 3) == 0 && al16(dx))
    hipLaunchKernelGGL(act_bwd_k<4>, dim3(grid_for(rows * C / 4)), dim3(256), 0, st, x, ldx, dy, lddy,
                       dx, lddx, (unsigned)rows, (unsigned)C, act, param);
  else
    hipLaunchKernelGGL(act_bwd_k<1>, dim3(grid_for(rows * C)), dim3(256), 0, st, x, ldx, dy, lddy, dx,
                       lddx, (unsigned)rows, (unsigned)C, act, param);
  return SO_LAUNCH_CHECK();
}

int so_act_bwd_add(const float* x, int ldx, const float* dy, int lddy, const float* res, int ldres, float* dx, int lddx,
                   long long rows, int C, int act, float param, void* stream) {
  if (rows * C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (VEC_OK2(x, ldx, dy, lddy, C) && (lddx & 3) == 0 && al16(dx) && (ldres & 3) == 0 && al16(res))
    hipLaunchKernelGGL(act_bwd_add_k<4>, dim3(grid_for(rows * C / 4)), dim3(256), 0, st, x, ldx, dy, lddy, res, ldres,
                       dx, lddx, (unsigned)rows, (unsigned)C, act, param);
  else
    hipLaunchKernelGGL(act_bwd_add_k<1>, dim3(grid_for(rows * C)), dim3(256), 0, st, x, ldx, dy, lddy, res, ldres, dx,
                       lddx, (unsigned)rows, (unsigned)C, act, param);
  return SO_LAUNCH_CHECK();
}

int so_copy2d(const float* src, int lds_, int Cs, float* dst, int ldd, int Cd, long long rows,
              int accumulate, void* stream) {
  if (rows * Cd <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (VEC_OK2(src, lds_, dst, ldd, Cd) && (Cs & 3) == 0)
    hipLaunchKernelGGL(copy2d_k<4>, dim3(grid_for(rows * Cd / 4)), dim3(256), 0, st, src, lds_,
                       (unsigned)Cs, dst, ldd, (unsigned)Cd, (unsigned)rows, accumulate);
  else
    hipLaunchKernelGGL(copy2d_k<1>, dim3(grid_for(rows * Cd)), dim3(256), 0, st, src, lds_,
                       (unsigned)Cs, dst, ldd, (unsigned)Cd, (unsigned)rows, accumulate);
  return SO_LAUNCH_CHECK();
}

int so_nchw_to_nhwc(const float* src, float* dst, int ldd, int Nb, int C, int Cd, int HW,
                    void* stream) {
  if (Nb * C * HW <= 0) return 0;
  if (Cd < C) return SO_ERR_SHAPE;
  dim3 grid(so_cdiv(HW, 32), so_cdiv(Cd, 32), Nb);
  hipLaunchKernelGGL(nchw_to_nhwc_k, grid, dim3(256), 0, (hipStream_t)stream, src, dst, ldd, C, Cd, HW);
  return SO_LAUNCH_CHECK();
}

int so_nhwc_to_nchw(const float* src, int lds_, float* dst, int Nb, int C, int HW, void* stream) {
  if (Nb * C * HW <= 0) return 0;
  dim3 grid(so_cdiv(HW, 32), so_cdiv(C, 32), Nb);
  hipLaunchKernelGGL(nhwc_to_nchw_k, grid, dim3(256), 0, (hipStream_t)stream, src, lds_, dst, C, HW);
  return SO_LAUNCH_CHECK();
}

int so_ohwi_to_ihwo(const float* w, float* wt, int Ko, int taps, int C, void* stream) {
  if (Ko <= 0 || taps <= 0 || C <= 0) return 0;
  dim3 grid(so_cdiv(C, 32), so_cdiv(Ko, 32), taps);
  hipLaunchKernelGGL(ohwi_to_ihwo_k, grid, dim3(256), 0, (hipStream_t)stream, w, wt, Ko, taps, C);
  return SO_LAUNCH_CHECK();
}

// SO_UPSAMPLE_TILED=0: the element-per-thread kernels for every level (A/B measurements)
static bool so_upsample_tiled_enabled() {
  static const bool on = [] { const char* e = getenv("SO_UPSAMPLE_TILED"); return !(e && e[0] == '0'); }();
  return on;
}

int so_upsample2x_cat_fwd(const float* x1, int ldx1, int C1, const float* x2, int ldx2, int C2, float* y, int ldy,
                          int Nb, int H, int W, int act, float act_param, void* stream) {
  const int C = C1 + (x2 ? C2 : 0);
  const long long total = (long long)Nb * H * W * 4 * C;
  if (total <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool v2 = !x2 || ((C1 & 3) == 0 && (C2 & 3) == 0 && (ldx2 & 3) == 0 && al16(x2));
  // large levels: source tile staged through LDS (activation once per source element, one global fetch per element)
  if (VEC_OK2(x1, ldx1, y, ldy, C) && v2 && C % UP_CC == 0 && (!x2 || C1 % UP_CC == 0) && (long long)H * W >= 192 &&
      so_upsample_tiled_enabled()) {
    const unsigned tiles_w = (W + UP_TW - 1) / UP_TW, tiles_h = (H + UP_TH - 1) / UP_TH, cchunks = C / UP_CC;
    hipLaunchKernelGGL(upsample2x_fwd_tiled_k, dim3((unsigned)Nb * tiles_h * tiles_w * cchunks), dim3(256), 0, st, x1, ldx1, y,
                       ldy, (unsigned)H, (unsigned)W, tiles_w, tiles_h * tiles_w, cchunks, act, act_param, x2, ldx2,
                       (unsigned)(x2 ? C1 : C));
    return SO_LAUNCH_CHECK();
  }
  if (VEC_OK2(x1, ldx1, y, ldy, C) && v2)
    hipLaunchKernelGGL(upsample2x_fwd_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, x1, ldx1, y,
                       ldy, (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C, act, act_param, x2, ldx2, (unsigned)C1);
  else
    hipLaunchKernelGGL(upsample2x_fwd_k<1>, dim3(grid_for(total)), dim3(256), 0, st, x1, ldx1, y, ldy,
                       (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C, act, act_param, x2, ldx2, (unsigned)C1);
  return SO_LAUNCH_CHECK();
}

int so_upsample2x_act_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C, int act,
                          float act_param, void* stream) {
  return so_upsample2x_cat_fwd(x, ldx, C, nullptr, 0, 0, y, ldy, Nb, H, W, act, act_param, stream);
}

int so_upsample2x_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C,
                      void* stream) {
  return so_upsample2x_cat_fwd(x, ldx, C, nullptr, 0, 0, y, ldy, Nb, H, W, SO_ACT_NONE, 0.f, stream);
}

int so_upsample2x_cat_bwd(const float* x1, int ldx1, int C1, const float* x2, int ldx2, int C2, const float* dy,
                          int lddy, float* dx1, int lddx1, float* dx2, int lddx2, int Nb, int H, int W, int act,
                          float act_param, void* stream) {
  const int C = C1 + (dx2 ? C2 : 0);
  const long long total = (long long)Nb * H * W * C;
  if (total <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool xok = act == SO_ACT_NONE || ((ldx1 & 3) == 0 && al16(x1) && (!dx2 || ((ldx2 & 3) == 0 && al16(x2))));
  const bool v2 = !dx2 || ((C1 & 3) == 0 && (C2 & 3) == 0 && (lddx2 & 3) == 0 && al16(dx2));
  if (VEC_OK2(dy, lddy, dx1, lddx1, C) && xok && v2 && C % UP_CC == 0 && (!dx2 || C1 % UP_CC == 0) && (long long)H * W >= 192 &&
      so_upsample_tiled_enabled()) {
    static const int bth = [] { const char* e = getenv("SO_UPSAMPLE_BWD_TH"); return e ? atoi(e) : 4; }();
    const unsigned tiles_w = (W + UP_TW - 1) / UP_TW, cchunks = C / UP_CC;
    if (bth == 8) {
      const unsigned tiles_h = (H + 7) / 8;
      hipLaunchKernelGGL(upsample2x_bwd_tiled_k<8>, dim3((unsigned)Nb * tiles_h * tiles_w * cchunks), dim3(256), 0, st, dy, lddy,
                         dx1, lddx1, (unsigned)H, (unsigned)W, tiles_w, tiles_h * tiles_w, cchunks, x1, ldx1, act, act_param, dx2,
                         lddx2, x2, ldx2, (unsigned)(dx2 ? C1 : C));
    } else {
      const unsigned tiles_h = (H + 3) / 4;
      hipLaunchKernelGGL(upsample2x_bwd_tiled_k<4>, dim3((unsigned)Nb * tiles_h * tiles_w * cchunks), dim3(256), 0, st, dy, lddy,
                         dx1, lddx1, (unsigned)H, (unsigned)W, tiles_w, tiles_h * tiles_w, cchunks, x1, ldx1, act, act_param, dx2,
                         lddx2, x2, ldx2, (unsigned)(dx2 ? C1 : C));
    }
    return SO_LAUNCH_CHECK();
  }
  if (VEC_OK2(dy, lddy, dx1, lddx1, C) && xok && v2)
    hipLaunchKernelGGL(upsample2x_bwd_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, dy, lddy, dx1,
                       lddx1, (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C, x1, ldx1, act, act_param, dx2, lddx2, x2,
                       ldx2, (unsigned)C1);
  else
    hipLaunchKernelGGL(upsample2x_bwd_k<1>, dim3(grid_for(total)), dim3(256), 0, st, dy, lddy, dx1,
                       lddx1, (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C, x1, ldx1, act, act_param, dx2, lddx2, x2,
                       ldx2, (unsigned)C1);
  return SO_LAUNCH_CHECK();
}

int so_upsample2x_act_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int Nb, int H,
                          int W, int C, int act, float act_param, void* stream) {
  return so_upsample2x_cat_bwd(x, ldx, C, nullptr, 0, 0, dy, lddy, dx, lddx, nullptr, 0, Nb, H, W, act, act_param, stream);
}

int so_upsample2x_bwd(const float* dy, int lddy, float* dx, int lddx, int Nb, int H, int W, int C,
                      void* stream) {
  return so_upsample2x_cat_bwd(nullptr, 0, C, nullptr, 0, 0, dy, lddy, dx, lddx, nullptr, 0, Nb, H, W, SO_ACT_NONE, 0.f, stream);
}

int so_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C,
                    void* stream) {
  if ((H & 1) || (W & 1)) return SO_ERR_SHAPE;
  const long long total = (long long)Nb * (H / 2) * (W / 2) * C;
  if (total <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (VEC_OK2(x, ldx, y, ldy, C))
    hipLaunchKernelGGL(maxpool2_fwd_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, x, ldx, y, ldy,
                       (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C);
  else
    hipLaunchKernelGGL(maxpool2_fwd_k<1>, dim3(grid_for(total)), dim3(256), 0, st, x, ldx, y, ldy,
                       (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C);
  return SO_LAUNCH_CHECK();
}

int so_maxpool2_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int Nb,
                    int H, int W, int C, int relu_gate, void* stream) {
  if ((H & 1) || (W & 1)) return SO_ERR_SHAPE;
  const long long total = (long long)Nb * (H / 2) * (W / 2) * C;
  if (total <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (VEC_OK2(x, ldx, dy, lddy, C) && (lddx & 3) == 0 && al16(dx))
    hipLaunchKernelGGL(maxpool2_bwd_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, x, ldx, dy,
                       lddy, dx, lddx, (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C, relu_gate);
  else
    hipLaunchKernelGGL(maxpool2_bwd_k<1>, dim3(grid_for(total)), dim3(256), 0, st, x, ldx, dy, lddy,
                       dx, lddx, (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C, relu_gate);
  return SO_LAUNCH_CHECK();
}

long long so_colsum_ws_floats(long long rows, int C) {
  const unsigned CPB = C >= 256 ? 256 : (unsigned)C;
  const unsigned RL = 256 / CPB;
  const long long chunk = 16LL * RL;
  return ((rows + chunk - 1) / chunk) * C;
}

int so_colsum(const float* x, int ldx, long long rows, int C, float* out, int accumulate, float* ws,
              void* stream) {
  if (rows <= 0 || C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 4096) {
    hipLaunchKernelGGL(colsum_small_k, dim3(so_cdiv(C, 32)), dim3(1024), 0, st, x, ldx, (unsigned)rows, (unsigned)C, out,
                       accumulate);
    return SO_LAUNCH_CHECK();
  }
  const unsigned CPB = C >= 256 ? 256 : (unsigned)C;
  const unsigned RL = 256 / CPB;
  const unsigned chunk = 16 * RL;  // short per-thread row runs: the loop is a chain of dependent-latency batches
  const unsigned nchunk = (unsigned)((rows + chunk - 1) / chunk);
  dim3 grid(nchunk, so_cdiv(C, CPB));
  hipLaunchKernelGGL(colsum_partial_k, grid, dim3(256), 0, st, x, ldx, (unsigned)rows, (unsigned)C,
                     chunk, ws);
  hipLaunchKernelGGL(colsum_final_k, dim3(so_cdiv(C, 16)), dim3(1024), 0, st, ws, nchunk, (unsigned)C,
                     out, accumulate);
  return SO_LAUNCH_CHECK();
}

// ws: >= 1024 floats
int so_l1_loss_fwd(const float* a, int lda, const float* b, int ldb, long long rows, int C,
                   float scale, float* out, int accumulate, float* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int blocks = grid_for(rows * C);
  if (blocks > 1024) blocks = 1024;
  if (VEC_OK2(a, lda, b, ldb, C))
    hipLaunchKernelGGL(l1_partial_k<4>, dim3(blocks), dim3(256), 0, st, a, lda, b, ldb, (unsigned)rows,
                       (unsigned)C, ws);
  else
    hipLaunchKernelGGL(l1_partial_k<1>, dim3(blocks), dim3(256), 0, st, a, lda, b, ldb, (unsigned)rows,
                       (unsigned)C, ws);
  hipLaunchKernelGGL(sum_final_k, dim3(1), dim3(256), 0, st, ws, (unsigned)blocks, scale, out,
                     accumulate);
  return SO_LAUNCH_CHECK();
}

// out[0] = ((a + b) + c) + d, left to right like the chain of scalar additions it replaces (c, d may be null)
__global__ void scalar_sum_k(const float* a, const float* b, const float* c, const float* d, float* out) {
  float s = a[0] + b[0];
  if (c) s += c[0];
  if (d) s += d[0];
  out[0] = s;
}

int so_scalar_sum(const float* a, const float* b, const float* c, const float* d, float* out, void* stream) {
  if (!a || !b || !out) return SO_ERR_SHAPE;
  hipLaunchKernelGGL(scalar_sum_k, dim3(1), dim3(1), 0, (hipStream_t)stream, a, b, c, d, out);
  return SO_LAUNCH_CHECK();
}

int so_l1_loss_bwd(const float* a, int lda, const float* b, int ldb, const float* gout, float scale,
                   float* da, int ldda, long long rows, int C, int accumulate, int relu_gate, void* stream) {
  if (rows * C <= 0) return 0;
  if (VEC_OK2(a, lda, b, ldb, C) && (ldda & 3) == 0 && al16(da))
    hipLaunchKernelGGL(l1_bwd_k<4>, dim3(grid_for(rows * C / 4)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                       ldb, gout, scale, da, ldda, (unsigned)rows, (unsigned)C, accumulate, relu_gate);
  else
    hipLaunchKernelGGL(l1_bwd_k<1>, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                       ldb, gout, scale, da, ldda, (unsigned)rows, (unsigned)C, accumulate, relu_gate);
  return SO_LAUNCH_CHECK();
}

int so_tryon_compose_fwd(const float* o, int ldo, const float* cloth, int ldc, float* rendered,
                         int ldr, float* mask, int ldm, float* tryon, int ldt, int tryon_pad,
                         long long pix, void* stream) {
  if (pix <= 0) return 0;
  hipLaunchKernelGGL(tryon_compose_fwd_k, dim3(grid_for(pix)), dim3(256), 0, (hipStream_t)stream, o,
                     ldo, cloth, ldc, rendered, ldr, mask, ldm, tryon, ldt, tryon_pad, (unsigned)pix);
  return SO_LAUNCH_CHECK();
}

int so_tryon_compose_bwd(const float* rendered, int ldr, const float* mask, int ldm,
                         const float* cloth, int ldc, const float* d_tryon, int lddt,
                         const float* d_rendered, int lddr, const float* d_mask, int lddm, float* d_o,
                         int lddo, long long pix, void* stream) {
  if (pix <= 0) return 0;
  hipLaunchKernelGGL(tryon_compose_bwd_k, dim3(grid_for(pix)), dim3(256), 0, (hipStream_t)stream,
                     rendered, ldr, mask, ldm, cloth, ldc, d_tryon, lddt, d_rendered, lddr, d_mask,
                     lddm, d_o, lddo, (unsigned)pix);
  return SO_LAUNCH_CHECK();
}

int so_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float b1,
                 float b2, float eps, int step, float grad_scale, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2s = sqrtf(1.f - powf(b2, (float)step));
  if ((n & 3) == 0 && al16(p) && al16(g) && al16(m) && al16(v))
    hipLaunchKernelGGL(adam_k<4>, dim3(grid_for(n / 4)), dim3(256), 0, st, p, g, m, v, (unsigned)n, lr,
                       b1, b2, eps, bc1, bc2s, grad_scale);
  else
    hipLaunchKernelGGL(adam_k<1>, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, (unsigned)n, lr, b1,
                       b2, eps, bc1, bc2s, grad_scale);
  return SO_LAUNCH_CHECK();
}

int so_scale_add(const float* a, int lda, const float* alpha, const float* b, int ldb, float* out,
                 int ldo, long long rows, int C, void* stream) {
  if (rows * C <= 0) return 0;
  hipLaunchKernelGGL(scale_add_k, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, a, lda,
                     alpha, b, ldb, out, ldo, (unsigned)rows, (unsigned)C);
  return SO_LAUNCH_CHECK();
}

int so_blend_fwd(const float* a, int lda, const float* b, int ldb, const float* m, int ldm, float* out,
                 int ldo, long long rows, int C, void* stream) {
  if (rows * C <= 0) return 0;
  hipLaunchKernelGGL(blend_fwd_k, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, a, lda,
                     b, ldb, m, ldm, out, ldo, (unsigned)rows, (unsigned)C);
  return SO_LAUNCH_CHECK();
}

int so_blend_bwd(const float* a, int lda, const float* b, int ldb, const float* m, int ldm,
                 const float* g, int ldg, float* da, int ldda, float* db, int lddb, float* dm, int lddm,
                 long long rows, int C, void* stream) {
  if (rows * C <= 0) return 0;
  hipLaunchKernelGGL(blend_bwd_k, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                     ldb, m, ldm, g, ldg, da, ldda, db, lddb, dm, lddm, (unsigned)rows, (unsigned)C);
  return SO_LAUNCH_CHECK();
}

int so_fill(float* p, long long n, float val, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fill_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, (unsigned)n, val);
  return SO_LAUNCH_CHECK();
}

int so_axpby(const float* x, float a, float* y, float b, long long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(axpby_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, a, y, b,
                     (unsigned)n);
  return SO_LAUNCH_CHECK();
}

// ---- stream hand-off in the MIDDLE of a replayed hipGraph (data-parallel gradient exchange overlapped with backward) ----
// External event nodes are not available to PyTorch-ROCm graphs, so "this gradient bucket is complete" is signalled through
// memory: a one-thread kernel captured at that point of the backward pass stores the step counter into an 8-byte signal
// word, and the communication stream executes hipStreamWaitValue32(word >= step) - a command-processor wait that occupies
// no compute unit - before its all-reduce.  Measured (tools/probes/stream_wait_value.hip): the waiting stream is released
// 0.6 ms into a 6.3 ms graph, i.e. when the signal node runs, not when the graph ends.
long long so_signal_alloc(void) {
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess) return 0;
  if (hipMemset(p, 0, 8) != hipSuccess) { (void)hipFree(p); return 0; }
  return (long long)(uintptr_t)p;
}

int so_signal_free(long long ptr) { return ptr ? (int)hipFree((void*)(uintptr_t)ptr) : 0; }

int so_signal_can_wait(void) {
  int can = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) return 0;
  return can;
}

int so_counter_bump(void* counter, void* stream) {
  hipLaunchKernelGGL(counter_bump_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)counter);
  return SO_LAUNCH_CHECK();
}

int so_signal_store(void* flag, const void* counter, int system_scope, void* stream) {
  hipLaunchKernelGGL(signal_store_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)flag, (const unsigned*)counter,
                     system_scope);
  return SO_LAUNCH_CHECK();
}

// mode 0: hipStreamWaitValue32 (command-processor wait, no CU occupied).  mode 1: a one-lane kernel that polls the word with
// AGENT-scope relaxed loads (the matching so_signal_store releases at agent scope) and s_sleep between polls (one wave slot on one CU, nothing for the command processor to do).
// Measured on MI355X (bench.py --config c3, one rank): with the CP wait queued right after the graph launch every one of
// the ~500 kernels of the step dispatches ~1.5 us later (6.64 vs 5.87 ms/step) - the command processor polls the word for
// the whole step; the spin kernel does not touch the dispatch path.
int so_stream_wait_ge(void* flag, int value, int mode, void* stream) {
  if (mode == 0)
    return (int)hipStreamWaitValue32((hipStream_t)stream, flag, (uint32_t)value, hipStreamWaitValueGte, 0xFFFFFFFFu);
  hipLaunchKernelGGL(wait_flag_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)flag, (unsigned)value,
                     (const unsigned*)nullptr, (unsigned*)nullptr, 0ull);
  return SO_LAUNCH_CHECK();
}

// The polling wait with an escape: `words` = so_hostwords_alloc(): words[0] abort request (host writes 1), words[1] status
// (kernel writes 1 = aborted, 2 = deadline of `max_ticks` 100 MHz ticks passed).
int so_stream_wait_ge_bounded(void* flag, int value, void* words, long long max_ticks, void* stream) {
  if (!words || max_ticks <= 0) return SO_ERR_SHAPE;
  hipLaunchKernelGGL(wait_flag_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)flag, (unsigned)value,
                     (const unsigned*)words, (unsigned*)words + 1, (unsigned long long)max_ticks);
  return SO_LAUNCH_CHECK();
}

// two zeroed 32-bit words of pinned, device-mapped, coherent host memory; the returned address is valid on host and device
long long so_hostwords_alloc(void) {
  void* p = nullptr;
  if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return 0;
  for (int i = 0; i < 16; ++i) ((volatile unsigned*)p)[i] = 0u;
  return (long long)(uintptr_t)p;
}

int so_hostwords_free(long long ptr) { return ptr ? (int)hipHostFree((void*)(uintptr_t)ptr) : 0; }

}  // extern "C"
