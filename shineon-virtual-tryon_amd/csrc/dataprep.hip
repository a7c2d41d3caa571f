// Dataset-side tensor preparation and the inter-stage image wire format on the GPU (gfx950).
//
// The reference builds every training sample on the CPU, per item, with PIL / numpy / torchvision
// (datasets/tryon_dataset.py:109-121,226-229,272-298,323-448, datasets/util.py:6-22) and its own comment calls the
// pose rasterisation "very expensive" (tryon_dataset.py:391).  Here the raw inputs (uint8 images, the uint8 LIP parse
// map, the 18x3 keypoints, the .flo payload) are uploaded once and every derived tensor is produced by the kernels
// below, whole batches at a time.  All of it is HBM-bound byte/integer work: one thread per pixel, coalesced planar
// stores, no LDS needed (the only reduction is the 33-tap PIL down-sampling window, re-read from L2).
//
// Bit-exactness: this file is compiled with -ffp-contract=off; every value is produced by the same sequence of
// IEEE operations as the reference (torchvision ToTensor/Normalize in fp32, Pillow's resampling coefficients in
// fp64 and its 22-bit fixed-point accumulation, Pillow's (int) truncation of rectangle corners).
#include "common.h"
#include "../../include/shineon_hip.h"

namespace {

__device__ __forceinline__ float normed(unsigned char u) {
  // transforms.ToTensor: float32(u) / 255 ; transforms.Normalize(0.5, 0.5): (t - 0.5) / 0.5
  return __fdiv_rn(__fsub_rn(__fdiv_rn((float)u, 255.0f), 0.5f), 0.5f);
}

// visualization.py:73-77: ((t + 1) * 0.5 * 255).clamp(0, 255) truncated to uint8
__global__ __launch_bounds__(256) void quantize_u8_k(const float* __restrict__ src, int ld, int chw,
                                                     unsigned char* __restrict__ dst, int C, int HW, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // pixel index over N*HW
  if (i >= total) return;
  const long long n = i / HW;
  const int pix = (int)(i - n * HW);
  for (int c = 0; c < C; ++c) {
    const float x = chw ? src[(n * C + c) * HW + pix] : src[i * ld + c];
    float v = __fmul_rn(__fmul_rn(__fadd_rn(x, 1.0f), 0.5f), 255.0f);
    v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);  // NaN -> 255 side never occurs for finite inputs
    dst[i * C + c] = (unsigned char)v;
  }
}

__global__ __launch_bounds__(256) void u8_to_normed_k(const unsigned char* __restrict__ src, int Cs,
                                                      float* __restrict__ dst, int C, int HW, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long n = i / HW;
  const int pix = (int)(i - n * HW);
  for (int c = 0; c < C; ++c) dst[(n * C + c) * HW + pix] = normed(src[i * Cs + c]);
}

// One-hot pose maps: P x H x W planes of -1 with an inclusive square [trunc(x-r), trunc(x+r)] x [trunc(y-r), trunc(y+r)]
// of +1 around every keypoint with x > 1 and y > 1 (PIL ImageDraw.rectangle truncates its float corners with (int)).
// draw_into_map = 0 reproduces the reference AS WRITTEN: tryon_dataset.py:417-424 converts the blank image to a
// tensor BEFORE drawing on it, so its pose_map planes stay -1 and only the im_cocopose visual carries the squares.
__global__ __launch_bounds__(256) void pose_map_k(const double* __restrict__ kp, float* __restrict__ pose_map,
                                                  float* __restrict__ im_pose, int P, int H, int W, int radius,
                                                  int draw_into_map, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over N*H*W
  if (i >= total) return;
  const int HW = H * W;
  const long long n = i / HW;
  const int pix = (int)(i - n * HW);
  const int y = pix / W, x = pix - y * W;
  bool any = false;
  // a sample without a detected person (pose_data is None in the reference) is passed as NaN keypoints: its planes keep
  // the zeros they were allocated with (tryon_dataset.py:403-405), the visual stays blank (-1)
  const bool none = kp[n * P * 3] != kp[n * P * 3];
  if (none) {
    for (int p = 0; p < P && pose_map; ++p) pose_map[(n * P + p) * HW + pix] = 0.0f;
    if (im_pose) im_pose[n * HW + pix] = -1.0f;
    return;
  }
  for (int p = 0; p < P; ++p) {
    const double px = kp[(n * P + p) * 3 + 0], py = kp[(n * P + p) * 3 + 1];
    bool in = false;
    if (px > 1.0 && py > 1.0) {
      int x0 = (int)(px - radius), x1 = (int)(px + radius), y0 = (int)(py - radius), y1 = (int)(py + radius);
      if (x0 > x1) { const int t = x0; x0 = x1; x1 = t; }
      if (y0 > y1) { const int t = y0; y0 = y1; y1 = t; }
      in = x >= x0 && x <= x1 && y >= y0 && y <= y1;
    }
    any |= in;
    if (pose_map) pose_map[(n * P + p) * HW + pix] = (in && draw_into_map) ? 1.0f : -1.0f;
  }
  if (im_pose) im_pose[n * HW + pix] = any ? 1.0f : -1.0f;
}

// im_head = im * m_head - (1 - m_head), im_cloth = im * m_cloth + (1 - m_cloth) with the masks taken from the LIP
// parse labels (bit `label` of head_bits / cloth_bits); shape = (label > 0) * 255 for the silhouette passes.
__global__ __launch_bounds__(256) void parse_compose_k(const unsigned char* __restrict__ parse,
                                                       const float* __restrict__ image, float* __restrict__ im_head,
                                                       long long s_head, float* __restrict__ im_cloth, long long s_cloth,
                                                       unsigned char* __restrict__ shape, unsigned head_bits,
                                                       unsigned cloth_bits, int HW, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long n = i / HW;
  const int pix = (int)(i - n * HW);
  const unsigned label = parse[i];
  const float mh = (label < 32 && ((head_bits >> label) & 1u)) ? 1.0f : 0.0f;
  const float mc = (label < 32 && ((cloth_bits >> label) & 1u)) ? 1.0f : 0.0f;
  for (int c = 0; c < 3; ++c) {
    const float v = image[(n * 3 + c) * HW + pix];
    if (im_head) im_head[n * s_head + (long long)c * HW + pix] = __fsub_rn(__fmul_rn(v, mh), __fsub_rn(1.0f, mh));
    if (im_cloth) im_cloth[n * s_cloth + (long long)c * HW + pix] = __fadd_rn(__fmul_rn(v, mc), __fsub_rn(1.0f, mc));
  }
  if (shape) shape[i] = label > 0 ? 255 : 0;
}

// One separable pass of Pillow's ImagingResample (8 bits per channel, BILINEAR) along x (horizontal = 1) or y.
//   scale = in / out; filterscale = max(scale, 1); support = 1.0 * filterscale
//   center = (o + 0.5) * scale; xmin = max(0, (int)(center - support + 0.5)); xmax = min(in, (int)(center + support + 0.5))
//   w_x = tri((x + xmin - center + 0.5) / filterscale), normalised by their sum, -> 22-bit fixed point, round half away
//   out = clip8((2^21 + sum_x pixel_x * k_x) >> 22)
// (Pillow src/libImaging/Resample.c: bilinear_filter, precompute_coeffs, normalize_coeffs_8bpc, ImagingResample*_8bpc).
// out_f != nullptr: the last pass writes ToTensor + Normalize(0.5, 0.5) of the byte instead of the byte.
#define SO_PRECISION_BITS 22
__global__ __launch_bounds__(256) void pil_bilinear_pass_k(const unsigned char* __restrict__ src,
                                                           unsigned char* __restrict__ dst, float* __restrict__ out_f,
                                                           long long s_out, int in_w, int in_h, int out_w, int out_h,
                                                           int horizontal, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over N * out_h * out_w
  if (i >= total) return;
  const int ohw = out_h * out_w;
  const long long n = i / ohw;
  const int opix = (int)(i - n * ohw);
  const int oy = opix / out_w, ox = opix - oy * out_w;
  const int in_size = horizontal ? in_w : in_h, out_size = horizontal ? out_w : out_h;
  const int o = horizontal ? ox : oy;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = (o + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double t = (x + xmin - center + 0.5) * ss;
    if (t < 0.0) t = -t;
    ww += t < 1.0 ? 1.0 - t : 0.0;
  }
  const unsigned char* base = src + n * (long long)in_w * in_h;
  int acc = 1 << (SO_PRECISION_BITS - 1);
  for (int x = 0; x < xmax; ++x) {
    double t = (x + xmin - center + 0.5) * ss;
    if (t < 0.0) t = -t;
    double w = t < 1.0 ? 1.0 - t : 0.0;
    if (ww != 0.0) w /= ww;
    const int k = (int)(w < 0.0 ? w * (double)(1 << SO_PRECISION_BITS) - 0.5 : w * (double)(1 << SO_PRECISION_BITS) + 0.5);
    const int pixel = horizontal ? base[oy * in_w + (x + xmin)] : base[(x + xmin) * in_w + ox];
    acc += pixel * k;
  }
  int v = acc >> SO_PRECISION_BITS;
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  if (out_f) out_f[n * s_out + opix] = normed((unsigned char)v);
  else dst[i] = (unsigned char)v;
}

// .flo payload [H][W][2] (u, v interleaved, Middlebury) -> planar (2, H, W), then transforms.Normalize(0.5, 0.5)
__global__ __launch_bounds__(256) void flow_decode_k(const float* __restrict__ payload, float* __restrict__ flow, int HW,
                                                     long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over N*HW
  if (i >= total) return;
  const long long n = i / HW;
  const int pix = (int)(i - n * HW);
  for (int c = 0; c < 2; ++c)
    flow[(n * 2 + c) * HW + pix] = __fdiv_rn(__fsub_rn(payload[i * 2 + c], 0.5f), 0.5f);
}

// get_input_cloth_mask (tryon_dataset.py:168-175): mask = (cloth[channel 0] >= threshold) ? 0 : 1, one channel.
__global__ __launch_bounds__(256) void threshold_mask_k(const float* __restrict__ x, int C, float threshold,
                                                        float* __restrict__ mask, int HW, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over N*HW
  if (i >= total) return;
  const long long n = i / HW;
  const int pix = (int)(i - n * HW);
  mask[i] = x[n * C * HW + pix] >= threshold ? 0.0f : 1.0f;
}

inline unsigned blocks_for(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" {

int so_quantize_u8(const float* src, int ld, int chw, void* dst, int Nb, int C, int HW, void* stream) {
  const long long total = (long long)Nb * HW;
  if (total <= 0 || C <= 0) return 0;
  if (!chw && ld < C) return SO_ERR_SHAPE;
  hipLaunchKernelGGL(quantize_u8_k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, ld, chw,
                     (unsigned char*)dst, C, HW, total);
  return SO_LAUNCH_CHECK();
}

int so_u8_to_normed(const void* src, int Cs, float* dst, int Nb, int C, int HW, void* stream) {
  const long long total = (long long)Nb * HW;
  if (total <= 0 || C <= 0) return 0;
  if (Cs < C) return SO_ERR_SHAPE;
  hipLaunchKernelGGL(u8_to_normed_k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src, Cs, dst, C, HW, total);
  return SO_LAUNCH_CHECK();
}

int so_pose_map(const double* keypoints, float* pose_map, float* im_pose, int Nb, int P, int H, int W, int radius,
                int draw_into_map, void* stream) {
  const long long total = (long long)Nb * H * W;
  if (total <= 0 || P <= 0) return 0;
  hipLaunchKernelGGL(pose_map_k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, keypoints, pose_map, im_pose,
                     P, H, W, radius, draw_into_map, total);
  return SO_LAUNCH_CHECK();
}

int so_parse_compose(const void* parse, const float* image, float* im_head, long long stride_head, float* im_cloth,
                     long long stride_cloth, void* shape_u8, int head_bits, int cloth_bits, int Nb, int HW, void* stream) {
  const long long total = (long long)Nb * HW;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(parse_compose_k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)parse, image, im_head, stride_head, im_cloth, stride_cloth,
                     (unsigned char*)shape_u8, (unsigned)head_bits, (unsigned)cloth_bits, HW, total);
  return SO_LAUNCH_CHECK();
}

long long so_silhouette_ws_bytes(int Nb, int H, int W, int factor) {
  const long long w2 = W / factor, h2 = H / factor;
  return (long long)Nb * (w2 * H + w2 * h2 + (long long)W * h2);
}

int so_silhouette(const void* shape_u8, float* silhouette, long long stride_out, void* ws, int Nb, int H, int W, int factor,
                  void* stream) {
  if (Nb <= 0) return 0;
  const int w2 = W / factor, h2 = H / factor;
  if (w2 <= 0 || h2 <= 0) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* a = (unsigned char*)ws;                     // [N][H][w2]   after the horizontal down pass
  unsigned char* b = a + (long long)Nb * w2 * H;             // [N][h2][w2]  after the vertical down pass
  unsigned char* c = b + (long long)Nb * w2 * h2;            // [N][h2][W]   after the horizontal up pass
  long long t = (long long)Nb * H * w2;
  hipLaunchKernelGGL(pil_bilinear_pass_k, dim3(blocks_for(t)), dim3(256), 0, st, (const unsigned char*)shape_u8, a,
                     (float*)nullptr, 0LL, W, H, w2, H, 1, t);
  t = (long long)Nb * h2 * w2;
  hipLaunchKernelGGL(pil_bilinear_pass_k, dim3(blocks_for(t)), dim3(256), 0, st, (const unsigned char*)a, b,
                     (float*)nullptr, 0LL, w2, H, w2, h2, 0, t);
  t = (long long)Nb * h2 * W;
  hipLaunchKernelGGL(pil_bilinear_pass_k, dim3(blocks_for(t)), dim3(256), 0, st, (const unsigned char*)b, c,
                     (float*)nullptr, 0LL, w2, h2, W, h2, 1, t);
  t = (long long)Nb * H * W;
  hipLaunchKernelGGL(pil_bilinear_pass_k, dim3(blocks_for(t)), dim3(256), 0, st, (const unsigned char*)c,
                     (unsigned char*)nullptr, silhouette, stride_out, W, h2, W, H, 0, t);
  return SO_LAUNCH_CHECK();
}

int so_threshold_mask(const float* x, int C, float threshold, float* mask, int Nb, int HW, void* stream) {
  const long long total = (long long)Nb * HW;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(threshold_mask_k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, C, threshold, mask, HW,
                     total);
  return SO_LAUNCH_CHECK();
}

int so_flow_decode(const float* payload, float* flow, int Nb, int HW, void* stream) {
  const long long total = (long long)Nb * HW;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(flow_decode_k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, payload, flow, HW, total);
  return SO_LAUNCH_CHECK();
}

}  // extern "C"
