// Geometric-matching (GMM / WarpModel) kernels and the small row-wise kernels of the SAGAN attention.
//
// Reference ops restated (file:line in /root/reference):
//   FeatureL2Norm            models/networks/cpvton/warp.py:39-50   f / sqrt(sum_c f^2 + 1e-6)
//   FeatureCorrelation       models/networks/cpvton/warp.py:53-67   (A transposed h<->w; the matmul itself is
//                            so_gemm_batched in igemm.hip, this file supplies the transposed write of A)
//   FeatureRegression.linear + tanh   models/networks/cpvton/warp.py:87,94-99 (flatten order is C,H,W)
//   TpsGridGen.apply_transformation    models/networks/cpvton/warp.py:191-318
//   F.grid_sample (bilinear, align_corners=False, padding border|zeros)   models/warp_model.py:85-86,143-145
//   Resample2d (flownet2 submodule, absent from the reference tree)       models/unet_mask_model.py:115-117
//   nn.Softmax(dim=-1) of the attention energies                          models/networks/attention/sagan.py:27,45
#include "common.h"
#include "../../include/shineon_hip.h"

namespace {

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ FeatureL2Norm
// One wavefront per pixel row.  transpose_hw: the output row of pixel (h, w) is w*H + h (the
// `.transpose(2,3).contiguous()` that FeatureCorrelation applies to feature A).
__global__ __launch_bounds__(256) void l2norm_fwd_k(const float* __restrict__ x, int ldx,
                                                    float* __restrict__ y, int ldy,
                                                    float* __restrict__ inv, unsigned Nb, unsigned H,
                                                    unsigned W, unsigned C, int transpose_hw) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned rows = Nb * H * W;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (unsigned c = lane; c < C; c += 64) { const float v = xr[c]; s += v * v; }
  s = so_wave_sum(s);
  const float r = 1.0f / sqrtf(s + 1e-6f);
  unsigned orow = row;
  if (transpose_hw) {
    const unsigned n = row / (H * W), rem = row - n * H * W;
    const unsigned h = rem / W, w = rem - h * W;
    orow = n * H * W + w * H + h;
  }
  float* yr = y + (size_t)orow * ldy;
  for (unsigned c = lane; c < C; c += 64) yr[c] = xr[c] * r;
  if (lane == 0) inv[row] = r;
}

// dx = (dy - y * <dy, y>) * inv      (y, dy read at the possibly transposed row)
__global__ __launch_bounds__(256) void l2norm_bwd_k(const float* __restrict__ y, int ldy,
                                                    const float* __restrict__ dy, int lddy,
                                                    const float* __restrict__ inv,
                                                    float* __restrict__ dx, int lddx, unsigned Nb,
                                                    unsigned H, unsigned W, unsigned C,
                                                    int transpose_hw) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned rows = Nb * H * W;
  if (row >= rows) return;
  unsigned orow = row;
  if (transpose_hw) {
    const unsigned n = row / (H * W), rem = row - n * H * W;
    const unsigned h = rem / W, w = rem - h * W;
    orow = n * H * W + w * H + h;
  }
  const float* yr = y + (size_t)orow * ldy;
  const float* gr = dy + (size_t)orow * lddy;
  float d = 0.f;
  for (unsigned c = lane; c < C; c += 64) d += yr[c] * gr[c];
  d = so_wave_sum(d);
  const float r = inv[row];
  float* o = dx + (size_t)row * lddx;
  for (unsigned c = lane; c < C; c += 64) o[c] = (gr[c] - yr[c] * d) * r;
}

// ------------------------------------------------------------------ row softmax (attention)
__global__ __launch_bounds__(256) void softmax_rows_fwd_k(const float* __restrict__ e, int lde,
                                                          float* __restrict__ a, int lda,
                                                          unsigned rows, unsigned Ncol) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* er = e + (size_t)row * lde;
  float mx = -INFINITY;
  for (unsigned c = lane; c < Ncol; c += 64) mx = fmaxf(mx, er[c]);
  mx = so_wave_max(mx);
  float s = 0.f;
  for (unsigned c = lane; c < Ncol; c += 64) s += expf(er[c] - mx);
  s = so_wave_sum(s);
  const float r = 1.0f / s;
  float* ar = a + (size_t)row * lda;
  for (unsigned c = lane; c < Ncol; c += 64) ar[c] = expf(er[c] - mx) * r;
}

// dE = A * (dA - sum_j dA_j A_j)
__global__ __launch_bounds__(256) void softmax_rows_bwd_k(const float* __restrict__ a, int lda,
                                                          const float* __restrict__ da, int ldda,
                                                          float* __restrict__ de, int ldde,
                                                          unsigned rows, unsigned Ncol) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* ar = a + (size_t)row * lda;
  const float* gr = da + (size_t)row * ldda;
  float d = 0.f;
  for (unsigned c = lane; c < Ncol; c += 64) d += ar[c] * gr[c];
  d = so_wave_sum(d);
  float* o = de + (size_t)row * ldde;
  for (unsigned c = lane; c < Ncol; c += 64) o[c] = ar[c] * (gr[c] - d);
}

// ------------------------------------------------------------------ dot product (attention d gamma)
// d gamma = <dout, o> is a sum of ~4e5 products of either sign that cancels to ~1e-3 of its absolute mass: accumulated
// in fp32 its round-off is ~10 % of the result.  The products and the sums are therefore carried in fp64 (exact products,
// fixed summation order); the fp64 pipes are idle on this path anyway.
__global__ __launch_bounds__(256) void dot_partial_k(const float* __restrict__ a, int lda,
                                                     const float* __restrict__ b, int ldb,
                                                     unsigned rows, unsigned C,
                                                     double* __restrict__ part) {
  __shared__ double red[4];
  const unsigned total = rows * C;
  double s = 0.0;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / C, c = idx - row * C;
    s += (double)a[(size_t)row * lda + c] * (double)b[(size_t)row * ldb + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void dot_final_k(const double* __restrict__ part, unsigned n, float scale,
                                                   float* __restrict__ out, int accumulate) {
  __shared__ double red[4];
  double s = 0.0;
  for (unsigned i = threadIdx.x; i < n; i += 256) s += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = (accumulate ? out[0] : 0.f) + (float)((double)scale * t);
  }
}

__global__ __launch_bounds__(256) void sum_final_k(const float* __restrict__ part, unsigned n,
                                                   float scale, float* __restrict__ out, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (unsigned i = threadIdx.x; i < n; i += 256) s += part[i];
  s = so_block_sum256(s, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * s;
}

// ------------------------------------------------------------------ regression head: Linear(C*P -> J) + tanh
// x: NHWC rows [b][p][c] (ldx); weight [J][C*P] with the reference's flatten order idx = c*P + p.
__global__ __launch_bounds__(256) void linear_chw_fwd_k(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, unsigned P, unsigned C,
                                                        unsigned J, int apply_tanh) {
  __shared__ float red[4];
  const unsigned b = blockIdx.y, j = blockIdx.x;
  const unsigned K = P * C;
  float s = 0.f;
  for (unsigned t = threadIdx.x; t < K; t += 256) {
    const unsigned p = t / C, c = t - p * C;  // NHWC order -> coalesced x reads
    s += x[((size_t)b * P + p) * ldx + c] * w[(size_t)j * K + c * P + p];
  }
  s = so_block_sum256(s, red);
  if (threadIdx.x == 0) {
    s += bias[j];
    y[(size_t)b * J + j] = apply_tanh ? tanhf(s) : s;
  }
}

// dz = dy * (1 - y^2) (if tanh) ; dw[j][c*P+p] = sum_b dz[b][j] x[b][p][c] ; dbias[j] = sum_b dz[b][j]
__global__ __launch_bounds__(256) void linear_chw_bwd_w_k(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ dw,
                                                          float* __restrict__ dbias, unsigned Nb,
                                                          unsigned P, unsigned C, unsigned J,
                                                          int apply_tanh) {
  const unsigned K = P * C;
  const unsigned total = J * K;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned j = idx / K, k = idx - j * K;
    const unsigned c = k / P, p = k - c * P;
    float s = 0.f, sb = 0.f;
    for (unsigned b = 0; b < Nb; ++b) {
      const float yv = y[(size_t)b * J + j];
      const float dz = dy[(size_t)b * J + j] * (apply_tanh ? (1.f - yv * yv) : 1.f);
      s += dz * x[((size_t)b * P + p) * ldx + c];
      sb += dz;
    }
    dw[idx] = s;
    if (k == 0) dbias[j] = sb;
  }
}

// dx[b][p][c] = sum_j dz[b][j] w[j][c*P+p]
__global__ __launch_bounds__(256) void linear_chw_bwd_x_k(const float* __restrict__ w,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ dx, int lddx,
                                                          unsigned Nb, unsigned P, unsigned C,
                                                          unsigned J, int apply_tanh) {
  const unsigned K = P * C;
  const unsigned total = Nb * K;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    // consecutive threads = consecutive p of one channel: the J weight rows are read along their contiguous (c * P + p) axis
    // (with c fastest every one of the J loads of a thread was a 4-byte access at a P-float stride); the single store pays
    const unsigned b = idx / K, t = idx - b * K;
    const unsigned c = t / P, p = t - c * P;
    float s = 0.f;
#pragma unroll 10
    for (unsigned j = 0; j < J; ++j) {   // (unrolled: ten rows' loads in flight per round; the sum keeps its order)
      const float yv = y[(size_t)b * J + j];
      const float dz = dy[(size_t)b * J + j] * (apply_tanh ? (1.f - yv * yv) : 1.f);
      s += dz * w[(size_t)j * K + c * P + p];
    }
    dx[((size_t)b * P + p) * lddx + c] = s;
  }
}

// ------------------------------------------------------------------ thin-plate-spline grid
// The TPS map amplifies rounding: Li (an fp32 LAPACK inverse, |entries| up to ~10) times (theta + P_base) cancels to the
// small non-linear weights W, and sum_k W_k U_k sums 25 terms of either sign with U up to ~17.  Evaluated in fp32 the grid
// lands ~2.5e-5 from the exact value of the reference's formula (the reference's own fp32 evaluation: ~9e-6), which the
// bilinear sampler turns into 1.4e-4 on the warped cloth - outside the 1e-4 the path is held to.  Both kernels therefore
// evaluate the reference's formula (warp.py:231-318, same operands: the fp32 constants Li, P, grid_X / grid_Y and theta)
// in FP64 and round once at the end: the result is the correctly rounded value any fp32 evaluation approximates, so the
// distance to the reference's CPU numbers is the reference's own rounding error.  Cost: 25 fp64 logs per PIXEL - U does not
// depend on theta, so each thread computes its 25 basis values once and reuses them for every sample.
//
// coef[b][axis][i], i < NP+3:  Li[i][0:NP] . (theta[b][axis*NP + k] + Pbase[axis][k])          (fp64)
// rows 0..NP-1 are the non-linear weights W, rows NP..NP+2 the affine part A (warp.py:231-268).
__global__ __launch_bounds__(64) void tps_coef_k(const float* __restrict__ theta,
                                                 const float* __restrict__ Li,
                                                 const float* __restrict__ px,
                                                 const float* __restrict__ py,
                                                 double* __restrict__ coef, int NP) {
  const int b = blockIdx.x;
  const int L = NP + 3;
  for (int t = threadIdx.x; t < 2 * L; t += 64) {
    const int axis = t / L, i = t - axis * L;
    const float* base = axis == 0 ? px : py;
    double s = 0.0;
    for (int k = 0; k < NP; ++k)
      s += (double)Li[i * L + k] * ((double)theta[(size_t)b * 2 * NP + axis * NP + k] + (double)base[k]);
    coef[((size_t)b * 2 + axis) * L + i] = s;
  }
}

__device__ __forceinline__ float tps_u(float x, float y, float pxk, float pyk) {
  const float dx = x - pxk, dy = y - pyk;
  float d2 = dx * dx + dy * dy;
  if (d2 == 0.f) d2 = 1.f;  // warp.py:288 (avoid log(0)); U = 0 there
  return d2 * logf(d2);
}

__device__ __forceinline__ double tps_u64(double x, double y, double pxk, double pyk) {
  const double dx = x - pxk, dy = y - pyk;
  double d2 = dx * dx + dy * dy;
  if (d2 == 0.0) d2 = 1.0;
  return d2 * log(d2);
}

// grid[b][h][w][0:2] = A0 + A1 x + A2 y + sum_k W_k U_k      (warp.py:304-318), fp64 inside, one rounding
constexpr int TPS_BG = 4;     // samples per block (blockIdx.y walks groups of TPS_BG samples)
constexpr int TPS_MAXNP = 61;
__global__ __launch_bounds__(256) void tps_grid_fwd_k(const double* __restrict__ coef,
                                                      const float* __restrict__ gx,
                                                      const float* __restrict__ gy,
                                                      const float* __restrict__ px,
                                                      const float* __restrict__ py,
                                                      float* __restrict__ grid, unsigned Nb, unsigned H,
                                                      unsigned W, int NP) {
  extern __shared__ double shd[];  // [TPS_BG][2*L] coef
  const int L = NP + 3;
  const unsigned b0 = blockIdx.y * TPS_BG;
  const unsigned nb = min((unsigned)TPS_BG, Nb - b0);
  for (unsigned t = threadIdx.x; t < nb * 2 * L; t += 256) shd[t] = coef[(size_t)b0 * 2 * L + t];
  __syncthreads();
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= H * W) return;
  const unsigned h = pix / W, w = pix - h * W;
  const double x = (double)gx[w], y = (double)gy[h];
  double sx[TPS_BG], sy[TPS_BG];
#pragma unroll
  for (int j = 0; j < TPS_BG; ++j) { sx[j] = 0.0; sy[j] = 0.0; }
  for (int k = 0; k < NP; ++k) {
    const double u = tps_u64(x, y, (double)px[k], (double)py[k]);
#pragma unroll
    for (int j = 0; j < TPS_BG; ++j) {
      if ((unsigned)j < nb) {
        sx[j] += shd[(j * 2 + 0) * L + k] * u;
        sy[j] += shd[(j * 2 + 1) * L + k] * u;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TPS_BG; ++j) {
    if ((unsigned)j < nb) {
      const double* cx = shd + (j * 2 + 0) * L;
      const double* cy = shd + (j * 2 + 1) * L;
      const double ox = cx[NP] + cx[NP + 1] * x + cx[NP + 2] * y + sx[j];
      const double oy = cy[NP] + cy[NP + 1] * x + cy[NP + 2] * y + sy[j];
      *reinterpret_cast<float2*>(grid + ((size_t)(b0 + j) * H * W + pix) * 2) = make_float2((float)ox, (float)oy);
    }
  }
}

// partial[b][blk][axis][i] = sum over the block's pixels of dgrid[b][pix][axis] * basis_i(pix)
// Basis functions are processed sixteen at a time in registers; each value is reduced inside its wavefront with
// shuffles and the four wave totals meet in LDS behind ONE barrier (the first version paid two block reductions per
// basis function: 112 barriers per block, 61 us per launch).
constexpr int TPS_PPT = 2;
__global__ __launch_bounds__(256) void tps_grid_bwd_partial_k(const float* __restrict__ dgrid,
                                                              const float* __restrict__ gx,
                                                              const float* __restrict__ gy,
                                                              const float* __restrict__ px,
                                                              const float* __restrict__ py,
                                                              float* __restrict__ part, unsigned H,
                                                              unsigned W, int NP) {
  __shared__ float red[4][2 * 64];
  __shared__ float spx[64], spy[64];
  const int L = NP + 3;
  const unsigned b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = threadIdx.x; t < NP; t += 256) { spx[t] = px[t]; spy[t] = py[t]; }
  __syncthreads();
  const unsigned HW = H * W;
  const unsigned p0 = blockIdx.x * 256u * TPS_PPT;
  float xs[TPS_PPT], ys[TPS_PPT];
  float2 gs[TPS_PPT];
#pragma unroll
  for (int q = 0; q < TPS_PPT; ++q) {
    const unsigned pix = p0 + q * 256u + threadIdx.x;
    xs[q] = 0.f; ys[q] = 0.f; gs[q] = make_float2(0.f, 0.f);  // zero gradient: contributes nothing
    if (pix < HW) {
      const unsigned h = pix / W, w = pix - h * W;
      xs[q] = gx[w]; ys[q] = gy[h];
      gs[q] = *reinterpret_cast<const float2*>(dgrid + ((size_t)b * HW + pix) * 2);
    }
  }
  for (int i0 = 0; i0 < L; i0 += 16) {
    float ax[16], ay[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u;
      float sx = 0.f, sy = 0.f;
      if (i < L) {
#pragma unroll
        for (int q = 0; q < TPS_PPT; ++q) {
          float basis;
          if (i < NP) basis = tps_u(xs[q], ys[q], spx[i], spy[i]);
          else if (i == NP) basis = 1.f;
          else if (i == NP + 1) basis = xs[q];
          else basis = ys[q];
          sx += gs[q].x * basis;
          sy += gs[q].y * basis;
        }
      }
      ax[u] = so_wave_sum(sx);
      ay[u] = so_wave_sum(sy);
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (i0 + u < L) { red[wave][i0 + u] = ax[u]; red[wave][L + i0 + u] = ay[u]; }
    }
  }
  __syncthreads();
  float* out = part + ((size_t)b * gridDim.x + blockIdx.x) * 2 * L;
  for (int t = threadIdx.x; t < 2 * L; t += 256) out[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// dtheta[b][axis*NP + k] = sum_i Li[i][k] * (sum_blk partial[b][blk][axis][i])
__global__ __launch_bounds__(64) void tps_grid_bwd_final_k(const float* __restrict__ part,
                                                           unsigned nblk,
                                                           const float* __restrict__ Li,
                                                           float* __restrict__ dtheta, int NP) {
  __shared__ float G[2 * 64];
  const int L = NP + 3;
  const unsigned b = blockIdx.x;
  for (int t = threadIdx.x; t < 2 * L; t += 64) {
    float s = 0.f;
#pragma unroll 8
    for (unsigned k = 0; k < nblk; ++k) s += part[((size_t)b * nblk + k) * 2 * L + t];
    G[t] = s;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * NP; t += 64) {
    const int axis = t / NP, k = t - axis * NP;
    float s = 0.f;
    for (int i = 0; i < L; ++i) s += Li[i * L + k] * G[axis * L + i];
    dtheta[(size_t)b * 2 * NP + t] = s;
  }
}

// ------------------------------------------------------------------ grid_sample (bilinear, align_corners=False)
// Index arithmetic follows ATen's CPU kernel (GridSamplerKernel.cpp): ix = (x + 1) * (W / 2) - 0.5 with one
// rounding per operation (no FMA contraction), border padding clamps the coordinate BEFORE floor.
struct GsTap {
  int x0, y0;          // north-west tap
  float wx, wy;        // fractional weights (east / south)
  bool gx_in, gy_in;   // coordinate strictly inside (0, max): gradient mask of the clamp
};

__device__ __forceinline__ GsTap gs_locate(float gxv, float gyv, int W, int H, int border) {
  GsTap t;
  float ix = __fsub_rn(__fmul_rn(__fadd_rn(gxv, 1.0f), (float)W * 0.5f), 0.5f);
  float iy = __fsub_rn(__fmul_rn(__fadd_rn(gyv, 1.0f), (float)H * 0.5f), 0.5f);
  t.gx_in = true; t.gy_in = true;
  if (border) {
    const float mx = (float)(W - 1), my = (float)(H - 1);
    ix = fminf(mx, fmaxf(0.f, ix));  // NaN -> 0, like clamp_min(0, in)
    iy = fminf(my, fmaxf(0.f, iy));
    t.gx_in = (ix != 0.f) && (ix != mx);
    t.gy_in = (iy != 0.f) && (iy != my);
  }
  const float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx; t.y0 = (int)fy;
  t.wx = __fsub_rn(ix, fx);
  t.wy = __fsub_rn(iy, fy);
  return t;
}

// input [B][C][H][W] planar, grid [B][Ho][Wo][2], out [B][C][Ho][Wo]; taps (optional) [B][Ho][Wo][2] int32 = (x0,y0)
__global__ __launch_bounds__(256) void grid_sample_fwd_k(const float* __restrict__ in,
                                                         const float* __restrict__ grid,
                                                         float* __restrict__ out,
                                                         int* __restrict__ taps, unsigned Nb,
                                                         unsigned C, unsigned H, unsigned W,
                                                         unsigned Ho, unsigned Wo, int border) {
  const unsigned total = Nb * Ho * Wo;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned b = idx / (Ho * Wo), pix = idx - b * Ho * Wo;
    const float2 g = *reinterpret_cast<const float2*>(grid + (size_t)idx * 2);
    const GsTap t = gs_locate(g.x, g.y, (int)W, (int)H, border);
    if (taps) { taps[(size_t)idx * 2] = t.x0; taps[(size_t)idx * 2 + 1] = t.y0; }
    const float e = 1.f - t.wx, s = 1.f - t.wy;
    const float nw = s * e, ne = s * t.wx, sw = t.wy * e, se = t.wy * t.wx;
    const bool x0ok = (unsigned)t.x0 < W, x1ok = (unsigned)(t.x0 + 1) < W;
    const bool y0ok = (unsigned)t.y0 < H, y1ok = (unsigned)(t.y0 + 1) < H;
    for (unsigned c = 0; c < C; ++c) {
      const float* p = in + ((size_t)b * C + c) * H * W;
      const float vnw = (x0ok && y0ok) ? p[(size_t)t.y0 * W + t.x0] : 0.f;
      const float vne = (x1ok && y0ok) ? p[(size_t)t.y0 * W + t.x0 + 1] : 0.f;
      const float vsw = (x0ok && y1ok) ? p[(size_t)(t.y0 + 1) * W + t.x0] : 0.f;
      const float vse = (x1ok && y1ok) ? p[(size_t)(t.y0 + 1) * W + t.x0 + 1] : 0.f;
      out[((size_t)b * C + c) * Ho * Wo + pix] = vnw * nw + vne * ne + vsw * sw + vse * se;
    }
  }
}

// d grid (and optionally d input through atomics).  dgrid[B][Ho][Wo][2].
__global__ __launch_bounds__(256) void grid_sample_bwd_k(const float* __restrict__ in,
                                                         const float* __restrict__ grid,
                                                         const float* __restrict__ dout,
                                                         float* __restrict__ dgrid,
                                                         float* __restrict__ din, unsigned Nb,
                                                         unsigned C, unsigned H, unsigned W,
                                                         unsigned Ho, unsigned Wo, int border) {
  const unsigned total = Nb * Ho * Wo;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned b = idx / (Ho * Wo), pix = idx - b * Ho * Wo;
    const float2 g = *reinterpret_cast<const float2*>(grid + (size_t)idx * 2);
    const GsTap t = gs_locate(g.x, g.y, (int)W, (int)H, border);
    const float e = 1.f - t.wx, s = 1.f - t.wy;
    const bool x0ok = (unsigned)t.x0 < W, x1ok = (unsigned)(t.x0 + 1) < W;
    const bool y0ok = (unsigned)t.y0 < H, y1ok = (unsigned)(t.y0 + 1) < H;
    float gxa = 0.f, gya = 0.f;
    for (unsigned c = 0; c < C; ++c) {
      const float go = dout[((size_t)b * C + c) * Ho * Wo + pix];
      const float* p = in + ((size_t)b * C + c) * H * W;
      const float vnw = (x0ok && y0ok) ? p[(size_t)t.y0 * W + t.x0] : 0.f;
      const float vne = (x1ok && y0ok) ? p[(size_t)t.y0 * W + t.x0 + 1] : 0.f;
      const float vsw = (x0ok && y1ok) ? p[(size_t)(t.y0 + 1) * W + t.x0] : 0.f;
      const float vse = (x1ok && y1ok) ? p[(size_t)(t.y0 + 1) * W + t.x0 + 1] : 0.f;
      gxa += go * ((vne - vnw) * s + (vse - vsw) * t.wy);
      gya += go * ((vsw - vnw) * e + (vse - vne) * t.wx);
      if (din) {
        float* q = din + ((size_t)b * C + c) * H * W;
        if (x0ok && y0ok) atomicAdd(q + (size_t)t.y0 * W + t.x0, go * s * e);
        if (x1ok && y0ok) atomicAdd(q + (size_t)t.y0 * W + t.x0 + 1, go * s * t.wx);
        if (x0ok && y1ok) atomicAdd(q + (size_t)(t.y0 + 1) * W + t.x0, go * t.wy * e);
        if (x1ok && y1ok) atomicAdd(q + (size_t)(t.y0 + 1) * W + t.x0 + 1, go * t.wy * t.wx);
      }
    }
    if (dgrid) {
      const float mx = t.gx_in ? (float)W * 0.5f : 0.f;
      const float my = t.gy_in ? (float)H * 0.5f : 0.f;
      *reinterpret_cast<float2*>(dgrid + (size_t)idx * 2) = make_float2(gxa * mx, gya * my);
    }
  }
}

// ------------------------------------------------------------------ Resample2d (flow warp, kernel_size 1, bilinear)
// Specification restated from the public NVIDIA flownet2-pytorch resample2d op (the reference's submodule
// is empty here: parity unpinned).  in [B][C][H][W], flow [B][2][H][W] in pixels.
__global__ __launch_bounds__(256) void resample2d_fwd_k(const float* __restrict__ in,
                                                        const float* __restrict__ flow,
                                                        float* __restrict__ out, unsigned Nb,
                                                        unsigned C, unsigned H, unsigned W) {
  const unsigned total = Nb * H * W;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned b = idx / (H * W), pix = idx - b * H * W;
    const unsigned y = pix / W, x = pix - y * W;
    const float xf = (float)x + flow[((size_t)b * 2) * H * W + pix];
    const float yf = (float)y + flow[((size_t)b * 2 + 1) * H * W + pix];
    const float fx = floorf(xf), fy = floorf(yf);
    const float a = xf - fx, bt = yf - fy;
    const int xL = max(min((int)fx, (int)W - 1), 0), xR = max(min((int)fx + 1, (int)W - 1), 0);
    const int yT = max(min((int)fy, (int)H - 1), 0), yB = max(min((int)fy + 1, (int)H - 1), 0);
    for (unsigned c = 0; c < C; ++c) {
      const float* p = in + ((size_t)b * C + c) * H * W;
      out[((size_t)b * C + c) * H * W + pix] =
          (1.f - a) * (1.f - bt) * p[(size_t)yT * W + xL] + a * (1.f - bt) * p[(size_t)yT * W + xR] +
          (1.f - a) * bt * p[(size_t)yB * W + xL] + a * bt * p[(size_t)yB * W + xR];
    }
  }
}

// The input gradient of Resample2d is a scatter-add (several output pixels may sample the same input pixel).  Float
// atomics would make the result depend on the order the hardware retires them; instead every contribution is rounded to
// a fixed-point grid (2^-42 of max|dout|, far below an fp32 ulp of the result) and summed with 64-bit INTEGER atomics,
// which are associative: the gradient is bit-identical from run to run.  At most 4*H*W < 2^20 contributions of
// magnitude <= max|dout| meet in one accumulator, so |sum| < 2^62.
// A non-finite dout must not be laundered into a finite gradient (fmaxf drops NaN, __float2ll_rn(NaN) is 0): the max pass
// reports +inf as soon as one element is NaN or Inf, and the final pass then poisons the whole input gradient with NaN,
// so a diverging generator shows up exactly as it would with the float scatter-add.
__global__ __launch_bounds__(256) void maxabs_partial_k(const float* __restrict__ x, size_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
    const float a = fabsf(x[i]);
    m = (a <= 3.402823466e+38f) ? fmaxf(m, a) : __int_as_float(0x7f800000);  // NaN / Inf -> +inf (sticky under fmaxf)
  }
  m = so_wave_max(m);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void maxabs_final_k(const float* __restrict__ part, unsigned n, float* __restrict__ scale) {
  __shared__ float red[4];
  float m = 0.f;
  for (unsigned i = threadIdx.x; i < n; i += 256u) m = fmaxf(m, part[i]);
  m = so_wave_max(m);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const bool finite = mx <= 3.402823466e+38f;
    scale[0] = (mx > 0.f && finite) ? 4398046511104.0f / mx : 0.f;  // 2^42 / max|dout|
    scale[1] = finite ? (mx > 0.f ? mx / 4398046511104.0f : 0.f) : __int_as_float(0x7fc00000);  // non-finite dout: NaN out
  }
}

__device__ __forceinline__ void fixed_add(long long* q, float v, float scale) {
  atomicAdd(reinterpret_cast<unsigned long long*>(q), (unsigned long long)__float2ll_rn(v * scale));
}

__global__ __launch_bounds__(256) void resample2d_bwd_k(const float* __restrict__ in,
                                                        const float* __restrict__ flow,
                                                        const float* __restrict__ dout,
                                                        long long* __restrict__ acc,
                                                        float* __restrict__ dflow, const float* __restrict__ scale_p,
                                                        unsigned Nb, unsigned C, unsigned H, unsigned W) {
  const unsigned total = Nb * H * W;
  const float scale = acc ? scale_p[0] : 0.f;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned b = idx / (H * W), pix = idx - b * H * W;
    const unsigned y = pix / W, x = pix - y * W;
    const float xf = (float)x + flow[((size_t)b * 2) * H * W + pix];
    const float yf = (float)y + flow[((size_t)b * 2 + 1) * H * W + pix];
    const float fx = floorf(xf), fy = floorf(yf);
    const float a = xf - fx, bt = yf - fy;
    const int xL = max(min((int)fx, (int)W - 1), 0), xR = max(min((int)fx + 1, (int)W - 1), 0);
    const int yT = max(min((int)fy, (int)H - 1), 0), yB = max(min((int)fy + 1, (int)H - 1), 0);
    float gfx = 0.f, gfy = 0.f;
    for (unsigned c = 0; c < C; ++c) {
      const float go = dout[((size_t)b * C + c) * H * W + pix];
      const float* p = in + ((size_t)b * C + c) * H * W;
      const float vTL = p[(size_t)yT * W + xL], vTR = p[(size_t)yT * W + xR];
      const float vBL = p[(size_t)yB * W + xL], vBR = p[(size_t)yB * W + xR];
      gfx += go * ((1.f - bt) * (vTR - vTL) + bt * (vBR - vBL));
      gfy += go * ((1.f - a) * (vBL - vTL) + a * (vBR - vTR));
      if (acc) {
        long long* q = acc + ((size_t)b * C + c) * H * W;
        fixed_add(q + (size_t)yT * W + xL, go * (1.f - a) * (1.f - bt), scale);
        fixed_add(q + (size_t)yT * W + xR, go * a * (1.f - bt), scale);
        fixed_add(q + (size_t)yB * W + xL, go * (1.f - a) * bt, scale);
        fixed_add(q + (size_t)yB * W + xR, go * a * bt, scale);
      }
    }
    if (dflow) {
      dflow[((size_t)b * 2) * H * W + pix] = gfx;
      dflow[((size_t)b * 2 + 1) * H * W + pix] = gfy;
    }
  }
}

__global__ __launch_bounds__(256) void fixed_to_float_k(const long long* __restrict__ acc, size_t n,
                                                        const float* __restrict__ scale_p, float* __restrict__ out) {
  const double unit = (double)scale_p[1];
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
    out[i] = (float)((double)acc[i] * unit);
}

}  // namespace

extern "C" {

int so_l2norm_fwd(const float* x, int ldx, float* y, int ldy, float* inv, int Nb, int H, int W, int C,
                  int transpose_hw, void* stream) {
  const long long rows = (long long)Nb * H * W;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(l2norm_fwd_k, dim3(so_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y,
                     ldy, inv, (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C, transpose_hw);
  return SO_LAUNCH_CHECK();
}

int so_l2norm_bwd(const float* y, int ldy, const float* dy, int lddy, const float* inv, float* dx,
                  int lddx, int Nb, int H, int W, int C, int transpose_hw, void* stream) {
  const long long rows = (long long)Nb * H * W;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(l2norm_bwd_k, dim3(so_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, ldy,
                     dy, lddy, inv, dx, lddx, (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)C,
                     transpose_hw);
  return SO_LAUNCH_CHECK();
}

int so_softmax_rows_fwd(const float* e, int lde, float* a, int lda, long long rows, int ncol,
                        void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(softmax_rows_fwd_k, dim3(so_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, e,
                     lde, a, lda, (unsigned)rows, (unsigned)ncol);
  return SO_LAUNCH_CHECK();
}

int so_softmax_rows_bwd(const float* a, int lda, const float* da, int ldda, float* de, int ldde,
                        long long rows, int ncol, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(softmax_rows_bwd_k, dim3(so_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a,
                     lda, da, ldda, de, ldde, (unsigned)rows, (unsigned)ncol);
  return SO_LAUNCH_CHECK();
}

// ws: >= 1024 floats
int so_dot(const float* a, int lda, const float* b, int ldb, long long rows, int C, float scale,
           float* out, int accumulate, float* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int blocks = grid_for(rows * C);
  if (blocks > 512) blocks = 512;  // 512 fp64 partials in the 1024-float scratch
  if ((((uintptr_t)ws) & 7) != 0) return SO_ERR_ALIGN;
  hipLaunchKernelGGL(dot_partial_k, dim3(blocks), dim3(256), 0, st, a, lda, b, ldb, (unsigned)rows,
                     (unsigned)C, reinterpret_cast<double*>(ws));
  hipLaunchKernelGGL(dot_final_k, dim3(1), dim3(256), 0, st, reinterpret_cast<const double*>(ws), (unsigned)blocks, scale,
                     out, accumulate);
  return SO_LAUNCH_CHECK();
}

int so_linear_chw_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int Nb,
                      int P, int C, int J, int apply_tanh, void* stream) {
  hipLaunchKernelGGL(linear_chw_fwd_k, dim3(J, Nb), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias,
                     y, (unsigned)P, (unsigned)C, (unsigned)J, apply_tanh);
  return SO_LAUNCH_CHECK();
}

int so_linear_chw_bwd(const float* x, int ldx, const float* w, const float* y, const float* dy,
                      float* dx, int lddx, float* dw, float* dbias, int Nb, int P, int C, int J,
                      int apply_tanh, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_chw_bwd_w_k, dim3(grid_for((long long)J * P * C)), dim3(256), 0, st, x, ldx,
                     y, dy, dw, dbias, (unsigned)Nb, (unsigned)P, (unsigned)C, (unsigned)J, apply_tanh);
  if (dx)
    hipLaunchKernelGGL(linear_chw_bwd_x_k, dim3(grid_for((long long)Nb * P * C)), dim3(256), 0, st, w,
                       y, dy, dx, lddx, (unsigned)Nb, (unsigned)P, (unsigned)C, (unsigned)J,
                       apply_tanh);
  return SO_LAUNCH_CHECK();
}

long long so_tps_ws_floats(int Nb, int H, int W, int NP) {
  const int nblk = so_cdiv((long long)H * W, 256 * TPS_PPT);
  // [fp64 coefficients: 2 floats each][backward partials]
  return 2ll * Nb * 2 * (NP + 3) + (long long)Nb * nblk * 2 * (NP + 3);
}

// theta [B][2*NP]; Li [(NP+3)][(NP+3)]; px,py [NP]; gx [W]; gy [H]; grid [B][H][W][2]
int so_tps_grid_fwd(const float* theta, const float* Li, const float* px, const float* py,
                    const float* gx, const float* gy, float* grid, int Nb, int H, int W, int NP,
                    float* ws, void* stream) {
  if (NP > 61) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (((uintptr_t)ws & 7) != 0) return SO_ERR_ALIGN;
  double* coef = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(tps_coef_k, dim3(Nb), dim3(64), 0, st, theta, Li, px, py, coef, NP);
  const size_t sh = (size_t)(TPS_BG * 2 * (NP + 3)) * sizeof(double);
  hipLaunchKernelGGL(tps_grid_fwd_k, dim3(so_cdiv((long long)H * W, 256), so_cdiv(Nb, TPS_BG)), dim3(256), sh, st, coef,
                     gx, gy, px, py, grid, (unsigned)Nb, (unsigned)H, (unsigned)W, NP);
  return SO_LAUNCH_CHECK();
}

int so_tps_grid_bwd(const float* dgrid, const float* Li, const float* px, const float* py,
                    const float* gx, const float* gy, float* dtheta, int Nb, int H, int W, int NP,
                    float* ws, void* stream) {
  if (NP > 61) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = so_cdiv((long long)H * W, 256 * TPS_PPT);
  float* part = ws + 2 * (size_t)Nb * 2 * (NP + 3);
  hipLaunchKernelGGL(tps_grid_bwd_partial_k, dim3(nblk, Nb), dim3(256), 0, st, dgrid, gx, gy, px, py,
                     part, (unsigned)H, (unsigned)W, NP);
  hipLaunchKernelGGL(tps_grid_bwd_final_k, dim3(Nb), dim3(64), 0, st, part, (unsigned)nblk, Li, dtheta,
                     NP);
  return SO_LAUNCH_CHECK();
}

int so_grid_sample_fwd(const float* in, const float* grid, float* out, int* taps, int Nb, int C, int H,
                       int W, int Ho, int Wo, int border, void* stream) {
  const long long total = (long long)Nb * Ho * Wo;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(grid_sample_fwd_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                     grid, out, taps, (unsigned)Nb, (unsigned)C, (unsigned)H, (unsigned)W,
                     (unsigned)Ho, (unsigned)Wo, border);
  return SO_LAUNCH_CHECK();
}

// din (optional) must be zero-filled by the caller; it is accumulated with atomics.
int so_grid_sample_bwd(const float* in, const float* grid, const float* dout, float* dgrid, float* din,
                       int Nb, int C, int H, int W, int Ho, int Wo, int border, void* stream) {
  const long long total = (long long)Nb * Ho * Wo;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(grid_sample_bwd_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                     grid, dout, dgrid, din, (unsigned)Nb, (unsigned)C, (unsigned)H, (unsigned)W,
                     (unsigned)Ho, (unsigned)Wo, border);
  return SO_LAUNCH_CHECK();
}

int so_resample2d_fwd(const float* in, const float* flow, float* out, int Nb, int C, int H, int W,
                      void* stream) {
  const long long total = (long long)Nb * H * W;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(resample2d_fwd_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                     flow, out, (unsigned)Nb, (unsigned)C, (unsigned)H, (unsigned)W);
  return SO_LAUNCH_CHECK();
}

long long so_resample2d_bwd_ws_floats(int Nb, int C, int H, int W) {
  return 2LL * Nb * C * H * W + 1024 + 8;  // int64 accumulators, max-abs partials, the two scale factors
}

// din, dflow optional.  din is overwritten (no pre-fill needed); ws: so_resample2d_bwd_ws_floats floats, 8-byte aligned.
int so_resample2d_bwd(const float* in, const float* flow, const float* dout, float* din, float* dflow,
                      int Nb, int C, int H, int W, float* ws, void* stream) {
  const long long total = (long long)Nb * H * W;
  if (total <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  long long* acc = nullptr;
  float* scale = nullptr;
  const size_t n = (size_t)Nb * C * H * W;
  if (din) {
    if (!ws || (((uintptr_t)ws) & 7)) return SO_ERR_ALIGN;
    acc = reinterpret_cast<long long*>(ws);
    float* part = ws + 2 * n;
    scale = part + 1024;
    int nb = grid_for((long long)n);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(maxabs_partial_k, dim3(nb), dim3(256), 0, st, dout, n, part);
    hipLaunchKernelGGL(maxabs_final_k, dim3(1), dim3(256), 0, st, (const float*)part, (unsigned)nb, scale);
    hipError_t e = hipMemsetAsync(acc, 0, n * sizeof(long long), st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(resample2d_bwd_k, dim3(grid_for(total)), dim3(256), 0, st, in, flow, dout, acc, dflow, (const float*)scale,
                     (unsigned)Nb, (unsigned)C, (unsigned)H, (unsigned)W);
  if (din)
    hipLaunchKernelGGL(fixed_to_float_k, dim3(grid_for((long long)n)), dim3(256), 0, st, (const long long*)acc, n,
                       (const float*)scale, din);
  return SO_LAUNCH_CHECK();
}

}  // extern "C"
