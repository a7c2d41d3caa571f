// SAMS-GAN specific kernels: everything in the SamsGenerator / PatchGAN discriminators that is
// not a convolution, a batch / instance norm, an activation or the SAGAN attention (those reuse the kernels of
// the try-on path).  All of it is HBM-bound elementwise / reduction work on NHWC rows.
//
//   nearest resize (label maps, nn.Upsample(scale_factor=2 | 0.5))      sams/spade.py:83, sams_generator.py:294-308
//   SPADE modulation  y = act(n * (1 + gamma) + beta)                  sams/spade.py:89,168-169
//   3x3 / stride 2 average pool, count_include_pad = False             discriminator.py:51-54
//   spectral norm: one power iteration + W / sigma, and its gradient   torch.nn.utils.spectral_norm (spade.py:149-153)
//   GAN losses (hinge | ls | original | w)                             loss.py:58-88
#include "common.h"
#include "../../include/shineon_hip.h"

namespace {

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ATen's nearest rule (UpSample.h nearest_neighbor_compute_source_index): min(floor(dst * scale), in - 1), fp32.
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  const int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

template <int VEC>
__global__ __launch_bounds__(256) void resize_nearest_fwd_k(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                            int ldy, unsigned Nb, unsigned Hi, unsigned Wi, unsigned Ho,
                                                            unsigned Wo, unsigned C, float sh, float sw) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const unsigned CQ = C / VEC;
  const unsigned total = Nb * Ho * Wo * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned cq = t % CQ; t /= CQ;
    const unsigned wo = t % Wo; t /= Wo;
    const unsigned ho = t % Ho;
    const unsigned n = t / Ho;
    const unsigned hi = nearest_src(ho, sh, Hi), wi = nearest_src(wo, sw, Wi);
    *reinterpret_cast<vec_t*>(y + ((size_t)(n * Ho + ho) * Wo + wo) * ldy + cq * VEC) =
        *reinterpret_cast<const vec_t*>(x + ((size_t)(n * Hi + hi) * Wi + wi) * ldx + cq * VEC);
  }
}

// Gather form of the adjoint: an input pixel collects every output pixel whose source index it is.  The candidates
// form a contiguous run starting no earlier than floor(hi / scale) - 1.
template <int VEC>
__global__ __launch_bounds__(256) void resize_nearest_bwd_k(const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                            int lddx, unsigned Nb, unsigned Hi, unsigned Wi, unsigned Ho,
                                                            unsigned Wo, unsigned C, float sh, float sw) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const unsigned CQ = C / VEC;
  const unsigned total = Nb * Hi * Wi * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned cq = t % CQ; t /= CQ;
    const int wi = t % Wi; t /= Wi;
    const int hi = t % Hi;
    const unsigned n = t / Hi;
    int h0 = (int)floorf((float)hi / sh) - 1, w0 = (int)floorf((float)wi / sw) - 1;
    if (h0 < 0) h0 = 0;
    if (w0 < 0) w0 = 0;
    vec_t acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int ho = h0; ho < (int)Ho; ++ho) {
      const int sh_i = nearest_src(ho, sh, Hi);
      if (sh_i > hi) break;
      if (sh_i < hi) continue;
      for (int wo = w0; wo < (int)Wo; ++wo) {
        const int sw_i = nearest_src(wo, sw, Wi);
        if (sw_i > wi) break;
        if (sw_i < wi) continue;
        acc += *reinterpret_cast<const vec_t*>(dy + ((size_t)(n * Ho + ho) * Wo + wo) * lddy + cq * VEC);
      }
    }
    *reinterpret_cast<vec_t*>(dx + ((size_t)(n * Hi + hi) * Wi + wi) * lddx + cq * VEC) = acc;
  }
}

// ------------------------------------------------------------------ residual add
template <int VEC>
__global__ __launch_bounds__(256) void add_k(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                             float* __restrict__ y, int ldy, unsigned rows, unsigned C) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, c0 = (idx - row * CQ) * VEC;
    *reinterpret_cast<vec_t*>(y + (size_t)row * ldy + c0) = *reinterpret_cast<const vec_t*>(a + (size_t)row * lda + c0) +
                                                            *reinterpret_cast<const vec_t*>(b + (size_t)row * ldb + c0);
  }
}

// ------------------------------------------------------------------ SPADE modulation
// mean != nullptr: `nrm` is the UN-normalised activation x and n = (x - mean[g][c]) * rstd[g][c] with g = row / R
// (batch norm: one group; instance norm: one group per sample) - the normalised tensor is never written to HBM.
template <int VEC>
__device__ __forceinline__ float __attribute__((ext_vector_type(VEC)))
spade_normalised(const float* __restrict__ nrm, int ldn, unsigned row, unsigned c0, const float* __restrict__ mean,
                 const float* __restrict__ rstd, unsigned R, unsigned C) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  vec_t n = *reinterpret_cast<const vec_t*>(nrm + (size_t)row * ldn + c0);
  if (mean) {
    const size_t s = (size_t)(row / R) * C + c0;
    n = (n - *reinterpret_cast<const vec_t*>(mean + s)) * *reinterpret_cast<const vec_t*>(rstd + s);
  }
  return n;
}

template <int VEC>
__global__ __launch_bounds__(256) void spade_fwd_k(const float* __restrict__ nrm, int ldn, const float* __restrict__ gamma,
                                                   int ldg, const float* __restrict__ beta, int ldb,
                                                   float* __restrict__ y, int ldy, unsigned rows, unsigned C, int act,
                                                   float param, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, unsigned R) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, c0 = (idx - row * CQ) * VEC;
    const vec_t n = spade_normalised<VEC>(nrm, ldn, row, c0, mean, rstd, R, C);
    const vec_t g = *reinterpret_cast<const vec_t*>(gamma + (size_t)row * ldg + c0);
    const vec_t b = *reinterpret_cast<const vec_t*>(beta + (size_t)row * ldb + c0);
    vec_t o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = so_actf(act, n[i] * (1.0f + g[i]) + b[i], param);
    *reinterpret_cast<vec_t*>(y + (size_t)row * ldy + c0) = o;
  }
}

// g = dy * act'(pre);  dn = g * (1 + gamma);  dgamma = g * n;  dbeta = g      (pre recomputed from n, gamma, beta)
// COLSUM (VEC == 4, C / 4 a power of two <= 256): the block also leaves the column sums of its dgamma | dbeta rows in
// part[blockIdx][2C] - the bias gradient of the convolution that produced gamma | beta, without a second pass over them.
// Every thread keeps one channel quad for the whole grid-stride loop (256 and the stride are multiples of C / 4).
template <int VEC, bool COLSUM>
__global__ __launch_bounds__(256) void spade_bwd_k(const float* __restrict__ nrm, int ldn, const float* __restrict__ gamma,
                                                   int ldg, const float* __restrict__ beta, int ldb,
                                                   const float* __restrict__ dy, int lddy, float* __restrict__ dn, int lddn,
                                                   float* __restrict__ dgamma, int lddg, float* __restrict__ dbeta, int lddb,
                                                   unsigned rows, unsigned C, int act, float param, float* __restrict__ part,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   unsigned R) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const unsigned CQ = C / VEC;
  const unsigned total = rows * CQ;
  vec_t sg, sb;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { sg[i] = 0.f; sb[i] = 0.f; }
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, c0 = (idx - row * CQ) * VEC;
    const vec_t n = spade_normalised<VEC>(nrm, ldn, row, c0, mean, rstd, R, C);
    const vec_t g = *reinterpret_cast<const vec_t*>(gamma + (size_t)row * ldg + c0);
    const vec_t b = *reinterpret_cast<const vec_t*>(beta + (size_t)row * ldb + c0);
    const vec_t d = *reinterpret_cast<const vec_t*>(dy + (size_t)row * lddy + c0);
    vec_t on, og, ob;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float gi = d[i] * so_actg(act, n[i] * (1.0f + g[i]) + b[i], param);
      on[i] = gi * (1.0f + g[i]);
      og[i] = gi * n[i];
      ob[i] = gi;
    }
    *reinterpret_cast<vec_t*>(dn + (size_t)row * lddn + c0) = on;
    *reinterpret_cast<vec_t*>(dgamma + (size_t)row * lddg + c0) = og;
    *reinterpret_cast<vec_t*>(dbeta + (size_t)row * lddb + c0) = ob;
    if constexpr (COLSUM) { sg += og; sb += ob; }
  }
  if constexpr (COLSUM) {
    __shared__ float red[256][2 * VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[threadIdx.x][i] = sg[i]; red[threadIdx.x][VEC + i] = sb[i]; }
    __syncthreads();
    if (threadIdx.x < CQ) {  // fixed order over the 256 / CQ threads that share this channel quad: deterministic
      float acc[2 * VEC];
#pragma unroll
      for (int i = 0; i < 2 * VEC; ++i) acc[i] = 0.f;
      for (unsigned t = threadIdx.x; t < 256u; t += CQ)
#pragma unroll
        for (int i = 0; i < 2 * VEC; ++i) acc[i] += red[t][i];
      float* out = part + (size_t)blockIdx.x * 2 * C + threadIdx.x * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) { out[i] = acc[i]; out[C + i] = acc[VEC + i]; }
    }
  }
}

// ------------------------------------------------------------------ avg_pool2d(3, stride 2, padding 1, count_include_pad=False)
__global__ __launch_bounds__(256) void avgpool3s2_fwd_k(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                        unsigned Nb, unsigned H, unsigned W, unsigned Ho, unsigned Wo,
                                                        unsigned C) {
  const unsigned total = Nb * Ho * Wo * C;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned c = t % C; t /= C;
    const int wo = t % Wo; t /= Wo;
    const int ho = t % Ho;
    const unsigned n = t / Ho;
    const int h0 = max(2 * ho - 1, 0), h1 = min(2 * ho + 2, (int)H);
    const int w0 = max(2 * wo - 1, 0), w1 = min(2 * wo + 2, (int)W);
    float s = 0.f;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) s += x[((size_t)(n * H + h) * W + w) * ldx + c];
    y[((size_t)(n * Ho + ho) * Wo + wo) * ldy + c] = s / (float)((h1 - h0) * (w1 - w0));
  }
}

__global__ __launch_bounds__(256) void avgpool3s2_bwd_k(const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                        int lddx, unsigned Nb, unsigned H, unsigned W, unsigned Ho,
                                                        unsigned Wo, unsigned C) {
  const unsigned total = Nb * H * W * C;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    unsigned t = idx;
    const unsigned c = t % C; t /= C;
    const int w = t % W; t /= W;
    const int h = t % H;
    const unsigned n = t / H;
    // windows [2o - 1, 2o + 1] containing h:  o in [ceil((h - 1) / 2), floor((h + 1) / 2)]
    const int ho0 = h / 2, ho1 = min((h + 1) / 2, (int)Ho - 1);
    const int wo0 = w / 2, wo1 = min((w + 1) / 2, (int)Wo - 1);
    float s = 0.f;
    for (int ho = ho0; ho <= ho1; ++ho) {
      const int hh = min(2 * ho + 2, (int)H) - max(2 * ho - 1, 0);
      for (int wo = wo0; wo <= wo1; ++wo) {
        const int ww = min(2 * wo + 2, (int)W) - max(2 * wo - 1, 0);
        s += dy[((size_t)(n * Ho + ho) * Wo + wo) * lddy + c] / (float)(hh * ww);
      }
    }
    dx[((size_t)(n * H + h) * W + w) * lddx + c] = s;
  }
}

// ------------------------------------------------------------------ spectral norm
// Weights are OHWI: row o holds K = RS * I floats, physical column j = rs * I + i; the power-iteration vector v keeps
// torch's (O, I, R, S) flattening, i.e. logical index i * RS + rs, so that `weight_v` is checkpoint-compatible.
__device__ __forceinline__ unsigned v_logical(unsigned j, unsigned I, unsigned RS) {
  const unsigned rs = j / I, i = j - rs * I;
  return i * RS + rs;
}

// tp[oc][j] = sum over the rows of chunk oc of W[o][j] * u[o]  (physical column order).  With one chunk (gridDim.y == 1)
// tp is t itself and part[block] = sum over the block's columns of t^2; otherwise sn_tsum_k finishes the job.
__global__ __launch_bounds__(256) void sn_wtu_k(const float* __restrict__ w, const float* __restrict__ u, unsigned O,
                                                unsigned K, unsigned rows_per_chunk, float* __restrict__ tp,
                                                float* __restrict__ part) {
  __shared__ float red[4];
  const unsigned j = blockIdx.x * 256u + threadIdx.x;
  const unsigned o0 = blockIdx.y * rows_per_chunk;
  const unsigned o1 = min(o0 + rows_per_chunk, O);
  float acc = 0.f;
  if (j < K) {
#pragma unroll 4
    for (unsigned o = o0; o < o1; ++o) acc += w[(size_t)o * K + j] * u[o];
    tp[(size_t)blockIdx.y * K + j] = acc;
  }
  if (gridDim.y == 1) {
    const float s = so_block_sum256(j < K ? acc * acc : 0.f, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
  }
}

// t[j] = sum_oc tp[oc][j] (fixed order: deterministic);  part[block] = sum over the block's columns of t^2
__global__ __launch_bounds__(256) void sn_tsum_k(const float* __restrict__ tp, unsigned chunks, unsigned K,
                                                 float* __restrict__ t, float* __restrict__ part) {
  __shared__ float red[4];
  const unsigned j = blockIdx.x * 256u + threadIdx.x;
  float acc = 0.f;
  if (j < K) {
    for (unsigned c = 0; c < chunks; ++c) acc += tp[(size_t)c * K + j];
    t[j] = acc;
  }
  const float s = so_block_sum256(j < K ? acc * acc : 0.f, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__device__ __forceinline__ float block_reduce_array(const float* __restrict__ a, unsigned n, float* red) {
  float s = 0.f;
  for (unsigned i = threadIdx.x; i < n; i += 256u) s += a[i];
  return so_block_sum256(s, red);
}

// s[o] = (W[o] . t) / max(|t|, eps)   (= W v with v = t / max(|t|, eps));   one wave per row, 4 rows per block.
// power_iter == 0: t is the stored v in LOGICAL order and is used as it is (norm 1 by construction).
__global__ __launch_bounds__(256) void sn_wv_k(const float* __restrict__ w, const float* __restrict__ t,
                                               const float* __restrict__ part, unsigned nparts, unsigned O, unsigned K,
                                               unsigned I, unsigned RS, float eps, int power_iter, float* __restrict__ s,
                                               float* __restrict__ scal) {
  __shared__ float red[4];
  float inv = 1.0f;
  if (power_iter) {
    const float nt = sqrtf(block_reduce_array(part, nparts, red));
    inv = 1.0f / fmaxf(nt, eps);
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = inv;
  }
  const unsigned lane = threadIdx.x & 63, o = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (o >= O) return;
  float acc = 0.f;
  const float* wr = w + (size_t)o * K;
  if (power_iter) {
    if ((K & 3u) == 0 && ((((uintptr_t)w) | ((uintptr_t)t)) & 15) == 0) {  // 16-byte loads: rows are K floats apart, K % 4 == 0
      const f32x4* w4 = reinterpret_cast<const f32x4*>(wr);
      const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
      for (unsigned j = lane; j < K / 4; j += 64u) {
        const f32x4 a = w4[j], b = t4[j];
        acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
    } else {
      for (unsigned j = lane; j < K; j += 64u) acc += wr[j] * t[j];
    }
  } else {
    for (unsigned j = lane; j < K; j += 64u) acc += wr[j] * t[v_logical(j, I, RS)];
  }
  acc = so_wave_sum(acc);
  if (lane == 0) s[o] = acc * inv;
}

// every block: |s| -> u = s / max(|s|, eps), sigma = u . s  (power_iter) or sigma = u_stored . s; then its slice of
// w_out = w / sigma and of v.  Block 0 also stores u and sigma.
__global__ __launch_bounds__(256) void sn_scale_k(const float* __restrict__ w, const float* __restrict__ t,
                                                  const float* __restrict__ s, unsigned O, unsigned K, unsigned I,
                                                  unsigned RS, float eps, int power_iter, float* __restrict__ u,
                                                  float* __restrict__ v, float* __restrict__ w_out,
                                                  const float* __restrict__ scal, float* __restrict__ sigma_out) {
  __shared__ float red[4];
  float sigma;
  if (power_iter) {
    float q = 0.f;
    for (unsigned i = threadIdx.x; i < O; i += 256u) q += s[i] * s[i];
    q = so_block_sum256(q, red);
    const float inv = 1.0f / fmaxf(sqrtf(q), eps);
    sigma = q * inv;
    if (blockIdx.x == 0)
      for (unsigned i = threadIdx.x; i < O; i += 256u) u[i] = s[i] * inv;
  } else {
    float q = 0.f;
    for (unsigned i = threadIdx.x; i < O; i += 256u) q += s[i] * u[i];
    sigma = so_block_sum256(q, red);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) sigma_out[0] = sigma;
  const size_t total = (size_t)O * K;
  if ((total & 3u) == 0 && ((((uintptr_t)w) | ((uintptr_t)w_out)) & 15) == 0) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    f32x4* o4 = reinterpret_cast<f32x4*>(w_out);
    for (size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x; idx < total / 4; idx += (size_t)gridDim.x * 256u) {
      const f32x4 a = w4[idx];
      const f32x4 r = {a[0] / sigma, a[1] / sigma, a[2] / sigma, a[3] / sigma};
      o4[idx] = r;
    }
  } else {
    for (size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256u)
      w_out[idx] = w[idx] / sigma;
  }
  if (power_iter) {
    const float inv_t = scal[0];
    for (unsigned j = blockIdx.x * 256u + threadIdx.x; j < K; j += gridDim.x * 256u) v[v_logical(j, I, RS)] = t[j] * inv_t;
  }
}

// part[block] = sum of g * w over the block's elements
__global__ __launch_bounds__(256) void sn_bwd_dot_k(const float* __restrict__ g, const float* __restrict__ w, size_t total,
                                                    float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256u) s += g[idx] * w[idx];
  s = so_block_sum256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// dW[o][j] = g[o][j] / sigma - (<g, W> / sigma^2) * u[o] * v[logical(j)]       (sigma = u^T W v, u and v constants)
__global__ __launch_bounds__(256) void sn_bwd_apply_k(const float* __restrict__ g, const float* __restrict__ part,
                                                      unsigned nparts, const float* __restrict__ u,
                                                      const float* __restrict__ v, const float* __restrict__ sigma_p,
                                                      unsigned O, unsigned K, unsigned I, unsigned RS,
                                                      float* __restrict__ dw, int accumulate) {
  __shared__ float red[4];
  const float sigma = sigma_p[0];
  const float coef = block_reduce_array(part, nparts, red) / (sigma * sigma);
  const size_t total = (size_t)O * K;
  for (size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256u) {
    const unsigned o = (unsigned)(idx / K), j = (unsigned)(idx - (size_t)o * K);
    const float val = g[idx] / sigma - coef * u[o] * v[v_logical(j, I, RS)];
    dw[idx] = accumulate ? dw[idx] + val : val;
  }
}

// ------------------------------------------------------------------ GAN losses
enum { GAN_ORIGINAL = 0, GAN_LS = 1, GAN_W = 2, GAN_HINGE = 3 };

__device__ __forceinline__ float gan_f(int mode, int real, int for_disc, float x) {
  const float t = real ? 1.0f : 0.0f;
  switch (mode) {
    case GAN_ORIGINAL: return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    case GAN_LS: return (x - t) * (x - t);
    case GAN_W: return real ? -x : x;
    default:
      if (!for_disc) return -x;
      return real ? -fminf(x - 1.0f, 0.f) : -fminf(-x - 1.0f, 0.f);
  }
}

// derivative; torch.min(a, 0) splits the gradient evenly on a tie (a == 0)
__device__ __forceinline__ float gan_g(int mode, int real, int for_disc, float x) {
  const float t = real ? 1.0f : 0.0f;
  switch (mode) {
    case GAN_ORIGINAL: return so_sigmoid(x) - t;
    case GAN_LS: return 2.0f * (x - t);
    case GAN_W: return real ? -1.0f : 1.0f;
    default: {
      if (!for_disc) return -1.0f;
      const float a = real ? x - 1.0f : -x - 1.0f;
      const float m = a < 0.f ? 1.0f : (a == 0.f ? 0.5f : 0.f);
      return real ? -m : m;
    }
  }
}

__global__ __launch_bounds__(256) void gan_partial_k(const float* __restrict__ x, int ldx, unsigned rows, unsigned C, int mode,
                                                     int real, int for_disc, float* __restrict__ part) {
  __shared__ float red[4];
  const unsigned total = rows * C;
  float s = 0.f;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned r = idx / C, c = idx - r * C;
    s += gan_f(mode, real, for_disc, x[(size_t)r * ldx + c]);
  }
  s = so_block_sum256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void gan_final_k(const float* __restrict__ part, unsigned nparts, float inv_count,
                                                   float* __restrict__ out) {
  __shared__ float red[4];
  const float s = block_reduce_array(part, nparts, red);
  if (threadIdx.x == 0) out[0] = s * inv_count;
}

__global__ __launch_bounds__(256) void gan_bwd_k(const float* __restrict__ x, int ldx, unsigned rows, unsigned C, int mode,
                                                 int real, int for_disc, const float* __restrict__ gout, float inv_count,
                                                 float* __restrict__ dx, int lddx) {
  const unsigned total = rows * C;
  const float go = gout[0] * inv_count;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned r = idx / C, c = idx - r * C;
    dx[(size_t)r * lddx + c] = go * gan_g(mode, real, for_disc, x[(size_t)r * ldx + c]);
  }
}

inline bool vec4(int C, int l1, int l2, const void* p1, const void* p2) {
  return (C & 3) == 0 && (l1 & 3) == 0 && (l2 & 3) == 0 && al16(p1) && al16(p2);
}

}  // namespace

extern "C" {

int so_resize_nearest_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int Hi, int Wi, int Ho, int Wo, int C,
                          float scale_h, float scale_w, void* stream) {
  const long long total = (long long)Nb * Ho * Wo * C;
  if (total <= 0) return 0;
  if (Hi <= 0 || Wi <= 0) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (vec4(C, ldx, ldy, x, y))
    hipLaunchKernelGGL(resize_nearest_fwd_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, x, ldx, y, ldy, (unsigned)Nb,
                       (unsigned)Hi, (unsigned)Wi, (unsigned)Ho, (unsigned)Wo, (unsigned)C, scale_h, scale_w);
  else
    hipLaunchKernelGGL(resize_nearest_fwd_k<1>, dim3(grid_for(total)), dim3(256), 0, st, x, ldx, y, ldy, (unsigned)Nb,
                       (unsigned)Hi, (unsigned)Wi, (unsigned)Ho, (unsigned)Wo, (unsigned)C, scale_h, scale_w);
  return SO_LAUNCH_CHECK();
}

int so_resize_nearest_bwd(const float* dy, int lddy, float* dx, int lddx, int Nb, int Hi, int Wi, int Ho, int Wo, int C,
                          float scale_h, float scale_w, void* stream) {
  const long long total = (long long)Nb * Hi * Wi * C;
  if (total <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (vec4(C, lddy, lddx, dy, dx))
    hipLaunchKernelGGL(resize_nearest_bwd_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, dy, lddy, dx, lddx, (unsigned)Nb,
                       (unsigned)Hi, (unsigned)Wi, (unsigned)Ho, (unsigned)Wo, (unsigned)C, scale_h, scale_w);
  else
    hipLaunchKernelGGL(resize_nearest_bwd_k<1>, dim3(grid_for(total)), dim3(256), 0, st, dy, lddy, dx, lddx, (unsigned)Nb,
                       (unsigned)Hi, (unsigned)Wi, (unsigned)Ho, (unsigned)Wo, (unsigned)C, scale_h, scale_w);
  return SO_LAUNCH_CHECK();
}

int so_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, long long rows, int C, void* stream) {
  if (rows <= 0 || C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (vec4(C, lda, ldb, a, b) && (ldy & 3) == 0 && al16(y))
    hipLaunchKernelGGL(add_k<4>, dim3(grid_for(rows * C / 4)), dim3(256), 0, st, a, lda, b, ldb, y, ldy, (unsigned)rows, (unsigned)C);
  else
    hipLaunchKernelGGL(add_k<1>, dim3(grid_for(rows * C)), dim3(256), 0, st, a, lda, b, ldb, y, ldy, (unsigned)rows, (unsigned)C);
  return SO_LAUNCH_CHECK();
}

static inline bool spade_colsum_ok(int C) { return (C & 3) == 0 && C / 4 <= 256 && ((C / 4) & (C / 4 - 1)) == 0; }

int so_spade_bwd_colsum_blocks(long long rows, int C) {
  if (!spade_colsum_ok(C)) return 0;
  int b = grid_for(rows * C / 4);
  return b > 1024 ? 1024 : b;
}

int so_spade_norm_fwd(const float* x, int ldx, const float* mean, const float* rstd, long long R, const float* gamma, int ldg,
                      const float* beta, int ldb, float* y, int ldy, long long rows, int C, int act, float act_param,
                      void* stream) {
  if (rows <= 0 || C <= 0) return 0;
  if (mean && (R <= 0 || rows % R)) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned Ru = mean ? (unsigned)R : 1u;
  if (vec4(C, ldx, ldy, x, y) && vec4(C, ldg, ldb, gamma, beta) && (!mean || (al16(mean) && al16(rstd))))
    hipLaunchKernelGGL(spade_fwd_k<4>, dim3(grid_for(rows * C / 4)), dim3(256), 0, st, x, ldx, gamma, ldg, beta, ldb, y, ldy,
                       (unsigned)rows, (unsigned)C, act, act_param, mean, rstd, Ru);
  else
    hipLaunchKernelGGL(spade_fwd_k<1>, dim3(grid_for(rows * C)), dim3(256), 0, st, x, ldx, gamma, ldg, beta, ldb, y, ldy,
                       (unsigned)rows, (unsigned)C, act, act_param, mean, rstd, Ru);
  return SO_LAUNCH_CHECK();
}

int so_spade_fwd(const float* nrm, int ldn, const float* gamma, int ldg, const float* beta, int ldb, float* y, int ldy,
                 long long rows, int C, int act, float act_param, void* stream) {
  return so_spade_norm_fwd(nrm, ldn, nullptr, nullptr, 0, gamma, ldg, beta, ldb, y, ldy, rows, C, act, act_param, stream);
}

int so_spade_norm_bwd(const float* x, int ldx, const float* mean, const float* rstd, long long R, const float* gamma, int ldg,
                      const float* beta, int ldb, const float* dy, int lddy, float* dn, int lddn, float* dgamma, int lddg,
                      float* dbeta, int lddb, long long rows, int C, int act, float act_param, float* colsum_part,
                      void* stream) {
  if (rows <= 0 || C <= 0) return 0;
  if (mean && (R <= 0 || rows % R)) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned Ru = mean ? (unsigned)R : 1u;
  const bool v4 = vec4(C, ldx, lddy, x, dy) && vec4(C, ldg, ldb, gamma, beta) && vec4(C, lddn, lddg, dn, dgamma) &&
                  vec4(C, lddb, lddb, dbeta, dbeta) && (!mean || (al16(mean) && al16(rstd)));
  if (colsum_part) {
    if (!v4 || !spade_colsum_ok(C)) return SO_ERR_ALIGN;
    hipLaunchKernelGGL((spade_bwd_k<4, true>), dim3(so_spade_bwd_colsum_blocks(rows, C)), dim3(256), 0, st, x, ldx, gamma, ldg,
                       beta, ldb, dy, lddy, dn, lddn, dgamma, lddg, dbeta, lddb, (unsigned)rows, (unsigned)C, act, act_param,
                       colsum_part, mean, rstd, Ru);
  } else if (v4) {
    hipLaunchKernelGGL((spade_bwd_k<4, false>), dim3(grid_for(rows * C / 4)), dim3(256), 0, st, x, ldx, gamma, ldg, beta, ldb, dy,
                       lddy, dn, lddn, dgamma, lddg, dbeta, lddb, (unsigned)rows, (unsigned)C, act, act_param, (float*)nullptr,
                       mean, rstd, Ru);
  } else {
    hipLaunchKernelGGL((spade_bwd_k<1, false>), dim3(grid_for(rows * C)), dim3(256), 0, st, x, ldx, gamma, ldg, beta, ldb, dy, lddy,
                       dn, lddn, dgamma, lddg, dbeta, lddb, (unsigned)rows, (unsigned)C, act, act_param, (float*)nullptr, mean,
                       rstd, Ru);
  }
  return SO_LAUNCH_CHECK();
}

int so_spade_bwd(const float* nrm, int ldn, const float* gamma, int ldg, const float* beta, int ldb, const float* dy, int lddy,
                 float* dn, int lddn, float* dgamma, int lddg, float* dbeta, int lddb, long long rows, int C, int act,
                 float act_param, float* colsum_part, void* stream) {
  return so_spade_norm_bwd(nrm, ldn, nullptr, nullptr, 0, gamma, ldg, beta, ldb, dy, lddy, dn, lddn, dgamma, lddg, dbeta, lddb,
                           rows, C, act, act_param, colsum_part, stream);
}

int so_avgpool3s2_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C, void* stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)Nb * Ho * Wo * C;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(avgpool3s2_fwd_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (unsigned)Nb,
                     (unsigned)H, (unsigned)W, (unsigned)Ho, (unsigned)Wo, (unsigned)C);
  return SO_LAUNCH_CHECK();
}

int so_avgpool3s2_bwd(const float* dy, int lddy, float* dx, int lddx, int Nb, int H, int W, int C, void* stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)Nb * H * W * C;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(avgpool3s2_bwd_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx, lddx,
                     (unsigned)Nb, (unsigned)H, (unsigned)W, (unsigned)Ho, (unsigned)Wo, (unsigned)C);
  return SO_LAUNCH_CHECK();
}

static inline unsigned sn_chunks(int O) {  // row chunks of the W^T u pass: enough blocks to fill the chip on big layers
  unsigned c = (unsigned)(O + 63) / 64;
  return c > 16 ? 16 : (c < 1 ? 1 : c);
}

long long so_spectral_norm_ws_floats(int O, int I, int RS) {
  const long long K = (long long)I * RS;
  // t, s, scalars (1 / |t|, sigma), partial sums (also the 1024 of the backward pass), row-chunk partials of W^T u
  return K + O + 8 + 1024 + (K + 255) / 256 + (long long)sn_chunks(O) * K;
}

int so_spectral_norm_fwd(const float* w_orig, int O, int I, int RS, float* u, float* v, float* w_out, float* sigma,
                         int power_iter, float eps, float* ws, void* stream) {
  if (O <= 0 || I <= 0 || RS <= 0) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned K = (unsigned)I * RS;
  float* t = ws;
  float* s = t + K;
  float* scal = s + O;
  float* part = scal + 8;
  const unsigned nparts = (K + 255) / 256;
  if (power_iter) {
    const unsigned chunks = sn_chunks(O);
    if (chunks == 1) {
      hipLaunchKernelGGL(sn_wtu_k, dim3(nparts, 1), dim3(256), 0, st, w_orig, (const float*)u, (unsigned)O, K, (unsigned)O, t, part);
    } else {
      float* tp = part + 1024 + nparts;
      const unsigned rpc = ((unsigned)O + chunks - 1) / chunks;
      hipLaunchKernelGGL(sn_wtu_k, dim3(nparts, chunks), dim3(256), 0, st, w_orig, (const float*)u, (unsigned)O, K, rpc, tp, part);
      hipLaunchKernelGGL(sn_tsum_k, dim3(nparts), dim3(256), 0, st, (const float*)tp, chunks, K, t, part);
    }
    hipLaunchKernelGGL(sn_wv_k, dim3((O + 3) / 4), dim3(256), 0, st, w_orig, (const float*)t, (const float*)part, nparts,
                       (unsigned)O, K, (unsigned)I, (unsigned)RS, eps, 1, s, scal);
  } else {
    hipLaunchKernelGGL(sn_wv_k, dim3((O + 3) / 4), dim3(256), 0, st, w_orig, (const float*)v, (const float*)part, 0u,
                       (unsigned)O, K, (unsigned)I, (unsigned)RS, eps, 0, s, scal);
  }
  hipLaunchKernelGGL(sn_scale_k, dim3(grid_for((long long)O * K)), dim3(256), 0, st, w_orig, (const float*)t, (const float*)s,
                     (unsigned)O, K, (unsigned)I, (unsigned)RS, eps, power_iter, u, v, w_out, (const float*)scal, sigma);
  return SO_LAUNCH_CHECK();
}

int so_spectral_norm_bwd(const float* g, const float* w_orig, const float* u, const float* v, const float* sigma, int O,
                         int I, int RS, float* dw, int accumulate, float* ws, void* stream) {
  if (O <= 0 || I <= 0 || RS <= 0) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)O * I * RS;
  int nparts = grid_for((long long)total);
  if (nparts > 1024) nparts = 1024;
  hipLaunchKernelGGL(sn_bwd_dot_k, dim3(nparts), dim3(256), 0, st, g, w_orig, total, ws);
  hipLaunchKernelGGL(sn_bwd_apply_k, dim3(grid_for((long long)total)), dim3(256), 0, st, g, (const float*)ws,
                     (unsigned)nparts, u, v, sigma, (unsigned)O, (unsigned)(I * RS), (unsigned)I, (unsigned)RS, dw,
                     accumulate);
  return SO_LAUNCH_CHECK();
}

int so_gan_loss_fwd(const float* x, int ldx, long long rows, int C, int mode, int target_is_real, int for_discriminator,
                    float* out, float* ws, void* stream) {
  if (rows <= 0 || C <= 0 || mode < 0 || mode > 3) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  int nparts = grid_for(rows * C);
  if (nparts > 1024) nparts = 1024;
  hipLaunchKernelGGL(gan_partial_k, dim3(nparts), dim3(256), 0, st, x, ldx, (unsigned)rows, (unsigned)C, mode, target_is_real,
                     for_discriminator, ws);
  hipLaunchKernelGGL(gan_final_k, dim3(1), dim3(256), 0, st, (const float*)ws, (unsigned)nparts,
                     1.0f / (float)((double)rows * C), out);
  return SO_LAUNCH_CHECK();
}

int so_gan_loss_bwd(const float* x, int ldx, long long rows, int C, int mode, int target_is_real, int for_discriminator,
                    const float* gout, float* dx, int lddx, void* stream) {
  if (rows <= 0 || C <= 0 || mode < 0 || mode > 3) return SO_ERR_SHAPE;
  hipLaunchKernelGGL(gan_bwd_k, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, (unsigned)rows,
                     (unsigned)C, mode, target_is_real, for_discriminator, gout, 1.0f / (float)((double)rows * C), dx, lddx);
  return SO_LAUNCH_CHECK();
}

}  // extern "C"
