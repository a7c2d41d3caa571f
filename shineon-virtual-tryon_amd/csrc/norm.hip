// InstanceNorm2d / BatchNorm2d (training and inference) forward + backward for NHWC rows (gfx950).
//
// Reference semantics (file:line in /root/reference):
//   nn.InstanceNorm2d(affine=False, eps=1e-5, no running stats)  models/networks/cpvton/unet.py:136,143-146
//        (U-Net norm layer: models/unet_mask_model.py:56)
//   nn.BatchNorm2d (train: batch statistics over N,H,W, biased variance for normalisation, unbiased for
//        running_var, momentum 0.1, eps 1e-5, affine)              models/networks/cpvton/warp.py:15,21,29,75-84
//
// A "group" is one statistics domain: one sample for InstanceNorm (G = N, R = H*W rows), the whole
// batch for BatchNorm (G = 1, R = N*H*W rows).  Statistics are column reductions of an [R][C] matrix:
// consecutive lanes own consecutive channels (coalesced), several row-lanes stride over the rows,
// partial (count, mean, M2) triples are merged with Chan's formula in a fixed order (deterministic,
// no cancellation: per-thread sums are shifted by the first element seen).
#include "common.h"
#include "../../include/shineon_hip.h"

namespace {

struct Mom {
  float n, mean, m2;
};

__device__ __forceinline__ Mom mom_merge(Mom a, Mom b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  Mom r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean;
  r.mean = a.mean + d * (b.n / r.n);
  r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
  return r;
}

// mom_merge with its two contractions written out.  hipcc (-ffp-contract=fast) fuses `a.mean + d * q` and `s + d * d * q2` in
// the rolled loops of this file; in the unrolled batch of stats_final_k it fused only some instances - one ulp in the mean,
// enough to flip ReLU gates downstream and move gradient tensors between acceptance routes.  Written out, the batch
// reproduces the rolled loop bit for bit (checked on MI355X against the previous build on eight shapes).
__device__ __forceinline__ Mom mom_merge_fma(Mom a, Mom b) {
#pragma clang fp contract(off)
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  Mom r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean;
  const float q = b.n / r.n;
  const float q2 = a.n * b.n / r.n;
  const float dd = d * d;
  const float s = a.m2 + b.m2;
  r.mean = __builtin_fmaf(d, q, a.mean);
  r.m2 = __builtin_fmaf(dd, q2, s);
  return r;
}

constexpr int EPT = 8;   // rows per thread per chunk (at least)

// part layout: [G][nchunk][3][C]
__global__ __launch_bounds__(256) void stats_partial_k(const float* __restrict__ x, int ldx,
                                                       unsigned R, unsigned C, unsigned chunk,
                                                       unsigned nchunk, float* __restrict__ part) {
  __shared__ float sn[256], sm[256], s2[256];
  const unsigned CPB = C >= 256 ? 256 : C;
  const unsigned RL = 256 / CPB;
  const unsigned tx = threadIdx.x % CPB, ty = threadIdx.x / CPB;
  const unsigned col = blockIdx.y * CPB + tx;
  const unsigned g = blockIdx.z;
  const unsigned r0 = blockIdx.x * chunk;
  unsigned r1 = r0 + chunk;
  if (r1 > R) r1 = R;
  const float* base = x + (size_t)g * R * ldx;
  float cnt = 0.f, shift = 0.f, s = 0.f, ss = 0.f;
  if (ty < RL && col < C) {
#pragma unroll 8
    for (unsigned r = r0 + ty; r < r1; r += RL) {
      const float v = base[(size_t)r * ldx + col];
      if (cnt == 0.f) shift = v;
      const float d = v - shift;
      s += d;
      ss += d * d;
      cnt += 1.f;
    }
  }
  float mean = 0.f, m2 = 0.f;
  if (cnt > 0.f) {
    mean = shift + s / cnt;
    m2 = ss - s * s / cnt;
    if (m2 < 0.f) m2 = 0.f;
  }
  sn[threadIdx.x] = cnt; sm[threadIdx.x] = mean; s2[threadIdx.x] = m2;
  __syncthreads();
  if (ty == 0 && col < C) {
    Mom acc = {0.f, 0.f, 0.f};
    for (unsigned l = 0; l < RL; ++l) {
      const unsigned i = l * CPB + tx;
      acc = mom_merge(acc, Mom{sn[i], sm[i], s2[i]});
    }
    float* o = part + ((size_t)g * nchunk + blockIdx.x) * 3 * C;
    o[col] = acc.n; o[C + col] = acc.mean; o[2 * C + col] = acc.m2;
  }
}

// mean[g][c], rstd[g][c]; optional running-stat update (BatchNorm, G must be 1).
__global__ __launch_bounds__(1024) void stats_final_k(const float* __restrict__ part, unsigned nchunk,
                                                     unsigned C, float eps, float* __restrict__ mean,
                                                     float* __restrict__ rstd,
                                                     float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float momentum) {
  // 16 columns x 64 chunk-lanes per block: lane l merges chunks l, l+64, ... then the lanes are merged in
  // a fixed order through LDS (deterministic).
  __shared__ float sn[1024], sm[1024], s2[1024];
  const unsigned tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const unsigned c = blockIdx.x * 16u + tx;
  const unsigned g = blockIdx.y;
  // The partials were written by other XCDs a moment ago (L2 misses): all of a thread's <= FB loads go out as ONE batch, the
  // merges then run in the same order as before (a batch of four at a time was four dependent memory latencies per launch).
  Mom acc = {0.f, 0.f, 0.f};
  if (c < C) {
    constexpr int FB = 16;   // chunking() keeps nchunk <= 1024 = 64 lanes x 16: one batch
    for (unsigned k0 = ty; k0 < nchunk; k0 += 64u * FB) {
      float pn[FB], pm[FB], p2[FB];
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        const unsigned k = k0 + 64u * j;
        const float* o = part + ((size_t)g * nchunk + (k < nchunk ? k : 0u)) * 3 * C;
        pn[j] = k < nchunk ? o[c] : 0.f;
        pm[j] = o[C + c];
        p2[j] = o[2 * C + c];
      }
#pragma unroll
      for (int j = 0; j < FB; ++j) acc = mom_merge_fma(acc, Mom{pn[j], pm[j], p2[j]});   // n == 0: identity
    }
  }
  sn[threadIdx.x] = acc.n; sm[threadIdx.x] = acc.mean; s2[threadIdx.x] = acc.m2;
  __syncthreads();
  // fixed binary tree over the 64 lanes of a column (6 levels instead of a 64-step serial chain of divisions)
  for (unsigned stride = 32; stride >= 1; stride >>= 1) {
    if (ty < stride) {
      const unsigned i0 = ty * 16 + tx, i1 = (ty + stride) * 16 + tx;
      const Mom m = mom_merge(Mom{sn[i0], sm[i0], s2[i0]}, Mom{sn[i1], sm[i1], s2[i1]});
      sn[i0] = m.n; sm[i0] = m.mean; s2[i0] = m.m2;
    }
    __syncthreads();
  }
  if (ty != 0 || c >= C) return;
  acc = Mom{sn[tx], sm[tx], s2[tx]};
  const float var = acc.m2 / acc.n;
  mean[(size_t)g * C + c] = acc.mean;
  rstd[(size_t)g * C + c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = acc.n > 1.f ? acc.m2 / (acc.n - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * acc.mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// y = (x - mean) * rstd * gamma + beta  (gamma/beta optional).  stat_is_var: `rstd` holds a variance
// (inference BatchNorm with running statistics) and eps is applied here.
template <int VEC>
__global__ __launch_bounds__(256) void norm_apply_k(const float* __restrict__ x, int ldx,
                                                    float* __restrict__ y, int ldy, unsigned G,
                                                    unsigned R, unsigned C,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int stat_is_var,
                                                    float eps, float* __restrict__ y2, int ldy2, int act2,
                                                    float act2_param) {
  const unsigned CQ = C / VEC;
  const unsigned total = G * R * CQ;
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, cq = idx - row * CQ;
    const unsigned g = row / R;
    const unsigned c0 = cq * VEC;
    const vec_t xv = *reinterpret_cast<const vec_t*>(x + (size_t)row * ldx + c0);
    const vec_t mu = *reinterpret_cast<const vec_t*>(mean + (size_t)g * C + c0);
    vec_t rs = *reinterpret_cast<const vec_t*>(rstd + (size_t)g * C + c0);
    vec_t o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float r = rs[i];
      if (stat_is_var) r = 1.0f / sqrtf(r + eps);
      float v = (xv[i] - mu[i]) * r;
      if (gamma) v = v * gamma[c0 + i] + beta[c0 + i];
      o[i] = v;
    }
    *reinterpret_cast<vec_t*>(y + (size_t)row * ldy + c0) = o;
    if (y2) {   // the consumer's activation of the same values (so_norm_act_fwd): second output, same launch
      vec_t o2;
#pragma unroll
      for (int i = 0; i < VEC; ++i) o2[i] = so_actf(act2, o[i], act2_param);
      *reinterpret_cast<vec_t*>(y2 + (size_t)row * ldy2 + c0) = o2;
    }
  }
}

// Backward partial sums per (g, chunk, c): s1 = sum dy, s2 = sum dy * xhat.   part: [G][nchunk][NS][C], NS = 2, or 5 with
// GATED = true: also the sums over the rows with x > 0 of dy, 1 and xhat - from them norm_bwd_final_k gets the column sums of
// the GATED input gradient (the bias gradient of the Conv -> ReLU in front of this BatchNorm, warp.py:15-31) without another
// pass over dx:  sum_r dx = rstd gamma (sum_{x>0} dy - s1/R #{x>0} - s2/R sum_{x>0} xhat).
template <bool GATED>
__global__ __launch_bounds__(256) void norm_bwd_partial_k(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ dy, int lddy,
                                                          unsigned R, unsigned C, unsigned chunk,
                                                          unsigned nchunk,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          float* __restrict__ part) {
  constexpr int NS = GATED ? 5 : 2;
  __shared__ float red[NS][256];
  const unsigned CPB = C >= 256 ? 256 : C;
  const unsigned RL = 256 / CPB;
  const unsigned tx = threadIdx.x % CPB, ty = threadIdx.x / CPB;
  const unsigned col = blockIdx.y * CPB + tx;
  const unsigned g = blockIdx.z;
  const unsigned r0 = blockIdx.x * chunk;
  unsigned r1 = r0 + chunk;
  if (r1 > R) r1 = R;
  float s[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) s[k] = 0.f;
  if (ty < RL && col < C) {
    const float mu = mean[(size_t)g * C + col], rs = rstd[(size_t)g * C + col];
    const float* bx = x + (size_t)g * R * ldx;
    const float* bd = dy + (size_t)g * R * lddy;
#pragma unroll 8
    for (unsigned r = r0 + ty; r < r1; r += RL) {
      const float d = bd[(size_t)r * lddy + col];
      const float xv = bx[(size_t)r * ldx + col];
      const float xh = (xv - mu) * rs;
      s[0] += d;
      s[1] += d * xh;
      if constexpr (GATED) {
        const bool on = xv > 0.f;
        s[2] += on ? d : 0.f;
        s[3] += on ? 1.f : 0.f;
        s[4] += on ? xh : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  if (ty == 0 && col < C) {
    float* o = part + ((size_t)g * nchunk + blockIdx.x) * NS * C;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      float a = 0.f;
      for (unsigned l = 0; l < RL; ++l) a += red[k][l * CPB + tx];
      o[k * C + col] = a;
    }
  }
}

// sums[g][2][C]; optional dgamma/dbeta (BatchNorm: G == 1) with accumulate flag; GATED: dbias (see norm_bwd_partial_k).
template <bool GATED>
__global__ __launch_bounds__(1024) void norm_bwd_final_k(const float* __restrict__ part,
                                                        unsigned nchunk, unsigned C,
                                                        float* __restrict__ sums,
                                                        float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int accumulate,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        float invR, float* __restrict__ dbias, int accumulate_bias) {
  constexpr int NS = GATED ? 5 : 2;
  __shared__ float sa[NS][1024];
  const unsigned tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const unsigned c = blockIdx.x * 16u + tx;
  const unsigned g = blockIdx.y;
  float a[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) a[k] = 0.f;
  if (c < C) {   // batched loads, then the adds in the old order (see stats_final_k); 80 values at once would spill
    constexpr int FB = GATED ? 8 : 16;
    for (unsigned k0 = ty; k0 < nchunk; k0 += 64u * FB) {
      float v[FB][NS];
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        const unsigned k = k0 + 64u * j;
        const float* o = part + ((size_t)g * nchunk + (k < nchunk ? k : 0u)) * NS * C;
#pragma unroll
        for (int q = 0; q < NS; ++q) v[j][q] = o[q * C + c];
      }
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        if (k0 + 64u * j < nchunk) {
#pragma unroll
          for (int q = 0; q < NS; ++q) a[q] += v[j][q];
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) sa[q][threadIdx.x] = a[q];
  __syncthreads();
  // the 64 lane sums of a column are added in lane order, one thread row per quantity (was: all NS chains in one thread)
  __shared__ float fin[NS][16];
  if (ty < (unsigned)NS) {
    float t = 0.f;
    for (unsigned l = 0; l < 64; ++l) t += sa[ty][l * 16 + tx];
    fin[ty][tx] = t;
  }
  __syncthreads();
  if (ty != 0 || c >= C) return;
#pragma unroll
  for (int q = 0; q < NS; ++q) a[q] = fin[q][tx];
  sums[(size_t)g * 2 * C + c] = a[0];
  sums[(size_t)g * 2 * C + C + c] = a[1];
  if (dgamma) {
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + a[1];
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + a[0];
  }
  if constexpr (GATED) {
    if (dbias) {   // G == 1
      const float sc = gamma ? rstd[c] * gamma[c] : rstd[c];
      const float v = sc * (a[2] - a[0] * invR * a[3] - a[1] * invR * a[4]);
      dbias[c] = (accumulate_bias ? dbias[c] : 0.f) + v;
    }
  }
}

// dx = rstd * gamma * (dy - s1/R - xhat * s2/R)
template <int VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply_k(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ dy, int lddy,
                                                        float* __restrict__ dx, int lddx, unsigned G,
                                                        unsigned R, unsigned C,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ sums, int relu_gate) {
  const unsigned CQ = C / VEC;
  const unsigned total = G * R * CQ;
  const float invR = 1.0f / (float)R;
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned row = idx / CQ, cq = idx - row * CQ;
    const unsigned g = row / R;
    const unsigned c0 = cq * VEC;
    const vec_t xq = *reinterpret_cast<const vec_t*>(x + (size_t)row * ldx + c0);
    const vec_t dq = *reinterpret_cast<const vec_t*>(dy + (size_t)row * lddy + c0);
    const vec_t mq = *reinterpret_cast<const vec_t*>(mean + (size_t)g * C + c0);
    const vec_t rq = *reinterpret_cast<const vec_t*>(rstd + (size_t)g * C + c0);
    const vec_t s1q = *reinterpret_cast<const vec_t*>(sums + (size_t)g * 2 * C + c0);
    const vec_t s2q = *reinterpret_cast<const vec_t*>(sums + (size_t)g * 2 * C + C + c0);
    vec_t o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float xh = (xq[i] - mq[i]) * rq[i];
      float v = dq[i] - s1q[i] * invR - xh * s2q[i] * invR;
      v *= rq[i];
      if (gamma) v *= gamma[c0 + i];
      if (relu_gate && !(xq[i] > 0.f)) v = 0.f;  // x is a ReLU output: also chain through that ReLU
      o[i] = v;
    }
    *reinterpret_cast<vec_t*>(dx + (size_t)row * lddx + c0) = o;
  }
}

// inference BatchNorm backward is never needed on the hot path (test_step runs without grad).

// ---- small statistics domains (R <= 1024 rows): everything in ONE launch ---------------------------------
// Sixteen of the 26 norm layers of a try-on step normalise over at most 768 rows (U-Net levels below 32x24, the
// 16x12 feature maps of the GMM and its regression head).  For those the three-kernel pipeline above is pure launch
// latency, so one block of 1024 threads = 32 channels x 32 row-lanes owns a (group, 32-channel) panel: pass 1
// accumulates shifted moments per thread, the row-lanes are merged in a fixed order through LDS (Chan), pass 2
// re-reads the panel (L2-resident, <= 128 KiB) and writes the result.  Same formulas as the large path.
__global__ __launch_bounds__(1024) void norm_small_fwd_k(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                         int ldy, unsigned R, unsigned C, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float momentum,
                                                         float* __restrict__ y2, int ldy2, int act2, float act2_param) {
  __shared__ float sn[32][33], sm[32][33], s2[32][33];
  __shared__ float bmean[32], brstd[32];
  const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const unsigned col = blockIdx.x * 32u + tx, g = blockIdx.y;
  const bool live = col < C;
  const float* bx = x + (size_t)g * R * ldx + col;
  float cnt = 0.f, shift = 0.f, s = 0.f, ss = 0.f;
  if (live) {
#pragma unroll 8
    for (unsigned r = ty; r < R; r += 32) {
      const float v = bx[(size_t)r * ldx];
      if (cnt == 0.f) shift = v;
      const float d = v - shift;
      s += d;
      ss += d * d;
      cnt += 1.f;
    }
  }
  float mu = 0.f, m2 = 0.f;
  if (cnt > 0.f) {
    mu = shift + s / cnt;
    m2 = ss - s * s / cnt;
    if (m2 < 0.f) m2 = 0.f;
  }
  sn[ty][tx] = cnt; sm[ty][tx] = mu; s2[ty][tx] = m2;
  __syncthreads();
  for (unsigned stride = 16; stride >= 1; stride >>= 1) {  // fixed binary tree over the 32 row-lanes
    if (ty < stride) {
      const Mom m = mom_merge(Mom{sn[ty][tx], sm[ty][tx], s2[ty][tx]},
                              Mom{sn[ty + stride][tx], sm[ty + stride][tx], s2[ty + stride][tx]});
      sn[ty][tx] = m.n; sm[ty][tx] = m.mean; s2[ty][tx] = m.m2;
    }
    __syncthreads();
  }
  if (ty == 0 && live) {
    const Mom acc = {sn[0][tx], sm[0][tx], s2[0][tx]};
    const float var = acc.m2 / acc.n;
    const float rs = 1.0f / sqrtf(var + eps);
    mean[(size_t)g * C + col] = acc.mean;
    rstd[(size_t)g * C + col] = rs;
    bmean[tx] = acc.mean; brstd[tx] = rs;
    if (running_mean) {
      const float unbiased = acc.n > 1.f ? acc.m2 / (acc.n - 1.f) : var;
      running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * acc.mean;
      running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
    }
  }
  __syncthreads();
  if (!live || !y) return;  // y == nullptr: statistics only
  const float m = bmean[tx], rs = brstd[tx];
  const float ga = gamma ? gamma[col] : 1.f, be = gamma ? beta[col] : 0.f;
  float* by = y + (size_t)g * R * ldy + col;
#pragma unroll 8
  for (unsigned r = ty; r < R; r += 32) {
    float v = (bx[(size_t)r * ldx] - m) * rs;
    if (gamma) v = v * ga + be;
    by[(size_t)r * ldy] = v;
    if (y2) y2[((size_t)g * R + r) * ldy2 + col] = so_actf(act2, v, act2_param);
  }
}

__global__ __launch_bounds__(1024) void norm_small_bwd_k(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                         int lddx, unsigned R, unsigned C,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int accumulate, int relu_gate,
                                                         float* __restrict__ dbias, int accumulate_bias) {
  __shared__ float p1[32][33], p2[32][33];
  __shared__ float b1[32], b2[32];
  const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const unsigned col = blockIdx.x * 32u + tx, g = blockIdx.y;
  const bool live = col < C;
  const float mu = live ? mean[(size_t)g * C + col] : 0.f, rs = live ? rstd[(size_t)g * C + col] : 0.f;
  const float* bx = x + (size_t)g * R * ldx + col;
  const float* bd = dy + (size_t)g * R * lddy + col;
  float s1 = 0.f, s2 = 0.f;
  if (live) {
#pragma unroll 8
    for (unsigned r = ty; r < R; r += 32) {
      const float d = bd[(size_t)r * lddy];
      const float xh = (bx[(size_t)r * ldx] - mu) * rs;
      s1 += d;
      s2 += d * xh;
    }
  }
  p1[ty][tx] = s1; p2[ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && live) {
    float a = 0.f, b = 0.f;
    for (unsigned l = 0; l < 32; ++l) { a += p1[l][tx]; b += p2[l][tx]; }
    b1[tx] = a; b2[tx] = b;
    if (dgamma) {
      dgamma[col] = (accumulate ? dgamma[col] : 0.f) + b;
      dbeta[col] = (accumulate ? dbeta[col] : 0.f) + a;
    }
  }
  __syncthreads();
  if (!live && !dbias) return;
  const float invR = 1.0f / (float)R;
  const float a = b1[tx] * invR, b = b2[tx] * invR;
  const float sc = (live && gamma) ? rs * gamma[col] : rs;
  float* bo = dx + (size_t)g * R * lddx + col;
  float sb = 0.f;
#pragma unroll 8
  for (unsigned r = ty; live && r < R; r += 32) {
    const float xv = bx[(size_t)r * ldx];
    const float xh = (xv - mu) * rs;
    float v = (bd[(size_t)r * lddy] - a - xh * b) * sc;
    if (relu_gate && !(xv > 0.f)) v = 0.f;
    bo[(size_t)r * lddx] = v;
    sb += v;
  }
  if (!dbias) return;    // kernel argument: uniform
  // column sums of the input gradient just written = the bias gradient of the convolution in front (G == 1)
  __syncthreads();
  p1[ty][tx] = sb;
  __syncthreads();
  if (ty == 0 && live) {
    float t = 0.f;
    for (unsigned l = 0; l < 32; ++l) t += p1[l][tx];
    dbias[col] = (accumulate_bias ? dbias[col] : 0.f) + t;
  }
}

constexpr long long kSmallRows = 1024;

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

inline void chunking(long long R, int C, unsigned& chunk, unsigned& nchunk) {
  const unsigned CPB = C >= 256 ? 256 : (unsigned)C;
  const unsigned RL = 256 / CPB;
  chunk = EPT * RL;
  // at most ~1024 partial blocks per column tile: every thread then walks a short run of rows (the loop is a chain of
  // dependent-latency batches), and the merge kernel spreads the partials over 64 lanes per column
  const unsigned min_chunk = (unsigned)((R + 1023) / 1024);
  if (chunk < min_chunk) chunk = (min_chunk + RL - 1) / RL * RL;
  nchunk = (unsigned)((R + chunk - 1) / chunk);
}

}  // namespace

extern "C" {

long long so_norm_ws_floats(int G, long long R, int C) {
  unsigned chunk, nchunk;
  chunking(R, C, chunk, nchunk);
  return (long long)G * nchunk * 5 * C + (long long)G * 2 * C;   // backward partials: up to 5 sums per (chunk, column)
}

// Training-mode statistics + normalisation.  mean/rstd: [G][C] outputs (saved for backward).
static int norm_fwd_launch(const float* x, int ldx, float* y, int ldy, float* y2, int ldy2, int act2, float act2_param, int G,
                           long long R, int C, float eps, const float* gamma, const float* beta, float* mean, float* rstd,
                           float* running_mean, float* running_var, float momentum, float* ws, void* stream) {
  if (G <= 0 || R <= 0 || C <= 0) return 0;
  if (running_mean && G != 1) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (R <= kSmallRows) {
    hipLaunchKernelGGL(norm_small_fwd_k, dim3(so_cdiv(C, 32), G), dim3(1024), 0, st, x, ldx, y, ldy, (unsigned)R,
                       (unsigned)C, eps, gamma, beta, mean, rstd, running_mean, running_var, momentum, y2, ldy2, act2, act2_param);
    return SO_LAUNCH_CHECK();
  }
  unsigned chunk, nchunk;
  chunking(R, C, chunk, nchunk);
  const unsigned CPB = C >= 256 ? 256 : (unsigned)C;
  dim3 g1(nchunk, so_cdiv(C, CPB), G);
  hipLaunchKernelGGL(stats_partial_k, g1, dim3(256), 0, st, x, ldx, (unsigned)R, (unsigned)C, chunk,
                     nchunk, ws);
  dim3 g2(so_cdiv(C, 16), G);
  hipLaunchKernelGGL(stats_final_k, g2, dim3(1024), 0, st, ws, nchunk, (unsigned)C, eps, mean, rstd,
                     running_mean, running_var, momentum);
  if (!y) return SO_LAUNCH_CHECK();  // statistics only: the caller normalises inside its own pass (so_spade_norm_fwd)
  const long long total = (long long)G * R * C;
  if ((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && al16(x) && al16(y) && al16(mean) && al16(rstd) &&
      (!y2 || ((ldy2 & 3) == 0 && al16(y2))))
    hipLaunchKernelGGL(norm_apply_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, x, ldx, y, ldy,
                       (unsigned)G, (unsigned)R, (unsigned)C, mean, rstd, gamma, beta, 0, eps, y2, ldy2, act2, act2_param);
  else
    hipLaunchKernelGGL(norm_apply_k<1>, dim3(grid_for(total)), dim3(256), 0, st, x, ldx, y, ldy,
                       (unsigned)G, (unsigned)R, (unsigned)C, mean, rstd, gamma, beta, 0, eps, y2, ldy2, act2, act2_param);
  return SO_LAUNCH_CHECK();
}

int so_norm_fwd(const float* x, int ldx, float* y, int ldy, int G, long long R, int C, float eps,
                const float* gamma, const float* beta, float* mean, float* rstd,
                float* running_mean, float* running_var, float momentum, float* ws, void* stream) {
  return norm_fwd_launch(x, ldx, y, ldy, nullptr, 0, 0, 0.f, G, R, C, eps, gamma, beta, mean, rstd, running_mean, running_var,
                         momentum, ws, stream);
}

// so_norm_fwd with a second output y2 = act(y) written by the same launch (the activation the consumer applies first)
int so_norm_act_fwd(const float* x, int ldx, float* y, int ldy, float* y2, int ldy2, int act, float act_param, int G,
                    long long R, int C, float eps, const float* gamma, const float* beta, float* mean, float* rstd,
                    float* running_mean, float* running_var, float momentum, float* ws, void* stream) {
  if (!y || !y2) return SO_ERR_SHAPE;
  return norm_fwd_launch(x, ldx, y, ldy, y2, ldy2, act, act_param, G, R, C, eps, gamma, beta, mean, rstd, running_mean,
                         running_var, momentum, ws, stream);
}

// Inference-mode normalisation with given statistics (BatchNorm eval: mean = running_mean,
// var = running_var, stat_is_var = 1).
int so_norm_apply(const float* x, int ldx, float* y, int ldy, int G, long long R, int C,
                  const float* mean, const float* stat, int stat_is_var, float eps,
                  const float* gamma, const float* beta, void* stream) {
  if (G <= 0 || R <= 0 || C <= 0) return 0;
  const long long total = (long long)G * R * C;
  hipLaunchKernelGGL(norm_apply_k<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     y, ldy, (unsigned)G, (unsigned)R, (unsigned)C, mean, stat, gamma, beta,
                     stat_is_var, eps, (float*)nullptr, 0, 0, 0.f);
  return SO_LAUNCH_CHECK();
}

static int norm_bwd_launch(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int G,
                           long long R, int C, const float* mean, const float* rstd, const float* gamma,
                           float* dgamma, float* dbeta, int accumulate, int relu_gate, float* dbias, int accumulate_bias,
                           float* ws, void* stream) {
  if (G <= 0 || R <= 0 || C <= 0) return 0;
  if ((dgamma || dbias) && G != 1) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (R <= kSmallRows) {
    hipLaunchKernelGGL(norm_small_bwd_k, dim3(so_cdiv(C, 32), G), dim3(1024), 0, st, x, ldx, dy, lddy, dx, lddx,
                       (unsigned)R, (unsigned)C, mean, rstd, gamma, dgamma, dbeta, accumulate, relu_gate, dbias, accumulate_bias);
    return SO_LAUNCH_CHECK();
  }
  unsigned chunk, nchunk;
  chunking(R, C, chunk, nchunk);
  const unsigned CPB = C >= 256 ? 256 : (unsigned)C;
  float* part = ws;
  float* sums = ws + (size_t)G * nchunk * 5 * C;
  dim3 g1(nchunk, so_cdiv(C, CPB), G);
  dim3 g2(so_cdiv(C, 16), G);
  const float invR = 1.0f / (float)R;
  if (dbias && relu_gate) {
    hipLaunchKernelGGL(norm_bwd_partial_k<true>, g1, dim3(256), 0, st, x, ldx, dy, lddy, (unsigned)R,
                       (unsigned)C, chunk, nchunk, mean, rstd, part);
    hipLaunchKernelGGL(norm_bwd_final_k<true>, g2, dim3(1024), 0, st, part, nchunk, (unsigned)C, sums, dgamma,
                       dbeta, accumulate, rstd, gamma, invR, dbias, accumulate_bias);
  } else {
    hipLaunchKernelGGL(norm_bwd_partial_k<false>, g1, dim3(256), 0, st, x, ldx, dy, lddy, (unsigned)R,
                       (unsigned)C, chunk, nchunk, mean, rstd, part);
    hipLaunchKernelGGL(norm_bwd_final_k<false>, g2, dim3(1024), 0, st, part, nchunk, (unsigned)C, sums, dgamma,
                       dbeta, accumulate, rstd, gamma, invR, (float*)nullptr, 0);
    if (dbias) {
      // ungated: the column sums of dx are -rstd gamma s2/R sum_r xhat = 0 up to round-off (sum_r xhat = 0): exact zero
      if (!accumulate_bias) (void)hipMemsetAsync(dbias, 0, (size_t)C * sizeof(float), st);
    }
  }
  const long long total = (long long)G * R * C;
  if ((C & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0 && (lddx & 3) == 0 && al16(x) && al16(dy) && al16(dx) && al16(mean) &&
      al16(rstd) && al16(sums))
    hipLaunchKernelGGL(norm_bwd_apply_k<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, x, ldx, dy,
                       lddy, dx, lddx, (unsigned)G, (unsigned)R, (unsigned)C, mean, rstd, gamma, sums, relu_gate);
  else
    hipLaunchKernelGGL(norm_bwd_apply_k<1>, dim3(grid_for(total)), dim3(256), 0, st, x, ldx, dy, lddy,
                       dx, lddx, (unsigned)G, (unsigned)R, (unsigned)C, mean, rstd, gamma, sums, relu_gate);
  return SO_LAUNCH_CHECK();
}

int so_norm_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int G,
                long long R, int C, const float* mean, const float* rstd, const float* gamma,
                float* dgamma, float* dbeta, int accumulate, int relu_gate, float* ws, void* stream) {
  return norm_bwd_launch(x, ldx, dy, lddy, dx, lddx, G, R, C, mean, rstd, gamma, dgamma, dbeta, accumulate, relu_gate, nullptr, 0,
                         ws, stream);
}

// so_norm_bwd that also leaves dbias[c] (+)= sum over the rows of dx[r][c] (G must be 1): the bias gradient of the
// convolution whose (ReLU-gated) output this BatchNorm normalises (Conv -> ReLU -> BatchNorm, models/networks/cpvton/warp.py:
// 15-31) - from the backward statistics pass itself instead of a column-sum pass over dx (two launches less per layer).
int so_norm_bwd_bias(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int G,
                     long long R, int C, const float* mean, const float* rstd, const float* gamma,
                     float* dgamma, float* dbeta, int accumulate, int relu_gate, float* dbias, int accumulate_bias,
                     float* ws, void* stream) {
  if (!dbias) return SO_ERR_SHAPE;
  return norm_bwd_launch(x, ldx, dy, lddy, dx, lddx, G, R, C, mean, rstd, gamma, dgamma, dbeta, accumulate, relu_gate, dbias,
                         accumulate_bias, ws, stream);
}

}  // extern "C"
