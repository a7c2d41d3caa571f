// SAGAN self-attention core, LDS-resident (reference: models/networks/attention/sagan.py:29-54).
//
//   energy[b,i,j] = sum_d q[b,i,d] k[b,j,d]      (no 1/sqrt(d))
//   a = softmax_j(energy);  o[b,i,c] = sum_j a[b,i,j] v[b,j,c];  out = gamma * o + x
//
// The 1x1 query/key/value projections stay ONE engine GEMM into a [B*N][E = 2d + C] buffer (q | k | v column slices,
// ops.py::_SelfAttentionQkvFn); everything after it used to be four launches (QK^T GEMM, softmax, AV GEMM, gamma*o + x)
// on 64x64 tiles at N = H*W <= 192, i.e. pinned at the launch floor.  Here one block owns 32 queries of one batch
// element and one 128-channel slice of the output: the 32 x N energy tile is produced by fp32 MFMAs straight into LDS,
// soft-maxed there, and multiplied with V (staged through LDS in 32-key chunks) without ever leaving the CU.
// fp32 MFMA 32x32x2 fragments are one float per lane, A[m = lane & 31][k = lane >> 5] and B[k = lane >> 5][n = lane & 31],
// so the attention tile (row pitch N + 1) and the V chunk ([key][channel]) are read conflict-free with ds_read_b32 and
// need no transposition.
//
// Small contractions (energy, da, dq, dk: one to six 32x32 tiles with K = 64 ... 512) are latency chains if one wave
// owns a tile; instead ALL four waves work on every tile, each on a quarter of the K steps, and the four partial tiles are
// summed through LDS in fixed wave order (deterministic).
//
// Backward: attn_bwd_q_k (per 32 queries: da = gamma * dout v^T, softmax backward, dq = de k) and attn_bwd_kv_k (per 32
// keys and 128-channel slice: dk = de^T q, dv = gamma * a^T dout) replace six launches by two; the sums over queries / keys
// stay inside one block, so there are no atomics.
#include "common.h"
#include "../../include/shineon_hip.h"

namespace {

constexpr int kQB = 32;      // queries (or keys) per block
constexpr int kMaxN = 192;   // LDS budget: the 32 x N tile + staging chunks
constexpr int kMaxT = 6;     // kMaxN / 32

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// out[32][32 * nt] (LDS, pitch ldo) = scale * A B for a K of 2 * ksteps, all four waves splitting the K steps.
//   A element (m, k): AT ? A[k * lda + m] : A[m * lda + k];   B element (k, n): BT ? B[n * ldb + k] : B[k * ldb + n]
// Ends with a __syncthreads(); `out` must not alias A or B.
template <bool AT, bool BT>
__device__ __forceinline__ void small_gemm_splitk(const float* A, int lda, const float* B, int ldb, int nt, int ksteps,
                                                  float* out, int ldo, float scale, int wave, int li, int lh) {
  f32x16 acc[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int s = wave; s < ksteps; s += 4) {
    const int k = 2 * s + lh;
    const float a = AT ? A[k * lda + li] : A[li * lda + k];
#pragma unroll
    for (int t = 0; t < kMaxT; ++t)
      if (t < nt) {
        const float bb = BT ? B[(t * 32 + li) * ldb + k] : B[k * ldb + t * 32 + li];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
      }
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < kMaxT; ++t)
        if (t < nt)
          for (int r = 0; r < 16; ++r) {
            float* dst = out + acc_row(r, lh) * ldo + t * 32 + li;
            *dst = (w == 0 ? 0.f : *dst) + scale * acc[t][r];
          }
    }
    __syncthreads();
  }
}

// grid (NP / 32, B, C / 128): 32 queries x 128 output channels per block
__global__ __launch_bounds__(256) void attn_fwd_k(const float* __restrict__ qkv, int E, int d, const float* __restrict__ x,
                                                  int ldx, const float* __restrict__ gamma, float* __restrict__ out, int ldo,
                                                  float* __restrict__ attn, float* __restrict__ o, int N, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NP = (N + 31) & ~31;
  const int ldq = d + 1, lde = NP + 1;
  float* qs = smem;                          // [32][d + 1]
  float* es = qs + kQB * ldq;                // [32][NP + 1]
  float* kv = es + kQB * lde;                // keys [NP][d + 1], later V chunks [32][128]
  kv = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(kv) + 15) & ~uintptr_t(15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, q0 = blockIdx.x * kQB, cz = blockIdx.z * 128;
  const float* base = qkv + (size_t)b * N * E;
  const int dq4 = d / 4;
  for (int idx = tid; idx < kQB * dq4; idx += 256) {
    const int i = idx / dq4, d4 = idx - i * dq4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q0 + i < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)(q0 + i) * E + d4 * 4);
    for (int u = 0; u < 4; ++u) qs[i * ldq + d4 * 4 + u] = v[u];
  }
  for (int idx = tid; idx < NP * dq4; idx += 256) {
    const int j = idx / dq4, d4 = idx - j * dq4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)j * E + d + d4 * 4);
    for (int u = 0; u < 4; ++u) kv[j * ldq + d4 * 4 + u] = v[u];
  }
  __syncthreads();
  small_gemm_splitk<false, true>(qs, ldq, kv, ldq, NP / 32, d / 2, es, lde, 1.0f, wave, li, lh);
  {  // softmax over the N valid keys of each row: 8 threads per row
    const int row = tid >> 3, sub = tid & 7;
    float m = -3.0e38f;
    for (int j = sub; j < N; j += 8) m = fmaxf(m, es[row * lde + j]);
    for (int off = 1; off < 8; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float sum = 0.f;
    for (int j = sub; j < N; j += 8) {
      const float e = expf(es[row * lde + j] - m);
      es[row * lde + j] = e;
      sum += e;
    }
    for (int off = 1; off < 8; off <<= 1) sum += __shfl_xor(sum, off, 64);
    const float inv = 1.0f / sum;
    const bool store = q0 + row < N && blockIdx.z == 0;
    for (int j = sub; j < NP; j += 8) {
      const float a = j < N ? es[row * lde + j] * inv : 0.f;
      es[row * lde + j] = a;
      if (store && j < N) attn[((size_t)b * N + q0 + row) * N + j] = a;
    }
  }
  __syncthreads();
  // o = a V for this block's 128 channels: wave w owns columns [32w, 32w + 32); V staged as [32 keys][128]
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kc = 0; kc < NP / 32; ++kc) {
    for (int idx = tid; idx < 32 * 32; idx += 256) {
      const int key = idx >> 5, c4 = idx & 31;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kc * 32 + key < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)(kc * 32 + key) * E + 2 * d + cz + c4 * 4);
      *reinterpret_cast<f32x4*>(kv + key * 128 + c4 * 4) = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int s = 0; s < 16; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(es[li * lde + kc * 32 + 2 * s + lh], kv[(2 * s + lh) * 128 + wave * 32 + li], acc, 0, 0, 0);
    __syncthreads();
  }
  const float g = gamma[0];
  const int col = cz + wave * 32 + li;
  for (int r = 0; r < 16; ++r) {
    const int row = q0 + acc_row(r, lh);
    if (row < N) {
      const size_t pix = (size_t)b * N + row;
      o[pix * C + col] = acc[r];
      out[pix * ldo + col] = g * acc[r] + x[pix * ldx + col];
    }
  }
}

// Per 32 queries: da = gamma * dout v^T (K = C, staged in 64-channel chunks), de = a * (da - sum_j a da), dq = de k.
__global__ __launch_bounds__(256) void attn_bwd_q_k(const float* __restrict__ qkv, int E, int d, const float* __restrict__ dout,
                                                    int ldg, const float* __restrict__ attn, const float* __restrict__ gamma,
                                                    float* __restrict__ de, float* __restrict__ dqkv, int N, int C) {
  constexpr int CH = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NP = (N + 31) & ~31;
  const int ldc = CH + 1, lde = NP + 1;
  float* gs = smem;                 // dout chunk [32][CH + 1]
  float* vs = gs + kQB * ldc;       // v chunk [NP][CH + 1], later k as [key][d]
  float* es = vs + NP * ldc;        // [32][NP + 1]: da, then de
  float* dqs = es + kQB * lde;      // [32][d + 1]: dq tile before it is written out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, q0 = blockIdx.x * kQB;
  const float* base = qkv + (size_t)b * N * E;
  const int nt = NP / 32;
  // every wave accumulates ALL nt tiles over its quarter of the K steps of each chunk
  f32x16 acc[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int c0 = 0; c0 < C; c0 += CH) {
    for (int idx = tid; idx < kQB * (CH / 4); idx += 256) {
      const int i = idx / (CH / 4), c4 = idx - i * (CH / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q0 + i < N) v = *reinterpret_cast<const f32x4*>(dout + ((size_t)b * N + q0 + i) * ldg + c0 + c4 * 4);
      for (int u = 0; u < 4; ++u) gs[i * ldc + c4 * 4 + u] = v[u];
    }
    for (int idx = tid; idx < NP * (CH / 4); idx += 256) {
      const int j = idx / (CH / 4), c4 = idx - j * (CH / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)j * E + 2 * d + c0 + c4 * 4);
      for (int u = 0; u < 4; ++u) vs[j * ldc + c4 * 4 + u] = v[u];
    }
    __syncthreads();
    for (int s = wave; s < CH / 2; s += 4) {
      const float a = gs[li * ldc + 2 * s + lh];
#pragma unroll
      for (int t = 0; t < kMaxT; ++t)
        if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vs[(t * 32 + li) * ldc + 2 * s + lh], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
  const float g = gamma[0];
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < kMaxT; ++t)
        if (t < nt)
          for (int r = 0; r < 16; ++r) {
            float* dst = es + acc_row(r, lh) * lde + t * 32 + li;
            *dst = (w == 0 ? 0.f : *dst) + g * acc[t][r];
          }
    }
    __syncthreads();
  }
  // keys as [key][d] for dq (B element (k = key, n = d)); the v chunk region is free now
  for (int idx = tid; idx < NP * (d / 4); idx += 256) {
    const int j = idx / (d / 4), d4 = idx - j * (d / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)j * E + d + d4 * 4);
    *reinterpret_cast<f32x4*>(vs + j * d + d4 * 4) = v;
  }
  {  // softmax backward per row: de = a * (da - sum_j a * da)
    const int row = tid >> 3, sub = tid & 7;
    const bool valid = q0 + row < N;
    const float* arow = attn + ((size_t)b * N + q0 + row) * N;
    float dotp = 0.f;
    if (valid)
      for (int j = sub; j < N; j += 8) dotp += arow[j] * es[row * lde + j];
    for (int off = 1; off < 8; off <<= 1) dotp += __shfl_xor(dotp, off, 64);
    for (int j = sub; j < NP; j += 8) {
      float v = 0.f;
      if (valid && j < N) {
        v = arow[j] * (es[row * lde + j] - dotp);
        de[((size_t)b * N + q0 + row) * N + j] = v;
      }
      es[row * lde + j] = v;
    }
  }
  __syncthreads();
  // dq[32][d] = de[32][NP] k[NP][d]
  small_gemm_splitk<false, false>(es, lde, vs, d, d / 32, NP / 2, dqs, d + 1, 1.0f, wave, li, lh);
  for (int idx = tid; idx < kQB * d; idx += 256) {
    const int i = idx / d, dd = idx - i * d;
    if (q0 + i < N) dqkv[((size_t)b * N + q0 + i) * E + dd] = dqs[i * (d + 1) + dd];
  }
}

// grid (NP / 32, B, C / 128).  Per 32 keys and 128-channel slice: dv[j][c] = gamma * sum_i a[i][j] dout[i][c]; the z = 0
// slice also produces dk[j][d] = sum_i de[i][j] q[i][d]  (K runs over the queries).
__global__ __launch_bounds__(256) void attn_bwd_kv_k(const float* __restrict__ qkv, int E, int d, const float* __restrict__ dout,
                                                     int ldg, const float* __restrict__ attn, const float* __restrict__ de,
                                                     const float* __restrict__ gamma, float* __restrict__ dqkv, int N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NP = (N + 31) & ~31;
  float* as = smem;                // a[:, j0:j0+32]  as [query][33]
  float* ds = as + NP * 33;        // de[:, j0:j0+32] as [query][33]
  float* qs = ds + NP * 33;        // q as [query][d]
  float* dks = qs + NP * d;        // dk tile [32][d + 1]
  float* gch = dks + kQB * (d + 1);  // dout chunk [32 queries][128]
  gch = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(gch) + 15) & ~uintptr_t(15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, j0 = blockIdx.x * kQB, cz = blockIdx.z * 128;
  const float* base = qkv + (size_t)b * N * E;
  const bool do_dk = blockIdx.z == 0;
  for (int idx = tid; idx < NP * 32; idx += 256) {
    const int i = idx >> 5, jj = idx & 31;
    const bool ok = i < N && j0 + jj < N;
    as[i * 33 + jj] = ok ? attn[((size_t)b * N + i) * N + j0 + jj] : 0.f;
    if (do_dk) ds[i * 33 + jj] = ok ? de[((size_t)b * N + i) * N + j0 + jj] : 0.f;
  }
  if (do_dk)
    for (int idx = tid; idx < NP * (d / 4); idx += 256) {
      const int i = idx / (d / 4), d4 = idx - i * (d / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)i * E + d4 * 4);
      *reinterpret_cast<f32x4*>(qs + i * d + d4 * 4) = v;
    }
  __syncthreads();
  if (do_dk) {  // block-uniform branch (the helper synchronises)
    small_gemm_splitk<true, false>(ds, 33, qs, d, d / 32, NP / 2, dks, d + 1, 1.0f, wave, li, lh);
    for (int idx = tid; idx < kQB * d; idx += 256) {
      const int jj = idx / d, dd = idx - jj * d;
      if (j0 + jj < N) dqkv[((size_t)b * N + j0 + jj) * E + d + dd] = dks[jj * (d + 1) + dd];
    }
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int qc = 0; qc < NP / 32; ++qc) {
    for (int idx = tid; idx < 32 * 32; idx += 256) {
      const int i = idx >> 5, c4 = idx & 31;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (qc * 32 + i < N) v = *reinterpret_cast<const f32x4*>(dout + ((size_t)b * N + qc * 32 + i) * ldg + cz + c4 * 4);
      *reinterpret_cast<f32x4*>(gch + i * 128 + c4 * 4) = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int s = 0; s < 16; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[(qc * 32 + 2 * s + lh) * 33 + li], gch[(2 * s + lh) * 128 + wave * 32 + li], acc, 0, 0, 0);
    __syncthreads();
  }
  const float g = gamma[0];
  const int col = cz + wave * 32 + li;
  for (int r = 0; r < 16; ++r) {
    const int key = j0 + acc_row(r, lh);
    if (key < N) dqkv[((size_t)b * N + key) * E + 2 * d + col] = g * acc[r];
  }
}

inline bool fused_ok(int N, int C, int d, int E, const void* p0) {
  return N >= 1 && N <= kMaxN && C >= 128 && C <= 512 && (C % 128) == 0 && (d == 32 || d == 64) && E == 2 * d + C &&
         (((uintptr_t)p0) & 15) == 0;
}

template <typename K>
int set_lds(K kern, size_t bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" {

int so_attention_supported(int N, int C, int d) {
  return N >= 1 && N <= kMaxN && C >= 128 && C <= 512 && (C % 128) == 0 && (d == 32 || d == 64);
}

int so_attention_fwd(const float* qkv, int E, int d, const float* x, int ldx, const float* gamma, float* out, int ldo,
                     float* attn, float* o, int B, int N, int C, void* stream) {
  if (!fused_ok(N, C, d, E, qkv)) return SO_ERR_SHAPE;
  const int NP = (N + 31) & ~31;
  const size_t kvf = (size_t)NP * (d + 1) > (size_t)32 * 128 ? (size_t)NP * (d + 1) : (size_t)32 * 128;
  const size_t lds = ((size_t)kQB * (d + 1) + (size_t)kQB * (NP + 1) + kvf + 4) * sizeof(float);
  int e = set_lds(attn_fwd_k, lds);
  if (e) return e;
  hipLaunchKernelGGL(attn_fwd_k, dim3(NP / 32, B, C / 128), dim3(256), lds, (hipStream_t)stream, qkv, E, d, x, ldx, gamma, out,
                     ldo, attn, o, N, C);
  return SO_LAUNCH_CHECK();
}

int so_attention_bwd(const float* qkv, int E, int d, const float* dout, int ldg, const float* attn, const float* gamma,
                     float* de, float* dqkv, int B, int N, int C, void* stream) {
  if (!fused_ok(N, C, d, E, qkv) || (ldg & 3) || (((uintptr_t)dout) & 15)) return SO_ERR_SHAPE;
  const int NP = (N + 31) & ~31;
  hipStream_t st = (hipStream_t)stream;
  {
    const size_t lds = ((size_t)kQB * 65 + (size_t)NP * 65 + (size_t)kQB * (NP + 1) + (size_t)kQB * (d + 1)) * sizeof(float);
    int e = set_lds(attn_bwd_q_k, lds);
    if (e) return e;
    hipLaunchKernelGGL(attn_bwd_q_k, dim3(NP / 32, B), dim3(256), lds, st, qkv, E, d, dout, ldg, attn, gamma, de, dqkv, N, C);
    e = SO_LAUNCH_CHECK();
    if (e) return e;
  }
  const size_t lds = ((size_t)NP * 33 * 2 + (size_t)NP * d + (size_t)kQB * (d + 1) + (size_t)32 * 128 + 4) * sizeof(float);
  int e = set_lds(attn_bwd_kv_k, lds);
  if (e) return e;
  hipLaunchKernelGGL(attn_bwd_kv_k, dim3(NP / 32, B, C / 128), dim3(256), lds, st, qkv, E, d, dout, ldg, attn, de, gamma, dqkv, N);
  return SO_LAUNCH_CHECK();
}

}  // extern "C"
