// Convolutions with FOUR channels on one side (gfx950).
//
// Three launches of a try-on step have a GEMM dimension of 4: the U-Net's last convolution (128 -> 4 channels at
// full resolution; reference models/networks/cpvton/unet.py:146-152), its weight gradient, and the input gradient
// of VGG conv1_1 (3 -> 64, RGB padded to 4; reference models/networks/vgg.py:6-36).  A 32x32 MFMA tile wastes 7/8
// of its rows or columns on them.  v_mfma_f32_4x4x1_16B_f32 runs sixteen independent 4x4 outer products per
// instruction at the full fp32 matrix rate, which fits exactly:
//   conv / dgrad : block b = 4 pixels; A = the 4 weight rows (same for all blocks), B = one input value per
//                  pixel  ->  lane l ends up with the 4 output channels of pixel l in its 4 accumulator registers
//                  (one 16-byte store, no cross-lane shuffle).
//   wgrad        : block b = 4 input channels; A = dy[pixel][0..3] (same for all blocks), B = x[pixel'][c]
//                  ->  lane l accumulates dw[0..3][tap][c0 + l]; one coalesced 256-byte row of x feeds R*S MFMAs.
// Register layout (checked on MI355X by tools/probes/mfma4x4.hip): D_b[i][j] is in lane 4b + j, register i;
// A_b[i] is read from lane 4b + i, B_b[j] from lane 4b + j.
#include "common.h"
#include "thin.h"
#include "../../include/shineon_hip.h"

namespace {

#define SO_OOB 0x80000000u
typedef int so_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 thin_bload(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  const so_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
  f32x4 r;
  r[0] = __int_as_float(v[0]); r[1] = __int_as_float(v[1]);
  r[2] = __int_as_float(v[2]); r[3] = __int_as_float(v[3]);
  return r;
}

struct ThinConv {
  const float* in;
  const float* w;
  const float* bias;
  float* y;
  unsigned in_bytes;
  int ldin, ldy, OH, OW, IH, IW, IC, R, S, pad, M, K, wrows, nbias, act, vec_store;
  float act_param;
};

// One thread per INPUT pixel of an 8 x 32 halo tile (a wave = two 32-pixel row segments).  Instead of gathering the
// R*S taps per output pixel (which would re-read every input line R*S times through L1), each lane reads the IC
// channels of its own pixel exactly once, in consecutive 16-byte quads (a 128-byte line is consumed by the lane
// that fetched it), and accumulates z[tap][0..3] = sum_c x[pix][c] * w[j][tap][c] for ALL taps: R*S independent
// MFMA chains per lane.  The output is the shifted sum out[a][b] = sum_tap z[(a,b) + tap][tap], exchanged through
// LDS (pixel pitch 36 floats: conflict-free b128).  The (8-R+1) x (32-S+1) interior lanes of the tile own outputs.
// Pixels outside the image read through the buffer descriptor at an out-of-range offset (zeros, branch-free).
template <bool FLIP, int R, int S>
__global__ __launch_bounds__(256) void thin_conv_k(const ThinConv p) {
  constexpr int T = R * S, ZP = T * 4;  // z pitch: 36 floats for 3x3
  constexpr int THo = 8 - (R - 1), TWo = 32 - (S - 1);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int KP = p.K + 4;
  float* wl = smem;
  float* zl = smem + 4 * KP;
  const int kq = p.K >> 2;
  for (int idx = tid; idx < 4 * kq; idx += 256) {
    const int row = idx / kq, q = idx - row * kq;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < p.wrows) v = *reinterpret_cast<const f32x4*>(p.w + (long long)row * p.K + q * 4);
    *reinterpret_cast<f32x4*>(wl + row * KP + q * 4) = v;
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const int tiles_w = (p.OW + TWo - 1) / TWo, tiles_h = (p.OH + THo - 1) / THo;
  const int n = blockIdx.x / (tiles_h * tiles_w);
  const int trem = blockIdx.x - n * (tiles_h * tiles_w);
  const int th = trem / tiles_w, tw = trem - th * tiles_w;
  const int oh0 = th * THo, ow0 = tw * TWo;
  const int lr = tid >> 5, lc = tid & 31;
  // input coordinates of this lane's halo pixel
  const int ih = FLIP ? oh0 + p.pad - (R - 1) + lr : oh0 - p.pad + lr;
  const int iw = FLIP ? ow0 + p.pad - (S - 1) + lc : ow0 - p.pad + lc;
  const bool ok = ((unsigned)ih < (unsigned)p.IH) & ((unsigned)iw < (unsigned)p.IW);
  const unsigned base = ok ? (unsigned)(((n * p.IH + ih) * p.IW + iw) * p.ldin) * 4u : SO_OOB;

  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wrow = wl + (lane & 3) * KP;
  const int Q = p.IC >> 2;
  // The lane's pixel is fetched PF quads (64 bytes) ahead: one 16-byte load per 36 MFMAs left a single request in flight
  // per wave and the loop ran at the HBM latency (58 us for the 128 -> 4 layer against 16 us of MFMA issue).  Quads past
  // the last channel read through an out-of-range offset (zeros) and are never multiplied.
  constexpr int PF = 4;
  f32x4 cur[PF], nxt[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) cur[j] = thin_bload(rs, j < Q ? base + (unsigned)j * 16u : SO_OOB);
  for (int q0 = 0; q0 < Q; q0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) nxt[j] = thin_bload(rs, q0 + PF + j < Q ? base + (unsigned)(q0 + PF + j) * 16u : SO_OOB);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      if (q0 + j < Q) {
        f32x4 wv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) wv[t] = *reinterpret_cast<const f32x4*>(wrow + t * p.IC + (q0 + j) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[t][k], cur[j][k], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) cur[j] = nxt[j];
  }

  f32x4 v = acc[0];
  if constexpr (T > 1) {
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(zl + tid * ZP + t * 4) = acc[t];
    __syncthreads();
    v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (lr < THo && lc < TWo) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int dr = FLIP ? R - 1 - r : r, ds = FLIP ? S - 1 - s : s;
          const f32x4 z = *reinterpret_cast<const f32x4*>(zl + ((lr + dr) * 32 + lc + ds) * ZP + (r * S + s) * 4);
          v[0] += z[0]; v[1] += z[1]; v[2] += z[2]; v[3] += z[3];
        }
    }
  }
  const int oh = oh0 + lr, ow = ow0 + lc;
  if (lr >= THo || lc >= TWo || oh >= p.OH || ow >= p.OW) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = v[i];
    if (p.bias && i < p.nbias) a += p.bias[i];
    v[i] = so_actf(p.act, a, p.act_param);
  }
  float* dst = p.y + ((long long)(n * p.OH + oh) * p.OW + ow) * p.ldy;
  if (p.vec_store) {
    *reinterpret_cast<f32x4*>(dst) = v;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = v[i];
  }
}

struct ThinWgrad {
  const float* dy;
  const float* x;
  float* ws;
  int lddy, ldx, Nb, H, W, C, Ho, Wo, pad, rpb, cgb, LW;
};

// Block = `rpb` consecutive input rows (n, hi) x up to 256 channels.  The R dy rows a row of x pairs with are
// staged (4 channels per pixel, zero halo) in LDS; wave w owns channel group w % cgb (64 channels, lane = channel)
// and every (4 / cgb)-th pixel of the row.  Partial sums go to slab[blockIdx.x * subs + sub][4][R*S*C]; the slabs
// are added in a fixed order by thin_reduce_k (deterministic, like the engine's split-K).
template <int R, int S>
__global__ __launch_bounds__(256) void thin_wgrad_k(const ThinWgrad p) {
  extern __shared__ __attribute__((aligned(16))) float dyl[];  // [R][LW][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = wave % p.cgb, sub = wave / p.cgb, subs = 4 / p.cgb;
  const int c0 = (blockIdx.y * p.cgb + cg) * 64;
  f32x4 acc[R * S];
#pragma unroll
  for (int t = 0; t < R * S; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int rows = p.Nb * p.H;
  for (int rr = 0; rr < p.rpb; ++rr) {
    const int row = blockIdx.x * p.rpb + rr;
    if (row >= rows) break;
    const int n = row / p.H, hi = row - n * p.H;
    __syncthreads();
    for (int idx = tid; idx < R * p.LW; idx += 256) {
      const int r = idx / p.LW, col = idx - r * p.LW;
      const int ho = hi + p.pad - r, wo = col - (S - 1);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo)
        v = *reinterpret_cast<const f32x4*>(p.dy + ((long long)(n * p.Ho + ho) * p.Wo + wo) * p.lddy);
      *reinterpret_cast<f32x4*>(dyl + idx * 4) = v;
    }
    __syncthreads();
    const float* xrow = p.x + (long long)(n * p.H + hi) * p.W * p.ldx + c0 + lane;
    const float* dl = dyl + (lane & 3);
    constexpr int U = 8;  // x values in flight per lane (4 left the loop waiting on HBM: 24 KB outstanding per CU)
    for (int wi0 = sub; wi0 < p.W; wi0 += U * subs) {
      float xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int wi = wi0 + u * subs;
        xv[u] = wi < p.W ? xrow[(long long)wi * p.ldx] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int wi = wi0 + u * subs;
        const int wic = wi < p.W ? wi : 0;  // xv is 0 beyond the row; keep the LDS address in range
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const float a = dl[(r * p.LW + wic + p.pad - s + (S - 1)) * 4];
            acc[r * S + s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, xv[u], acc[r * S + s], 0, 0, 0);
          }
      }
    }
  }
  const long long E = 4LL * R * S * p.C;
  float* slab = p.ws + (long long)(blockIdx.x * subs + sub) * E;
#pragma unroll
  for (int t = 0; t < R * S; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) slab[((long long)i * R * S + t) * p.C + c0 + lane] = acc[t][i];
}

// out[e] = (accumulate ? out[e] : 0) + sum_slab ws[slab][e]; block = 64 elements x 16 slab groups.
__global__ __launch_bounds__(1024) void thin_reduce_k(const float* ws, int nslab, long long E, float* out, int accumulate) {
  __shared__ float part[16][64];
  const int ex = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + ex;
  float s = 0.f;
  if (e < E) {
#pragma unroll 8
    for (int k = g; k < nslab; k += 16) s += ws[(long long)k * E + e];
  }
  part[g][ex] = s;
  __syncthreads();
  if (g == 0 && e < E) {
    float t = accumulate ? out[e] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][ex];
    out[e] = t;
  }
}

// ---- four INPUT channels -> N outputs (3x3, stride 1, pad 1): K = 36 --------------------------------------------------
// VGG conv1_1 (RGB padded to 4 -> 64) and the input gradient of the U-Net's last convolution (4 -> 128).  In the general
// engine these are all prologue and epilogue (one or two k-steps per 64x64 tile: 73 / 58 us against ~25 us of output
// stores).  Here a block stages the (8 + 2) x (32 + 2) pixel halo once (16 bytes per pixel), a wave owns 32-pixel row
// segments and builds the A fragments of v_mfma_f32_32x32x2 straight from it: lane (m, kh) supplies
// A[m][k = 2i + kh] = in[row + tap / 3][m + tap % 3][c] with tap = i / 2, c = 2 (i % 2) + kh - one ds_read_b128 per tap.
// B[k][n] sits in LDS transposed once per block (36 x N floats) and in registers per 64-column pass; the 32x32 accumulator
// tile is stored row by row (32 lanes = one 128-byte line of an output pixel).
struct ThinExpand {
  const float* in;     // [Nb][H][W][4]
  const float* w;
  const float* bias;   // N entries or nullptr
  float* y;
  int ldy, Nb, H, W, N, flip, wmode;
  float slope;         // epilogue v > 0 ? v : v * slope  (1: none, 0: ReLU, a: LeakyReLU) - a runtime activation switch inlined
};                     // at 32 store sites made the kernel 12 000 instructions long (instruction-cache bound: 63 us instead of 30)

constexpr int TE_ROWS = 8, TE_COLS = 32, TE_HP = TE_COLS + 2;

__global__ __launch_bounds__(256) void thin_expand_k(const ThinExpand p) {
  extern __shared__ __attribute__((aligned(16))) float te_smem[];
  f32x4* halo = reinterpret_cast<f32x4*>(te_smem);                 // [(TE_ROWS + 2)][TE_HP] pixels
  float* wl = te_smem + (TE_ROWS + 2) * TE_HP * 4;                 // [36][N + 32]
  const int NP = p.N + 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_w = (p.W + TE_COLS - 1) / TE_COLS, tiles_h = (p.H + TE_ROWS - 1) / TE_ROWS;
  const int n_img = blockIdx.x / (tiles_h * tiles_w);
  const int trem = blockIdx.x - n_img * (tiles_h * tiles_w);
  const int h0 = (trem / tiles_w) * TE_ROWS, w0 = (trem % tiles_w) * TE_COLS;
  for (int i = tid; i < (TE_ROWS + 2) * TE_HP; i += 256) {
    const int h = h0 - 1 + i / TE_HP, w = w0 - 1 + i % TE_HP;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
      v = *reinterpret_cast<const f32x4*>(p.in + ((long long)(n_img * p.H + h) * p.W + w) * 4);
    halo[i] = v;
  }
  // weights -> wl[k = tap * 4 + c][n]; wmode 0: w[n][tap][c] (OHWI, C = 4: forward); 1: w[c][tap][n] (OHWI with four output
  // rows, read for the input gradient: k = tap * 4 + ko).  flip: tap -> 8 - tap (gradient taps run backwards).
  // A thread's <= 9 quads (N <= 256) are fetched in ONE batch of 16-byte loads (was a scalar load -> ds_write loop of 18 rounds
  // per thread): 11.6 -> 10.2 us per launch on a 4 x 16 x 12 map, 34.8 -> 32.0 on 4 x 256 x 192 (profiles/r06_thin4_launches.txt).
  // p.w is 16-byte aligned (so_thin_expand checks; both callers in igemm2.hip already require it).
  constexpr int WQ = 9;
  const int nq = 9 * p.N;
  f32x4 wq[WQ];
#pragma unroll
  for (int j = 0; j < WQ; ++j) {
    const int q = tid + 256 * j;
    wq[j] = q < nq ? *reinterpret_cast<const f32x4*>(p.w + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < WQ; ++j) {
    const int q = tid + 256 * j;
    if (q >= nq) continue;
    if (p.wmode == 0) {   // quad = the four channels of one (n, tap)
      const int n = q / 9, tap = q - n * 9;
      const int kk = (p.flip ? 8 - tap : tap) << 2;
#pragma unroll
      for (int c = 0; c < 4; ++c) wl[(kk + c) * NP + n] = wq[j][c];
    } else {              // quad = four consecutive n of one (c, tap) row
      const int row = (4 * q) / p.N, n = 4 * q - row * p.N;
      const int tap = row % 9, c = row / 9;
      const int kk = ((p.flip ? 8 - tap : tap) << 2) + c;
      *reinterpret_cast<f32x4*>(wl + kk * NP + n) = wq[j];
    }
  }
  __syncthreads();
  const int m = lane & 31, kh = lane >> 5;
  for (int n0 = 0; n0 < p.N; n0 += 64) {
    float b0[18], b1[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      b0[i] = wl[(2 * i + kh) * NP + n0 + m];
      b1[i] = wl[(2 * i + kh) * NP + n0 + 32 + m];
    }
    const float bias0 = p.bias ? p.bias[n0 + m] : 0.f, bias1 = p.bias ? p.bias[n0 + 32 + m] : 0.f;
    for (int row = wave; row < TE_ROWS; row += 4) {
      const int h = h0 + row;
      if (h >= p.H) break;
      f32x16 acc0, acc1;
#pragma unroll
      for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const f32x4 v = halo[(row + tap / 3) * TE_HP + m + tap % 3];
        const float ae = kh ? v[1] : v[0], ao = kh ? v[3] : v[2];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b0[2 * tap], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b1[2 * tap], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, b0[2 * tap + 1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, b1[2 * tap + 1], acc1, 0, 0, 0);
      }
      float* dst = p.y + ((long long)(n_img * p.H + h) * p.W + w0) * p.ldy + n0 + m;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int pm = (j >> 2) * 8 + kh * 4 + (j & 3);   // pixel (row of the 32x32 tile) held by accumulator register j
        if (w0 + pm < p.W) {
          const float v0 = acc0[j] + bias0, v1 = acc1[j] + bias1;
          dst[(long long)pm * p.ldy] = v0 > 0.f ? v0 : v0 * p.slope;
          dst[(long long)pm * p.ldy + 32] = v1 > 0.f ? v1 : v1 * p.slope;
        }
      }
    }
  }
}

}  // namespace

int so_thin_expand(int flip, int wmode, const float* in, const float* w, const float* bias, float* y, int ldy, int Nb, int H,
                   int W, int N, int act, float act_param, hipStream_t stream) {
  if ((N & 63) || N > 256 || (long long)Nb * H * W >= (1 << 30) || (((uintptr_t)w) & 15)) return 1;
  if (act != SO_ACT_NONE && act != SO_ACT_RELU && act != SO_ACT_LEAKY) return 1;
  ThinExpand p = {};
  p.in = in; p.w = w; p.bias = bias; p.y = y; p.ldy = ldy; p.Nb = Nb; p.H = H; p.W = W; p.N = N; p.flip = flip; p.wmode = wmode;
  p.slope = act == SO_ACT_NONE ? 1.f : (act == SO_ACT_RELU ? 0.f : act_param);
  const size_t lds = ((size_t)(TE_ROWS + 2) * TE_HP * 4 + (size_t)36 * (N + 32)) * sizeof(float);
  const long long M = (long long)Nb * H * W;
  const int slot = so_prof_begin((flip ? 1 : 0) * 8 + 6, 2.0 * (double)M * N * 36.0, (int)M, N, 36, stream);
  const dim3 grid((unsigned)((long long)Nb * so_cdiv(H, TE_ROWS) * so_cdiv(W, TE_COLS)));
  hipLaunchKernelGGL(thin_expand_k, grid, dim3(256), lds, stream, p);
  so_prof_end(slot, stream);
  return SO_LAUNCH_CHECK();
}

int so_thin_conv(int flip, const float* in, int ldin, const float* w, int wrows, const float* bias, int nbias,
                 float* y, int ldy, int Nb, int OH, int OW, int IH, int IW, int IC, int R, int S, int pad, int act,
                 float act_param, hipStream_t stream) {
  const long long K = (long long)R * S * IC;
  const long long M = (long long)Nb * OH * OW;
  const long long in_bytes = ((long long)Nb * IH * IW - 1) * ldin * 4 + (long long)IC * 4;
  if (!((R == 3 && S == 3) || (R == 1 && S == 1)) || (IC & 3) || (ldin & 3) || K > 4092 || M <= 0 || M >= (1 << 30) || in_bytes >= 0x7fffffffLL || wrows < 1 || wrows > 4)
    return 1;
  ThinConv p = {};
  p.in = in; p.w = w; p.bias = bias; p.y = y;
  p.in_bytes = (unsigned)in_bytes;
  p.ldin = ldin; p.ldy = ldy; p.OH = OH; p.OW = OW; p.IH = IH; p.IW = IW; p.IC = IC; p.R = R; p.S = S; p.pad = pad;
  p.M = (int)M; p.K = (int)K; p.wrows = wrows; p.nbias = nbias; p.act = act; p.act_param = act_param;
  p.vec_store = ((ldy & 3) == 0 && (((uintptr_t)y) & 15) == 0) ? 1 : 0;
  const bool k3 = R == 3 && S == 3;
  const int THo = k3 ? 6 : 8, TWo = k3 ? 30 : 32;
  const size_t lds = ((size_t)4 * (K + 4) + (k3 ? 256 * 36 : 0)) * sizeof(float);
  if (lds > 65536) return 1;
  const int slot = so_prof_begin((flip ? 1 : 0) * 8 + 6, 2.0 * M * 4.0 * (double)K, (int)M, 4, (int)K, stream);
  const dim3 grid((unsigned)((long long)Nb * so_cdiv(OH, THo) * so_cdiv(OW, TWo)));
  if (k3) {
    if (flip)
      hipLaunchKernelGGL((thin_conv_k<true, 3, 3>), grid, dim3(256), lds, stream, p);
    else
      hipLaunchKernelGGL((thin_conv_k<false, 3, 3>), grid, dim3(256), lds, stream, p);
  } else {
    if (flip)
      hipLaunchKernelGGL((thin_conv_k<true, 1, 1>), grid, dim3(256), lds, stream, p);
    else
      hipLaunchKernelGGL((thin_conv_k<false, 1, 1>), grid, dim3(256), lds, stream, p);
  }
  so_prof_end(slot, stream);
  return SO_LAUNCH_CHECK();
}

int so_thin_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int accumulate, int Nb, int H, int W,
                  int C, int Ho, int Wo, int R, int S, int pad, float* ws, long long ws_bytes, hipStream_t stream) {
  if ((C & 63) || (lddy & 3) || !ws || !((R == 3 && S == 3) || (R == 1 && S == 1))) return 1;
  const int cgb = C >= 256 ? 4 : C / 64;      // channel groups per block: 1, 2 or 4
  if (cgb == 3 || C % (64 * cgb)) return 1;
  const int subs = 4 / cgb;
  const long long E = 4LL * R * S * C;
  const int rows = Nb * H;
  int rpb = rows / 512 > 0 ? rows / 512 : 1;  // aim for >= 512 blocks, then shrink the slab count to fit ws
  while (rpb <= 64 && (long long)so_cdiv(rows, rpb) * subs * E * 4 > ws_bytes) rpb *= 2;
  if (rpb > 64) return 1;
  ThinWgrad p = {};
  p.dy = dy; p.x = x; p.ws = ws;
  p.lddy = lddy; p.ldx = ldx; p.Nb = Nb; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.pad = pad;
  p.rpb = rpb; p.cgb = cgb; p.LW = W + pad + S;
  const dim3 grid((unsigned)so_cdiv(rows, rpb), (unsigned)(C / (64 * cgb)));
  const size_t lds = (size_t)R * p.LW * 4 * sizeof(float);
  const long long Kpix = (long long)Nb * Ho * Wo;
  const int slot = so_prof_begin(2 * 8 + 6, 2.0 * 4.0 * (double)(R * S * C) * (double)Kpix, 4, R * S * C, (int)Kpix, stream);
  if (R == 3)
    hipLaunchKernelGGL((thin_wgrad_k<3, 3>), grid, dim3(256), lds, stream, p);
  else
    hipLaunchKernelGGL((thin_wgrad_k<1, 1>), grid, dim3(256), lds, stream, p);
  so_prof_end(slot, stream);
  int err = SO_LAUNCH_CHECK();
  if (err) return err;
  hipLaunchKernelGGL(thin_reduce_k, dim3((unsigned)so_cdiv(E, 64)), dim3(1024), 0, stream, (const float*)ws,
                     (int)(grid.x * subs), E, dw, accumulate);
  return SO_LAUNCH_CHECK();
}
