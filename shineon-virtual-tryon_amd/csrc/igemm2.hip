// fp32 MFMA implicit-GEMM engine for gfx950 (MI355X), second structure.
//
// One kernel template covers every contraction on the try-on hot path:
//   FPROP : y[pix][ko]      = sum_{r,s,c} x[pix@(r,s)][c] * w[ko][r][s][c]      (Conv2d forward)
//   DGRAD : dx[pix][c]      = sum_{r,s,ko} dy[pix'@(r,s)][ko] * w[ko][r][s][c]  (Conv2d input gradient;
//                             strided convs are split in stride^2 parity classes so no zero taps are multiplied)
//   WGRAD : dw[ko][r][s][c] = sum_{pix} dy[pix][ko] * x[pix@(r,s)][c]           (Conv2d weight gradient)
//   GEMM  : batched C = op(A) op(B) for attention (QK^T, AV and their gradients) and the correlation volume.
// Reference ops replaced: torch conv2d / bmm as used by models/networks/cpvton/unet.py:129-174,
// models/networks/cpvton/warp.py:9-99, models/networks/attention/sagan.py:38-50, models/networks/vgg.py:6-36.
//
// Design (CDNA4):
//   * activations NHWC, weights OHWI -> the GEMM K axis (channels within a filter tap) is contiguous in HBM for
//     both operands: every global load is a 16-byte quad, 8 lanes cover one 128-byte line.
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  K may be permuted freely as long as A and B agree,
//     so a lane-half reads FOUR consecutive k with one ds_read_b128 and feeds four MFMAs
//     (k = kbase + 4*(lane>>5) + t for MFMA t).
//   * LDS stages are filled by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write (round 5).  The LDS
//     image of a tile is the order in which the lanes fetch it:
//       KC ("k-contiguous" memory)  : tile [rows][32 k], rows unpadded, k quads XOR-swizzled on the source side;
//                                     fragments = ds_read_b128
//       MC ("mn-contiguous" memory) : tile [32 k][rows] as it lies in memory, NOT transposed; fragments = 4 x ds_read_b32
//     FPROP = KC x KC, DGRAD = KC x KC on transposed weights (or KC x MC in place), WGRAD = MC x MC.
//   * BRANCH-FREE staging: operands are read through buffer descriptors; a lane that must see a zero (conv
//     padding, ragged tile edge, K tail) issues the same fill with an out-of-range offset and the hardware writes 0.
//   * two LDS stages: the fill of tile t+1 is issued at the top of tile t and drained by the barrier that ends tile t.
//   * index decode (k -> tap/channel, pixel -> n/h/w) uses a reciprocal-multiply division (operands < 2^24).
//   * 256 threads = 4 waves (2x2); block tiles 64x64, 128x64, 64x128, 128x128; BK = 32.
//   * deterministic split-K (slabs in a workspace + reduce kernel carrying the fused epilogue).
//   * fused epilogue: alpha (device scalar, attention gamma) * acc + bias[n] + residual, then activation.
#include "common.h"
#include "../../include/shineon_hip.h"
#include "thin.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

enum { MODE_FPROP = 0, MODE_DGRAD = 1, MODE_WGRAD = 2, MODE_GEMM = 3 };

struct SoIgemm {
  const float* a;
  const float* b;
  float* c;
  float* ws;
  const float* bias;
  const float* alpha;
  const float* res;
  const float* gate;  // optional ReLU gate (same pitch as c): out = gate > 0 ? out : 0, applied last
  unsigned a_bytes, b_bytes;  // buffer extents for the descriptors (per batch matrix in GEMM mode)
  int M, N, K;
  int lda, ldb, ldc, ldres;
  int Nb, H, W, C;   // input-side tensor (x / dx)
  int Ho, Wo, Ko;    // output-side tensor (y / dy)
  int R, S, stride, pad;
  int TS;            // DGRAD: taps per class along W (= S / stride)
  int H2, W2;        // DGRAD: class grid (= ceil(H / stride), ceil(W / stride))
  int act;
  float act_param;
  int splitk, ktps, nclass;
  int nbias;         // bias entries available (columns >= nbias get no bias: zero channel padding)
  long long sa, sb, sc, sres;  // GEMM batch strides in elements
};

// pins memory ops (global, LDS) and MFMAs in source order; VALU/SALU address arithmetic may float across
#define SO_SB() __builtin_amdgcn_sched_barrier(0x006)
#define SO_OOB 0x80000000u  // byte offset beyond any operand (< 2 GiB each): buffer_load returns 0

// floor(x / d) and x % d for 0 <= x < 2^24, d >= 1, with inv = 1.0f / d: one multiply + one correction step.
__device__ __forceinline__ void so_divmod(unsigned x, unsigned d, float inv, unsigned& q, unsigned& r) {
  q = (unsigned)(__uint2float_rz(x) * inv);
  int rem = (int)x - (int)(q * d);
  const int neg = rem < 0 ? 1 : 0;   // branch-free corrections (selects)
  q -= (unsigned)neg;
  rem += neg ? (int)d : 0;
  const int big = rem >= (int)d ? 1 : 0;
  q += (unsigned)big;
  rem -= big ? (int)d : 0;
  r = (unsigned)rem;
}

// The engine's epilogues apply none / ReLU / LeakyReLU only: the generic activation switch (erf, tanh, sin, ...) inlined at
// every store site made the 64x64 instantiations 17 000 and the 128x128 ones 33-37 000 instructions long (2.5 min of
// compile time, a 7.7 MB library).  so_launch applies any other activation as a second, element-wise pass over the output
// (same fp32 values in, so the results are bit-identical to the fused form).
__device__ __forceinline__ float so_act_epi(const SoIgemm& p, float v) {
  return (p.act == SO_ACT_NONE || v > 0.f) ? v : (p.act == SO_ACT_RELU ? 0.f : v * p.act_param);
}

__device__ __forceinline__ float so_epilogue(const SoIgemm& p, float v, long long res_off, int n) {
  if (p.alpha) v *= p.alpha[0];
  if (p.bias && n < p.nbias) v += p.bias[n];
  if (p.res) v += p.res[res_off + n];
  return so_act_epi(p, v);
}

// Row index of the GEMM -> element offset of that output row (and of the residual row); off < 0: skip the row.
template <int MODE>
__device__ __forceinline__ void so_row_offset(const SoIgemm& p, int cls, int m, long long& off,
                                              long long& roff) {
  if constexpr (MODE == MODE_DGRAD) {
    if (p.nclass > 1) {
      const int ph = cls / p.stride, pw = cls - ph * p.stride;
      const int hw2 = p.H2 * p.W2;
      const int n = m / hw2;
      const int rem = m - n * hw2;
      const int h2 = rem / p.W2;
      const int w2 = rem - h2 * p.W2;
      const int hi = h2 * p.stride + ph, wi = w2 * p.stride + pw;
      if (hi >= p.H || wi >= p.W) {  // ragged class grid (H or W not a multiple of the stride)
        off = -1;
        roff = 0;
        return;
      }
      const long long pix = ((long long)n * p.H + hi) * p.W + wi;
      off = pix * p.ldc;
      roff = pix * p.ldres;
      return;
    }
    off = (long long)m * p.ldc;
    roff = (long long)m * p.ldres;
  } else if constexpr (MODE == MODE_GEMM) {
    off = (long long)cls * p.sc + (long long)m * p.ldc;
    roff = (long long)cls * p.sres + (long long)m * p.ldres;
  } else {
    off = (long long)m * p.ldc;
    roff = (long long)m * p.ldres;
  }
}


// Sum of the split-K slabs of one output quad in a fixed order (deterministic): groups of eight as
// ((0+1)+(2+3))+((4+5)+(6+7)), then a group of four, then singles.
template <typename vec_t>
__device__ __forceinline__ vec_t so_sum_slabs(const float* src, int splitk, long long mn) {
  vec_t v = {};
  int sidx = 0;
  for (; sidx + 8 <= splitk; sidx += 8) {
    vec_t t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const vec_t*>(src + (long long)(sidx + u) * mn);
    v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  if (sidx + 4 <= splitk) {
    vec_t t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const vec_t*>(src + (long long)(sidx + u) * mn);
    v += (t[0] + t[1]) + (t[2] + t[3]);
    sidx += 4;
  }
  for (; sidx < splitk; ++sidx) v += *reinterpret_cast<const vec_t*>(src + (long long)sidx * mn);
  return v;
}

// The fused epilogue on one quad of four consecutive output columns (alpha, bias, residual, activation, ReLU gate).
template <int MODE>
__device__ __forceinline__ void so_epilogue_quad(const SoIgemm& p, f32x4 v, long long off, long long roff, int n) {
  if (p.alpha) { const float al = p.alpha[0]; v[0] *= al; v[1] *= al; v[2] *= al; v[3] *= al; }
  if (p.bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (n + k < p.nbias) v[k] += p.bias[n + k];
  }
  if (p.res) {
    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + roff + n);
    v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
  }
  if (p.act != SO_ACT_NONE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = so_act_epi(p, v[k]);
  }
  if (p.gate) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(p.gate + off + n);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = gv[k] > 0.f ? v[k] : 0.f;
  }
  *reinterpret_cast<f32x4*>(p.c + off + n) = v;
}

typedef int so_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 so_bload(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  const so_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
  f32x4 r;
  r[0] = __int_as_float(v[0]); r[1] = __int_as_float(v[1]);
  r[2] = __int_as_float(v[2]); r[3] = __int_as_float(v[3]);
  return r;
}

// KC tiles ([rows][32 k], k-contiguous memory) use an unpadded 32-float row whose eight 16-byte k-quads are XOR-swizzled with
// so_swz(row): found by exhaustive search over bit-linear swizzles against the gfx950 bank / lane-group rules, it makes the
// ds_read_b128 of a 32-row tile conflict-free.
__device__ __forceinline__ int so_swz(int row) {
  return ((row >> 2) & 1) | (((row >> 3) & 1) << 1) | ((((row >> 1) ^ (row >> 4)) & 1) << 2);
}

// NW = waves per block: 4 (2x2 wave grid) or 8 (2x4, BN = 128 only: twice the waves per SIMD for the same LDS
// footprint, which hides the staging bubbles of the large tiles).
template <int MODE, bool A_MC, bool B_MC, int BM, int BN, int NW>
__global__ __launch_bounds__(NW * 64, 2) void so_igemm_kernel(const SoIgemm p) {
  constexpr int BK = 32;
  constexpr int NT = NW * 64;
  constexpr int LDK = 36;  // row pitch of the epilogue's wave-private transposition patches
  // Staging (round 5): every LDS stage is filled by LDS-DMA (`buffer_load_dwordx4 ... lds`) - no staging registers, no
  // ds_write.  One instruction writes 64 lanes x 16 bytes to one contiguous KB of LDS (wave-uniform base in M0 + lane * 16),
  // i.e. the LDS image of a tile IS the order in which the lanes fetch it:
  //   KC operand (k-contiguous memory): tile [rows][32 k]; an instruction = 8 rows x 128 B.  Rows are unpadded and their eight
  //       k-quads XOR-swizzled with so_swz(row) on the SOURCE side (lane l fetches the quad that belongs in physical quad
  //       l & 7 of its row); fragments = ds_read_b128 (4 consecutive k per lane half), conflict-free.
  //   MC operand (memory contiguous along the GEMM row / column: both wgrad operands, in-place dgrad weights, NN / TN GEMMs):
  //       tile [32 k][rows] exactly as it lies in memory - NO transposition; an instruction = 256 / rows k-rows.  A lane
  //       reads its fragment as four ds_read_b32 (k, k+1, k+2, k+3 at its own row: 32 lanes = 32 consecutive dwords of one
  //       k-row, conflict-free; the compiler pairs them into ds_read2_b32).  Rounds 1-4 transposed these operands in
  //       registers (ds_write_b64 / b128 of assembled pairs; the ablation builds put 25 % of the weight-gradient kernels'
  //       time on that path, profiles/r05_igemm_ablation.txt).
  constexpr int LDA = 32, LDB = 32;   // KC row pitch (floats); MC tiles have row pitch BM / BN
  constexpr int A_STAGE = BM * LDA;
  constexpr int B_STAGE = BN * LDB;
  constexpr int WGN = NW / 2;                          // waves along N
  constexpr int WTM = BM / 2, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
  constexpr int RPP = NT / 8;                          // KC mode: tile rows covered per pass
  constexpr int AJ = BM / RPP, BJ = BN / RPP;          // 16-byte quads staged per thread per K tile
  constexpr int AQPR = BM / 4, BQPR = BN / 4;          // MC mode: quads per k-row
  constexpr int APASS = BK / AJ, BPASS = BK / BJ;      // MC mode: k-rows covered by the block per pass j

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;

  const int tiles_n = (p.N + BN - 1) / BN;
  // XCD-aware block -> tile map.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each
  // with its own 4 MiB L2; handing XCD x the x-th CONTIGUOUS eighth of the (split, tile_m, tile_n) order keeps the
  // activation rows (and their 3x3 halo) of neighbouring tiles in ONE L2 instead of all eight.  Any bijection is
  // correct; this one only changes which L2 a tile's operands are fetched into.
  const unsigned gx = gridDim.x, lin = blockIdx.x + gx * blockIdx.z, tot = gx * gridDim.z;
  const unsigned xper = tot >> 3, xrem = tot & 7, xcd = lin & 7;
  const unsigned lg = xcd * xper + (xcd < xrem ? xcd : xrem) + (lin >> 3);
  const int bz = (int)(lg / gx), bx = (int)(lg - (unsigned)bz * gx);
  const int tile_m = bx / tiles_n;
  const int tile_n = bx - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int cls = bz / p.splitk;
  const int split = bz - cls * p.splitk;

  const float* gA = p.a;
  const float* gB = p.b;
  if constexpr (MODE == MODE_GEMM) {
    gA += (long long)cls * p.sa;
    gB += (long long)cls * p.sb;
  }
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)gA, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)gB, 0, (int)p.b_bytes, 0x00020000);

  // DGRAD parity class constants
  int d_r0 = 0, d_s0 = 0, d_oh = 0, d_ow = 0;
  if constexpr (MODE == MODE_DGRAD) {
    const int ph = cls / p.stride, pw = cls - ph * p.stride;
    d_r0 = (ph + p.pad) % p.stride;
    d_s0 = (pw + p.pad) % p.stride;
    d_oh = (ph + p.pad - d_r0) / p.stride;
    d_ow = (pw + p.pad - d_s0) / p.stride;
  }

  // reciprocals for the index decodes (all dividends < 2^24)
  const float invC = 1.0f / (float)(p.C > 0 ? p.C : 1), invS = 1.0f / (float)(p.S > 0 ? p.S : 1);
  const float invKo = 1.0f / (float)(p.Ko > 0 ? p.Ko : 1), invTS = 1.0f / (float)(p.TS > 0 ? p.TS : 1);
  const float invWo = 1.0f / (float)(p.Wo > 0 ? p.Wo : 1);
  const float invHWo = 1.0f / (float)(p.Ho * p.Wo > 0 ? p.Ho * p.Wo : 1);

  // ---------------- per-thread loader state (K-tile invariant) -----------------
  const int krow8 = tid >> 3;  // KC: row within a 32-row pass
  // KC: the lane's PHYSICAL quad is tid & 7 (LDS-DMA is lane-linear); it holds logical quad (tid & 7) ^ so_swz(row) - the same
  // for every pass j, since so_swz only looks at row bits 1..4 and passes are 32+ rows apart
  const int kq = (tid & 7) ^ so_swz(krow8);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // in an SGPR: LDS-DMA bases (M0) are wave-uniform
  typedef __attribute__((address_space(3))) void* so_lds_ptr;
  // one staged quad of pass j -> LDS stage `fill_st`, KB number wave + NW * j of the tile (KC: rows 8 * that ..+7; MC: k-rows
  // (256 / rows) * that ..)
#define SO_EMIT_A(j, off) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (so_lds_ptr)(As + fill_st * A_STAGE + (wave_u + NW * (j)) * 256), 16, (int)(off), 0, 0, 0)
#define SO_EMIT_B(j, off) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (so_lds_ptr)(Bs + fill_st * B_STAGE + (wave_u + NW * (j)) * 256), 16, (int)(off), 0, 0, 0)
  // A operand, KC: element offset of the row's origin pixel and its (h0, w0); invalid rows get h0 = -2^28
  int a_org[AJ], a_h0[AJ], a_w0[AJ];
  (void)a_org; (void)a_h0; (void)a_w0;
  if constexpr (!A_MC) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = m0 + krow8 + RPP * j;
      a_org[j] = 0; a_h0[j] = -(1 << 28); a_w0[j] = 0;
      if (m < p.M) {
        if constexpr (MODE == MODE_FPROP) {
          const int hw = p.Ho * p.Wo;
          const int n = m / hw;
          const int rem = m - n * hw;
          const int ho = rem / p.Wo;
          const int wo = rem - ho * p.Wo;
          a_h0[j] = ho * p.stride - p.pad;
          a_w0[j] = wo * p.stride - p.pad;
          a_org[j] = ((n * p.H + a_h0[j]) * p.W + a_w0[j]) * p.lda;  // may be negative; only used when in range
        } else if constexpr (MODE == MODE_DGRAD) {
          const int hw2 = p.H2 * p.W2;
          const int n = m / hw2;
          const int rem = m - n * hw2;
          const int h2 = rem / p.W2;
          const int w2 = rem - h2 * p.W2;
          const int ph_ = cls / p.stride, pw_ = cls - ph_ * p.stride;
          if (h2 * p.stride + ph_ < p.H && w2 * p.stride + pw_ < p.W) {
            a_h0[j] = h2 + d_oh;
            a_w0[j] = w2 + d_ow;
            a_org[j] = ((n * p.Ho + a_h0[j]) * p.Wo + a_w0[j]) * p.lda;
          }
        } else {  // GEMM KC
          a_org[j] = m * p.lda;
          a_h0[j] = 0;
        }
      }
    }
  }
  // MC mapping: pass j of wave w is KB number w + NW * j = k-rows (w + NW * j) * (256 / rows) ..; within it lane l holds quad
  // l % QPR of k-row l / QPR.  => the thread's k-row in pass j = a_kr + APASS * j
  const int a_mq = lane % AQPR, a_kr = wave * (64 / AQPR) + lane / AQPR;
  const int b_mq = lane % BQPR, b_kr = wave * (64 / BQPR) + lane / BQPR;
  // B operand, KC (weights / plain rows): element offset of each row, -1 if out of range
  int b_row[BJ];
  (void)b_row;
  if constexpr (!B_MC) {
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int n = n0 + krow8 + RPP * j;
      b_row[j] = n < p.N ? n * p.ldb : -1;
    }
  }
  // WGRAD B: fixed filter tap / channel of this thread's column quad
  int w_r = 0, w_s = 0, w_c = 0;
  bool w_colvalid = true;
  if constexpr (MODE == MODE_WGRAD) {
    const int nn = n0 + b_mq * 4;
    w_colvalid = nn < p.N;
    const int tap = nn / p.C;
    w_c = nn - tap * p.C;
    w_r = tap / p.S;
    w_s = tap - w_r * p.S;
  }

  const int nkt = (p.K + BK - 1) / BK;
  const int kt_begin = split * p.ktps;
  const int kt_end = (kt_begin + p.ktps > nkt) ? nkt : kt_begin + p.ktps;

  // WGRAD B: output pixel (n, ho, wo) of the thread's next k row, advanced incrementally (no per-quad division)
  int wg_n = 0, wg_ho = 0, wg_wo = 0, wg_dw = 0, wg_dh = 0, wg_dn = 0;
  (void)wg_n; (void)wg_ho; (void)wg_wo; (void)wg_dw; (void)wg_dh; (void)wg_dn;
  // DGRAD with in-place weights (B_MC): (ko, ts, tr) of the thread's next k row
  int dg_ko = 0, dg_ts = 0, dg_tr = 0, dg_dko = 0, dg_dts = 0, dg_dtr = 0;
  (void)dg_ko; (void)dg_ts; (void)dg_tr; (void)dg_dko; (void)dg_dts; (void)dg_dtr;
  // two strides: from pass j to pass j + 1 (BPASS k-rows) and from the last pass of a tile to the first of the next
  int dg_pko = 0, dg_pts = 0, dg_ptr = 0, wg_pw = 0, wg_ph = 0, wg_pn = 0;
  (void)dg_pko; (void)dg_pts; (void)dg_ptr; (void)wg_pw; (void)wg_ph; (void)wg_pn;
  if constexpr (MODE == MODE_DGRAD && B_MC) {
    const int kk = kt_begin * BK + b_kr;
    const int tapi = kk / p.Ko;
    dg_ko = kk - tapi * p.Ko;
    dg_tr = tapi / p.TS;
    dg_ts = tapi - dg_tr * p.TS;
    const int adv = BK - (BJ - 1) * BPASS;
    const int q1 = adv / p.Ko;
    dg_dko = adv - q1 * p.Ko;
    dg_dtr = q1 / p.TS;
    dg_dts = q1 - dg_dtr * p.TS;
    const int q2 = BPASS / p.Ko;
    dg_pko = BPASS - q2 * p.Ko;
    dg_ptr = q2 / p.TS;
    dg_pts = q2 - dg_ptr * p.TS;
  }
  if constexpr (MODE == MODE_WGRAD) {
    const int kk = kt_begin * BK + b_kr;
    const int hw = p.Ho * p.Wo;
    wg_n = kk / hw;
    const int rem = kk - wg_n * hw;
    wg_ho = rem / p.Wo;
    wg_wo = rem - wg_ho * p.Wo;
    const int adv = BK - (BJ - 1) * BPASS;  // from the last pass of a tile to the first pass of the next
    const int q1 = adv / p.Wo;
    wg_dw = adv - q1 * p.Wo;
    wg_dn = q1 / p.Ho;
    wg_dh = q1 - wg_dn * p.Ho;
    const int q2 = BPASS / p.Wo;
    wg_pw = BPASS - q2 * p.Wo;
    wg_pn = q2 / p.Ho;
    wg_ph = q2 - wg_pn * p.Ho;
  }

  // Channel counts that are a multiple of BK (every layer but the image-facing ones) put a whole K tile inside ONE
  // filter tap: the (tap row, tap column, first channel) of the tile is the same for every thread, and since the loaders
  // are called for consecutive tiles it is carried from call to call in block-uniform scalars instead of being
  // re-derived per thread by two reciprocal divisions (the per-thread part shrinks to `channel = c0 + 4 * quad`).
  const bool uni_k = (MODE == MODE_FPROP) ? (p.C % BK == 0) : (MODE == MODE_DGRAD ? (p.Ko % BK == 0) : false);
  int u_r = 0, u_s = 0, u_c0 = 0;        // FPROP: tap (r, s), first channel;  DGRAD: class tap (tr, ts), first ko
  int u_cur_r = 0, u_cur_s = 0, u_cur_c0 = 0;  // decode of the tile the last load_a call was issued for (load_b reuses it)
  if constexpr (MODE == MODE_FPROP || MODE == MODE_DGRAD) {
    if (uni_k) {
      const int inner = (MODE == MODE_FPROP) ? p.C : p.Ko, width = (MODE == MODE_FPROP) ? p.S : p.TS;
      const int kk0 = kt_begin * BK;
      const int tap = kk0 / inner;
      u_c0 = kk0 - tap * inner;
      u_r = tap / width;
      u_s = tap - u_r * width;
    }
  }
  auto advance_uni = [&]() {
    const int inner = (MODE == MODE_FPROP) ? p.C : p.Ko, width = (MODE == MODE_FPROP) ? p.S : p.TS;
    u_cur_r = u_r; u_cur_s = u_s; u_cur_c0 = u_c0;
    u_c0 += BK;
    if (u_c0 >= inner) {
      u_c0 -= inner;
      u_s += 1;
      if (u_s == width) { u_s = 0; u_r += 1; }
    }
  };
  (void)u_cur_r; (void)u_cur_s; (void)u_cur_c0;

  // Issue the (branch-free) global loads of K tile `kt` into ra/rb.  Tiles at or beyond kt_end read as zeros
  // without touching memory (every lane goes out of range), which lets the main loop run without tail branches.
  auto load_a = [&](int kt, int fill_st) {
    const int k0 = kt * BK;
    const int Klim = kt < kt_end ? p.K : 0;
    if constexpr (!A_MC) {
      const unsigned kk = (unsigned)(k0 + kq * 4);
      const bool kvalid = (int)kk < Klim;
      if constexpr (MODE == MODE_FPROP) {
        unsigned tap, c, r, s;
        if (uni_k) {
          advance_uni();
          r = (unsigned)u_cur_r; s = (unsigned)u_cur_s; c = (unsigned)(u_cur_c0 + kq * 4);
        } else {
          so_divmod(kk, (unsigned)p.C, invC, tap, c);
          so_divmod(tap, (unsigned)p.S, invS, r, s);
        }
        const int tap_off = ((int)r * p.W + (int)s) * p.lda + (int)c;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const int hi = a_h0[j] + (int)r, wi = a_w0[j] + (int)s;
          const bool ok = kvalid & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
          SO_EMIT_A(j, (((unsigned)(a_org[j] + tap_off) * 4u) | (ok ? 0u : SO_OOB)));
        }
      } else if constexpr (MODE == MODE_DGRAD) {
        unsigned tapi, ko, tr, ts;
        if (uni_k) {
          advance_uni();
          tr = (unsigned)u_cur_r; ts = (unsigned)u_cur_s; ko = (unsigned)(u_cur_c0 + kq * 4);
        } else {
          so_divmod(kk, (unsigned)p.Ko, invKo, tapi, ko);
          so_divmod(tapi, (unsigned)p.TS, invTS, tr, ts);
        }
        const int tap_off = -((int)tr * p.Wo + (int)ts) * p.lda + (int)ko;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const int ho = a_h0[j] - (int)tr, wo = a_w0[j] - (int)ts;
          const bool ok = kvalid & ((unsigned)ho < (unsigned)p.Ho) & ((unsigned)wo < (unsigned)p.Wo);
          SO_EMIT_A(j, (((unsigned)(a_org[j] + tap_off) * 4u) | (ok ? 0u : SO_OOB)));
        }
      } else {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const bool ok = kvalid & (a_h0[j] == 0);
          SO_EMIT_A(j, (((unsigned)(a_org[j] + (int)kk) * 4u) | (ok ? 0u : SO_OOB)));
        }
      }
    } else {
      const int col = m0 + a_mq * 4;
      const bool colok = col < p.M;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int kk = k0 + a_kr + APASS * j;
        const bool ok = colok & (kk < Klim);
        SO_EMIT_A(j, (((unsigned)(kk * p.lda + col) * 4u) | (ok ? 0u : SO_OOB)));
      }
    }
  };
  auto load_b = [&](int kt, int fill_st) {
    const int k0 = kt * BK;
    const int Klim = kt < kt_end ? p.K : 0;
    if constexpr (!B_MC) {
      const int kk = k0 + kq * 4;
      const bool kvalid = kk < Klim;
      int koff = kk;
      if constexpr (MODE == MODE_DGRAD) {
        // transposed weights wt[c][r][s][ko]: k index (class tap, ko) -> ((r0 + st*tr) * S + s0 + st*ts) * Ko + ko
        unsigned tapi, ko, tr, ts;
        if (uni_k) {  // same tile as the load_a call just before: its decode is reused
          tr = (unsigned)u_cur_r; ts = (unsigned)u_cur_s; ko = (unsigned)(u_cur_c0 + kq * 4);
        } else {
          so_divmod((unsigned)kk, (unsigned)p.Ko, invKo, tapi, ko);
          so_divmod(tapi, (unsigned)p.TS, invTS, tr, ts);
        }
        koff = ((d_r0 + p.stride * (int)tr) * p.S + d_s0 + p.stride * (int)ts) * p.Ko + (int)ko;
      }
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const bool ok = kvalid & (b_row[j] >= 0);
        SO_EMIT_B(j, (((unsigned)(b_row[j] + koff) * 4u) | (ok ? 0u : SO_OOB)));
      }
    } else {
      const int col = n0 + b_mq * 4;
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const int kk = k0 + b_kr + BPASS * j;
        if constexpr (MODE == MODE_DGRAD) {
          if (uni_k) {
            // Ko % 32 == 0: the whole K tile sits in one class tap (decoded once per tile by load_a, block-uniform);
            // only the output channel ko = first ko of the tile + this thread's k row differs between threads
            const int r = d_r0 + p.stride * u_cur_r, s = d_s0 + p.stride * u_cur_s;
            const int ko = u_cur_c0 + b_kr + BPASS * j;
            const bool ok = (kk < Klim) & (col < p.N);
            SO_EMIT_B(j, (((unsigned)((ko * (p.R * p.S) + (r * p.S + s)) * p.ldb + col) * 4u) | (ok ? 0u : SO_OOB)));
            continue;
          }
          // in-place OHWI weights: k row kk = (class tap (tr, ts), ko), carried incrementally like the WGRAD pixel
          const int r = d_r0 + p.stride * dg_tr, s = d_s0 + p.stride * dg_ts;
          const bool ok = (kk < Klim) & (col < p.N);
          SO_EMIT_B(j, (((unsigned)((dg_ko * (p.R * p.S) + (r * p.S + s)) * p.ldb + col) * 4u) | (ok ? 0u : SO_OOB)));
          {   // next pass (BPASS k-rows further) / first pass of the next tile
            const int sko = j + 1 < BJ ? dg_pko : dg_dko, sts = j + 1 < BJ ? dg_pts : dg_dts, str_ = j + 1 < BJ ? dg_ptr : dg_dtr;
            dg_ko += sko;
            const bool c1 = dg_ko >= p.Ko;
            dg_ko -= c1 ? p.Ko : 0;
            dg_ts += sts + (c1 ? 1 : 0);
            const bool c2 = dg_ts >= p.TS;
            dg_ts -= c2 ? p.TS : 0;
            dg_tr += str_ + (c2 ? 1 : 0);
          }
        } else if constexpr (MODE == MODE_WGRAD) {
          // (wg_n, wg_ho, wg_wo) = output pixel of k row kk, carried from quad to quad and from tile to tile
          const int hi = wg_ho * p.stride - p.pad + w_r;
          const int wi = wg_wo * p.stride - p.pad + w_s;
          const bool ok = (kk < Klim) & w_colvalid & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
          SO_EMIT_B(j, (((unsigned)(((wg_n * p.H + hi) * p.W + wi) * p.ldb + w_c) * 4u) | (ok ? 0u : SO_OOB)));
          {   // next pass (BPASS output pixels further) / first pass of the next tile
            const int sw = j + 1 < BJ ? wg_pw : wg_dw, sh = j + 1 < BJ ? wg_ph : wg_dh, sn = j + 1 < BJ ? wg_pn : wg_dn;
            wg_wo += sw;
            const bool cw = wg_wo >= p.Wo;
            wg_wo -= cw ? p.Wo : 0;
            wg_ho += sh + (cw ? 1 : 0);
            const bool ch = wg_ho >= p.Ho;
            wg_ho -= ch ? p.Ho : 0;
            wg_n += sn + (ch ? 1 : 0);
          }
        } else {
          const bool ok = (kk < Klim) & (col < p.N);
          SO_EMIT_B(j, (((unsigned)(kk * p.ldb + col) * 4u) | (ok ? 0u : SO_OOB)));
        }
      }
    }
  };

  // A wave that owns a single 32x32 output tile would issue every MFMA on the SAME accumulator; an instruction
  // slotted between two such dependent MFMAs costs a ~43-cycle bubble (MI355X_MICROARCH.md).  Such waves split the
  // K sum over two accumulators (even / odd k-pairs), added once in the epilogue.
  // (Round 1 measured no gain and left it off.  Round 3 PMC passes - profiles/r03_e_*_pmc.csv - show the 64x64 / 4-wave
  //  instantiations at 42 % MFMA-pipe utilisation with the waves issue-stalled 59 % of the time, the 128x128 ones at 75 %:
  //  re-measured with the split on (SO_KSPLIT=2): 57.4 vs 56.9 us on the 512x4608x768 weight gradient, 550.3 vs 550.6
  //  frames/s for the step - still no gain, so the dependent chain is not the limiter; what the counters show instead is
  //  tile quantisation (576 tiles on 256 CUs: the CUs holding 3 tiles run at 58 %, the average is 42 %).  Left off.)
#ifndef SO_KSPLIT
#define SO_KSPLIT 1
#endif
  constexpr int KS = (TM * TN == 1) ? SO_KSPLIT : 1;
  f32x16 acc[TM][TN], acc2[1];  // acc2 is dead code unless KS == 2
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[0][r] = 0.f;

  // Fragments of one 8-k chunk (one ds_read_b128 per 32-row tile per operand) and the 4 MFMA steps they feed.
  // Two fragment sets alternate so that the LDS reads of chunk c+1 are issued BEFORE the MFMAs of chunk c.
  f32x4 fa[2][TM], fb[2][TN];
  auto read_frag = [&](int st, int kc, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
    const float* as = As + st * A_STAGE;
    const float* bs = Bs + st * B_STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * WTM + i * 32 + li;
      if constexpr (A_MC) {   // [32 k][BM rows]: four ds_read_b32 at k, k+1, k+2, k+3 of this lane half
        const float* q = as + (kc * 8 + lh * 4) * BM + row;
        af[i][0] = q[0]; af[i][1] = q[BM]; af[i][2] = q[2 * BM]; af[i][3] = q[3 * BM];
      } else {
        af[i] = *reinterpret_cast<const f32x4*>(as + row * 32 + ((kc * 8) ^ (lh * 4) ^ (so_swz(row) << 2)));
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * WTN + j * 32 + li;
      if constexpr (B_MC) {
        const float* q = bs + (kc * 8 + lh * 4) * BN + row;
        bf[j][0] = q[0]; bf[j][1] = q[BN]; bf[j][2] = q[2 * BN]; bf[j][3] = q[3 * BN];
      } else {
        bf[j] = *reinterpret_cast<const f32x4*>(bs + row * 32 + ((kc * 8) ^ (lh * 4) ^ (so_swz(row) << 2)));
      }
    }
  };
  auto mma = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
    if constexpr (KS == 2) {
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][t], bf[0][t], acc[0][0], 0, 0, 0);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][t + 1], bf[0][t + 1], acc2[0], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
    }
  };

#ifndef SO_ABLATE
#define SO_ABLATE 0
#endif
  // SO_ABLATE (tools/ablate_igemm.sh, measurement builds only; results are WRONG for any non-zero value): drop one ingredient
  // of the K loop to see what the loop is bound by - 1: MFMAs, 4: the LDS-DMA fills, 8: the barrier, 16: LDS reads.
#if SO_ABLATE & 16
#define SO_READ_FRAG(...)
#else
#define SO_READ_FRAG(...) read_frag(__VA_ARGS__)
#endif
#if SO_ABLATE & 1
#define SO_MMA(a, b) do { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) \
    acc[i_][j_][0] += a[i_][0] * b[j_][0] + a[i_][1] * b[j_][1] + a[i_][2] * b[j_][2] + a[i_][3] * b[j_][3]; } while (0)
#else
#define SO_MMA(a, b) mma(a, b)
#endif
#if SO_ABLATE & 4
#define SO_FILL(kt, st)
#else
#define SO_FILL(kt, st) do { load_a(kt, st); load_b(kt, st); } while (0)
#endif
#if SO_ABLATE & 8
#define SO_SYNC()
#else
#define SO_SYNC() do { SO_DMA_DRAIN(); __syncthreads(); } while (0)
#endif

  // K loop.  Two LDS stages; the fill of tile t+1 is issued at the top of tile t into the stage all waves left at the barrier
  // that ended tile t-1, and is drained by SO_DMA_DRAIN (vmcnt(0), common.h) in front of the barrier that ends tile t (an LDS-DMA is a
  // pending LDS write on the VM counter): one whole tile - 16+ MFMAs per wave, times the blocks sharing the CU - to land.
  // sched_barrier(0) around that barrier keeps the MFMAs (which touch no memory) from sinking below it; the loop is unrolled
  // by two so that both stages are addressed statically, with no exit from the middle of the body (a second exit makes the
  // accumulators live across a merge point and hipcc copies them).  The loaders see the tiles in order (their incremental
  // index state relies on it).  History of this loop: rounds 1-4 staged through registers (loads consumed 512 cycles after
  // issue; vmcnt(3..0) in front of every ds_write), round 5 first deepened that to two register stages, then removed the
  // registers altogether (profiles/r05_igemm_ablation.txt, r05_*_ab.txt).
  const bool to_ws = p.splitk > 1;
  // Wide path: each 32x32 accumulator tile is transposed through a wave-private LDS patch so that a lane owns
  // four consecutive columns of one row and issues 16-byte stores (8 rows x 128 B per instruction) instead of
  // sixteen 4-byte stores per tile.  Needs 16-byte aligned rows; otherwise the scalar path below is used.
  const bool wide = (p.N & 3) == 0 &&
                    (to_ws || ((p.ldc & 3) == 0 && (((uintptr_t)p.c) & 15) == 0 && (MODE != MODE_GEMM || (p.sc & 3) == 0) &&
                               (!p.res || ((p.ldres & 3) == 0 && (((uintptr_t)p.res) & 15) == 0 &&
                                           (MODE != MODE_GEMM || (p.sres & 3) == 0))) &&
                               (!p.gate || (((uintptr_t)p.gate) & 15) == 0)));

  // Weight gradients accumulate into the optimizer's gradient slab (res == c): the epilogue used to read the old values
  // right before it stores - a dependent HBM / L2 latency per tile at the very end of a kernel whose K loop is short
  // (K = the layer's pixel count).  With the staging registers gone there is room to fetch them up front.
  constexpr bool PRE_RES = MODE == MODE_WGRAD && TM * TN <= 2;
  f32x4 rres[PRE_RES ? TM : 1][PRE_RES ? TN : 1][4];
  (void)rres;
  bool pre_res = false;
  if constexpr (PRE_RES) {
    pre_res = wide && !to_ws && p.res != nullptr && (long long)p.M * p.ldres < (1ll << 29);
    if (pre_res) {
      const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = m0 + wm * WTM + i * 32 + (lane >> 3) + 8 * q;
            const int n = n0 + wn * WTN + j * 32 + (lane & 7) * 4;
            const bool ok = (m < p.M) & (n < p.N);
            rres[i][j][q] = so_bload(rR, (((unsigned)(m * p.ldres + n) * 4u) | (ok ? 0u : SO_OOB)));
          }
    }
  }

  load_a(kt_begin, 0);
  load_b(kt_begin, 0);
#if SO_ABLATE & 16
  read_frag(0, 0, fa[0], fb[0]);
  read_frag(0, 1, fa[1], fb[1]);
#endif
  SO_DMA_DRAIN();
  __syncthreads();

#define SO_KTILE(CUR, kt_next)                                      \
  do {                                                              \
    if ((kt_next) < kt_end) SO_FILL(kt_next, (CUR) ^ 1);            \
    __builtin_amdgcn_sched_barrier(0);                              \
    SO_READ_FRAG(CUR, 0, fa[0], fb[0]);                             \
    SO_READ_FRAG(CUR, 1, fa[1], fb[1]);                             \
    SO_SB();                                                        \
    SO_MMA(fa[0], fb[0]);                                           \
    SO_SB();                                                        \
    SO_READ_FRAG(CUR, 2, fa[0], fb[0]);                             \
    SO_SB();                                                        \
    SO_MMA(fa[1], fb[1]);                                           \
    SO_SB();                                                        \
    SO_READ_FRAG(CUR, 3, fa[1], fb[1]);                             \
    SO_SB();                                                        \
    SO_MMA(fa[0], fb[0]);                                           \
    SO_SB();                                                        \
    SO_MMA(fa[1], fb[1]);                                           \
    __builtin_amdgcn_sched_barrier(0);                              \
    SO_SYNC();                                                      \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)

  {
    int kt = kt_begin;
    for (; kt + 1 < kt_end; kt += 2) {
      SO_KTILE(0, kt + 1);
      SO_KTILE(1, kt + 2);
    }
    if (kt < kt_end) SO_KTILE(0, kt + 1);
  }

  // ---------------- epilogue ----------------
  if constexpr (KS == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[0][r];
  }
  if (wide) {
    float* stg = smem + wave * (32 * LDK);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDK + li] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (lane >> 3) + 8 * q, col4 = (lane & 7) * 4;
          f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * LDK + col4);
          const int m = m0 + wm * WTM + i * 32 + row;
          const int n = n0 + wn * WTN + j * 32 + col4;
          if (m < p.M && n < p.N) {
            if (to_ws) {
              *reinterpret_cast<f32x4*>(p.ws + ((long long)bz * p.M + m) * p.N + n) = v;
            } else {
              long long off, roff;
              so_row_offset<MODE>(p, cls, m, off, roff);
              if (off >= 0) {
                if (p.alpha) { const float al = p.alpha[0]; v[0] *= al; v[1] *= al; v[2] *= al; v[3] *= al; }
                if (p.bias) {
#pragma unroll
                  for (int k = 0; k < 4; ++k) if (n + k < p.nbias) v[k] += p.bias[n + k];
                }
                if (p.res) {
                  f32x4 rv;
                  if (PRE_RES && pre_res) rv = rres[PRE_RES ? i : 0][PRE_RES ? j : 0][q];
                  else rv = *reinterpret_cast<const f32x4*>(p.res + roff + n);
                  v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
                }
                if (p.act != SO_ACT_NONE) {
#pragma unroll
                  for (int k = 0; k < 4; ++k) v[k] = so_act_epi(p, v[k]);
                }
                if (p.gate) {
                  const f32x4 gv = *reinterpret_cast<const f32x4*>(p.gate + off + n);
#pragma unroll
                  for (int k = 0; k < 4; ++k) v[k] = gv[k] > 0.f ? v[k] : 0.f;
                }
                *reinterpret_cast<f32x4*>(p.c + off + n) = v;
              }
            }
          }
        }
        __syncthreads();
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = m0 + wm * WTM + i * 32 + row;
      if (m >= p.M) continue;
      if (to_ws) {
        float* dst = p.ws + ((long long)bz * p.M + m) * p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * WTN + j * 32 + li;
          if (n < p.N) dst[n] = acc[i][j][r];
        }
      } else {
        long long off, roff;
        so_row_offset<MODE>(p, cls, m, off, roff);
        if (off < 0) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * WTN + j * 32 + li;
          if (n < p.N) {
            const float e = so_epilogue(p, acc[i][j][r], roff, n);
            p.c[off + n] = (!p.gate || p.gate[off + n] > 0.f) ? e : 0.f;
          }
        }
      }
    }
  }
}

// Sums the split-K slabs ws[cls][split][M][N] in a fixed order (deterministic) and applies the epilogue.  VEC = 4:
// a thread owns four consecutive columns (16-byte loads / stores); eight slab loads are kept in flight per thread, the
// kernel is otherwise one dependent-latency chain per split.
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void so_splitk_reduce_kernel(const SoIgemm p) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const long long mn = (long long)p.M * p.N;
  const long long total = (long long)p.nclass * mn / VEC;
  const int nq = p.N / VEC;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int cls = (int)(idx / (mn / VEC));
    const long long remq = idx - (long long)cls * (mn / VEC);
    const int m = (int)(remq / nq);
    const int n = (int)(remq - (long long)m * nq) * VEC;
    const float* src = p.ws + (long long)cls * p.splitk * mn + (long long)m * p.N + n;
    vec_t v = so_sum_slabs<vec_t>(src, p.splitk, mn);
    long long off, roff;
    so_row_offset<MODE>(p, cls, m, off, roff);
    if (off < 0) continue;
    if constexpr (VEC == 4) {
      if (p.alpha) v *= p.alpha[0];
      if (p.bias) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (n + k < p.nbias) v[k] += p.bias[n + k];
      }
      if (p.res) v += *reinterpret_cast<const vec_t*>(p.res + roff + n);
      if (p.act != SO_ACT_NONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = so_act_epi(p, v[k]);
      }
      if (p.gate) {
        const vec_t gv = *reinterpret_cast<const vec_t*>(p.gate + off + n);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gv[k] > 0.f ? v[k] : 0.f;
      }
      *reinterpret_cast<vec_t*>(p.c + off + n) = v;
    } else {
      const float e = so_epilogue(p, v[0], roff, n);
      p.c[off + n] = (!p.gate || p.gate[off + n] > 0.f) ? e : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Host side: tile / split-K selection and launch.
// ------------------------------------------------------------------------------------------------
struct SoPlan {
  int bm, bn, splitk, ktps, nw;
};

static bool so_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Cost model: blocks are dealt round-robin to 256 CUs; a CU's time is (#blocks it owns) x (K tiles per block +
// prologue/epilogue) x (tile FLOPs / per-CU MFMA rate) x a per-shape factor calibrated with tools/igemm_bench.py.
// {BM, BN, waves}: tile index 0-3 = 4 waves, 4-5 = 8 waves (BN = 128 only)
constexpr int kNTiles = 6;
static const int kTiles[kNTiles][3] = {{64, 64, 4}, {128, 64, 4}, {64, 128, 4}, {128, 128, 4}, {128, 128, 8}, {64, 128, 8}};
static double g_tile_cost[kNTiles] = {1.0, 1.12, 1.15, 1.2, 1.1, 1.1};
static int g_force_bm = 0, g_force_bn = 0, g_force_splitk = 0, g_force_nw = 0;

static SoPlan so_plan(const SoIgemm& p, long long ws_floats) {
  const int nkt = so_cdiv(p.K, 32);
  SoPlan best = {64, 64, 1, nkt, 4};
  double best_cost = 1e30;
  for (int ti = 0; ti < kNTiles; ++ti) {
    const int bm = kTiles[ti][0], bn = kTiles[ti][1], nw = kTiles[ti][2];
    if (g_force_bm && (bm != g_force_bm || bn != g_force_bn || nw != (g_force_nw ? g_force_nw : 4))) continue;
    const long long tiles = (long long)so_cdiv(p.M, bm) * so_cdiv(p.N, bn) * p.nclass;
    for (int sk = 1; sk <= 512; sk *= 2) {
      if (g_force_splitk && sk != g_force_splitk) continue;
      if (sk > nkt) break;
      const int ktps = so_cdiv(nkt, sk);
      const int sk_eff = so_cdiv(nkt, ktps);
      if (sk_eff > 1 && (long long)sk_eff * p.nclass * p.M * p.N > ws_floats) continue;
      const long long blocks = tiles * sk_eff;
      const double t_kt = 2.0 * bm * bn * 32 / (0.6e12 * 0.8) * g_tile_cost[ti];
      const double waves = (double)((blocks + 255) / 256);
      double cost = waves * (ktps + 3) * t_kt;
      if (sk_eff > 1) cost += 3e-6 + (double)(sk_eff + 1) * p.nclass * p.M * p.N * 4.0 / 3e12;
      if (cost < best_cost) {
        best_cost = cost;
        best = {bm, bn, sk_eff, ktps, nw};
      }
    }
  }
  return best;
}

// ---- optional live timing of every MFMA launch with HIP events (bench.py's roofline figure) --------
// Events are recorded on the launch stream around the main kernel only (not the split-K reduce).
struct SoProfRec {
  hipEvent_t e0, e1;
  int key;  // MODE * 8 + tile index (0: 64x64, 1: 128x64, 2: 64x128, 3: 128x128, 4/5: 8-wave tiles, 6: thin.hip)
  double flops;
  double bytes;  // algorithmic HBM bytes of the launch: every operand read once + the result written once (0: not stated)
  int M, N, K, nclass, splitk;
};
static const char* g_prof_dump_path = nullptr;
static bool g_prof_on = false;
static std::vector<SoProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t so_prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;  // callers skip the record (the launch itself still runs)
  return e;
}

int so_prof_begin(int key, double flops, int M, int N, int K, hipStream_t stream) {
  if (!g_prof_on) return -1;
  SoProfRec rec;
  rec.e0 = so_prof_event();
  rec.e1 = so_prof_event();
  if (!rec.e0 || !rec.e1) return -1;  // no events available: this launch is simply not timed
  rec.key = key;
  rec.flops = flops;
  rec.bytes = 0.0;
  rec.M = M; rec.N = N; rec.K = K; rec.nclass = 1; rec.splitk = 1;
  (void)hipEventRecord(rec.e0, stream);
  g_prof.push_back(rec);
  return (int)g_prof.size() - 1;
}

void so_prof_bytes(int slot, double bytes) {
  if (slot >= 0 && slot < (int)g_prof.size()) g_prof[slot].bytes = bytes;
}

void so_prof_end(int slot, hipStream_t stream) {
  if (slot >= 0 && slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].e1, stream);
}

template <int MODE, bool A_MC, bool B_MC, int BM, int BN, int NW>
static int so_launch_tile(const SoIgemm& p_in, hipStream_t stream) {
  SoIgemm p = p_in;
  constexpr int A_STAGE = BM * 32;   // floats per LDS stage, either layout ([rows][32 k] or [32 k][rows])
  constexpr int B_STAGE = BN * 32;
  constexpr size_t lds = (size_t)(2 * (A_STAGE + B_STAGE)) * sizeof(float);
  auto kern = so_igemm_kernel<MODE, A_MC, B_MC, BM, BN, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long long tiles = (long long)so_cdiv(p.M, BM) * so_cdiv(p.N, BN);
  dim3 grid((unsigned)tiles, 1, (unsigned)(p.nclass * p.splitk));
  // split-K: 16-byte epilogue possible?  (the slabs themselves are always written wide when N % 4 == 0)
  const bool wide = (p.N & 3) == 0 && (p.ldc & 3) == 0 && so_aligned16(p.c) && so_aligned16(p.ws) &&
                    (MODE != MODE_GEMM || (p.sc & 3) == 0) &&
                    (!p.res || ((p.ldres & 3) == 0 && so_aligned16(p.res) && (MODE != MODE_GEMM || (p.sres & 3) == 0))) &&
                    (!p.gate || so_aligned16(p.gate));
  SoProfRec rec;
  bool timed = g_prof_on;
  if (timed) {
    rec.e0 = so_prof_event();
    rec.e1 = so_prof_event();
    timed = rec.e0 && rec.e1;
  }
  if (timed) {
    rec.key = MODE * 8 + (NW == 8 ? (BM == 128 ? 4 : 5) : (BM == 128 ? 1 : 0) + (BN == 128 ? 2 : 0));
    // batched GEMMs with >= 16 matrices are the Winograd-domain GEMMs of csrc/wino.hip (16 or 36 transform points): their own
    // key group (4), so that they are not averaged with the ~10 us attention GEMMs that share the 64x64 instantiation
    if (MODE == MODE_GEMM && p.nclass >= 16) rec.key += 8;
    rec.flops = 2.0 * p.M * p.N * (double)p.K * p.nclass;
    // convolution modes (fprop / dgrad / wgrad alike): the input-side tensor, the output-side tensor and the filter, each
    // touched once; batched GEMM: A, B and C of every matrix
    rec.bytes = MODE == MODE_GEMM
                    ? 4.0 * p.nclass * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N)
                    : 4.0 * ((double)p.Nb * p.H * p.W * p.C + (double)p.Nb * p.Ho * p.Wo * p.Ko + (double)p.Ko * p.R * p.S * p.C);
    rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.nclass = p.nclass; rec.splitk = p.splitk;
    (void)hipEventRecord(rec.e0, stream);
  }
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, p);
  if (timed) {
    (void)hipEventRecord(rec.e1, stream);
    g_prof.push_back(rec);
  }
  int err = SO_LAUNCH_CHECK();
  if (err) return err;
  if (p.splitk > 1) {
    const long long total = (long long)p.nclass * p.M * p.N;
    int blocks = so_cdiv(wide ? total / 4 : total, 256);
    if (blocks > 4096) blocks = 4096;
    if (wide)
      hipLaunchKernelGGL((so_splitk_reduce_kernel<MODE, 4>), dim3(blocks), dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL((so_splitk_reduce_kernel<MODE, 1>), dim3(blocks), dim3(256), 0, stream, p);
    err = SO_LAUNCH_CHECK();
  }
  return err;
}

template <int MODE, bool A_MC, bool B_MC>
static int so_launch_plan(SoIgemm& p, const SoPlan& plan, hipStream_t stream) {
  p.splitk = plan.splitk;
  p.ktps = plan.ktps;
  if (plan.nw == 8 && plan.bn == 128) {
    if (plan.bm == 128) return so_launch_tile<MODE, A_MC, B_MC, 128, 128, 8>(p, stream);
    return so_launch_tile<MODE, A_MC, B_MC, 64, 128, 8>(p, stream);
  }
  if (plan.bm == 128 && plan.bn == 128) return so_launch_tile<MODE, A_MC, B_MC, 128, 128, 4>(p, stream);
  if (plan.bm == 128) return so_launch_tile<MODE, A_MC, B_MC, 128, 64, 4>(p, stream);
  if (plan.bn == 128) return so_launch_tile<MODE, A_MC, B_MC, 64, 128, 4>(p, stream);
  return so_launch_tile<MODE, A_MC, B_MC, 64, 64, 4>(p, stream);
}

// ---- measured plans -------------------------------------------------------------------------------------
// The layer shapes of a training run repeat every step, so the first time a problem shape is seen (outside a
// stream capture) every (tile, split-K) candidate is timed once with HIP events on the real operands and the
// fastest is cached; afterwards the lookup is a map access.  Tuning launches write to scratch behind the split-K
// slabs, never to the caller's output (which may be an accumulating gradient slab).
#include <map>
#include <array>
static int g_autotune = 0;
static std::map<std::array<int, 12>, SoPlan> g_plan_cache;

template <int MODE, bool A_MC, bool B_MC>
static int so_launch_inner(SoIgemm& p, long long ws_bytes, hipStream_t stream);

// Activations other than none / ReLU / LeakyReLU run as a second pass over the output (see so_act_epi).
template <int MODE, bool A_MC, bool B_MC>
static int so_launch(SoIgemm& p, long long ws_bytes, hipStream_t stream) {
  const int act = p.act;
  const bool deferred = act != SO_ACT_NONE && act != SO_ACT_RELU && act != SO_ACT_LEAKY;
  if (deferred) {
    if (p.gate || (MODE != MODE_FPROP && MODE != MODE_GEMM)) return SO_ERR_SHAPE;
    p.act = SO_ACT_NONE;
  }
  const int err = so_launch_inner<MODE, A_MC, B_MC>(p, ws_bytes, stream);
  if (err || !deferred || p.M <= 0 || p.N <= 0) return err;
  if (MODE == MODE_GEMM && p.nclass > 1 && p.sc != (long long)p.M * p.ldc) {
    for (int b = 0; b < p.nclass; ++b) {
      const int e = so_act_fwd(p.c + b * p.sc, p.ldc, p.c + b * p.sc, p.ldc, p.M, p.N, act, p.act_param, (void*)stream);
      if (e) return e;
    }
    return 0;
  }
  const long long rows = (long long)p.M * (MODE == MODE_GEMM ? p.nclass : 1);
  return so_act_fwd(p.c, p.ldc, p.c, p.ldc, rows, p.N, act, p.act_param, (void*)stream);
}

template <int MODE, bool A_MC, bool B_MC>
static int so_launch_inner(SoIgemm& p, long long ws_bytes, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0) return 0;
  const long long ws_floats = p.ws ? ws_bytes / 4 : 0;
  if (g_force_bm || g_force_splitk || !g_autotune) return so_launch_plan<MODE, A_MC, B_MC>(p, so_plan(p, ws_floats), stream);
  const std::array<int, 12> key = {MODE, (int)A_MC * 2 + (int)B_MC, p.M, p.N, p.K, p.nclass, p.R, p.S, p.stride, p.C, p.Ko,
                                   p.lda * 31 + p.ldb};
  auto it = g_plan_cache.find(key);
  const long long mn = (long long)p.nclass * p.M * p.N;
  if (it != g_plan_cache.end()) {
    // a cached / loaded plan was measured against some workspace: re-check its slab requirement against the
    // workspace actually handed in (a plans file from another configuration must never overrun the slab)
    if (it->second.splitk <= 1 || (long long)it->second.splitk * mn <= ws_floats)
      return so_launch_plan<MODE, A_MC, B_MC>(p, it->second, stream);
    return so_launch_plan<MODE, A_MC, B_MC>(p, so_plan(p, ws_floats), stream);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  if (cap != hipStreamCaptureStatusNone || ws_floats < 2 * mn)
    return so_launch_plan<MODE, A_MC, B_MC>(p, so_plan(p, ws_floats), stream);

  const int nkt = so_cdiv(p.K, 32);
  SoPlan best = so_plan(p, ws_floats);
  float best_ms = 1e30f;
  const bool was_prof = g_prof_on;
  g_prof_on = false;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    g_prof_on = was_prof;
    return so_launch_plan<MODE, A_MC, B_MC>(p, best, stream);  // cannot measure: cost-model plan, not cached
  }
  SoIgemm q = p;
  q.res = nullptr;
  q.ldres = 0;
  int err = 0;
  const int timed_reps = g_autotune >= 2 ? 6 : 2;
  if (g_autotune >= 2) {
    // Thorough mode (tools/make_plans.py, which writes the committed plans file): the chip clocks up to its power budget
    // only after tens of milliseconds of load, and a candidate timed on a cold chip loses to whatever is timed after
    // it.  Keep the matrix pipes busy with the cost-model plan for ~40 ms before the first measurement.
    SoPlan warm = best;
    const long long mnw = (long long)(warm.splitk > 1 ? warm.splitk : 0) * mn;
    if (mnw + mn <= ws_floats) {
      q.c = p.ws + mnw;
      if constexpr (MODE == MODE_GEMM) q.sc = (long long)p.M * p.N;
      q.ldc = (MODE == MODE_DGRAD && p.nclass > 1) ? p.ldc : p.N;
      if (MODE == MODE_DGRAD && p.nclass > 1) q.c = p.c;
      const double flops = 2.0 * p.M * p.N * (double)p.K * p.nclass;
      int n_warm = (int)(40e-3 / (flops / 60e12 + 8e-6));
      if (n_warm > 4000) n_warm = 4000;
      for (int r = 0; r < n_warm && !err; ++r) err = so_launch_plan<MODE, A_MC, B_MC>(q, warm, stream);
    }
  }
  for (int ti = 0; ti < kNTiles && !err; ++ti) {
    int last_ktps = -1;
    // split-K candidates: not only powers of two, so that tiles x splits can land near a multiple of the
    // 1024 block slots (256 CUs x 4 resident blocks) of the chip
    static const int kSplits[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 112, 128,
                                  160, 192, 224, 256};
    for (int si = 0; si < (int)(sizeof(kSplits) / sizeof(kSplits[0])) && !err; ++si) {
      const int sk = kSplits[si];
      if (sk > nkt) break;
      const int ktps = so_cdiv(nkt, sk);
      const int sk_eff = so_cdiv(nkt, ktps);
      if (ktps == last_ktps) continue;                          // same plan as the previous power of two
      last_ktps = ktps;
      if ((long long)(sk_eff > 1 ? sk_eff : 0) * mn + mn > ws_floats) continue;
      const long long blocks = (long long)so_cdiv(p.M, kTiles[ti][0]) * so_cdiv(p.N, kTiles[ti][1]) * p.nclass * sk_eff;
      if (blocks > 65536 || (sk_eff > 1 && blocks > 8192)) continue;
      SoPlan cand = {kTiles[ti][0], kTiles[ti][1], sk_eff, ktps, kTiles[ti][2]};
      q.c = p.ws + (long long)(sk_eff > 1 ? sk_eff : 0) * mn;   // scratch output behind the slabs
      if constexpr (MODE == MODE_GEMM) q.sc = (long long)p.M * p.N;
      q.ldc = (MODE == MODE_DGRAD && p.nclass > 1) ? p.ldc : p.N;
      if (MODE == MODE_DGRAD && p.nclass > 1) q.c = p.c;        // class-scattered rows: idempotent overwrite of dx
      float ms = 0.f;
      for (int rep = 0; rep < 1 + timed_reps && !err; ++rep) {  // one warm-up, then the fastest of the timed launches
        float t = 0.f;
        (void)hipEventRecord(e0, stream);
        err = so_launch_plan<MODE, A_MC, B_MC>(q, cand, stream);
        (void)hipEventRecord(e1, stream);
        if (!err) {
          const hipError_t se = hipEventSynchronize(e1);
          if (se != hipSuccess) err = (int)se;  // a failed tuning launch is an error, not a silent cost-model fallback
          else (void)hipEventElapsedTime(&t, e0, e1);
        }
        if (rep > 0 && t > 0.f && (ms == 0.f || t < ms)) ms = t;
      }
      if (!err && ms > 0.f && ms < best_ms) {
        best_ms = ms;
        best = cand;
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  g_prof_on = was_prof;
  if (err) return err;
  g_plan_cache[key] = best;
  return so_launch_plan<MODE, A_MC, B_MC>(p, best, stream);
}

// a forced tile / split-K (tests, tools) always means the general engine
static bool so_forced() { return g_force_bm || g_force_splitk || g_force_nw; }

// extent in bytes of a [rows][ld] fp32 operand; 0 -> too large for the 31-bit offsets used by the loaders
static unsigned so_extent(long long rows, long long ld) {
  const long long b = rows * ld * 4;
  return (b > 0 && b < 0x7FFFFFF0LL) ? (unsigned)b : 0u;
}

extern "C" {

void so_igemm_force(int bm, int bn, int splitk) {
  g_force_bm = bm;
  g_force_bn = bn;
  g_force_splitk = splitk;
  g_force_nw = 0;
}

void so_igemm_force_waves(int nw) { g_force_nw = nw; }

void so_igemm_tile_cost(float c64x64, float c128x64, float c64x128, float c128x128) {
  g_tile_cost[0] = c64x64; g_tile_cost[1] = c128x64; g_tile_cost[2] = c64x128; g_tile_cost[3] = c128x128;
}

void so_igemm_autotune(int on) { g_autotune = on; }
int so_igemm_plan_count(void) { return (int)g_plan_cache.size(); }

// Plan cache persistence (one text line per shape: 12 key ints, then bm bn splitk ktps).  Returns the number
// of plans written / read, or -1 if the file cannot be opened.
int so_igemm_plans_save(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  for (const auto& kv : g_plan_cache) {
    for (int v : kv.first) fprintf(f, "%d ", v);
    fprintf(f, "%d %d %d %d %d\n", kv.second.bm, kv.second.bn, kv.second.splitk, kv.second.ktps, kv.second.nw);
  }
  fclose(f);
  return (int)g_plan_cache.size();
}

int so_igemm_plans_load(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int n = 0;
  for (;;) {
    std::array<int, 12> key;
    SoPlan pl;
    bool ok = true;
    for (int i = 0; i < 12 && ok; ++i) ok = fscanf(f, "%d", &key[i]) == 1;
    ok = ok && fscanf(f, "%d %d %d %d %d", &pl.bm, &pl.bn, &pl.splitk, &pl.ktps, &pl.nw) == 5;
    if (!ok) break;
    if ((pl.bm == 64 || pl.bm == 128) && (pl.bn == 64 || pl.bn == 128) && pl.splitk >= 1 && pl.ktps >= 1 &&
        (pl.nw == 4 || (pl.nw == 8 && pl.bn == 128))) {
      g_plan_cache[key] = pl;
      ++n;
    }
  }
  fclose(f);
  return n;
}

void so_prof_enable(int on) { g_prof_on = on != 0; }

// Waits for every recorded launch, then fills per-key totals (key = group*8 + tile index, 40 keys; groups: fprop, dgrad,
// wgrad, gemm, Winograd-domain gemm):
// out_ms[k] = summed kernel time in ms, out_flops[k] = summed algorithmic FLOPs, out_count[k] = launches.
// Clears the record list.  Returns the number of launches collected.
int so_prof_collect_bytes(float* out_ms, float* out_flops, int* out_count, double* out_bytes);
int so_prof_collect(float* out_ms, float* out_flops, int* out_count) {
  return so_prof_collect_bytes(out_ms, out_flops, out_count, nullptr);
}

// + out_bytes[k] = summed algorithmic HBM bytes (operands once + result once) of the launches under key k (may be null)
int so_prof_collect_bytes(float* out_ms, float* out_flops, int* out_count, double* out_bytes) {
  for (int k = 0; k < 40; ++k) { out_ms[k] = 0.f; out_flops[k] = 0.f; out_count[k] = 0; if (out_bytes) out_bytes[k] = 0.0; }
  int n = 0;
  FILE* dump = nullptr;
  if (const char* path = getenv("SO_PROF_DUMP")) dump = fopen(path, "w");
  if (dump) fprintf(dump, "key,M,N,K,nclass,splitk,us,tflops\n");
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      if (dump)
        fprintf(dump, "%d,%d,%d,%d,%d,%d,%.2f,%.2f\n", r.key, r.M, r.N, r.K, r.nclass, r.splitk, ms * 1e3,
                ms > 0 ? r.flops / (ms * 1e-3) / 1e12 : 0.0);
      out_ms[r.key] += ms;
      out_flops[r.key] += (float)r.flops;
      if (out_bytes) out_bytes[r.key] += r.bytes;
      out_count[r.key] += 1;
      ++n;
    }
    g_prof_pool.push_back(r.e0);
    g_prof_pool.push_back(r.e1);
  }
  if (dump) fclose(dump);
  g_prof.clear();
  return n;
}

// Ko output columns are computed and written; only the first Kw of them have weight rows / bias entries, the
// remaining columns come out as act(0) (zero channel padding of the output for channel counts that are not a
// multiple of 4: the weight descriptor simply ends after Kw rows and the hardware returns zeros beyond it).
int so_conv2d_fprop_padded(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                           int Nb, int H, int W, int C, int Ko, int Kw, int R, int S, int stride, int pad,
                           int act, float act_param, float* ws, long long ws_bytes, void* stream) {
  if ((C & 3) || (ldx & 3) || !so_aligned16(x) || !so_aligned16(w)) return SO_ERR_ALIGN;
  if (Kw > Ko || Kw <= 0) return SO_ERR_SHAPE;
  if (Ko == 4 && stride == 1 && !so_forced()) {  // four output channels: 4x4x1 MFMA kernel (thin.hip)
    const int r = so_thin_conv(0, x, ldx, w, Kw, bias, Kw, y, ldy, Nb, H + 2 * pad - R + 1, W + 2 * pad - S + 1, H, W, C, R,
                               S, pad, act, act_param, (hipStream_t)stream);
    if (r != 1) return r;
  }
  if (C == 4 && ldx == 4 && R == 3 && S == 3 && stride == 1 && pad == 1 && Kw == Ko && !so_forced()) {  // K = 36: thin.hip
    const int r = so_thin_expand(0, 0, x, w, bias, y, ldy, Nb, H, W, Ko, act, act_param, (hipStream_t)stream);
    if (r != 1) return r;
  }
  SoIgemm p = {};
  p.a = x; p.b = w; p.c = y; p.ws = ws; p.bias = bias; p.nbias = Kw;
  p.Nb = Nb; p.H = H; p.W = W; p.C = C;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  p.Ko = Ko; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.M = Nb * p.Ho * p.Wo; p.N = Ko; p.K = R * S * C;
  p.lda = ldx; p.ldb = p.K; p.ldc = ldy; p.ldres = 0;
  p.a_bytes = so_extent((long long)Nb * H * W, ldx);
  p.b_bytes = so_extent(Kw, p.K);
  if (!p.a_bytes || !p.b_bytes || p.K >= (1 << 24) || p.M >= (1 << 24)) return SO_ERR_SHAPE;
  p.act = act; p.act_param = act_param; p.nclass = 1;
  return so_launch<MODE_FPROP, false, false>(p, ws_bytes, (hipStream_t)stream);
}

int so_conv2d_fprop(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                    int Nb, int H, int W, int C, int Ko, int R, int S, int stride, int pad, int act,
                    float act_param, float* ws, long long ws_bytes, void* stream) {
  return so_conv2d_fprop_padded(x, ldx, w, bias, y, ldy, Nb, H, W, C, Ko, Ko, R, S, stride, pad, act, act_param, ws,
                                ws_bytes, stream);
}

// Input gradient reading the OHWI weights in place (KC x MC mode: no transposed copy needed).
int so_conv2d_dgrad(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nb, int H,
                    int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                    long long ws_bytes, void* stream) {
  if ((Ko & 3) || (lddy & 3) || (C & 3) || !so_aligned16(dy) || !so_aligned16(w)) return SO_ERR_ALIGN;
  if ((R % stride) || (S % stride)) return SO_ERR_SHAPE;
  if (Ko == 4 && lddy == 4 && R == 3 && S == 3 && stride == 1 && pad == 1 && !so_forced()) {  // four dy channels, K = 36: thin.hip
    const int r = so_thin_expand(1, 1, dy, w, nullptr, dx, lddx, Nb, H, W, C, SO_ACT_NONE, 0.f, (hipStream_t)stream);
    if (r != 1) return r;
  }
  SoIgemm p = {};
  p.a = dy; p.b = w; p.c = dx; p.ws = ws;
  p.Nb = Nb; p.H = H; p.W = W; p.C = C;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  p.Ko = Ko; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.TS = S / stride; p.H2 = (H + stride - 1) / stride; p.W2 = (W + stride - 1) / stride;
  p.nclass = stride * stride;
  p.M = Nb * p.H2 * p.W2; p.N = C; p.K = (R / stride) * (S / stride) * Ko;
  p.lda = lddy; p.ldb = C; p.ldc = lddx; p.ldres = 0;
  p.a_bytes = so_extent((long long)Nb * p.Ho * p.Wo, lddy);
  p.b_bytes = so_extent((long long)Ko * R * S, C);
  if (!p.a_bytes || !p.b_bytes || p.K >= (1 << 24) || p.M >= (1 << 24)) return SO_ERR_SHAPE;
  p.act = SO_ACT_NONE;
  return so_launch<MODE_DGRAD, false, true>(p, ws_bytes, (hipStream_t)stream);
}

// Input gradient with TRANSPOSED weights wt[c][r][s][ko] (so_ohwi_to_ihwo): both operands are k-contiguous, the
// kernel runs in the same KC x KC mode as the forward convolution (ds_read_b128 fragments on both sides).
int so_conv2d_dgrad_t_gated(const float* dy, int lddy, const float* wt, float* dx, int lddx, const float* gate,
                            int Nb, int H, int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                            long long ws_bytes, void* stream) {
  if ((Ko & 3) || (lddy & 3) || !so_aligned16(dy) || !so_aligned16(wt)) return SO_ERR_ALIGN;
  if ((R % stride) || (S % stride)) return SO_ERR_SHAPE;
  if (C == 4 && stride == 1 && !so_forced() && !gate) {  // four input channels (RGB + pad): thin.hip
    const int r = so_thin_conv(1, dy, lddy, wt, 4, nullptr, 0, dx, lddx, Nb, H, W, H + 2 * pad - R + 1, W + 2 * pad - S + 1,
                               Ko, R, S, pad, SO_ACT_NONE, 0.f, (hipStream_t)stream);
    if (r != 1) return r;
  }
  SoIgemm p = {};
  p.a = dy; p.b = wt; p.c = dx; p.ws = ws; p.gate = gate;
  p.Nb = Nb; p.H = H; p.W = W; p.C = C;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  p.Ko = Ko; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.TS = S / stride; p.H2 = (H + stride - 1) / stride; p.W2 = (W + stride - 1) / stride;
  p.nclass = stride * stride;
  p.M = Nb * p.H2 * p.W2; p.N = C; p.K = (R / stride) * (S / stride) * Ko;
  p.lda = lddy; p.ldb = R * S * Ko; p.ldc = lddx; p.ldres = 0;
  p.a_bytes = so_extent((long long)Nb * p.Ho * p.Wo, lddy);
  p.b_bytes = so_extent(C, (long long)R * S * Ko);
  if (!p.a_bytes || !p.b_bytes || p.K >= (1 << 24) || p.M >= (1 << 24)) return SO_ERR_SHAPE;
  p.act = SO_ACT_NONE;
  return so_launch<MODE_DGRAD, false, false>(p, ws_bytes, (hipStream_t)stream);
}

int so_conv2d_dgrad_t(const float* dy, int lddy, const float* wt, float* dx, int lddx, int Nb, int H,
                      int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                      long long ws_bytes, void* stream) {
  return so_conv2d_dgrad_t_gated(dy, lddy, wt, dx, lddx, nullptr, Nb, H, W, C, Ko, R, S, stride, pad, ws, ws_bytes, stream);
}

int so_conv2d_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int Nb, int H,
                    int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                    long long ws_bytes, void* stream) {
  if ((Ko & 3) || (lddy & 3) || (C & 3) || (ldx & 3) || !so_aligned16(dy) || !so_aligned16(x))
    return SO_ERR_ALIGN;
  if (Ko == 4 && stride == 1 && !so_forced()) {
    const int r = so_thin_wgrad(dy, lddy, x, ldx, dw, 0, Nb, H, W, C, H + 2 * pad - R + 1, W + 2 * pad - S + 1, R, S, pad, ws,
                                ws ? ws_bytes : 0, (hipStream_t)stream);
    if (r != 1) return r;
  }
  SoIgemm p = {};
  p.a = dy; p.b = x; p.c = dw; p.ws = ws;
  p.Nb = Nb; p.H = H; p.W = W; p.C = C;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  p.Ko = Ko; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.M = Ko; p.N = R * S * C; p.K = Nb * p.Ho * p.Wo;
  p.lda = lddy; p.ldb = ldx; p.ldc = p.N; p.ldres = 0;
  p.a_bytes = so_extent(p.K, lddy);
  p.b_bytes = so_extent((long long)Nb * H * W, ldx);
  if (!p.a_bytes || !p.b_bytes || p.K >= (1 << 24)) return SO_ERR_SHAPE;
  p.act = SO_ACT_NONE; p.nclass = 1;
  return so_launch<MODE_WGRAD, true, true>(p, ws_bytes, (hipStream_t)stream);
}

// Same as so_conv2d_wgrad but dw += (accumulate into an existing gradient slab).
int so_conv2d_wgrad_acc(const float* dy, int lddy, const float* x, int ldx, float* dw, int Nb, int H,
                        int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                        long long ws_bytes, void* stream) {
  if ((Ko & 3) || (lddy & 3) || (C & 3) || (ldx & 3) || !so_aligned16(dy) || !so_aligned16(x))
    return SO_ERR_ALIGN;
  if (Ko == 4 && stride == 1 && !so_forced()) {
    const int r = so_thin_wgrad(dy, lddy, x, ldx, dw, 1, Nb, H, W, C, H + 2 * pad - R + 1, W + 2 * pad - S + 1, R, S, pad, ws,
                                ws ? ws_bytes : 0, (hipStream_t)stream);
    if (r != 1) return r;
  }
  SoIgemm p = {};
  p.a = dy; p.b = x; p.c = dw; p.ws = ws; p.res = dw;
  p.Nb = Nb; p.H = H; p.W = W; p.C = C;
  p.Ho = (H + 2 * pad - R) / stride + 1;
  p.Wo = (W + 2 * pad - S) / stride + 1;
  p.Ko = Ko; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.M = Ko; p.N = R * S * C; p.K = Nb * p.Ho * p.Wo;
  p.lda = lddy; p.ldb = ldx; p.ldc = p.N; p.ldres = p.N;
  p.a_bytes = so_extent(p.K, lddy);
  p.b_bytes = so_extent((long long)Nb * H * W, ldx);
  if (!p.a_bytes || !p.b_bytes || p.K >= (1 << 24)) return SO_ERR_SHAPE;
  p.act = SO_ACT_NONE; p.nclass = 1;
  return so_launch<MODE_WGRAD, true, true>(p, ws_bytes, (hipStream_t)stream);
}

int so_gemm_batched(int transa, int transb, int M, int N, int K, const float* A, int lda,
                    long long sa, const float* B, int ldb, long long sb, float* C, int ldc,
                    long long sc, int batch, const float* alpha, const float* bias,
                    const float* res, int ldres, long long sres, int act, float act_param,
                    float* ws, long long ws_bytes, void* stream) {
  // C[b] (MxN) = act(alpha * opA(A[b]) opB(B[b]) + bias[n] + res[b])
  //   transa == 0: A is [M][K] row-major (lda);  transa == 1: A is [K][M] row-major
  //   transb == 0: B is [K][N] row-major (ldb);  transb == 1: B is [N][K] row-major
  if (!so_aligned16(A) || !so_aligned16(B) || (lda & 3) || (ldb & 3) || (sa & 3) || (sb & 3))
    return SO_ERR_ALIGN;
  if ((transa ? (M & 3) : (K & 3)) || (transb ? (K & 3) : (N & 3))) return SO_ERR_ALIGN;
  SoIgemm p = {};
  p.a = A; p.b = B; p.c = C; p.ws = ws; p.alpha = alpha; p.bias = bias; p.res = res;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres;
  p.sa = sa; p.sb = sb; p.sc = sc; p.sres = sres;
  p.act = act; p.act_param = act_param; p.nclass = batch;
  p.nbias = N;
  p.stride = 1;
  p.a_bytes = so_extent(transa ? K : M, lda);
  p.b_bytes = so_extent(transb ? N : K, ldb);
  if (!p.a_bytes || !p.b_bytes || K >= (1 << 24)) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (!transa && transb) return so_launch<MODE_GEMM, false, false>(p, ws_bytes, st);
  if (!transa && !transb) return so_launch<MODE_GEMM, false, true>(p, ws_bytes, st);
  if (transa && !transb) return so_launch<MODE_GEMM, true, true>(p, ws_bytes, st);
  return SO_ERR_SHAPE;
}

}  // extern "C"
