// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / padding 1 convolutions of the try-on hot path (fp32 throughout).
//
// Reference ops replaced: torch conv2d at models/networks/vgg.py:9-23 (the frozen VGG19 chain of the perceptual loss,
// models/networks/loss.py:106-122: 57 % of the step's FLOPs) and its input gradient (autograd).
//
// On CDNA4 the fp32-input MFMA runs at the VECTOR rate (157 TFLOP/s), so the only way past the direct convolution's
// roofline in exact-type fp32 arithmetic is to multiply less: F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input
// tile with 16 multiplications per (input channel, output channel) instead of 36 - 2.25x fewer MFMA FLOPs.  All
// transform coefficients are 0, +-1, +-1/2, so the arithmetic stays fp32 with round-off of the same order as the
// direct sum (measured by tests/test_ops_gpu.py against the fp64 convolution).
//
//   V[xi][tile][c]  = (B^T d B)[xi]      input transform   (HBM-bound: reads x once, writes 4x its size)
//   M[xi][tile][ko] = sum_c V[xi][tile][c] * U[xi][ko][c]   16 independent GEMMs = ONE batched launch of the fp32
//                                                           MFMA engine (igemm2.hip, batch = 16)
//   y[2x2 of tile]  = A^T M A + bias -> activation -> ReLU gate    output transform (HBM-bound)
//   U[xi][ko][c]    = (G g G^T)[xi]      weight transform (once for frozen weights)
//
// The input gradient of such a convolution is the same convolution with flipped taps and the channel roles swapped:
// U'[xi][c][ko] = G flip(g) G^T, so one code path serves forward and backward.
//
// This non-fused form pays 16/4 = 4x the activation size in V and in M (write + read each); it wins where the GEMM
// dominates that traffic (C, Ko >= 128).  See DESIGN.md 3.7 for the measured crossover.
#include "common.h"
#include "../../include/shineon_hip.h"
#include "thin.h"   // so_prof_begin / so_prof_end (live HIP-event timing shared with the implicit-GEMM engine)

extern "C" int so_pgemm_nt(int M, int N, int K, const float* A, int lda, long long sa, const float* B, int ldb, long long sb, float* C,
                           int ldc, long long sc, int batch, void* stream);   // csrc/pgemm.hip: 1 = done, 0 = not applicable
extern "C" int so_gemm_batched(int transa, int transb, int M, int N, int K, const float* A, int lda, long long sa,
                               const float* B, int ldb, long long sb, float* C, int ldc, long long sc, int batch,
                               const float* alpha, const float* bias, const float* res, int ldres, long long sres, int act,
                               float act_param, float* ws, long long ws_bytes, void* stream);

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// one thread = (tile, channel quad): 16 16-byte loads, 32 vector adds, 16 16-byte stores
__global__ __launch_bounds__(256) void wino_input_k(const float* __restrict__ x, int ldx, float* __restrict__ V, int Nb, int H,
                                                    int W, int C, int th, int tw) {
  const int cq = C >> 2;
  const long long total = (long long)Nb * th * tw * cq;
  const long long T = (long long)Nb * th * tw;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long tile = idx / cq;
    const int c = (int)(idx - tile * cq) * 4;
    const int n = (int)(tile / (th * tw));
    const int rem = (int)(tile - (long long)n * th * tw);
    const int ty = rem / tw, tx = rem - ty * tw;
    const int h0 = 2 * ty - 1, w0 = 2 * tx - 1;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h = h0 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = w0 + j;
        const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        d[i][j] = ok ? ld4(x + ((long long)(n * H + h) * W + w) * ldx + c) : z;
      }
    }
    // B^T d : combine rows
    f32x4 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0][j] - d[2][j];
      t[1][j] = d[1][j] + d[2][j];
      t[2][j] = d[2][j] - d[1][j];
      t[3][j] = d[1][j] - d[3][j];
    }
    // (B^T d) B : combine columns
    float* out = V + tile * C + c;
    const long long sx = T * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(out + (i * 4 + 0) * sx) = t[i][0] - t[i][2];
      *reinterpret_cast<f32x4*>(out + (i * 4 + 1) * sx) = t[i][1] + t[i][2];
      *reinterpret_cast<f32x4*>(out + (i * 4 + 2) * sx) = t[i][2] - t[i][1];
      *reinterpret_cast<f32x4*>(out + (i * 4 + 3) * sx) = t[i][1] - t[i][3];
    }
  }
}

// one thread = (tile, output-channel quad): 16 loads, A^T M A, bias / activation / gate, four 16-byte pixel stores
__global__ __launch_bounds__(256) void wino_output_k(const float* __restrict__ Mx, const float* __restrict__ bias, int nbias,
                                                     const float* __restrict__ gate, float* __restrict__ y, int ldy, int Nb,
                                                     int H, int W, int Ko, int th, int tw, int act, float act_param) {
  const int kq = Ko >> 2;
  const long long T = (long long)Nb * th * tw;
  const long long total = T * kq;
  const long long sx = T * Ko;
  const float slope = act == SO_ACT_RELU ? 0.f : (act == SO_ACT_LEAKY ? act_param : 1.f);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long tile = idx / kq;
    const int k = (int)(idx - tile * kq) * 4;
    const int n = (int)(tile / (th * tw));
    const int rem = (int)(tile - (long long)n * th * tw);
    const int ty = rem / tw, tx = rem - ty * tw;
    const float* src = Mx + tile * Ko + k;
    f32x4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[i][j] = ld4(src + (i * 4 + j) * sx);
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[0][j] = m[0][j] + m[1][j] + m[2][j];
      s[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = (k + q < nbias) ? bias[k + q] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int h = 2 * ty + a;
      if (h >= H) continue;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int w = 2 * tx + bb;
        if (w >= W) continue;
        f32x4 v = (bb == 0 ? s[a][0] + s[a][1] + s[a][2] : s[a][1] - s[a][2] - s[a][3]) + b;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * slope;   // none / ReLU / LeakyReLU
        const long long off = ((long long)(n * H + h) * W + w) * ldy + k;
        if (gate) {
          const f32x4 g = ld4(gate + off);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = g[q] > 0.f ? v[q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + off) = v;
      }
    }
  }
}

// one thread = one (ko, c) filter: U = G g G^T
__global__ __launch_bounds__(256) void wino_weights_k(const float* __restrict__ w, float* __restrict__ U, int Ko, int C,
                                                      int Kw, int flip_transpose) {
  const long long total = (long long)Ko * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int ko, c;   // consecutive threads = consecutive elements of the OUTPUT row (see wino4_weights_k)
    if (flip_transpose) { c = (int)(idx / Ko); ko = (int)(idx - (long long)c * Ko); }
    else { ko = (int)(idx / C); c = (int)(idx - (long long)ko * C); }
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int rr = flip_transpose ? 2 - r : r, ss = flip_transpose ? 2 - s : s;
        g[r][s] = ko < Kw ? w[((long long)ko * 9 + rr * 3 + ss) * C + c] : 0.f;
      }
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = g[0][s];
      t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
      t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
      t[3][s] = g[2][s];
    }
    const long long sx = (long long)Ko * C;
    float* out = flip_transpose ? U + (long long)c * Ko + ko : U + (long long)ko * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[(i * 4 + 0) * sx] = t[i][0];
      out[(i * 4 + 1) * sx] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      out[(i * 4 + 2) * sx] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      out[(i * 4 + 3) * sx] = t[i][2];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// FUSED F(2x2, 3x3): input transform, the 16 GEMMs and the output transform in ONE kernel - V and M never exist in HBM.
//
// Block = 256 threads (4 waves), TWO blocks per CU (54.5 KB LDS, <= 256 VGPRs): a patch of 8 x 4 Winograd tiles (16 x 8
// output pixels of one image, 18 x 10 input pixels) x KB = 32 output channels x all 16 transform points.  Wave w owns
// transformed ROW i = w (points xi = 4 i + j, j = 0..3): four independent 32 x 32 accumulators, MFMA M axis = the 32 tiles,
// N axis = output channels, K axis = input channels.  (First version: 512 threads x 64 output channels, one block per CU -
// nothing overlapped a block's prologue / epilogue and the C = 64 layers ran at 43 % of the MFMA rate; two independent
// blocks per CU cover each other's ramps.)
// K loop, 8 input channels per step, double-buffered LDS, two register stages in flight (below):
//   * A stage: the RAW 18 x 10 input patch, 8 channels per pixel (straight 16-byte copies; image borders / padding are
//     out-of-range buffer loads = 0).  The input transform happens when a wave builds its fragments ("transform at read"):
//     row i of B^T d B needs two patch rows (a1, a2) and all four patch columns - 8 ds_read_b128 and 8 vector adds yield the
//     four A fragments (one per j) for a lane's tile; each transformed value is consumed by exactly one wave, so nothing is
//     computed twice.  Even / odd patch rows and columns are stored in separate halves (slot = x>>1 + (x&1)*9) with a
//     12-float pixel pitch and 24 slots per row: the stride-2 tile walk of the 16 lanes of a ds_read_b128 group then hits
//     16 distinct bank quads (derivation in DESIGN.md 3.7).
//   * B stage: U[c/8][xi][ko][8] (so_wino_fused_weights): the block's slice of one step is 16 contiguous 1 KB runs.
// Epilogue: A^T M A = column combination in registers (over the wave's four j accumulators), row combination through
// LDS (8 planes of 32 x 32 floats), then bias / activation / ReLU gate and 16-byte pixel stores (128 B per pixel).
struct WinoP {
  const float* x;
  const float* U;
  const float* bias;
  const float* gate;
  float* y;
  float* yp;           // optional: max over each 2x2 output tile (= nn.MaxPool2d(2, 2) of y: an F(2x2) tile is one pooling window)
  unsigned x_bytes, u_bytes;
  int ldx, ldy, Nb, H, W, C, Ko, nbias, pbx, pby, nkb, nks, act;
  int ldyp, n_keep;    // images >= n_keep do not store y at all (their un-pooled activations are never read)
  float act_param;
  int kb_major, npatch;   // block order: 0 = (patch, ko block) with the ko block fastest, 1 = (ko block, patch) with the patch fastest
};

#define WF_SB() __builtin_amdgcn_sched_barrier(0x006)
#define WF_OOB 0x80000000u
typedef int wf_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 wf_bload(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  const wf_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
  f32x4 r;
  r[0] = __int_as_float(v[0]); r[1] = __int_as_float(v[1]); r[2] = __int_as_float(v[2]); r[3] = __int_as_float(v[3]);
  return r;
}

constexpr int WF_NT = 256;                 // threads per block
// Raw A stage, compact form: 8 floats per pixel slot, 20 slots per patch row, and the two channel quads of a pixel swapped
// on odd row slots (physical quad = q ^ (row slot & 1)).  Bank quad of a lane's read = (2 * slot + quad) mod 16: the
// stride-2 tile walk gives tx = 0..7 -> the 8 even quads, a tile row adds 2 * 20 = 8 (mod 16) and flips the parity, so the
// hardware's 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} cover all 16 quads once (same property as the padded
// 12-float / 24-slot layout of the first version, in 6.25 KB instead of 11.25 KB: 44.5 KB per block, three blocks per CU).
constexpr int WF_APIX = 8;                 // floats per pixel slot in the raw A stage
constexpr int WF_AROW = 20;                // pixel slots per patch row
constexpr int WF_ASTAGE = 10 * WF_AROW * WF_APIX;   // 1600 floats = 6.25 KB
// NKG = 32-channel output groups per wave (output channels per block KB = 32 * NKG):
//   NKG = 1: 16 MFMAs per wave and K step, 44.5 KB LDS, <= 168 VGPRs, THREE blocks per CU;
//   NKG = 2: the four A fragments of a lane's tile (8 LDS reads + 32 VALU of input transform) feed 32 MFMAs instead of 16,
//            76.5 KB LDS, <= 256 VGPRs, TWO blocks per CU - for Ko >= 64.
// DMA = true (round 5, the default): both stages are filled by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers,
// no ds_write instructions - the ablation builds (profiles/r05_wino_ablation.txt) put 25 % of the kernel's time on the
// register round trip (ds_write_b128 issue + LDS store bandwidth 15 %, the loads' waits 10 %).  An LDS-DMA instruction writes
// 64 lanes x 16 bytes to ONE contiguous KB of LDS (wave-uniform base in M0 + lane * 16), so the swizzles of both stages move
// to the SOURCE side: lane l of a fill instruction fetches whatever quad belongs in slot l.  The A stage is rounded up to
// 7 full instructions (448 quad slots, 400 used), image borders / padding / channel tails are out-of-range offsets (the
// hardware writes zeros).  The fill of step s + 1 is issued at the top of step s - all waves have left the stage it
// overwrites at the barrier that ended step s - 1 - and is drained by the vmcnt(0) of the barrier that ends step s: a whole
// step (16 NKG MFMAs per wave) to land.
constexpr int WF_ASTAGE_DMA = 448 * 4;
template <int NKG, bool DMA>
__global__ __launch_bounds__(WF_NT, NKG == 1 ? 3 : 2) void wino_fused_k(const WinoP p) {
  constexpr int WF_KB = 32 * NKG;
  constexpr int WF_BSTAGE = 16 * WF_KB * 8;
  constexpr int WF_BQ = 16 * WF_KB * 2 / WF_NT;       // B quads per thread per step (4 or 8)
  constexpr int A_STAGE = DMA ? WF_ASTAGE_DMA : WF_ASTAGE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, in an SGPR: per-wave choices become scalar branches
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware block -> (patch, ko block) map: XCD x works on the x-th contiguous eighth of the (patch, kb) order, so the
  // ko blocks of one patch (which re-read the same input pixels) and neighbouring patches (halo) share one L2
  const unsigned tot = gridDim.x, lin = blockIdx.x;
  const unsigned xper = tot >> 3, xrem = tot & 7, xcd = lin & 7;
  const unsigned lg = xcd * xper + (xcd < xrem ? xcd : xrem) + (lin >> 3);
  // Two orders of the (patch, ko block) items along that list.  ko block fastest (VGG: large activations, <= 1 MB of
  // Winograd-domain filters): the ko blocks of one patch re-read the same input pixels out of L2.  Patch fastest (round 6; the
  // 512-1024-channel layers of the SAMS generator: 17-67 MB of filters, small maps): an XCD then works through ONE ko block's
  // filter slice (<= 4 MB, L2-resident) for all patches instead of streaming the whole filter tensor once per patch -
  // measured on SAMS' most frequent layers as 2 x the kernel's algorithmic bytes (profiles/r05 traffic_ratio 1.82).
  const int kb = p.kb_major ? (int)(lg / (unsigned)p.npatch) : (int)(lg % (unsigned)p.nkb);
  const int patch = p.kb_major ? (int)(lg % (unsigned)p.npatch) : (int)(lg / (unsigned)p.nkb);
  const int pxb = patch % p.pbx;
  const int t2 = patch / p.pbx;
  const int pyb = t2 % p.pby;
  const int n = t2 / p.pby;
  const int k0 = kb * WF_KB;
  const int h_org = 8 * pyb - 1, w_org = 16 * pxb - 1;   // input pixel of patch position (0, 0)

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, (int)p.u_bytes, 0x00020000);

  // ---- loader state ------------------------------------------------------------------------------------------------
  // A: 360 quads (pixel = id >> 1, channel quad = id & 1) of the 18 x 10 patch: ids tid and tid + 256
  int a_src[2], a_dst[2];
  const int aq = tid & 1;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int id = tid + WF_NT * m;
    a_src[m] = -1;
    a_dst[m] = -1;
    if (id >= 360) {   // no second quad for this thread: it repeats its first one (same load, same store - idempotent and
      a_src[m] = a_src[0];   // branch-free, where an `if` around the second ds_write split the K step into basic blocks)
      a_dst[m] = a_dst[0];
    }
    if (id < 360) {
      const int pix = id >> 1;
      const int ppy = pix / 18, ppx = pix - ppy * 18;
      const int h = h_org + ppy, w = w_org + ppx;
      if ((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) a_src[m] = ((n * p.H + h) * p.W + w) * p.ldx + aq * 4;
      const int rs = (ppy >> 1) + (ppy & 1) * 5, cs = (ppx >> 1) + (ppx & 1) * 9;
      a_dst[m] = (rs * WF_AROW + cs) * WF_APIX + ((aq ^ (rs & 1)) << 2);
    }
  }
  // B: 16 * KB * 2 quads per step, WF_BQ per thread: quad id = tid + 256 m -> row = (tid >> 1) + 128 m = xi * KB + ko_local.
  // 128 rows further = 128 / KB transform points further at the same ko_local (and the same swizzle bit): one base each.
  const int b_row0 = tid >> 1, b_q = tid & 1;
  const int b_kol = b_row0 % WF_KB;
  const bool b_ok = k0 + b_kol < p.Ko;
  const int b_src0 = ((b_row0 / WF_KB) * p.Ko + k0 + b_kol) * 8 + b_q * 4;
  const int b_src_m = (128 / WF_KB) * p.Ko * 8;                 // per m
  const int b_dst0 = b_row0 * 8 + ((b_q ^ ((b_row0 >> 3) & 1)) << 2);
  const int b_step = 16 * p.Ko * 8;   // floats per K step in U

  auto load_a = [&](int s, f32x4 (&dst)[2]) {
    const int c = 8 * s + aq * 4;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const bool ok = (a_src[m] >= 0) & (c < p.C);
      dst[m] = wf_bload(rX, ok ? (unsigned)(a_src[m] + 8 * s) * 4u : WF_OOB);
    }
  };
  auto load_b = [&](int s, f32x4 (&dst)[WF_BQ]) {
    const bool ok = b_ok & (s < p.nks);
    const int base = b_src0 + s * b_step;
#pragma unroll
    for (int m = 0; m < WF_BQ; ++m) dst[m] = wf_bload(rU, ok ? (unsigned)(base + m * b_src_m) * 4u : WF_OOB);
  };
  auto store_a = [&](int st, const f32x4 (&v)[2]) {
    *reinterpret_cast<f32x4*>(As + st * A_STAGE + a_dst[0]) = v[0];
    *reinterpret_cast<f32x4*>(As + st * A_STAGE + a_dst[1]) = v[1];
  };
  auto store_b = [&](int st, const f32x4 (&v)[WF_BQ]) {
    float* dst = Bs + st * WF_BSTAGE + b_dst0;
#pragma unroll
    for (int m = 0; m < WF_BQ; ++m) *reinterpret_cast<f32x4*>(dst + m * 128 * 8) = v[m];
  };

  // ---- LDS-DMA fill (DMA = true) --------------------------------------------------------------------------------------
  // A: 7 instructions of 64 quad slots (slot = (row slot * 20 + column slot) * 2 + physical quad); wave w issues j = w and
  // j = w + 4.  B: 16 NKG instructions of 1 KB (32 rows = 32 output channels of one transform point); wave w issues the
  // 4 NKG instructions of its own transformed row (xi = 4 w + j) - any assignment would do, the barrier publishes all of them.
  constexpr int NA_DMA = 2, NB_DMA = 4 * NKG;
  int da_src[NA_DMA];      // element offset of the lane's quad at step 0, or -1 (border / padding slot / no instruction)
  int da_q4[NA_DMA];       // first channel of that quad within the step's 8 channels (0 or 4)
  int db_src[NB_DMA];      // element offset in U at step 0, or -1 (output channel beyond Ko)
  (void)da_src; (void)da_q4; (void)db_src;
  if constexpr (DMA) {
#pragma unroll
    for (int m = 0; m < NA_DMA; ++m) {
      const int j = wi + 4 * m;                       // instruction index (block-uniform per wave)
      const int slot = 64 * j + lane;                 // quad slot in the stage
      da_src[m] = -1;
      da_q4[m] = 0;
      if (j < 7 && slot < 400) {
        const int pq = slot & 1, ps = slot >> 1;      // physical quad, pixel slot
        const int rs = ps / WF_AROW, cs = ps - rs * WF_AROW;
        const int ppy = rs < 5 ? 2 * rs : 2 * (rs - 5) + 1;
        const int ppx = cs < 9 ? 2 * cs : 2 * (cs - 9) + 1;
        const int h = h_org + ppy, w = w_org + ppx;
        const int q = pq ^ (rs & 1);                  // logical channel quad stored in this physical quad
        da_q4[m] = q * 4;
        if (cs < 18 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) da_src[m] = ((n * p.H + h) * p.W + w) * p.ldx + q * 4;
      }
    }
#pragma unroll
    for (int m = 0; m < NB_DMA; ++m) {
      const int row = (4 * wi * WF_KB) + 32 * m + (lane >> 1);     // row of the B stage = xi * KB + ko_local
      const int xi = row / WF_KB, kol = row - xi * WF_KB;
      const int q = (lane & 1) ^ ((row >> 3) & 1);                 // logical quad stored in this physical quad
      db_src[m] = (k0 + kol < p.Ko) ? (xi * p.Ko + k0 + kol) * 8 + q * 4 : -1;
    }
  }
  auto dma_fill = [&](int st, int s) {
    if constexpr (DMA) {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      float* as = As + st * A_STAGE;
      float* bs = Bs + st * WF_BSTAGE;
      const int c0 = 8 * s;
#pragma unroll
      for (int m = 0; m < NA_DMA; ++m) {
        const int j = wi + 4 * m;
        if (j < 7) {   // wave-uniform
          const bool ok = (da_src[m] >= 0) & (c0 + da_q4[m] < p.C);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr)(as + 256 * j), 16, ok ? (unsigned)(da_src[m] + c0) * 4u : WF_OOB, 0, 0, 0);
        }
      }
      const int ub = s * b_step;
#pragma unroll
      for (int m = 0; m < NB_DMA; ++m) {
        const bool ok = db_src[m] >= 0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rU, (lds_ptr)(bs + (4 * wi * WF_KB + 32 * m) * 8), 16,
                                                 ok ? (unsigned)(db_src[m] + ub) * 4u : WF_OOB, 0, 0, 0);
      }
    }
  };

  // ---- fragment addresses (K-step invariant) -------------------------------------------------------------------------
  // transformed row i = wi:  t[b] = d[a1][b] + sgn * d[a2][b]   (B^T rows: (1,0,-1,0) (0,1,1,0) (0,-1,1,0) (0,1,0,-1))
  const int a1 = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int a2 = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float sgn = wi == 1 ? 1.f : -1.f;
  const int ttx = li & 7, tty = li >> 3;
  // all fragment addresses are ONE base per operand row plus compile-time offsets (ds_read immediates): patch column b adds
  // ((b >> 1) + (b & 1) * 9) slots; transform point j adds j * KB rows and group g 32 rows of B (neither changes the swizzle)
  const int r1 = tty + (a1 >> 1) + (a1 & 1) * 5, r2 = tty + (a2 >> 1) + (a2 & 1) * 5;
  const int fa1_0 = (r1 * WF_AROW + ttx) * WF_APIX + ((lh ^ (r1 & 1)) << 2);
  const int fa2_0 = (r2 * WF_AROW + ttx) * WF_APIX + ((lh ^ (r2 & 1)) << 2);
  const int fb_row0 = 4 * wi * WF_KB + li;
  const int fb_0 = fb_row0 * 8 + ((lh ^ ((li >> 3) & 1)) << 2);

  f32x16 acc[NKG][4];
#pragma unroll
  for (int g = 0; g < NKG; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][j][r] = 0.f;

  // Register-side prefetch.  WF_RING = 2: two stages in flight (the loads of step s + 3 are issued during step s and written
  // to LDS during step s + 2; K loop unrolled by two so the ring is indexed statically) - needed while ONE or TWO blocks
  // shared a CU.  WF_RING = 1 (default with three blocks per CU): one stage (issued at the end of step s, written during
  // step s + 1); the other two blocks of the CU cover the latency and the kernel fits 168 VGPRs without spilling.
  // Past-the-end loads return 0, so trailing steps need no branches.
#ifndef WF_RING
#define WF_RING 1
#endif
#ifndef WF_ABLATE
#define WF_ABLATE 0
#endif
  // WF_ABLATE (measurement builds only, results WRONG): 1 no MFMAs, 2 no LDS stores, 4 no global loads, 8 no barrier,
  // 16 fragments (LDS reads + input transform) built once instead of every step, 32 LDS reads kept but no transform VALU
  f32x4 v[4], bf[NKG][4];
  auto build_frags = [&](int rd) {
    const float* as = As + rd * A_STAGE;
    const float* bs = Bs + rd * WF_BSTAGE;
    f32x4 d1[4], d2[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      d1[b] = *reinterpret_cast<const f32x4*>(as + fa1_0 + ((b >> 1) + (b & 1) * 9) * WF_APIX);
      d2[b] = *reinterpret_cast<const f32x4*>(as + fa2_0 + ((b >> 1) + (b & 1) * 9) * WF_APIX);
    }
#pragma unroll
    for (int g = 0; g < NKG; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[g][j] = *reinterpret_cast<const f32x4*>(bs + fb_0 + (j * WF_KB + g * 32) * 8);
    WF_SB();
#if WF_ABLATE & 32
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = d1[j] + d2[j];
#else
    f32x4 t[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[b][e] = __builtin_fmaf(sgn, d2[b][e], d1[b][e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[0][e] = t[0][e] - t[2][e];
      v[1][e] = t[1][e] + t[2][e];
      v[2][e] = t[2][e] - t[1][e];
      v[3][e] = t[1][e] - t[3][e];
    }
#endif
  };
#if WF_ABLATE & 1
#define WF_MMA(e_)                                                                                 \
  _Pragma("unroll") for (int g = 0; g < NKG; ++g) _Pragma("unroll") for (int j = 0; j < 4; ++j)   \
      acc[g][j][0] += v[j][e_] * bf[g][j][e_];
#else
#define WF_MMA(e_)                                                                                 \
  _Pragma("unroll") for (int g = 0; g < NKG; ++g) _Pragma("unroll") for (int j = 0; j < 4; ++j)   \
      acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][e_], bf[g][j][e_], acc[g][j], 0, 0, 0);
#endif
  auto kstep = [&](int rd, f32x4 (&ra)[2], f32x4 (&rb)[WF_BQ], int s_next) {
#if !(WF_ABLATE & 16)
    build_frags(rd);
#endif
    WF_MMA(0)
    WF_MMA(1)
    WF_SB();
#if WF_ABLATE & 2
    asm volatile("" :: "v"(ra[0][0]), "v"(ra[1][3]), "v"(rb[0][0]), "v"(rb[WF_BQ - 1][3]));
#else
    store_a(rd ^ 1, ra);     // the stage consumed by the NEXT step
    store_b(rd ^ 1, rb);
#endif
    WF_SB();
    WF_MMA(2)
    WF_SB();
    // Round 5: the refill is issued HERE, right behind the stores that vacated the registers (it used to be the last thing
    // of the step): a load now has a quarter of this step + the fragment phase and first half of the next step to land
    // (12+ MFMAs) instead of 8 MFMAs; the ISA had vmcnt(5) .. vmcnt(0) in front of the six ds_writes of every step.
#if !(WF_ABLATE & 4)
    load_a(s_next, ra);
    load_b(s_next, rb);
#endif
    WF_SB();
    WF_MMA(3)
    WF_SB();
#if !(WF_ABLATE & 8)
    __syncthreads();
#endif
  };
  if constexpr (DMA) {
    // sched_barrier(0) on both sides of the closing barrier: MFMAs touch no memory, so without it the compiler sinks most of
    // the step's MFMAs BELOW `s_waitcnt vmcnt(0); s_barrier` - i.e. drains the fill it has just issued (seen in the ISA)
    auto kstep_dma = [&](int rd, int s_next) {
#if !(WF_ABLATE & 4)
      if (s_next < p.nks) dma_fill(rd ^ 1, s_next);   // block-uniform; the stage it overwrites was left at the last barrier
#endif
      __builtin_amdgcn_sched_barrier(0);
#if !(WF_ABLATE & 16)
      build_frags(rd);
#endif
      WF_MMA(0)
      WF_MMA(1)
      WF_SB();
      WF_MMA(2)
      WF_MMA(3)
      __builtin_amdgcn_sched_barrier(0);
#if !(WF_ABLATE & 8)
      SO_DMA_DRAIN();    // every wave's own fills have landed before the barrier publishes the stage (common.h)
      __syncthreads();
#endif
      __builtin_amdgcn_sched_barrier(0);
    };
    dma_fill(0, 0);
    SO_DMA_DRAIN();
    __syncthreads();
#if WF_ABLATE & 16
    build_frags(0);
#endif
    // (no exit from the middle of the body: the accumulators would be live across a merge point and get copied)
    int s = 0;
    for (; s + 1 < p.nks; s += 2) {
      kstep_dma(0, s + 1);
      kstep_dma(1, s + 2);
    }
    if (s < p.nks) kstep_dma(0, s + 1);
  } else {
#if WF_RING == 2
  f32x4 ra0[2], ra1[2], rb0[WF_BQ], rb1[WF_BQ];
  {
    f32x4 a0[2], b0[WF_BQ];
    load_a(0, a0);
    load_b(0, b0);
    load_a(1, ra0);
    load_b(1, rb0);
    load_a(2, ra1);
    load_b(2, rb1);
    store_a(0, a0);
    store_b(0, b0);
  }
  __syncthreads();
  for (int s = 0; s < p.nks; s += 2) {
    kstep(0, ra0, rb0, s + 3);
    kstep(1, ra1, rb1, s + 4);
  }
#else
  f32x4 ra0[2], rb0[WF_BQ];
  {
    f32x4 a0[2], b0[WF_BQ];
    load_a(0, a0);
    load_b(0, b0);
    load_a(1, ra0);
    load_b(1, rb0);
    store_a(0, a0);
    store_b(0, b0);
  }
  __syncthreads();
#if WF_ABLATE & 16
  build_frags(0);
#endif
  int cur = 0;
  for (int s = 0; s < p.nks; ++s) {
    kstep(cur, ra0, rb0, s + 2);
    cur ^= 1;
  }
#endif
  }

  // ---- epilogue: A^T M A -----------------------------------------------------------------------------------------------
  // columns (in registers): P[i][0] = M[i][0] + M[i][1] + M[i][2],  P[i][1] = M[i][1] - M[i][2] - M[i][3]
  float* Ps = smem;   // [i 4][b 2][tile 32][ko KB]
#pragma unroll
  for (int g = 0; g < NKG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float p0 = acc[g][0][r] + acc[g][1][r] + acc[g][2][r];
      const float p1 = acc[g][1][r] - acc[g][2][r] - acc[g][3][r];
      Ps[((wi * 2 + 0) * 32 + tile) * WF_KB + g * 32 + li] = p0;
      Ps[((wi * 2 + 1) * 32 + tile) * WF_KB + g * 32 + li] = p1;
    }
  __syncthreads();
  // rows (through LDS): y[0][b] = P[0][b] + P[1][b] + P[2][b],  y[1][b] = P[1][b] - P[2][b] - P[3][b]
#pragma unroll
  for (int rep = 0; rep < NKG; ++rep) {
    const int task = tid + WF_NT * rep;                  // (tile, output-channel quad): 32 x (KB / 4) tasks
    const int tile = task / (WF_KB / 4), kq = (task % (WF_KB / 4)) * 4;
    const int ty = 4 * pyb + (tile >> 3), tx = 8 * pxb + (tile & 7);
    const int ko = k0 + kq;
    if (ko < p.Ko) {
      f32x4 P[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b) P[i][b] = *reinterpret_cast<const f32x4*>(Ps + ((i * 2 + b) * 32 + tile) * WF_KB + kq);
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = (ko + q < p.nbias) ? p.bias[ko + q] : 0.f;
      }
      // the convolutions that reach this kernel fuse no activation, ReLU or LeakyReLU only (ops.conv2d): one select
      const float slope = p.act == SO_ACT_RELU ? 0.f : (p.act == SO_ACT_LEAKY ? p.act_param : 1.f);
      f32x4 vmax;
      const bool keep = n < p.n_keep;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int h = 2 * ty + a;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int w = 2 * tx + b;
          if (h < p.H && w < p.W) {
            f32x4 v = (a == 0 ? P[0][b] + P[1][b] + P[2][b] : P[1][b] - P[2][b] - P[3][b]) + bv;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * slope;   // none / ReLU / LeakyReLU
            const long long off = ((long long)(n * p.H + h) * p.W + w) * p.ldy + ko;
            if (p.gate) {
              const f32x4 g = ld4(p.gate + off);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = g[q] > 0.f ? v[q] : 0.f;
            }
            if (keep) *reinterpret_cast<f32x4*>(p.y + off) = v;
            if (a == 0 && b == 0) vmax = v;
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) vmax[q] = fmaxf(vmax[q], v[q]);
            }
          }
        }
      }
      // pooled output (launcher: H and W even, so a tile is inside the image entirely or not at all)
      if (p.yp && 2 * ty < p.H && 2 * tx < p.W)
        *reinterpret_cast<f32x4*>(p.yp + ((long long)(n * (p.H >> 1) + ty) * (p.W >> 1) + tx) * p.ldyp + ko) = vmax;
    }
  }
}

// weights in the fused kernel's order: U[c/8][xi][ko][8] (channels beyond C zero-filled up to a multiple of 8)
__global__ __launch_bounds__(256) void wino_weights_fused_k(const float* __restrict__ w, float* __restrict__ U, int Ko, int C,
                                                            int Kw, int flip_transpose) {
  // flip_transpose = 0: GEMM (N, K) = (ko, c) of w[ko][tap][c];  1: (N, K) = (c, ko), taps flipped (input gradient)
  const int Nn = flip_transpose ? C : Ko, Kk = flip_transpose ? Ko : C;
  const int K8 = (Kk + 7) / 8 * 8;
  const long long total = (long long)Nn * K8;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int nn = (int)(idx / K8), kk = (int)(idx - (long long)nn * K8);
    const int ko = flip_transpose ? kk : nn, c = flip_transpose ? nn : kk;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int rr = flip_transpose ? 2 - r : r, ss = flip_transpose ? 2 - s : s;
        g[r][s] = (ko < Kw && kk < Kk) ? w[((long long)ko * 9 + rr * 3 + ss) * C + c] : 0.f;
      }
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = g[0][s];
      t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
      t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
      t[3][s] = g[2][s];
    }
    // element (step = kk / 8, xi, nn, kk % 8)
    float* out = U + ((long long)(kk >> 3) * 16 * Nn + nn) * 8 + (kk & 7);
    const long long sx = (long long)Nn * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[(i * 4 + 0) * sx] = t[i][0];
      out[(i * 4 + 1) * sx] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      out[(i * 4 + 2) * sx] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      out[(i * 4 + 3) * sx] = t[i][2];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// F(4x4, 3x3), non-fused: 36 transform points per 4x4 output tile (6x6 input tile) - 2.25 multiplications per output
// instead of 4 (F(2x2)) or 9 (direct), and only 36/16 = 2.25x the activation size in transformed operands (F(2x2): 4x).
// Used for the deep VGG layers (>= 256 channels, <= 64x48 pixels), where the 36-GEMM batch fills the chip even at 96 tiles.
// Interpolation points {0, 1, -1, 2, -1/2, inf}: a search over small rational point sets with this kernel's arithmetic
// (fp64 transforms, one fp32 rounding of U, V, M; profiles/r03_wino_points.txt) gives 1.7x lower rms and 2.6x lower maximum
// error than the textbook {0, +-1, +-2, inf} (Lavin & Gray), in line with Barabasz et al. 2018.
//   B^T rows: (1,3/2,-2,-3/2,1,0) (0,-1,-5/2,-1/2,1,0) (0,1,1/2,-5/2,1,0) (0,-1/2,-1,1/2,1,0) (0,2,-1,-2,1,0) (0,1,3/2,-2,-3/2,1)
//   G rows  : (1,0,0) (-1/3,-1/3,-1/3) (1/3,-1/3,1/3) (1/15,2/15,4/15) (-16/15,8/15,-4/15) (0,0,1)
//   A^T rows: (1,1,1,1,1,0) (0,1,-1,2,-1/2,0) (0,1,1,4,1/4,0) (0,1,-1,8,-1/8,1)
// The transforms themselves run in fp64 (exact for these small-integer combinations of fp32 values) and round ONCE when the
// transformed operand is stored: V, M and y then carry a single fp32 rounding each instead of the accumulated round-off of
// two 6-point passes with coefficients up to 8.  The kernels are HBM-bound; the fp64 vector rate is not a limit.
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 w4_d(const f32x4& v) { f64x4 r = {(double)v[0], (double)v[1], (double)v[2], (double)v[3]}; return r; }
__device__ __forceinline__ f32x4 w4_f(const f64x4& v) { f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; return r; }

__device__ __forceinline__ void w4_bt(const f64x4 (&d)[6], f64x4 (&t)[6]) {
  t[0] = d[0] + 1.5 * (d[1] - d[3]) - 2.0 * d[2] + d[4];
  t[1] = d[4] - d[1] - 2.5 * d[2] - 0.5 * d[3];
  t[2] = d[4] + d[1] + 0.5 * d[2] - 2.5 * d[3];
  t[3] = d[4] - d[2] + 0.5 * (d[3] - d[1]);
  t[4] = d[4] - d[2] + 2.0 * (d[1] - d[3]);
  t[5] = d[1] + 1.5 * (d[2] - d[4]) - 2.0 * d[3] + d[5];
}

// one thread = (tile, channel PAIR-of-pairs): 36 16-byte loads; the 6x6 fp64 intermediate is kept per column pass
__global__ __launch_bounds__(256) void wino4_input_k(const float* __restrict__ x, int ldx, float* __restrict__ V, int Nb, int H,
                                                     int W, int C, int th, int tw) {
  const int cq = C >> 2;
  const long long T = (long long)Nb * th * tw;
  const long long total = T * cq;
  const long long sx = T * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long tile = idx / cq;
    const int c = (int)(idx - tile * cq) * 4;
    const int n = (int)(tile / (th * tw));
    const int rem = (int)(tile - (long long)n * th * tw);
    const int ty = rem / tw, tx = rem - ty * tw;
    const int h0 = 4 * ty - 1, w0 = 4 * tx - 1;
    f64x4 t[6][6];   // t[i][j] = (B^T d)[i][j]: columns first
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f64x4 d[6], o[6];
      const int w = w0 + j;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int h = h0 + i;
        const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        d[i] = w4_d(ok ? ld4(x + ((long long)(n * H + h) * W + w) * ldx + c) : z);
      }
      w4_bt(d, o);
#pragma unroll
      for (int i = 0; i < 6; ++i) t[i][j] = o[i];
    }
    float* out = V + tile * C + c;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      f64x4 o[6];
      w4_bt(t[i], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(out + (i * 6 + j) * sx) = w4_f(o[j]);
    }
  }
}

__device__ __forceinline__ void w4_at(const f64x4 (&m)[6], f64x4 (&y)[4]) {
  const f64x4 s12 = m[1] + m[2], d12 = m[1] - m[2];
  y[0] = m[0] + s12 + m[3] + m[4];
  y[1] = d12 + 2.0 * m[3] - 0.5 * m[4];
  y[2] = s12 + 4.0 * m[3] + 0.25 * m[4];
  y[3] = d12 + 8.0 * m[3] - 0.125 * m[4] + m[5];
}

__global__ __launch_bounds__(256) void wino4_output_k(const float* __restrict__ Mx, const float* __restrict__ bias, int nbias,
                                                      const float* __restrict__ gate, float* __restrict__ y, int ldy, int Nb,
                                                      int H, int W, int Ko, int th, int tw, int act, float act_param) {
  const int kq = Ko >> 2;
  const long long T = (long long)Nb * th * tw;
  const long long total = T * kq;
  const long long sx = T * Ko;
  const float slope = act == SO_ACT_RELU ? 0.f : (act == SO_ACT_LEAKY ? act_param : 1.f);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long tile = idx / kq;
    const int k = (int)(idx - tile * kq) * 4;
    const int n = (int)(tile / (th * tw));
    const int rem = (int)(tile - (long long)n * th * tw);
    const int ty = rem / tw, tx = rem - ty * tw;
    const float* src = Mx + tile * Ko + k;
    f64x4 s[4][6];   // s[a][j] = (A^T M)[a][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f64x4 m[6], o[4];
#pragma unroll
      for (int i = 0; i < 6; ++i) m[i] = w4_d(ld4(src + (i * 6 + j) * sx));
      w4_at(m, o);
#pragma unroll
      for (int a = 0; a < 4; ++a) s[a][j] = o[a];
    }
    f64x4 b = {0.0, 0.0, 0.0, 0.0};
    if (bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = (k + q < nbias) ? (double)bias[k + q] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int h = 4 * ty + a;
      f64x4 o[4];
      w4_at(s[a], o);
      if (h >= H) continue;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int w = 4 * tx + bb;
        if (w >= W) continue;
        f32x4 v = w4_f(o[bb] + b);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * slope;
        const long long off = ((long long)(n * H + h) * W + w) * ldy + k;
        if (gate) {
          const f32x4 g = ld4(gate + off);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = g[q] > 0.f ? v[q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + off) = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void wino4_weights_k(const float* __restrict__ w, float* __restrict__ U, int Ko, int C,
                                                       int Kw, int flip_transpose) {
  const long long total = (long long)Ko * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    // consecutive threads follow the OUTPUT's fastest index (c forward, ko for the transposed U'): the 36 stores of a thread
    // are then 36 coalesced rows; the transposed form pays with 9 strided loads instead (it used to pay with 36 strided stores)
    int ko, c;
    if (flip_transpose) { c = (int)(idx / Ko); ko = (int)(idx - (long long)c * Ko); }
    else { ko = (int)(idx / C); c = (int)(idx - (long long)ko * C); }
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int rr = flip_transpose ? 2 - r : r, ss = flip_transpose ? 2 - s : s;
        g[r][s] = ko < Kw ? w[((long long)ko * 9 + rr * 3 + ss) * C + c] : 0.f;
      }
    double t[6][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const double g0 = g[0][s], g1 = g[1][s], g2 = g[2][s];
      t[0][s] = g0;
      t[1][s] = -(g0 + g1 + g2) / 3.0;
      t[2][s] = (g0 - g1 + g2) / 3.0;
      t[3][s] = (g0 + 2.0 * g1 + 4.0 * g2) / 15.0;
      t[4][s] = (-16.0 * g0 + 8.0 * g1 - 4.0 * g2) / 15.0;
      t[5][s] = g2;
    }
    const long long sx = (long long)Ko * C;
    float* out = flip_transpose ? U + (long long)c * Ko + ko : U + (long long)ko * C + c;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double g0 = t[i][0], g1 = t[i][1], g2 = t[i][2];
      out[(i * 6 + 0) * sx] = (float)g0;
      out[(i * 6 + 1) * sx] = (float)(-(g0 + g1 + g2) / 3.0);
      out[(i * 6 + 2) * sx] = (float)((g0 - g1 + g2) / 3.0);
      out[(i * 6 + 3) * sx] = (float)((g0 + 2.0 * g1 + 4.0 * g2) / 15.0);
      out[(i * 6 + 4) * sx] = (float)((-16.0 * g0 + 8.0 * g1 - 4.0 * g2) / 15.0);
      out[(i * 6 + 5) * sx] = (float)g2;
    }
  }
}

inline int grid_for(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace

extern "C" {

long long so_wino_ws_floats(int Nb, int H, int W, int C, int Ko) {
  const long long T = (long long)Nb * ((H + 1) / 2) * ((W + 1) / 2);
  return 16 * T * ((long long)C + Ko);
}

// Ko rows are written; rows >= Kw are zero (output-channel padding).  flip_transpose = 0: U[16][Ko][C] for the forward
// convolution; 1: U'[16][C][Ko] built from the flipped taps, for the input gradient (which is then the same convolution with
// the roles of C and Ko swapped).
int so_wino_weights(const float* w, float* U, int Ko, int Kw, int C, int flip_transpose, void* stream) {
  if (Ko <= 0 || C <= 0 || Kw > Ko) return SO_ERR_SHAPE;
  hipLaunchKernelGGL(wino_weights_k, dim3(grid_for((long long)Ko * C)), dim3(256), 0, (hipStream_t)stream, w, U, Ko, C, Kw,
                     flip_transpose);
  return SO_LAUNCH_CHECK();
}

int so_wino_conv3x3(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                    int ldy, int Nb, int H, int W, int C, int Ko, int act, float act_param, float* wino_ws,
                    long long wino_ws_bytes, float* ws, long long ws_bytes, void* stream) {
  if ((C & 3) || (Ko & 3) || (ldx & 3) || (ldy & 3) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 15) || (((uintptr_t)U) & 15) ||
      (((uintptr_t)wino_ws) & 15) || (gate && (((uintptr_t)gate) & 15)))
    return SO_ERR_ALIGN;
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  const long long T = (long long)Nb * th * tw;
  if (T <= 0 || T >= (1 << 24) || so_wino_ws_floats(Nb, H, W, C, Ko) * 4 > wino_ws_bytes) return SO_ERR_SHAPE;
  if (act != SO_ACT_NONE && act != SO_ACT_RELU && act != SO_ACT_LEAKY) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  float* V = wino_ws;
  float* Mx = wino_ws + 16 * T * C;
  hipLaunchKernelGGL(wino_input_k, dim3(grid_for(T * (C >> 2))), dim3(256), 0, st, x, ldx, V, Nb, H, W, C, th, tw);
  int err = SO_LAUNCH_CHECK();
  if (err) return err;
  err = so_pgemm_nt((int)T, Ko, C, V, C, T * C, U, C, (long long)Ko * C, Mx, Ko, T * Ko, 16, stream);   // persistent short-K form
  if (err == SO_NOT_APPLICABLE)   // declined (shape / alignment / switched off): the general engine
  err = so_gemm_batched(0, 1, (int)T, Ko, C, V, C, T * C, U, C, (long long)Ko * C, Mx, Ko, T * Ko, 16, nullptr, nullptr, nullptr,
                        0, 0, SO_ACT_NONE, 0.f, ws, ws_bytes, stream);
  if (err) return err;
  hipLaunchKernelGGL(wino_output_k, dim3(grid_for(T * (Ko >> 2))), dim3(256), 0, st, (const float*)Mx, bias, nbias, gate, y, ldy,
                     Nb, H, W, Ko, th, tw, act, act_param);
  return SO_LAUNCH_CHECK();
}

// ---- F(4x4, 3x3), non-fused ------------------------------------------------------------------------------------------------
long long so_wino4_ws_floats(int Nb, int H, int W, int C, int Ko) {
  const long long T = (long long)Nb * ((H + 3) / 4) * ((W + 3) / 4);
  return 36 * T * ((long long)C + Ko);
}

// U[36][Ko][C] (flip_transpose = 0) or U'[36][C][Ko] from the flipped taps (input gradient)
int so_wino4_weights(const float* w, float* U, int Ko, int Kw, int C, int flip_transpose, void* stream) {
  if (Ko <= 0 || C <= 0 || Kw > Ko) return SO_ERR_SHAPE;
  hipLaunchKernelGGL(wino4_weights_k, dim3(grid_for((long long)Ko * C)), dim3(256), 0, (hipStream_t)stream, w, U, Ko, C, Kw,
                     flip_transpose);
  return SO_LAUNCH_CHECK();
}

int so_wino4_conv3x3(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                     int ldy, int Nb, int H, int W, int C, int Ko, int act, float act_param, float* wino_ws,
                     long long wino_ws_bytes, float* ws, long long ws_bytes, void* stream) {
  if ((C & 3) || (Ko & 3) || (ldx & 3) || (ldy & 3) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 15) || (((uintptr_t)U) & 15) ||
      (((uintptr_t)wino_ws) & 15) || (gate && (((uintptr_t)gate) & 15)))
    return SO_ERR_ALIGN;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const long long T = (long long)Nb * th * tw;
  if (T <= 0 || T >= (1 << 24) || so_wino4_ws_floats(Nb, H, W, C, Ko) * 4 > wino_ws_bytes) return SO_ERR_SHAPE;
  if (act != SO_ACT_NONE && act != SO_ACT_RELU && act != SO_ACT_LEAKY) return SO_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  float* V = wino_ws;
  float* Mx = wino_ws + 36 * T * C;
  hipLaunchKernelGGL(wino4_input_k, dim3(grid_for(T * (C >> 2))), dim3(256), 0, st, x, ldx, V, Nb, H, W, C, th, tw);
  int err = SO_LAUNCH_CHECK();
  if (err) return err;
  err = so_pgemm_nt((int)T, Ko, C, V, C, T * C, U, C, (long long)Ko * C, Mx, Ko, T * Ko, 36, stream);   // persistent short-K form
  if (err == SO_NOT_APPLICABLE)   // declined (shape / alignment / switched off): the general engine
  err = so_gemm_batched(0, 1, (int)T, Ko, C, V, C, T * C, U, C, (long long)Ko * C, Mx, Ko, T * Ko, 36, nullptr, nullptr, nullptr,
                        0, 0, SO_ACT_NONE, 0.f, ws, ws_bytes, stream);
  if (err) return err;
  hipLaunchKernelGGL(wino4_output_k, dim3(grid_for(T * (Ko >> 2))), dim3(256), 0, st, (const float*)Mx, bias, nbias, gate, y, ldy,
                     Nb, H, W, Ko, th, tw, act, act_param);
  return SO_LAUNCH_CHECK();
}

// ---- fused kernel --------------------------------------------------------------------------------------------------------
// floats of the fused-layout weights: [ceil(K/8)][16][N][8] with (N, K) = (Ko, C) forward, (C, Ko) for the input gradient
long long so_wino_fused_weight_floats(int Ko, int C, int flip_transpose) {
  const long long Nn = flip_transpose ? C : Ko, Kk = flip_transpose ? Ko : C;
  return (Kk + 7) / 8 * 16 * Nn * 8;
}

int so_wino_fused_weights(const float* w, float* U, int Ko, int Kw, int C, int flip_transpose, void* stream) {
  if (Ko <= 0 || C <= 0 || Kw > Ko) return SO_ERR_SHAPE;
  const long long total = so_wino_fused_weight_floats(Ko, C, flip_transpose) / 16;
  hipLaunchKernelGGL(wino_weights_fused_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, U, Ko, C, Kw,
                     flip_transpose);
  return SO_LAUNCH_CHECK();
}

// y = gate(act(conv3x3_s1_p1(x) + bias)) in ONE launch; U from so_wino_fused_weights with (N, K) = (Ko, C) of THIS call
// (for an input gradient call it with x = dy, C = the convolution's Ko, Ko = its C and the flip_transpose = 1 weights).
// Output channels per block: 32 (three blocks per CU) or 64 for Ko >= 64 (two blocks per CU; the A fragments of a lane's
// tile - 8 LDS reads + 32 VALU - feed 32 MFMAs instead of 16).  With LDS-DMA staging the 64-channel form wins wherever its grid
// still covers the chip twice over (VGG conv1_2 / conv2_x on 8 images: 163 vs 169 us, 146 vs 155 us; equal on 4 images; it
// loses on the small layers, 51 vs 48 us at 144 blocks - profiles/r05_wino_dma_ab.txt) and, since round 6, has >= 128 input
// channels: chosen per launch.
// so_wino_fused_force_kb32: 1 = always 32, 0 = 64 whenever Ko >= 64, -1 (default) = that rule.
static int g_wino_force_nkg1 = -1;
void so_wino_fused_force_kb32(int on) { g_wino_force_nkg1 = on; }
static int g_wino_kb_major = -1;   // -1: by filter size (default); 0 / 1: forced (A/B measurements, tools/one_layer.py)
void so_wino_fused_kb_major(int mode) { g_wino_kb_major = mode; }
static int g_wino_dma = 1;
void so_wino_fused_dma(int on) { g_wino_dma = on; }   // 1 (default): LDS-DMA staging; 0: the register-staged form (A/B measurements)
// (A barrier-free variant - every wave staging only its own operands: B fragments straight from L2 into registers, its 8 of
//  the 10 patch rows in a wave-private LDS region - was built and measured in round 3 (commit 3c3fa6b): correct, but 12 % SLOWER
//  (conv1_2 176 vs 162 us, conv2_2 150 vs 131 us, step 576 vs 589 frames/s: 2x the patch loads, 5 instead of 1.4 loads per
//  lane and step, spills at 168 VGPRs), i.e. the per-step barrier is not what holds the kernel at 53 % of the pipe.  Removed.)

static int wino_fused_launch(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                             int ldy, float* ypool, int ldyp, int n_keep, int Nb, int H, int W, int C, int Ko, int act,
                             float act_param, void* stream);

int so_wino_fused_conv3x3(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                          int ldy, int Nb, int H, int W, int C, int Ko, int act, float act_param, void* stream) {
  return wino_fused_launch(x, ldx, U, bias, nbias, gate, y, ldy, nullptr, 0, Nb, Nb, H, W, C, Ko, act, act_param, stream);
}

// The same convolution with nn.MaxPool2d(2, 2) of its output written by the epilogue (ypool: [Nb][H/2][W/2][ldyp]); y itself is
// stored for the first n_keep images only (VGG19 on [prediction | target]: the target's un-pooled activations are never read).
int so_wino_fused_conv3x3_pool(const float* x, int ldx, const float* U, const float* bias, int nbias, float* y, int ldy,
                               int n_keep, float* ypool, int ldyp, int Nb, int H, int W, int C, int Ko, int act,
                               float act_param, void* stream) {
  if (!ypool || (H & 1) || (W & 1) || (ldyp & 3) || (((uintptr_t)ypool) & 15) || n_keep < 0 || n_keep > Nb) return SO_ERR_SHAPE;
  return wino_fused_launch(x, ldx, U, bias, nbias, nullptr, y, ldy, ypool, ldyp, n_keep, Nb, H, W, C, Ko, act, act_param, stream);
}

static int wino_fused_launch(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                             int ldy, float* ypool, int ldyp, int n_keep, int Nb, int H, int W, int C, int Ko, int act,
                             float act_param, void* stream) {
  if ((C & 3) || (Ko & 3) || (ldx & 3) || (ldy & 3) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 15) || (((uintptr_t)U) & 15) ||
      (gate && (((uintptr_t)gate) & 15)))
    return SO_ERR_ALIGN;
  if (act != SO_ACT_NONE && act != SO_ACT_RELU && act != SO_ACT_LEAKY) return SO_ERR_SHAPE;
  const long long xb = (long long)Nb * H * W * ldx * 4, ub = so_wino_fused_weight_floats(Ko, C, 0) * 4;
  if (xb <= 0 || xb >= 0x7FFFFFF0LL || ub >= 0x7FFFFFF0LL) return SO_ERR_SHAPE;
  WinoP p = {};
  p.x = x; p.U = U; p.bias = bias; p.gate = gate; p.y = y;
  p.yp = ypool; p.ldyp = ldyp; p.n_keep = n_keep;
  p.x_bytes = (unsigned)xb; p.u_bytes = (unsigned)ub;
  p.ldx = ldx; p.ldy = ldy; p.Nb = Nb; p.H = H; p.W = W; p.C = C; p.Ko = Ko; p.nbias = nbias;
  p.pbx = ((W + 1) / 2 + 7) / 8; p.pby = ((H + 1) / 2 + 3) / 4;
  int nkg = 1;   // output channels per block: 32, or 64 (two groups per wave)
  if (Ko >= 64 && g_wino_force_nkg1 <= 0) {
    const long long blocks64 = (long long)Nb * p.pbx * p.pby * ((Ko + 63) / 64);
    // (round 6, warm clocks, 8 images: 64 -> 64 at 256x192 136.5 vs 133.1 us, 64 -> 128 at 128x96 69.8 vs 68.4: with only
    //  8 K steps per block the third resident block of the 32-channel form covers more of the per-block prologue / epilogue
    //  than the shared A fragments save; 128 -> 128 at 128x96 120.5 vs 123.7 the other way - profiles/r06_wino_stagger_experiment.txt)
    if (g_wino_force_nkg1 == 0 || (g_wino_dma && blocks64 >= 1024 && C >= 128)) nkg = 2;
  }
  const int KB = 32 * nkg;
  p.nkb = (Ko + KB - 1) / KB; p.nks = (C + 7) / 8;
  p.act = act; p.act_param = act_param;
  p.npatch = Nb * p.pbx * p.pby;
  // filters far beyond an XCD's 4 MB L2 (>= 512 x 512 channels): walk the patches inside a ko block (see the kernel).  Measured,
  // warm clocks, 4 images: 1024 -> 1024 at 32x24 372 -> 354 us, at 16x12 118 -> 110, 512 -> 512 at 64x48 282 -> 278; at 4 MB
  // (256 -> 256 at 128x96) the other order wins, 264 vs 269
  p.kb_major = (g_wino_kb_major < 0 ? (ub > (8ll << 20) && p.nkb >= 2) : g_wino_kb_major) ? 1 : 0;
  const long long blocks = (long long)Nb * p.pbx * p.pby * p.nkb;
  if (blocks <= 0 || blocks > 0x7FFFFFFF) return SO_ERR_SHAPE;
  const bool dma = g_wino_dma != 0;
  const size_t lds = (size_t)(2 * ((dma ? WF_ASTAGE_DMA : WF_ASTAGE) + 16 * KB * 8)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipSuccess;
    const void* kerns[4] = {reinterpret_cast<const void*>(wino_fused_k<1, false>), reinterpret_cast<const void*>(wino_fused_k<2, false>),
                            reinterpret_cast<const void*>(wino_fused_k<1, true>), reinterpret_cast<const void*>(wino_fused_k<2, true>)};
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
      e = hipFuncSetAttribute(kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((2 * ((i >= 2 ? WF_ASTAGE_DMA : WF_ASTAGE) + 16 * 32 * (1 + (i & 1)) * 8)) * sizeof(float)));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  // live timing (bench.py roofline): key "fprop, tile slot 7" = the fused Winograd kernel.  FLOPs recorded = the ALGORITHMIC
  // FLOPs of the convolution it computes (2 * pixels * Ko * 9C, SURVEY 8d's unit); the matrix pipe executes 1 / 2.25 of them
  // (16 instead of 36 multiplications per 2x2 tile and channel pair) - bench.py reports both.
  const int slot = so_prof_begin(0 * 8 + 7, 2.0 * (double)Nb * H * W * (double)Ko * 9.0 * (double)C, Nb * H * W, Ko, 9 * C,
                                 (hipStream_t)stream);
  // algorithmic bytes: x and y once + the 16 Winograd-domain filter planes this kernel reads (16 / 9 of the 3x3 filter)
  so_prof_bytes(slot, 4.0 * ((double)Nb * H * W * ((double)C + (double)Ko) + 16.0 * (double)C * (double)Ko));
  if (nkg == 2 && dma)
    hipLaunchKernelGGL((wino_fused_k<2, true>), dim3((unsigned)blocks), dim3(WF_NT), lds, (hipStream_t)stream, p);
  else if (nkg == 2)
    hipLaunchKernelGGL((wino_fused_k<2, false>), dim3((unsigned)blocks), dim3(WF_NT), lds, (hipStream_t)stream, p);
  else if (dma)
    hipLaunchKernelGGL((wino_fused_k<1, true>), dim3((unsigned)blocks), dim3(WF_NT), lds, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((wino_fused_k<1, false>), dim3((unsigned)blocks), dim3(WF_NT), lds, (hipStream_t)stream, p);
  so_prof_end(slot, (hipStream_t)stream);
  return SO_LAUNCH_CHECK();
}

}  // extern "C"
