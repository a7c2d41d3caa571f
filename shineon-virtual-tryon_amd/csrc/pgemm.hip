// Persistent batched NT GEMM for the Winograd-domain products of csrc/wino.hip:  C[b] (M x N) = A[b] (M x K) * B[b]^T (N x K),
// b = 0 .. batch-1 (16 or 36 transform points), plain store - no alpha / bias / residual / activation.
//
// Why a second GEMM kernel next to the engine of csrc/igemm2.hip: these products have a SHORT K loop (K = input channels =
// 64 .. 1024, i.e. 2 .. 32 tiles of 32) and thousands of 64x64 output tiles.  In the engine every tile is its own workgroup:
// a prologue that waits for the first fill with nothing to multiply, then the loop, then an epilogue during which the
// workgroup issues no fills.  The ablation builds put the bare MFMA loop + those two ramps at 0.70 of the fp32 MFMA peak for
// K = 256 (profiles/r05_igemm_dma_ablation.txt); the engine reaches 0.59-0.67 on these shapes.  Here a workgroup walks a list
// of tiles and keeps the LDS-DMA pipeline running ACROSS tile boundaries: the first fill of tile i+1 is issued during the last
// K tile of tile i, so it travels while tile i's epilogue runs, and the epilogue has its own LDS patches (the stages stay free
// for that fill).
//
// Same data path as the engine's KC x KC mode (igemm2.hip): 64x64 tile, 4 waves (2x2, one 32x32 accumulator each), BK = 32,
// two LDS stages [rows][32 k] with the so_swz XOR swizzle applied on the source side of the LDS-DMA fills, fragments by
// ds_read_b128, v_mfma_f32_32x32x2_f32 with the same k permutation - the results are bit-identical to the engine's.
// Requires K % 64 == 0 (an even number of K tiles: the stage of every tile is static) and N % 4 == 0; anything else, and any
// epilogue work, stays on the engine.
#include "common.h"
#include "../../include/shineon_hip.h"
#include "thin.h"   // so_prof_begin / so_prof_end / so_prof_bytes (igemm2.hip)

namespace {

struct PGemmP {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc, batch;
  long long sa, sb, sc;     // batch strides in elements
  unsigned a_bytes, b_bytes;
  int tiles_m, tiles_n;
  long long items;          // batch * tiles_m * tiles_n
};

__device__ __forceinline__ int pg_swz(int row) {   // = so_swz of igemm2.hip
  return ((row >> 2) & 1) | (((row >> 3) & 1) << 1) | ((((row >> 1) ^ (row >> 4)) & 1) << 2);
}

#define PG_SB() __builtin_amdgcn_sched_barrier(0x006)
constexpr unsigned PG_OOB = 0x80000000u;
constexpr int PG_STAGE = 64 * 32;          // floats per operand stage
constexpr int PG_LDK = 36;                 // row pitch of the epilogue patches
constexpr int PG_PATCH = 32 * PG_LDK;      // floats per wave patch
constexpr size_t PG_LDS = (size_t)(4 * PG_STAGE + 4 * PG_PATCH) * sizeof(float);   // 32 KB stages + 18 KB patches

typedef __attribute__((address_space(3))) void* pg_lds_ptr;

__global__ __launch_bounds__(256, 3) void pgemm_nt_k(const PGemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][64][32]
  float* Bs = smem + 2 * PG_STAGE;       // [2][64][32]
  float* Ps = smem + 4 * PG_STAGE;       // [4 waves][32][36]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int krow8 = tid >> 3;                        // row within a 32-row pass (two passes per operand)
  const int kq = (tid & 7) ^ pg_swz(krow8);          // logical k quad held by this lane's physical quad

  // XCD-aware item order: workgroups are dealt round-robin to the 8 XCDs; XCD x walks the x-th contiguous eighth of the
  // (batch, tile_m, tile_n) list, so one transform point's operands stay in one 4 MiB L2
  const unsigned G = gridDim.x, gx = G >> 3;                   // G is a multiple of 8 (host)
  const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long long per = p.items / 8, extra = p.items % 8;
  const long long lo = xcd * per + (xcd < extra ? xcd : extra);
  const long long hi = lo + per + (xcd < extra ? 1 : 0);
  const int nk = p.K >> 5;                                     // even (host)
  const int tmn = p.tiles_m * p.tiles_n;

  // per-item loader state: byte offsets of this thread's two quads per operand at K tile 0 (OOB if the row is outside)
  unsigned a_off[2], b_off[2], na_off[2], nb_off[2];
  int m0 = 0, n0 = 0, nm0 = 0, nn0 = 0, bz = 0, nbz = 0;
  auto setup = [&](long long it, unsigned (&ao)[2], unsigned (&bo)[2], int& mm0, int& nn0_, int& bb) {
    bb = (int)(it / tmn);
    const int r = (int)(it - (long long)bb * tmn);
    const int tm = r / p.tiles_n, tn = r - tm * p.tiles_n;
    mm0 = tm * 64;
    nn0_ = tn * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = mm0 + krow8 + 32 * j, n = nn0_ + krow8 + 32 * j;
      ao[j] = m < p.M ? (unsigned)(m * p.lda + kq * 4) * 4u : PG_OOB;
      bo[j] = n < p.N ? (unsigned)(n * p.ldb + kq * 4) * 4u : PG_OOB;
    }
  };
  auto fill = [&](int bb, const unsigned (&ao)[2], const unsigned (&bo)[2], int kt, int st) {
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)bb * p.sa), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)bb * p.sb), 0, (int)p.b_bytes, 0x00020000);
    const unsigned koff = (unsigned)kt * 128u;   // 32 floats
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (pg_lds_ptr)(As + st * PG_STAGE + (wave + 4 * j) * 256), 16,
                                               (int)(ao[j] == PG_OOB ? PG_OOB : ao[j] + koff), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (pg_lds_ptr)(Bs + st * PG_STAGE + (wave + 4 * j) * 256), 16,
                                               (int)(bo[j] == PG_OOB ? PG_OOB : bo[j] + koff), 0, 0, 0);
    }
  };

  f32x16 acc;
  f32x4 fa[2], fb[2];
  const int arow = wm * 32 + li, brow = wn * 32 + li;
  const int aswz = pg_swz(arow) << 2, bswz = pg_swz(brow) << 2;
  auto read_frag = [&](int st, int kc, f32x4& af, f32x4& bf) {
    af = *reinterpret_cast<const f32x4*>(As + st * PG_STAGE + arow * 32 + ((kc * 8) ^ (lh * 4) ^ aswz));
    bf = *reinterpret_cast<const f32x4*>(Bs + st * PG_STAGE + brow * 32 + ((kc * 8) ^ (lh * 4) ^ bswz));
  };
  auto mma = [&](const f32x4& af, const f32x4& bf) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf[t], acc, 0, 0, 0);
  };
  // one K tile out of stage ST; FILL is issued first (into the other stage: all waves left it at the previous barrier)
#define PG_TILE(ST, FILL)                       \
  do {                                          \
    FILL;                                       \
    __builtin_amdgcn_sched_barrier(0);          \
    read_frag(ST, 0, fa[0], fb[0]);             \
    read_frag(ST, 1, fa[1], fb[1]);             \
    PG_SB();                                    \
    mma(fa[0], fb[0]);                          \
    PG_SB();                                    \
    read_frag(ST, 2, fa[0], fb[0]);             \
    PG_SB();                                    \
    mma(fa[1], fb[1]);                          \
    PG_SB();                                    \
    read_frag(ST, 3, fa[1], fb[1]);             \
    PG_SB();                                    \
    mma(fa[0], fb[0]);                          \
    PG_SB();                                    \
    mma(fa[1], fb[1]);                          \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)

  long long it = lo + slot;
  if (it >= hi) return;
  setup(it, a_off, b_off, m0, n0, bz);
  fill(bz, a_off, b_off, 0, 0);
  for (; it < hi; it += gx) {
    const long long nxt = it + gx;
    const bool has_next = nxt < hi;   // block-uniform
    if (has_next) setup(nxt, na_off, nb_off, nm0, nn0, nbz);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    SO_DMA_DRAIN();                   // K tile 0 of this item: every wave's own fills have landed (common.h) ...
    __syncthreads();                  // ... before the barrier publishes the stage
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt + 2 < nk; kt += 2) {
      PG_TILE(0, fill(bz, a_off, b_off, kt + 1, 1));
      SO_DMA_DRAIN();
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      PG_TILE(1, fill(bz, a_off, b_off, kt + 2, 0));
      SO_DMA_DRAIN();
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
    // the last two K tiles: the second one starts the NEXT item's pipeline
    PG_TILE(0, fill(bz, a_off, b_off, nk - 1, 1));
    SO_DMA_DRAIN();
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    PG_TILE(1, if (has_next) fill(nbz, na_off, nb_off, 0, 0));
    // ---- epilogue: 32x32 accumulator -> wave-private LDS patch -> 16-byte row stores; no block barrier needed (the patch is
    // this wave's own, the stages are not touched)
    {
      float* stg = Ps + wave * PG_PATCH;
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * PG_LDK + li] = acc[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float* cb = p.C + (long long)bz * p.sc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 3) + 8 * q, col4 = (lane & 7) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * PG_LDK + col4);
        const int m = m0 + wm * 32 + row, n = n0 + wn * 32 + col4;
        if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(cb + (long long)m * p.ldc + n) = v;
      }
      __builtin_amdgcn_wave_barrier();   // the patch is rewritten by this wave's next epilogue only after these reads
    }
    a_off[0] = na_off[0]; a_off[1] = na_off[1]; b_off[0] = nb_off[0]; b_off[1] = nb_off[1];
    m0 = nm0; n0 = nn0; bz = nbz;
  }
}

static int g_pgemm = 1;

}  // namespace

extern "C" {

void so_pgemm_enable(int on) { g_pgemm = on; }

// 0 = launched; SO_NOT_APPLICABLE = declined, nothing launched (the caller uses the general engine); anything else = failure
int so_pgemm_nt(int M, int N, int K, const float* A, int lda, long long sa, const float* B, int ldb, long long sb, float* C, int ldc,
                long long sc, int batch, void* stream) {
  if (!g_pgemm || batch < 16 || (K & 63) || (N & 3) || (lda & 3) || (ldb & 3) || (ldc & 3) || K < 128) return SO_NOT_APPLICABLE;
  if ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C)) & 15) return SO_NOT_APPLICABLE;
  if ((sa & 3) || (sb & 3) || (sc & 3)) return SO_NOT_APPLICABLE;
  const long long ab = (long long)M * lda * 4, bb = (long long)N * ldb * 4;
  if (ab <= 0 || bb <= 0 || ab >= 0x7FFFFFF0LL || bb >= 0x7FFFFFF0LL) return SO_NOT_APPLICABLE;
  PGemmP p = {};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.batch = batch;
  p.sa = sa; p.sb = sb; p.sc = sc; p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
  p.tiles_m = (M + 63) / 64; p.tiles_n = (N + 63) / 64;
  p.items = (long long)batch * p.tiles_m * p.tiles_n;
  if (p.items < 768) return SO_NOT_APPLICABLE;     // fewer items than resident workgroups: nothing to pipeline across
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pgemm_nt_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PG_LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const unsigned grid = 768;   // 256 CUs x 3 resident workgroups (50 KB LDS each), a multiple of 8
  const int slot = so_prof_begin(4 * 8 + 7 /* bench.py KEY_NAMES[39] = winograd_pgemm */, 2.0 * M * N * (double)K * batch, M, N, K, (hipStream_t)stream);
  so_prof_bytes(slot, 4.0 * batch * ((double)M * K + (double)K * N + (double)M * N));
  hipLaunchKernelGGL(pgemm_nt_k, dim3(grid), dim3(256), PG_LDS, (hipStream_t)stream, p);
  so_prof_end(slot, (hipStream_t)stream);
  const int err = SO_LAUNCH_CHECK();
  return err;
}

}  // extern "C"
