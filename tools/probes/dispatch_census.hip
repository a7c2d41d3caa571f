// Where does the hardware dispatcher put the workgroups of a grid that is SMALLER than the chip's resident capacity?
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dispatch_census.hip -o tools/probes/dispatch_census.bin
//   ./dispatch_census.bin [grid=576] [lds_bytes=32768] [spin_ticks=4000 (100 MHz wall clock: 40 us)] [threads=256]
//
// Every block records (XCC id, HW_ID, start, end) and spins for `spin_cycles` so that all blocks of the first wave are
// resident together.  The host prints the histogram "blocks resident per CU" - if the dispatcher balanced, a 576-block grid
// on 256 CUs would show 2-3 blocks on every CU; if it packs CUs up to their occupancy limit, some CUs hold
// floor(160 KiB / lds_bytes) blocks and others none.  (csrc/igemm2.hip: 64x64 tiles use 32-36 KiB of LDS, i.e. 4 per CU.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

__global__ void census_k(Rec* out, int spin) {
  extern __shared__ float smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) smem[0] = 1.f;
  __syncthreads();
  unsigned long long t = t0;
  while ((long long)(t - t0) < spin) { __builtin_amdgcn_s_sleep(8); t = __builtin_amdgcn_s_memrealtime(); }
  if (threadIdx.x == 0) {
    Rec r;
    r.xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
    r.hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
    r.t0 = t0; r.t1 = t;
    out[blockIdx.x] = r;
  }
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 576;
  const int lds = argc > 2 ? atoi(argv[2]) : 32768;
  const int spin = argc > 3 ? atoi(argv[3]) : 4000;
  const int threads = argc > 4 ? atoi(argv[4]) : 256;
  Rec* d; hipMalloc(&d, sizeof(Rec) * grid);
  hipFuncSetAttribute((const void*)census_k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(census_k, dim3(grid), dim3(threads), lds, 0, d, spin);
    hipDeviceSynchronize();
  }
  std::vector<Rec> h(grid);
  hipMemcpy(h.data(), d, sizeof(Rec) * grid, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, first_end = ~0ull;
  for (auto& r : h) { tmin = std::min(tmin, r.t0); first_end = std::min(first_end, r.t1); }
  std::map<unsigned long long, int> per_cu;   // blocks that started before ANY block ended = co-resident first wave
  std::map<unsigned, int> per_xcc;
  int first_wave = 0;
  for (auto& r : h) {
    if (r.t0 < first_end) {
      const unsigned cu_key = (r.hwid >> 8) & 0xFF;     // cu[3:0] sh se[2:0]
      per_cu[((unsigned long long)(r.xcc & 0xF) << 8) | cu_key] += 1;
      per_xcc[r.xcc & 0xF] += 1;
      ++first_wave;
    }
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second] += 1;
  printf("grid %d, %d threads, %d B LDS per block (max %d blocks/CU by LDS), spin %d ticks of 10 ns\n", grid, threads, lds, 163840 / (lds > 0 ? lds : 1), spin);
  printf("  first-wave blocks %d on %zu distinct CUs (of 256); idle CUs %d\n", first_wave, per_cu.size(), 256 - (int)per_cu.size());
  for (auto& kv : hist) printf("  CUs holding %d blocks: %d\n", kv.first, kv.second);
  printf("  per XCC:");
  for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
  unsigned long long tmax0 = 0;
  for (auto& r : h) if (r.t0 < first_end) tmax0 = std::max(tmax0, r.t0);
  printf("\n  first-wave start skew %.2f us\n", (double)(tmax0 - tmin) / 100.0);
  hipFree(d);
  return 0;
}
