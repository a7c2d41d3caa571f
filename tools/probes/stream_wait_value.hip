// Can another stream start work in the MIDDLE of a replayed hipGraph?  External event nodes are rejected by PyTorch-ROCm
// ("External events are disallowed in rocm"), so the hand-off is done with HIP stream memory operations instead:
//   * a one-thread kernel INSIDE the captured graph stores the step number into a flag word (signal memory);
//   * the other stream executes hipStreamWaitValue32(flag >= step) - a command-processor wait, no CU is occupied -
//     followed by its own work.
// Prints when the side stream's kernel finished relative to the end of the graph (3 replays, increasing step numbers).
// Build: hipcc --offload-arch=gfx950 -O2 stream_wait_value.hip -o stream_wait_value.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void busy(float* p, int iters) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 1e-9f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
__global__ void signal_k(unsigned* flag, const unsigned* step) {
  __threadfence_system();
  __hip_atomic_store(flag, step[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void bump_k(unsigned* step) { step[0] += 1; }
__global__ void probe_k(const float* src, float* dst) { dst[0] = src[0]; }

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned* flag = nullptr;
  hipError_t fe = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);   // signal memory: exactly 8 bytes
  printf("hipExtMallocWithFlags(8, hipMallocSignalMemory) -> %s\n", hipGetErrorString(fe));
  if (fe != hipSuccess) {
    fe = hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped);   // fine-grained host memory
    printf("hipHostMalloc(coherent) -> %s\n", hipGetErrorString(fe));
    if (fe != hipSuccess) return 1;
  }
  CK(hipMemset(flag, 0, 8));
  unsigned* step = nullptr;
  CK(hipMalloc(&step, 4));
  CK(hipMemset(step, 0, 4));
  float *buf = nullptr, *marker = nullptr, *seen = nullptr;
  CK(hipMalloc(&buf, 256 * 1024 * 4));
  CK(hipMemset(buf, 0, 256 * 1024 * 4));
  CK(hipMalloc(&marker, 4));
  CK(hipMalloc(&seen, 4));
  hipStream_t main_s, side;
  CK(hipStreamCreate(&main_s));
  CK(hipStreamCreate(&side));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeGlobal));
  bump_k<<<1, 1, 0, main_s>>>(step);                       // step counter lives on the device: the graph is replay-invariant
  busy<<<1024, 256, 0, main_s>>>(buf, 20000);              // "first part of the backward pass"
  signal_k<<<1, 1, 0, main_s>>>(flag, step);               // bucket ready
  for (int i = 0; i < 10; ++i) busy<<<1024, 256, 0, main_s>>>(buf, 20000);   // "rest of the backward pass"
  CK(hipStreamEndCapture(main_s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (unsigned it = 1; it <= 3; ++it) {
    hipEvent_t t0, t_side, t_main;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t_side)); CK(hipEventCreate(&t_main));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, main_s));
    CK(hipStreamWaitEvent(side, t0, 0));
    CK(hipGraphLaunch(exec, main_s));
    CK(hipStreamWaitValue32(side, flag, it, hipStreamWaitValueGte, 0xFFFFFFFFu));
    probe_k<<<1, 1, 0, side>>>(buf, seen);
    CK(hipEventRecord(t_side, side));
    CK(hipEventRecord(t_main, main_s));
    CK(hipDeviceSynchronize());
    float a = 0, b = 0;
    CK(hipEventElapsedTime(&a, t0, t_side));
    CK(hipEventElapsedTime(&b, t0, t_main));
    unsigned f = 0;
    CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
    printf("replay %u: side stream released at %.3f ms, graph finished at %.3f ms, flag = %u\n", it, a, b, f);
  }
  return 0;
}
