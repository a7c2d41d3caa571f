// Probe: register layout of v_mfma_f32_4x4x1_16B_f32 on gfx950.
// Hypothesis: block b = lane/4; A_b[i] comes from lane 4b+i, B_b[j] from lane 4b+j; D_b[i][j] lands in lane 4b+j, reg i.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), (float)(100 + l), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main() {
  float* d;
  hipMalloc(&d, 256 * sizeof(float));
  probe<<<1, 64>>>(d);
  float h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const float want = (float)(1 + 4 * (l / 4) + r) * (float)(100 + l);
      if (h[l * 4 + r] != want) ++bad;
    }
  printf("layout hypothesis mismatches: %d\n", bad);
  if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
