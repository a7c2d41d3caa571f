// Shared device/host helpers for the ShineOn MI355X (gfx950) kernels.
// Everything in csrc/ is written for CDNA4 only: 64-wide wavefronts, fp32 MFMA,
// 160 KiB LDS per CU.  Tensors are NHWC ("pixel rows x channel columns") with an
// explicit pixel stride `ld` so that channel slices of a wider buffer (the U-Net
// skip concatenations) are first-class operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SO_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Activation codes shared by host and device (include/shineon_hip.h mirrors them).
enum SoAct {
  SO_ACT_NONE = 0,
  SO_ACT_RELU = 1,
  SO_ACT_LEAKY = 2,   // negative slope in act_param
  SO_ACT_GELU = 3,    // exact erf form (reference: nn.GELU(), unet.py:205)
  SO_ACT_SWISH = 4,   // x * sigmoid(x)            (reference: activation.py:13-18)
  SO_ACT_SINE = 5,    // sin(30 x)                 (reference: activation.py:4-10)
  SO_ACT_TANH = 6,
  SO_ACT_SIGMOID = 7,
};

__device__ __forceinline__ float so_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float so_actf(int act, float x, float param) {
  switch (act) {
    case SO_ACT_RELU: return x > 0.f ? x : 0.f;
    case SO_ACT_LEAKY: return x > 0.f ? x : x * param;
    case SO_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case SO_ACT_SWISH: return x * so_sigmoid(x);
    case SO_ACT_SINE: return sinf(30.0f * x);
    case SO_ACT_TANH: return tanhf(x);
    case SO_ACT_SIGMOID: return so_sigmoid(x);
    default: return x;
  }
}

// d act(x) / dx evaluated at the *input* x.
__device__ __forceinline__ float so_actg(int act, float x, float param) {
  switch (act) {
    case SO_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case SO_ACT_LEAKY: return x > 0.f ? 1.f : param;
    case SO_ACT_GELU: {
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case SO_ACT_SWISH: {
      const float s = so_sigmoid(x);
      return s + x * s * (1.f - s);
    }
    case SO_ACT_SINE: return 30.0f * cosf(30.0f * x);
    case SO_ACT_TANH: {
      const float t = tanhf(x);
      return 1.f - t * t;
    }
    case SO_ACT_SIGMOID: {
      const float s = so_sigmoid(x);
      return s * (1.f - s);
    }
    default: return 1.f;
  }
}

__device__ __forceinline__ float so_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float so_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves). `red` is >= 4 floats of LDS.
__device__ __forceinline__ float so_block_sum256(float v, float* red) {
  v = so_wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// A wave's LDS-DMA fills (`buffer_load ... lds`) are pending VM operations of THAT wave only: before a workgroup barrier
// publishes an LDS stage to the other waves, every wave must have drained its own vmcnt.  hipcc happens to emit that
// s_waitcnt in front of s_barrier; stating it keeps a compiler or flag change from turning every staged tile into a silent
// cross-wave race (it merges with the compiler's own wait: no extra instruction in today's ISA).
#define SO_DMA_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)   /* vmcnt(0); expcnt / lgkmcnt left at their maxima */

static inline int so_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

#define SO_LAUNCH_CHECK() ((int)hipGetLastError())
