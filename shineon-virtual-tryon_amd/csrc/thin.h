// Internal (not part of the C ABI): kernels for convolutions with FOUR channels on one side, and the live-timing
// hooks they share with the implicit-GEMM engine.  See thin.hip.
#pragma once
#include <hip/hip_runtime.h>

// Records a [begin, end) HIP-event pair on `stream` under so_prof key `key` when live timing is enabled; returns
// a slot index (or -1 when timing is off) to hand to so_prof_end.  Defined in igemm2.hip.
int so_prof_begin(int key, double flops, int M, int N, int K, hipStream_t stream);
void so_prof_end(int slot, hipStream_t stream);
void so_prof_bytes(int slot, double bytes);  // algorithmic HBM bytes of the launch recorded under `slot` (operands + result once)

// y[pix][0..3] = act(bias + sum_{r,s,c} in[pix @ (r,s)][c] * w[j][(r*S+s)*IC + c]) for a stride-1 convolution whose
// OUTPUT has four channel columns.  flip = 0: forward conv (in = x, tap offset r - pad); flip = 1: input gradient
// with transposed weights (in = dy, tap offset pad - r).  `wrows` (<= 4) weight rows exist, the rest read as 0.
// Returns 0 on success, 1 if the shape is not covered (caller falls back to the general engine), < 0 on error.
int so_thin_conv(int flip, const float* in, int ldin, const float* w, int wrows, const float* bias, int nbias,
                 float* y, int ldy, int Nb, int OH, int OW, int IH, int IW, int IC, int R, int S, int pad, int act,
                 float act_param, hipStream_t stream);

// dw[ko < 4][r][s][c] (+)= sum_pix dy[pix][ko] * x[pix @ (r,s)][c] for a stride-1 convolution with FOUR output
// channels.  Same return convention as so_thin_conv.
int so_thin_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int accumulate, int Nb, int H, int W,
                  int C, int Ho, int Wo, int R, int S, int pad, float* ws, long long ws_bytes, hipStream_t stream);

// y[pix][0..N) = act(bias + sum_{tap, c < 4} in[pix @ tap][c] * B[tap * 4 + c][n]) for a 3x3 / stride 1 / pad 1 convolution whose
// INPUT has exactly four channel columns (row stride 4) and N (multiple of 64, <= 256) outputs.  wmode 0: w = [n][tap][4]
// (forward, OHWI); wmode 1: w = [4][tap][n] (OHWI with four output rows, used for the input gradient: in = dy), flip = 1
// reverses the taps.  Same return convention as so_thin_conv.
int so_thin_expand(int flip, int wmode, const float* in, const float* w, const float* bias, float* y, int ldy, int Nb, int H,
                   int W, int N, int act, float act_param, hipStream_t stream);
