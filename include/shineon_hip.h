/* C ABI of libshineon_hip.so — the MI355X (gfx950) kernels behind the ShineOn try-on hot path.
 *
 * The reference (andrewjong/ShineOn-Virtual-Tryon) has no FFI of its own: its hot path calls PyTorch
 * ops from Python.  Each entry point below therefore names the reference call site (file:line under the
 * reference tree) whose torch op it replaces; INTEGRATION.md shows the ctypes stub a maintainer would
 * add.  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise; nothing is allocated,
 *     freed or synchronised inside the library (two documented exceptions, neither on the path of a single-rank step:
 *     so_signal_alloc's 8-byte signal words and so_hostwords_alloc's pinned abort / status words);
 *     `stream` is a hipStream_t (NULL = default stream);
 *   - return value: 0 on success, a hipError_t (>0) from the launch, or a negative SO_ERR_* code;
 *   - activations are "rows x channels" NHWC matrices: element (pixel p, channel c) lives at
 *     ptr[p * ld + c]; `ld` (floats) may exceed the channel count so that a channel slice of a wider
 *     buffer (a U-Net skip concatenation) is a valid operand;
 *   - convolution weights are OHWI: w[ko][r][s][c] (a torch (O,I,H,W) tensor in channels_last memory
 *     format), gradients are produced in the same layout;
 *   - `ws`/`ws_bytes` is caller-owned scratch; the so_*_ws_floats helpers give the required size.
 */
#ifndef SHINEON_HIP_H_
#define SHINEON_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SO_ERR_ALIGN (-1) /* pointer not 16-byte aligned / channel count or stride not a multiple of 4 */
#define SO_ERR_SHAPE (-2) /* unsupported shape combination */
#define SO_NOT_APPLICABLE (-3) /* a specialised entry point declines this problem: nothing was launched, use the general one */

/* activation codes (models/networks/cpvton/unet.py:132-135,201-211; models/networks/activation.py) */
#define SO_ACT_NONE_ 0
#define SO_ACT_RELU_ 1
#define SO_ACT_LEAKY_ 2
#define SO_ACT_GELU_ 3
#define SO_ACT_SWISH_ 4
#define SO_ACT_SINE_ 5
#define SO_ACT_TANH_ 6
#define SO_ACT_SIGMOID_ 7

/* ---- convolution / matmul on fp32 MFMA (csrc/igemm.hip) ------------------------------------------ */

/* nn.Conv2d forward (unet.py:129-131,139-174; warp.py:13-31,73-85; sagan.py:12-20; vgg.py:9-23):
 * y = act(conv(x, w) + bias).  x: [Nb*H*W][C] (ldx), y: [Nb*Ho*Wo][Ko] (ldy), C % 4 == 0.
 * act: any SO_ACT_*; none / ReLU / LeakyReLU are fused into the kernel's epilogue, the others run as a second element-wise
 * pass over y on the same stream (same values). */
int so_conv2d_fprop(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                    int Nb, int H, int W, int C, int Ko, int R, int S, int stride, int pad, int act,
                    float act_param, float* ws, long long ws_bytes, void* stream);

/* same with output-channel zero padding: Ko columns are written, only the first Kw have weights/bias (the
 * rest are act(0)); used when a channel count is not a multiple of 4 (n_frames_total > 1: ngf = 134, 167...). */
int so_conv2d_fprop_padded(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                           int Nb, int H, int W, int C, int Ko, int Kw, int R, int S, int stride, int pad,
                           int act, float act_param, float* ws, long long ws_bytes, void* stream);

/* input gradient of the same convolution (autograd of the call sites above): dx: [Nb*H*W][C]. */
int so_conv2d_dgrad(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nb, int H,
                    int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                    long long ws_bytes, void* stream);

/* the same input gradient computed from TRANSPOSED weights wt[c][r][s][ko] (made by so_ohwi_to_ihwo): both
 * GEMM operands are then k-contiguous, like the forward pass.  C need not be a multiple of 4 here. */
int so_conv2d_dgrad_t(const float* dy, int lddy, const float* wt, float* dx, int lddx, int Nb, int H,
                      int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                      long long ws_bytes, void* stream);
/* so_conv2d_dgrad_t with the backward of the ReLU that produced the convolution's input fused into the epilogue:
 * dx[p][c] = gate[p][c] > 0 ? dx[p][c] : 0, `gate` = that ReLU's output ([Nb*H*W][C], pitch lddx).  Used by the frozen
 * VGG19 chain of the perceptual loss (loss.py:106-122), where conv -> ReLU -> conv repeats 13 times. */
int so_conv2d_dgrad_t_gated(const float* dy, int lddy, const float* wt, float* dx, int lddx, const float* gate,
                            int Nb, int H, int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                            long long ws_bytes, void* stream);
/* w[ko][tap][c] (OHWI) -> wt[c][tap][ko] (IHWO) */
int so_ohwi_to_ihwo(const float* w, float* wt, int Ko, int taps, int C, void* stream);

/* weight gradient: dw[ko][r][s][c] (OHWI, dense). */
int so_conv2d_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int Nb, int H,
                    int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                    long long ws_bytes, void* stream);

/* same, accumulating: dw += ... (writes straight into the flat gradient slab the optimizer owns) */
int so_conv2d_wgrad_acc(const float* dy, int lddy, const float* x, int ldx, float* dw, int Nb, int H,
                        int W, int C, int Ko, int R, int S, int stride, int pad, float* ws,
                        long long ws_bytes, void* stream);

/* ---- Winograd F(2x2, 3x3) for 3x3 / stride 1 / padding 1 convolutions (csrc/wino.hip) -------------------------------
 * Replaces torch conv2d at models/networks/vgg.py:9-23 (frozen VGG19 chain of the perceptual loss, loss.py:106-122) and its
 * autograd input gradient: 16 multiplications per 2x2 output tile and channel pair instead of 36, all in fp32.
 * so_wino_weights: U[16][Ko][C] = G g G^T from OHWI weights w[Kw][3][3][C] (rows >= Kw zero); flip_transpose = 1 builds
 *   U'[16][C][Ko] from the flipped taps - the input gradient is then the same convolution with C and Ko swapped.
 * so_wino_conv3x3: y = gate(act(conv(x) + bias)); x [Nb*H*W][C] (ldx), y [Nb*H*W][Ko] (ldy), C % 4 == Ko % 4 == 0;
 *   `gate` (optional, pitch ldy): y = gate > 0 ? y : 0 (ReLU backward fused, as so_conv2d_dgrad_t_gated);
 *   wino_ws: so_wino_ws_floats(...) floats (V[16][tiles][C] and M[16][tiles][Ko]); ws: split-K scratch of the GEMM. */
long long so_wino_ws_floats(int Nb, int H, int W, int C, int Ko);
int so_wino_weights(const float* w, float* U, int Ko, int Kw, int C, int flip_transpose, void* stream);
int so_wino_conv3x3(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                    int ldy, int Nb, int H, int W, int C, int Ko, int act, float act_param, float* wino_ws,
                    long long wino_ws_bytes, float* ws, long long ws_bytes, void* stream);

/* F(4x4, 3x3), same three-kernel structure with 36 transform points per 4x4 output tile (2.25 multiplications per output
 * instead of 4, and 2.25x instead of 4x the activation size in transformed operands): the deep VGG19 layers.  Same
 * argument meaning as so_wino_weights / so_wino_conv3x3; U is [36][Ko][C] (or [36][C][Ko] with flip_transpose = 1). */
long long so_wino4_ws_floats(int Nb, int H, int W, int C, int Ko);
int so_wino4_weights(const float* w, float* U, int Ko, int Kw, int C, int flip_transpose, void* stream);
int so_wino4_conv3x3(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                     int ldy, int Nb, int H, int W, int C, int Ko, int act, float act_param, float* wino_ws,
                     long long wino_ws_bytes, float* ws, long long ws_bytes, void* stream);

/* Fused form: input transform, the 16 GEMMs and the output transform in ONE launch (transformed operands never touch HBM).
 * Weights in the kernel's own order U[ceil(K/8)][16][N][8] with (N, K) = (Ko, C) (flip_transpose = 0) or (C, Ko) built from
 * the flipped taps (flip_transpose = 1: the input gradient, called with x = dy, C <- Ko, Ko <- C).  Same reference call
 * sites and argument meaning as so_wino_conv3x3; no workspace. */
long long so_wino_fused_weight_floats(int Ko, int C, int flip_transpose);
int so_wino_fused_weights(const float* w, float* U, int Ko, int Kw, int C, int flip_transpose, void* stream);
int so_wino_fused_conv3x3(const float* x, int ldx, const float* U, const float* bias, int nbias, const float* gate, float* y,
                          int ldy, int Nb, int H, int W, int C, int Ko, int act, float act_param, void* stream);
/* The same convolution followed by nn.MaxPool2d(2, 2) (models/networks/vgg.py:14-25: conv1_2 -> pool, conv2_2 -> pool): an
 * F(2x2) output tile is one pooling window, so the epilogue also writes ypool [Nb][H/2][W/2][ldyp] (H, W even); y itself is
 * stored for the first n_keep images only (y may be NULL when n_keep = 0). */
int so_wino_fused_conv3x3_pool(const float* x, int ldx, const float* U, const float* bias, int nbias, float* y, int ldy,
                               int n_keep, float* ypool, int ldyp, int Nb, int H, int W, int C, int Ko, int act,
                               float act_param, void* stream);
/* -1 (default) = 64 output channels per block (two blocks per CU) when Ko >= 64 and that grid still has >= 1024 blocks, else 32
 * (three per CU); 1 = always 32; 0 = 64 whenever Ko >= 64 */
void so_wino_fused_force_kb32(int on);
/* block order of the fused kernel: -1 (default) = the ko blocks of a patch adjacent unless the Winograd-domain filters exceed
 * 8 MB, then all patches of a ko block adjacent (its filter slice stays in the XCD's L2); 0 / 1 force either (A/B measurements) */
void so_wino_fused_kb_major(int mode);
/* csrc/attn.hip: SAGAN self-attention core for n = H*W <= 256 positions (sagan.py:38-52) on qkv = [q | k | v] rows ([B n][E], the
 * output of ONE projection GEMM).  so_attn_supported: 1 when the fused kernels take the shape (C % 32 == 0, d % 4 == 0, LDS fits);
 * so_attn_fwd: energy -> softmax -> attention x V -> gamma * o + x in one launch (a: [B][n][n], o: [B n][C], both may be NULL);
 * so_attn_bwd: dqkv = [dq | dk | dv] from dout in two launches (ws: so_attn_ws_floats floats, 8-byte aligned);
 * so_attn_tail: db[0:E] (+)= column sums of dqkv and dgamma (+)= <dout, o> (fp64 partials left in ws by so_attn_bwd). */
int so_attn_supported(int B, int n, int C, int d);
long long so_attn_ws_floats(int B, int n);
int so_attn_fwd(const float* qkv, int E, int d, const float* x, int ldx, const float* gamma, float* a, float* o, float* out,
                int ldout, int B, int n, int C, void* stream);
int so_attn_bwd(const float* qkv, int E, int d, const float* a, const float* o, const float* dout, int lddout,
                const float* gamma, float* dqkv, float* ws, int B, int n, int C, void* stream);
int so_attn_tail(const float* dqkv, int E, const float* ws, int B, int n, float* db, int acc_db, float* dgamma, int acc_gamma,
                 void* stream);
/* csrc/pgemm.hip: the Winograd-domain batched products (C[b] = A[b] B[b]^T, K % 64 == 0, >= 16 matrices, >= 768 output tiles) run
 * on a PERSISTENT kernel that keeps its LDS-DMA pipeline running across output tiles; 0 = launched, SO_NOT_APPLICABLE = declined,
 * nothing launched (callers fall back to so_gemm_batched), anything else = an error (hipError_t / SO_ERR_*).  so_pgemm_enable(0) switches it off (A/B measurements). */
int so_pgemm_nt(int M, int N, int K, const float* A, int lda, long long sa, const float* B, int ldb, long long sb, float* C, int ldc,
                long long sc, int batch, void* stream);
void so_pgemm_enable(int on);
/* 1 (default): the fused kernel's two LDS stages are filled by LDS-DMA (buffer_load ... lds); 0: through registers + ds_write */
void so_wino_fused_dma(int on);

/* torch.bmm replacement (sagan.py:44,50; warp.py:63):
 * C[b] = act(alpha[0] * opA(A[b]) opB(B[b]) + bias[n] + res[b]),  alpha/bias/res optional (NULL).
 * transa 0: A [M][K]; 1: A [K][M].  transb 0: B [K][N]; 1: B [N][K].  (transa=1,transb=1 unsupported) */
int so_gemm_batched(int transa, int transb, int M, int N, int K, const float* A, int lda,
                    long long sa, const float* B, int ldb, long long sb, float* C, int ldc,
                    long long sc, int batch, const float* alpha, const float* bias,
                    const float* res, int ldres, long long sres, int act, float act_param,
                    float* ws, long long ws_bytes, void* stream);

/* test / tuning hooks: force the block tile (bm, bn in {64, 128}; bm = 0 -> auto) and the split-K factor
 * (0 = auto); set the per-FLOP cost factors of the four tile shapes used by the tile planner. */
void so_igemm_force(int bm, int bn, int splitk);
void so_igemm_force_waves(int nw); /* after so_igemm_force: 8 selects the 8-wave variant of a BN = 128 tile */
void so_igemm_tile_cost(float c64x64, float c128x64, float c64x128, float c128x128);

/* measured plans: when enabled, the first launch of every new problem shape (outside stream capture) times all
 * (tile, split-K) candidates with HIP events on the real operands and caches the fastest; later launches of the
 * shape (including captured ones) use the cached plan.  so_igemm_plan_count = number of cached shapes. */
void so_igemm_autotune(int on);
int so_igemm_plan_count(void);
/* persist / restore the measured plans (text file); return the number of plans, -1 on I/O error */
int so_igemm_plans_save(const char* path);
int so_igemm_plans_load(const char* path);

/* measurement hook (bench.py): when enabled, every MFMA launch is bracketed by HIP events on its own
 * stream.  so_prof_collect waits for them and fills HOST arrays of 40 entries, key = group*8 + tile
 * (group 0 fprop, 1 dgrad, 2 wgrad, 3 gemm, 4 Winograd-domain batched gemm (batch >= 16); tile 0 64x64, 1 128x64,
 * 2 64x128, 3 128x128, 4 128x128/8 waves, 5 64x128/8 waves, 6 the 4x4x1-MFMA kernels for four-channel convolutions,
 * 7 (group 0) the fused Winograd kernel): summed milliseconds, summed algorithmic FLOPs (2*M*N*K per launch; the fused
 * Winograd kernel: the direct convolution's), launch count.  Returns the number of launches collected and clears the list. */
void so_prof_enable(int on);
int so_prof_collect(float* out_ms, float* out_flops, int* out_count);
/* the same + out_bytes[k] (HOST array of 40 doubles; may be NULL): summed ALGORITHMIC HBM bytes of the launches under key k -
 * every operand read once and the result written once (convolutions: input-side tensor + output-side tensor + filter;
 * batched GEMM: A + B + C); bench.py's roofline.traffic / algorithmic_bytes ratio divides the PMC bytes by this */
int so_prof_collect_bytes(float* out_ms, float* out_flops, int* out_count, double* out_bytes);

/* ---- normalisation (csrc/norm.hip) ---------------------------------------------------------------- */

/* floats of scratch needed by so_norm_fwd / so_norm_bwd */
long long so_norm_ws_floats(int G, long long R, int C);

/* InstanceNorm2d (G = batch, R = H*W; unet.py:136,146) / BatchNorm2d training (G = 1, R = N*H*W;
 * warp.py:15,21,29,75-84): y = (x - mean) * rstd [* gamma + beta]; mean/rstd [G][C] are outputs kept for
 * backward; running_mean/var (optional) get the momentum update with the unbiased variance.  y = NULL: statistics
 * only (the caller normalises inside its own pass, so_spade_norm_fwd). */
int so_norm_fwd(const float* x, int ldx, float* y, int ldy, int G, long long R, int C, float eps,
                const float* gamma, const float* beta, float* mean, float* rstd,
                float* running_mean, float* running_var, float momentum, float* ws, void* stream);
/* so_norm_fwd with a second output y2 = act(y) from the same launch: the activation that the consumer applies first (the
 * U-Net's [norm] -> submodule [GELU -> conv], models/networks/cpvton/unet.py:132-147) while y itself stays for the skip path */
int so_norm_act_fwd(const float* x, int ldx, float* y, int ldy, float* y2, int ldy2, int act, float act_param, int G,
                    long long R, int C, float eps, const float* gamma, const float* beta, float* mean, float* rstd,
                    float* running_mean, float* running_var, float momentum, float* ws, void* stream);

/* BatchNorm2d in eval mode (test_step): statistics are inputs; stat_is_var = 1 -> `stat` is a variance. */
int so_norm_apply(const float* x, int ldx, float* y, int ldy, int G, long long R, int C,
                  const float* mean, const float* stat, int stat_is_var, float eps,
                  const float* gamma, const float* beta, void* stream);

/* dgamma/dbeta (BatchNorm affine, optional): overwritten, or added to when accumulate != 0 (gradient slabs).
 * relu_gate != 0: x is the output of a ReLU (FeatureExtraction's Conv -> ReLU -> BatchNorm, warp.py:13-31) and dx is
 * the gradient in front of that ReLU (zero where x == 0), saving the separate mask pass. */
int so_norm_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int G,
                long long R, int C, const float* mean, const float* rstd, const float* gamma,
                float* dgamma, float* dbeta, int accumulate, int relu_gate, float* ws, void* stream);

/* so_norm_bwd that also leaves dbias[c] (+)= sum over the rows of dx[r][c] (G == 1): the bias gradient of the convolution in
 * front of a Conv -> ReLU -> BatchNorm group (warp.py:15-31), taken from the backward statistics pass (two launches less). */
int so_norm_bwd_bias(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int G,
                     long long R, int C, const float* mean, const float* rstd, const float* gamma,
                     float* dgamma, float* dbeta, int accumulate, int relu_gate, float* dbias, int accumulate_bias,
                     float* ws, void* stream);

/* ---- pointwise / resampling / reductions (csrc/elementwise.hip) ---------------------------------- */

/* LeakyReLU / ReLU / GELU / Swish / Sine / tanh / sigmoid (unet.py:132-135,201-211) */
int so_act_fwd(const float* x, int ldx, float* y, int ldy, long long rows, int C, int act,
               float param, void* stream);
int so_act_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx,
               long long rows, int C, int act, float param, void* stream);
/* dx = res + dy * act'(x): gradient of a U-Net block input, which feeds the skip concatenation (gradient res) and the block's
 * down activation (gradient dy of its output) - unet.py:187-198; replaces act_bwd + autograd's accumulation add, same bits. */
int so_act_bwd_add(const float* x, int ldx, const float* dy, int lddy, const float* res, int ldres, float* dx, int lddx,
                   long long rows, int C, int act, float param, void* stream);

/* torch.cat(dim=1) / channel slicing / zero channel padding (unet.py:198; unet_mask_model.py:69):
 * dst[row][0:Cd] = src[row][0:Cs] (zero beyond Cs); accumulate != 0: dst += src. */
int so_copy2d(const float* src, int lds_, int Cs, float* dst, int ldd, int Cd, long long rows,
              int accumulate, void* stream);

/* boundary layout changes: planar NCHW (the reference's batch tensors) <-> NHWC rows */
int so_nchw_to_nhwc(const float* src, float* dst, int ldd, int Nb, int C, int Cd, int HW,
                    void* stream); /* channels C..Cd-1 of dst are zero-filled */
int so_nhwc_to_nchw(const float* src, int lds_, float* dst, int Nb, int C, int HW, void* stream);

/* nn.Upsample(scale_factor=2, mode="bilinear") (unet.py:138,155,166), align_corners=False */
int so_upsample2x_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C,
                      void* stream);
int so_upsample2x_bwd(const float* dy, int lddy, float* dx, int lddx, int Nb, int H, int W, int C,
                      void* stream);
/* the up path's `act -> Upsample` pair (unet.py:137-138,154-155,165-166) in one pass: y = upsample(act(x));
 * backward dx = act'(x) * upsample^T(dy) (x = the activation's input) */
int so_upsample2x_act_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C, int act,
                          float act_param, void* stream);
int so_upsample2x_act_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int Nb, int H,
                          int W, int C, int act, float act_param, void* stream);
/* the same pair applied to the channel concatenation [x1 | x2] of a U-Net skip connection (unet.py:198 feeding
 * unet.py:137-138 of the enclosing block) without materialising torch.cat: y = upsample(act(cat(x1, x2)));
 * backward writes the two input gradients separately.  x2 / dx2 == NULL: single source. */
int so_upsample2x_cat_fwd(const float* x1, int ldx1, int C1, const float* x2, int ldx2, int C2, float* y, int ldy,
                          int Nb, int H, int W, int act, float act_param, void* stream);
int so_upsample2x_cat_bwd(const float* x1, int ldx1, int C1, const float* x2, int ldx2, int C2, const float* dy,
                          int lddy, float* dx1, int lddx1, float* dx2, int lddx2, int Nb, int H, int W, int act,
                          float act_param, void* stream);

/* MaxPool2d(2, 2) of VGG19 (vgg.py:9-23) */
int so_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C,
                    void* stream);
/* relu_gate != 0: x is a ReLU output; the gradient is also chained through that ReLU (zero where the window max is 0) */
int so_maxpool2_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int Nb,
                    int H, int W, int C, int relu_gate, void* stream);

/* bias gradient: out[c] (+)= sum_rows x[row][c] */
long long so_colsum_ws_floats(long long rows, int C);
int so_colsum(const float* x, int ldx, long long rows, int C, float* out, int accumulate, float* ws,
              void* stream);

/* F.l1_loss / nn.L1Loss (warp_model.py:88; unet_mask_model.py:174-184; loss.py:110,121):
 * out[0] (+)= scale * sum |a - b| (scale = weight / numel); ws >= 1024 floats.
 * backward: da (+)= sign(a - b) * gout[0] * scale; relu_gate != 0: `a` is a ReLU output and da is the gradient in
 * front of that ReLU (zero where a == 0). */
int so_l1_loss_fwd(const float* a, int lda, const float* b, int ldb, long long rows, int C,
                   float scale, float* out, int accumulate, float* ws, void* stream);
int so_l1_loss_bwd(const float* a, int lda, const float* b, int ldb, const float* gout, float scale,
                   float* da, int ldda, long long rows, int C, int accumulate, int relu_gate, void* stream);
/* out[0] = ((a[0] + b[0]) + c[0]) + d[0] (c, d may be NULL): the sum of the loss terms (unet_mask_model.py:190) */
int so_scalar_sum(const float* a, const float* b, const float* c, const float* d, float* out, void* stream);

/* tanh / sigmoid / mask blend of UnetMaskModel.forward (unet_mask_model.py:84-86,126-129), one frame:
 * o: [pix][>=4] network output; cloth: [pix][>=3]; outputs rendered [pix][3], mask [pix][1],
 * tryon [pix][tryon_pad] (channels >= 3 zero-filled so it can feed the VGG conv directly). */
int so_tryon_compose_fwd(const float* o, int ldo, const float* cloth, int ldc, float* rendered,
                         int ldr, float* mask, int ldm, float* tryon, int ldt, int tryon_pad,
                         long long pix, void* stream);
int so_tryon_compose_bwd(const float* rendered, int ldr, const float* mask, int ldm,
                         const float* cloth, int ldc, const float* d_tryon, int lddt,
                         const float* d_rendered, int lddr, const float* d_mask, int lddm, float* d_o,
                         int lddo, long long pix, void* stream);

/* torch.optim.Adam step on a flat parameter slab (base_model.py:165-168); g is scaled by grad_scale. */
int so_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float b1,
                 float b2, float eps, int step, float grad_scale, void* stream);

/* out = alpha[0] * a + b with a device-resident alpha (SelfAttention: gamma * out + x, sagan.py:53) */
int so_scale_add(const float* a, int lda, const float* alpha, const float* b, int ldb, float* out,
                 int ldo, long long rows, int C, void* stream);

/* out = (1 - m) * a + m * b, m a one-channel mask (multi-frame blends, unet_mask_model.py:118-129);
 * backward: da, db, dm are optional (NULL to skip). */
int so_blend_fwd(const float* a, int lda, const float* b, int ldb, const float* m, int ldm, float* out,
                 int ldo, long long rows, int C, void* stream);
int so_blend_bwd(const float* a, int lda, const float* b, int ldb, const float* m, int ldm,
                 const float* g, int ldg, float* da, int ldda, float* db, int lddb, float* dm, int lddm,
                 long long rows, int C, void* stream);

int so_fill(float* p, long long n, float val, void* stream);
int so_axpby(const float* x, float a, float* y, float b, long long n, void* stream);

/* ---- stream hand-off inside a replayed hipGraph (trainer.BucketedExchange; reference: torch DDP's bucketed all-reduce
 * overlapped with backward, train.py:76-85).  so_signal_alloc: one 8-byte signal word (hipMallocSignalMemory; the one
 * allocation this library makes - signal memory has no torch equivalent), returned as an integer device address, 0 on failure;
 * so_counter_bump: counter[0] += 1 (first node of the captured step); so_signal_store: flag[0] = counter[0] with a release at
 * system scope (system_scope = 1, for the mode-0 waiter) or agent scope ("everything in front of me in this stream is done"); so_stream_wait_ge: host call, makes `stream` wait until
 * flag[0] >= value (mode 0: hipStreamWaitValue32, no compute unit occupied; mode 1: a one-lane polling kernel with s_sleep,
 * which keeps the command processor out of it); so_signal_can_wait: 1 if the device supports mode 0. */
long long so_signal_alloc(void);
int so_signal_free(long long ptr);
int so_signal_can_wait(void);
int so_counter_bump(void* counter, void* stream);
int so_signal_store(void* flag, const void* counter, int system_scope, void* stream);
int so_stream_wait_ge(void* flag, int value, int mode, void* stream);
/* The polling wait with an escape (no unbounded device spin): `words` = so_hostwords_alloc() - two zeroed 32-bit words of
 * pinned, device-mapped host memory, valid on host and device: words[0] = abort request (the host stores 1 on any exception
 * path; no stream needed), words[1] = status written by the waiter (1 = aborted, 2 = `max_ticks` ticks of the 100 MHz wall
 * clock passed).  On either the waiter RETURNS - the stream and every peer inside the collective queued behind it move on -
 * and the host raises when it next looks at words[1] (BucketedExchange.finish). */
int so_stream_wait_ge_bounded(void* flag, int value, void* words, long long max_ticks, void* stream);
long long so_hostwords_alloc(void);
int so_hostwords_free(long long ptr);

/* ---- geometric matching + attention row kernels (csrc/gmm.hip) ------------------------------------ */

/* FeatureL2Norm (warp.py:39-50); inv[pix] = 1/sqrt(sum_c x^2 + 1e-6) kept for backward;
 * transpose_hw writes pixel (h, w) to row w*H + h (FeatureCorrelation's transpose of A, warp.py:60). */
int so_l2norm_fwd(const float* x, int ldx, float* y, int ldy, float* inv, int Nb, int H, int W, int C,
                  int transpose_hw, void* stream);
int so_l2norm_bwd(const float* y, int ldy, const float* dy, int lddy, const float* inv, float* dx,
                  int lddx, int Nb, int H, int W, int C, int transpose_hw, void* stream);

/* nn.Softmax(dim=-1) on the attention energies (sagan.py:45) */
int so_softmax_rows_fwd(const float* e, int lde, float* a, int lda, long long rows, int ncol,
                        void* stream);
int so_softmax_rows_bwd(const float* a, int lda, const float* da, int ldda, float* de, int ldde,
                        long long rows, int ncol, void* stream);

/* out[0] (+)= scale * sum a*b (gradient of the attention gamma, sagan.py:53); ws >= 1024 floats */
int so_dot(const float* a, int lda, const float* b, int ldb, long long rows, int C, float scale,
           float* out, int accumulate, float* ws, void* stream);

/* FeatureRegression.linear + tanh (warp.py:87,96-98): x NHWC [Nb][P][C] flattened in the reference's
 * (C, H, W) order, w [J][C*P], y [Nb][J]. */
int so_linear_chw_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int Nb,
                      int P, int C, int J, int apply_tanh, void* stream);
int so_linear_chw_bwd(const float* x, int ldx, const float* w, const float* y, const float* dy,
                      float* dx, int lddx, float* dw, float* dbias, int Nb, int P, int C, int J,
                      int apply_tanh, void* stream);

/* TpsGridGen.forward (warp.py:159-167,191-318): theta [Nb][2*NP] -> grid [Nb][H][W][2] (x, y).
 * Li [(NP+3)^2] inverse TPS system, px/py [NP] control points, gx [W] / gy [H] the regular grid. */
long long so_tps_ws_floats(int Nb, int H, int W, int NP);
int so_tps_grid_fwd(const float* theta, const float* Li, const float* px, const float* py,
                    const float* gx, const float* gy, float* grid, int Nb, int H, int W, int NP,
                    float* ws, void* stream);
int so_tps_grid_bwd(const float* dgrid, const float* Li, const float* px, const float* py,
                    const float* gx, const float* gy, float* dtheta, int Nb, int H, int W, int NP,
                    float* ws, void* stream);

/* F.grid_sample(bilinear, align_corners=False; border = 1 -> padding_mode="border", 0 -> "zeros")
 * (warp_model.py:85-86,143-145).  in [Nb][C][H][W] planar, grid [Nb][Ho][Wo][2], out [Nb][C][Ho][Wo];
 * taps (optional int32 [Nb][Ho][Wo][2]) receives the north-west tap (x0, y0) of every output pixel. */
int so_grid_sample_fwd(const float* in, const float* grid, float* out, int* taps, int Nb, int C, int H,
                       int W, int Ho, int Wo, int border, void* stream);
int so_grid_sample_bwd(const float* in, const float* grid, const float* dout, float* dgrid, float* din,
                       int Nb, int C, int H, int W, int Ho, int Wo, int border, void* stream);

/* Resample2d of the flownet2 submodule (unet_mask_model.py:115-117): out = bilinear(in, pixel + flow) */
int so_resample2d_fwd(const float* in, const float* flow, float* out, int Nb, int C, int H, int W,
                      void* stream);
/* backward: din (optional, overwritten) is a scatter-add; it is accumulated in 64-bit fixed point (integer atomics are
 * associative), so the result is bit-identical from run to run.  ws: so_resample2d_bwd_ws_floats floats, 8-byte aligned. */
long long so_resample2d_bwd_ws_floats(int Nb, int C, int H, int W);
int so_resample2d_bwd(const float* in, const float* flow, const float* dout, float* din, float* dflow,
                      int Nb, int C, int H, int W, float* ws, void* stream);

/* ---- dataset-side tensor preparation + inter-stage image wire format (csrc/dataprep.hip) ---------------- */

/* visualization.py:73-77 (save_images): byte = trunc(clamp((x + 1) * 0.5 * 255, 0, 255)).  src is planar
 * [Nb][C][HW] (chw = 1) or NHWC rows with pixel pitch ld (chw = 0); dst is [Nb][HW][C] bytes (PIL's HWC). */
int so_quantize_u8(const float* src, int ld, int chw, void* dst, int Nb, int C, int HW, void* stream);

/* transforms.ToTensor + Normalize(0.5, 0.5) (tryon_dataset.py:109-118, used for every image the datasets open,
 * e.g. the warp-cloth PNGs read back by the try-on stage, vvt_dataset.py:139-150): src [Nb][HW][Cs] bytes,
 * dst planar [Nb][C][HW] = (byte / 255 - 0.5) / 0.5 for the first C of the Cs interleaved channels. */
int so_u8_to_normed(const void* src, int Cs, float* dst, int Nb, int C, int HW, void* stream);

/* convert_pose_data_to_pose_map_and_vis (tryon_dataset.py:388-448): keypoints [Nb][P][3] (x, y, confidence; fp64 as
 * json.load yields) -> pose_map [Nb][P][H][W] and the 1-channel visual im_pose [Nb][1][H][W], both in {-1, +1}; a
 * keypoint with x > 1 and y > 1 paints the inclusive square [(int)(x - r), (int)(x + r)] x [(int)(y - r), (int)(y + r)].
 * draw_into_map = 0 is the reference as written (its planes are converted BEFORE the square is drawn and stay -1, only
 * im_pose shows the squares); 1 paints the squares into the one-hot planes (CP-VTON's intent).  Either output may be NULL. */
int so_pose_map(const double* keypoints, float* pose_map, float* im_pose, int Nb, int P, int H, int W, int radius,
                int draw_into_map, void* stream);

/* get_person_head (tryon_dataset.py:323-345) and segment_cloths_from_image (datasets/util.py:6-22): with m = 1 where the
 * LIP label's bit is set in head_bits / cloth_bits, im_head = im * m - (1 - m), im_cloth = im * m + (1 - m);
 * parse [Nb][HW] bytes, image planar [Nb][3][HW]; outputs are planar with batch strides (elements) so they can land
 * inside a wider buffer (agnostic = [silhouette | im_head]); shape_u8 [Nb][HW] receives (label > 0) * 255, the input
 * of so_silhouette.  Any output may be NULL. */
int so_parse_compose(const void* parse, const float* image, float* im_head, long long stride_head, float* im_cloth,
                     long long stride_cloth, void* shape_u8, int head_bits, int cloth_bits, int Nb, int HW, void* stream);

/* get_person_body_silhouette (tryon_dataset.py:347-369): PIL resize BILINEAR to (W/factor, H/factor) and back (Pillow's
 * 8-bit separable resampling, 22-bit fixed point), then ToTensor + Normalize; bit-exact.  ws: so_silhouette_ws_bytes. */
long long so_silhouette_ws_bytes(int Nb, int H, int W, int factor);
int so_silhouette(const void* shape_u8, float* silhouette, long long stride_out, void* ws, int Nb, int H, int W, int factor,
                  void* stream);

/* get_input_cloth_mask (tryon_dataset.py:168-175): mask [Nb][1][HW] = (x[:, 0] >= threshold) ? 0 : 1 for planar x [Nb][C][HW].
 * NB the reference compares the NORMALISED cloth (values in [-1, 1]) with its default threshold 240, so its mask is all ones. */
int so_threshold_mask(const float* x, int C, float threshold, float* mask, int Nb, int HW, void* stream);

/* get_person_flow (tryon_dataset.py:272-298): .flo payload [Nb][HW][2] (u, v interleaved) -> planar [Nb][2][HW]
 * followed by transforms.Normalize((0.5, 0.5), (0.5, 0.5)). */
int so_flow_decode(const float* payload, float* flow, int Nb, int HW, void* stream);

/* ---- split-bf16 3x3 convolution for the frozen VGG19 chain (csrc/sb16.hip; opt-in, non-headline) ----------------- */
/* fp32 = hi + mid (two bf16 planes); a*b ~= hi*hi + hi*mid + mid*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * (vgg.py:6-36 conv3x3 + ReLU; weights are frozen there, vgg.py:25-27, so they are split once).
 * so_sb16_split: x [rows][ldx] fp32 -> hi / mid planes [rows][C] bf16.
 * so_sb16_prep_weights: OHWI fp32 [Ko][9][Cw] -> planes [Ko][9][C] (transpose = 0) or the input-gradient weights
 *   [C][9][Ko] with flipped taps (transpose = 1).
 * so_sb16_conv3x3: stride 1, pad 1, C % 32 == 0, Ko % 64 == 0; y fp32 [rows][ldy] = act(conv + bias), optional ReLU gate
 *   (gate [rows][Ko] > 0), optional output planes yh / ym [rows][Ko] for the next convolution. */
int so_sb16_split(const float* x, int ldx, int C, void* hi, void* mid, long long rows, void* stream);
int so_sb16_prep_weights(const float* w_ohwi, int Ko, int C, int Cw, int transpose, void* hi, void* mid, void* stream);
int so_sb16_conv3x3(const void* xh, const void* xm, const void* wh, const void* wm, const float* bias, const float* gate,
                    float* y, int ldy, void* yh, void* ym, int Nb, int H, int W, int C, int Ko, int relu, void* stream);

/* ---- SAMS-GAN (csrc/sams.hip) ------------------------------------------------------ */

/* F.interpolate(mode="nearest") of the SPADE label maps (models/networks/sams/spade.py:83) and nn.Upsample(scale_factor=2 |
 * 0.5) of the generator (models/networks/sams/sams_generator.py:294-308): y[ho][wo] = x[min(floor(ho * scale_h), Hi - 1)][..]
 * with ATen's fp32 rule; scale = Hi / Ho for `size=`, 1 / scale_factor for `scale_factor=`.  _bwd is the adjoint
 * (dx overwritten). */
int so_resize_nearest_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int Hi, int Wi, int Ho, int Wo, int C,
                          float scale_h, float scale_w, void* stream);
int so_resize_nearest_bwd(const float* dy, int lddy, float* dx, int lddx, int Nb, int Hi, int Wi, int Ho, int Wo, int C,
                          float scale_h, float scale_w, void* stream);

/* y = a + b: the residual connection of AnySpadeResBlock (models/networks/sams/spade.py:171) */
int so_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, long long rows, int C, void* stream);

/* SPADE modulation (models/networks/sams/spade.py:89) fused with the activation AnySpadeResBlock applies to it
 * (spade.py:168-169): y = act(nrm * (1 + gamma) + beta).  gamma / beta: [rows][C] each (two halves of one conv output
 * are fine: same ld, pointers C apart).  _bwd recomputes the pre-activation and writes dn, dgamma, dbeta. */
int so_spade_fwd(const float* nrm, int ldn, const float* gamma, int ldg, const float* beta, int ldb, float* y, int ldy,
                 long long rows, int C, int act, float act_param, void* stream);
int so_spade_bwd(const float* nrm, int ldn, const float* gamma, int ldg, const float* beta, int ldb, const float* dy, int lddy,
                 float* dn, int lddn, float* dgamma, int lddg, float* dbeta, int lddb, long long rows, int C, int act,
                 float act_param, float* colsum_part, void* stream);
/* The same two passes with the parameter-free normalisation folded in (spade.py:80 + :89): `x` is the un-normalised
 * activation, n = (x - mean[g][c]) * rstd[g][c], g = row / R (mean / rstd [rows / R][C] from so_norm_fwd(y = NULL)); the
 * normalised tensor is never materialised.  dn is the gradient with respect to n: feed it to so_norm_bwd. */
int so_spade_norm_fwd(const float* x, int ldx, const float* mean, const float* rstd, long long R, const float* gamma, int ldg,
                      const float* beta, int ldb, float* y, int ldy, long long rows, int C, int act, float act_param,
                      void* stream);
int so_spade_norm_bwd(const float* x, int ldx, const float* mean, const float* rstd, long long R, const float* gamma, int ldg,
                      const float* beta, int ldb, const float* dy, int lddy, float* dn, int lddn, float* dgamma, int lddg,
                      float* dbeta, int lddb, long long rows, int C, int act, float act_param, float* colsum_part,
                      void* stream);
/* colsum_part (optional): [so_spade_bwd_colsum_blocks(rows, C)][2C] per-block column sums of dgamma | dbeta - summed over
 * the blocks they are the bias gradient of the convolution that produced gamma | beta.  0 blocks: C not eligible. */
int so_spade_bwd_colsum_blocks(long long rows, int C);

/* F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False) between the scales of the multiscale discriminator
 * (models/networks/discriminator.py:51-54): Ho = (H - 1) / 2 + 1. */
int so_avgpool3s2_fwd(const float* x, int ldx, float* y, int ldy, int Nb, int H, int W, int C, void* stream);
int so_avgpool3s2_bwd(const float* dy, int lddy, float* dx, int lddx, int Nb, int H, int W, int C, void* stream);

/* torch.nn.utils.spectral_norm (applied at models/networks/sams/spade.py:149-153 and models/networks/normalization.py:
 * 24-25), dim 0, one power iteration: v <- normalize(W^T u), u <- normalize(W v), sigma = u^T W v, w_out = W / sigma.
 * w_orig / w_out: dense OHWI [O][RS][I]; u: [O]; v: [I * RS] in torch's (I, R, S) flattening (checkpoint layout).
 * power_iter = 0 (eval): u and v are only read.  sigma: 1 float, kept for the backward pass.
 * _bwd: dw = g / sigma - (<g, W> / sigma^2) u v^T with the u, v of that forward (the caller keeps copies). */
long long so_spectral_norm_ws_floats(int O, int I, int RS);
int so_spectral_norm_fwd(const float* w_orig, int O, int I, int RS, float* u, float* v, float* w_out, float* sigma,
                         int power_iter, float eps, float* ws, void* stream);
int so_spectral_norm_bwd(const float* g, const float* w_orig, const float* u, const float* v, const float* sigma, int O,
                         int I, int RS, float* dw, int accumulate, float* ws, void* stream);

/* GANLoss.loss (models/networks/loss.py:58-88) with labels 1 / 0: mean over all elements of
 * mode 0 "original": BCE with logits; 1 "ls": (x - t)^2; 2 "w": -x (real) / x (fake); 3 "hinge": -min(x - 1, 0) (real),
 * -min(-x - 1, 0) (fake) for the discriminator, -x for the generator.  out: 1 float; ws: 1024 floats.
 * _bwd: dx = gout[0] / count * f'(x). */
int so_gan_loss_fwd(const float* x, int ldx, long long rows, int C, int mode, int target_is_real, int for_discriminator,
                    float* out, float* ws, void* stream);
int so_gan_loss_bwd(const float* x, int ldx, long long rows, int C, int mode, int target_is_real, int for_discriminator,
                    const float* gout, float* dx, int lddx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SHINEON_HIP_H_ */
