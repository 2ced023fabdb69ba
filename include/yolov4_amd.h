/*
 * yolov4_amd.h -- C ABI of libyolov4_amd.so: the MI355X (gfx950) hot path of YOLOv4.
 *
 * The reference (zjykzj/YOLOv4) has no FFI of its own: its hot path is a set of Python
 * nn.Module classes whose arithmetic runs inside PyTorch ATen.  Each entry
 * point below therefore cites the reference Python interface (file:line under /root/reference)
 * whose arithmetic it replaces; the Python classes of the same names in yolov4_amd/ bind these
 * symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - activations are NHWC fp32 with an explicit pixel pitch `ld*` (elements between two
 *     consecutive pixels, >= channel count), so a channel slice of a wider buffer is a
 *     valid operand (zero-copy concat / split);
 *   - conv filters are KRSC fp32: w[Cout][k][k][Cin], i.e. the reference's
 *     nn.Conv2d.weight [Cout,Cin,k,k] held in torch.channels_last memory format;
 *   - `stream` is a hipStream_t passed as void*; work is only enqueued, no call synchronises,
 *     allocates or frees (workspaces are caller-provided; *_workspace() return their size);
 *   - return value: Y4_OK or a Y4_ERR_* code; y4_strerror() names it.  Shapes are validated on
 *     the host before any launch.
 */
#ifndef YOLOV4_AMD_H
#define YOLOV4_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y4_OK 0
#define Y4_ERR_SHAPE 1      /* unsupported / inconsistent shape or pitch            */
#define Y4_ERR_NULL 2       /* required pointer is NULL                             */
#define Y4_ERR_LAUNCH 3     /* hipLaunch / hipMemsetAsync reported an error         */
#define Y4_ERR_WORKSPACE 4  /* workspace too small                                  */
#define Y4_ERR_NODEVICE 5   /* no gfx950 device visible                             */

/* activation ids: darknet/darknet.py:41-51 ('relu' | 'leaky_relu' (0.1) | 'mish' | 'linear') */
#define Y4_ACT_LINEAR 0
#define Y4_ACT_LEAKY 1
#define Y4_ACT_MISH 2
#define Y4_ACT_RELU 3

const char* y4_strerror(int code);
int y4_version(void);
/* number of visible HIP devices whose arch is gfx950 (0 if none) */
int y4_device_count(void);

/* Arithmetic of the conv implicit GEMMs (process-wide):
 *   1  "bf16x3": every fp32 operand is split EXACTLY into three bf16 pieces (8+8+8
 *      mantissa bits) while its tile is staged into LDS; a product is the six leading terms
 *      a1b1+a1b2+a2b1+a1b3+a2b2+a3b1 on v_mfma_f32_32x32x16_bf16 (products exact, fp32 accumulate).
 *      Dropped terms <= 2^-23 |a*b|: measured error vs an fp64 convolution is at or below that of
 *      mode 0 (rms 1.06e-6 vs 1.19e-6 of the output range at K = 4608), at 6/16 of the MFMA cost.
 *   0  v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fp32 fma chain.
 *   2  plain bf16: operands rounded (RN) to bf16 while staged, one bf16 MFMA per product, fp32
 *      accumulate; activations / gradients / BN / loss / NMS stay fp32 (BASELINE config 5, mixed
 *      precision -- NOT fp32-grade: ~3 significant digits per product).
 *   3  "f16x2" (default): every fp32 operand is scaled by a power of two taken from its tensor's max|x| (so the
 *      maximum lands in [2^14, 2^15)) and split into two fp16 pieces hi = RN(sx), lo = RN((sx-hi) 2^11)
 *      (11 + 11 significant bits, |error| <= 2^-22 |x|); a product is THREE fp16 MFMAs
 *      (hi*hi -> acc0; hi*lo + lo*hi -> acc1; result (acc0 + 2^-11 acc1)/(s_a s_b)), fp32 accumulate.
 *      Per-product error ~2^-22, i.e. 2.4e-7 of the rms of a sum of any length: below the rounding of
 *      an fp32 fma chain and of the MFMA's own fp32 accumulation; half the MFMAs of mode 1.  Operand
 *      maxima (`*_amax` arguments: device words holding the bit pattern of max|finite element|,
 *      upper bounds are fine) come from the producing kernels (y4_bn_act_fwd_f32 / y4_bn_act_bwd_f32
 *      out_amax) or, when NULL, from one extra pass over the operand inside the call. */
int y4_set_conv_mode(int mode);
int y4_get_conv_mode(void);
/* BASELINE configs[4] ("bf16 MFMA conv, fp32 loss / NMS") as shipped: conv mode 3 with this switch on -- the layers that run on
 * the DMA-fed plane kernels (72 of the 107 BatchNorm layers, 9/10 of the conv flops) take plain bf16 operands written by the
 * BatchNorm sweeps and run one bf16 MFMA per product (y4_conv2d_*_planes_f32, y4_bn_act_fwd_f32 z_planes == 3); every other
 * layer keeps the fp32-grade f16x2 kernels of mode 3, which are HBM-bound anyway.  (Mode 2 rounds EVERY conv operand to bf16.) */
int y4_set_planes_bf16(int on);
int y4_get_planes_bf16(void);
/* 1 if EVERY plane kernel a conv layer of this geometry (x: [B][H][W][Cin]; k in {1,3}; stride 1, or 2 with k = 3 on an even
 * grid) would launch in the current mode can address its operands -- the DMA kernels use 32-bit buffer windows (an image span
 * of a 256-row tile, a wgrad block's pixel range, the whole dx of a stride-2 dgrad: each < 4 GiB) -- else 0.  dgrad_planes: the
 * layer's dgrad runs on the plane kernels too.  The host asks this before it lets a producer write its result pre-split: such
 * an operand has no fp32 form for the register-staged kernels to fall back to (yolov4_amd/darknet/darknet.py takes_planes()). */
int y4_conv_planes_fit(int B, int H, int W, int Cin, int Cout, int k, int stride, int dgrad_planes);
/* ---------------------------------------------------------------- convolution
 * Replaces nn.Conv2d inside ConvBNAct.forward, darknet/darknet.py:31-36,53-54
 * (k in {1,3}, stride in {1,2}, pad=(k-1)//2, dilation 1, groups 1).
 *
 * y[b,ho,wo,n] = act( (sum_{r,q,c} x[b,ho*s-pad+r,wo*s-pad+q,c] * w[n,r,q,c]) * scale[n] + shift[n] )
 *                + residual[b,ho,wo,n]
 * scale/shift/residual may be NULL (identity / 0).  Eval-mode BatchNorm folds into
 * scale/shift (darknet.py:55), the bias of the 3 head convs into shift (yolov4.py:237,243,249),
 * ResBlock's skip into residual (darknet.py:76-80).  Implicit GEMM on
 * v_mfma_f32_32x32x2_f32 (exact fp32).  Requires Cin % 32 == 0, or Cin == 3 (stem,
 * yolov4.py:30: direct kernel, x given with arbitrary element strides).
 * workspace: y4_conv2d_fwd_workspace() bytes of caller-owned device memory, 16-B aligned, private to the call in stream
 * order (modes 1-3: the filter's maximum and its pre-split planes; 6 B per filter element + 4160 B).  The library keeps
 * no state between calls: convs may be issued from any number of streams, devices and host threads at once.
 */
size_t y4_conv2d_fwd_workspace(int Cin, int Cout, int k);
int y4_conv2d_fwd_f32(const float* x, int ldx, const float* w, float* y, int ldy,
                      int B, int H, int W, int Cin, int Cout, int k, int stride,
                      const float* scale, const float* shift, int act,
                      const float* residual, int ldr, const unsigned* x_amax /* mode 3, nullable */,
                      unsigned* y_amax /* mode 3, nullable: max|finite y| folded in with atomicMax */,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Inference, conv mode 3: keep a KRSC filter [Cout][K = k*k*Cin] in the form the forward kernels consume -- a 64-B header
 * (word 0: bit pattern of max|w|; word 1: scratch of the conv call; words 2-5: 64-bit fingerprint, valid flag, ticket), 4 KiB of
 * fingerprint partials, then the fp16 hi/lo planes interleaved per 32-deep K tile; y4_conv2d_prepared_bytes(Cout, K) bytes in
 * all, whose first 64 bytes the caller ZEROES once after allocating.  y4_conv2d_prepare_filter_f32 is a REFRESH: it reads the
 * filter once (maximum + a 64-bit positional hash of its bit patterns; collision probability 2^-64) and re-splits it only if
 * either differs from what the buffer holds -- decided on the device, so a stale buffer cannot be used whatever wrote the weights (optimizer kernels,
 * `.data` writes, load_state_dict).  Call it before every y4_conv2d_fwd_prepared_f32: same arithmetic and results as
 * y4_conv2d_fwd_f32 at one read pass instead of two reads + one write pass over the filter.  (The reference has no
 * counterpart: nn.Conv2d re-reads its weight every call.) */
size_t y4_conv2d_prepared_bytes(int Cout, int K);
int y4_conv2d_prepare_filter_f32(const float* w, int Cout, int K, void* prepared, size_t prepared_bytes, void* stream);
int y4_conv2d_fwd_prepared_f32(const float* x, int ldx, void* w_prepared, float* y, int ldy,
                               int B, int H, int W, int Cin, int Cout, int k, int stride,
                               const float* scale, const float* shift, int act,
                               const float* residual, int ldr, const unsigned* x_amax, unsigned* y_amax, void* stream);
/* max |finite element| over the first C channels of an NHWC tensor (pitch ldx), as a bit pattern (mode 3 operand
 * maximum).  y4_amax_f32 overwrites *amax_bits; y4_amax_merge_u32 folds *src into *dst (a concat buffer's maximum is
 * the maximum of its parts).  Integer atomicMax: order independent. */
int y4_amax_f32(const float* x, int ldx, long long M, int C, unsigned* amax_bits, void* stream);
int y4_amax_merge_u32(unsigned* dst, const unsigned* src, void* stream);

/* Measurement aid (bench.py): copies the symbol of the conv kernel the calling host thread launched last, spelled as
 * rocprofv3 --stats prints it, into buf (NUL-terminated, at most cap bytes) and clears it; empty when the last launch
 * was not an f16x2-mode kernel.  No reference counterpart. */
int y4_last_conv_kernel(char* buf, int cap);

/* Training-mode variant: raw conv output + BatchNorm batch statistics fused into the epilogue.
 * partials receives one row [2][Cout] (column sums, sums of squares) per M-tile; *nparts_host
 * (HOST int64) is set to the number of rows written; feed both to y4_bn_finalize_partials_f32.
 * partial_bytes >= y4_conv2d_bnstats_workspace(). */
size_t y4_conv2d_bnstats_workspace(int B, int H, int W, int Cin, int Cout, int k, int stride);
int y4_conv2d_fwd_bnstats_f32(const float* x, int ldx, const float* w, float* y, int ldy,
                              int B, int H, int W, int Cin, int Cout, int k, int stride,
                              float* partials, size_t partial_bytes, long long* nparts_host,
                              const unsigned* x_amax /* mode 3, nullable */,
                              void* workspace, size_t workspace_bytes /* as y4_conv2d_fwd_f32 */,
                              void* dgrad_filter /* nullable; mode 3: also receives the transposed filter planes, in the layout of
                                 y4_conv2d_dgrad_f32's workspace -- that call then takes this buffer as its workspace with
                                 w == NULL and launches no filter kernels (one split launch serves forward and backward) */,
                              size_t dgrad_filter_bytes /* >= y4_conv2d_dgrad_workspace(Cin, Cout, k) */, void* stream);

/* ---- conv mode 3 over PRE-SPLIT activations ("planes", csrc/conv_planes.hip): the input arrives as the two fp16 pieces
 * of the f16x2 split, per pixel and 32-channel K tile [64 B: 32 hi halfs | 64 B: 32 scaled-lo halfs] (4 bytes per element,
 * pitch 4 Cin bytes, Cin % 32 == 0), scaled by the power of two that *x_amax (an upper bound of max|x|) implies; that is
 * the form the BatchNorm sweeps can emit directly, and the form LDS-DMA can stage without touching the VALU.
 * y4_planes_split_f32 converts an fp32 NHWC tensor (tests, tensors produced by kernels that do not emit planes).
 * y4_conv2d_fwd_planes_f32 = y4_conv2d_fwd_bnstats_f32 on such an input (raw output + per-M-tile column sums, 256- or 128-row tiles;
 * partials may be NULL); workspace as y4_conv2d_fwd_f32.  Replaces the same nn.Conv2d, darknet/darknet.py:31-36,53-54. */
int y4_planes_split_f32(const float* x, int ldx, long long M, int C, const unsigned* amax, void* planes, void* stream);
/* The same into a CHANNEL SLICE of a wider pre-split tensor (a concat buffer whose other slices their producers write pre-split,
 * y4_bn_planes_bound_f32): `planes` points at the slice's first tile in pixel 0's row (128-B aligned; bf16 mode: byte 2 c of the
 * row for channel offset c), ld_planes = channels per pixel row of the whole tensor; *amax = the tensor's joint scale word, which
 * must dominate max|x|.  c_valid (0: C): source channels >= c_valid are PAD of x's rows (ldx >= C) and leave as zeros -- the
 * 255-channel head gradient as 256 pre-split channels.  Replaces the copy half of torch.cat, yolo/model/yolov4.py:176,183 (PANBlock). */
int y4_planes_split_into_f32(const float* x, int ldx, long long M, int C, const unsigned* amax, void* planes, int ld_planes, int c_valid,
                             void* stream);
int y4_conv2d_fwd_planes_f32(const void* x_planes, const float* w, float* y, int ldy,
                             int B, int H, int W, int Cin, int Cout, int k, int stride,
                             float* partials, size_t partial_bytes, long long* nparts_host, const unsigned* x_amax,
                             void* workspace, size_t workspace_bytes,
                             void* dgrad_filter /* nullable: as in y4_conv2d_fwd_bnstats_f32 -- stride 1: for
                                y4_conv2d_dgrad_planes_f32 (mirrored taps); stride 2: for y4_conv2d_dgrad_f32 (the register-staged
                                parity-class dgrad, taps as they are) */,
                             size_t dgrad_filter_bytes,
                             int y_bf16 /* != 0 (Cout % 32 == 0): y leaves as plain bf16 (RN) in the FIRST HALF of each fp32-sized row
                                (pitch 4 ldy bytes), the column sums are those of the rounded values; the BatchNorm sweeps read it
                                with y4_bn_act_fwd_f32 z_planes + 16 / y4_bn_act_bwd_f32 frozen_stats bit 2 */,
                             const float* bias /* nullable [Cout]: added to every result row (a conv WITHOUT BatchNorm -- the head's
                                output convs, yolo/model/yolov4.py:235-239 -- reading a pre-split input; partials NULL, y_bf16 0) */,
                             void* stream);

/* dgrad / wgrad (stride 1, and the 3x3 stride-2 layers on even maps) of a conv over planes (dy, and for wgrad also
 * x, pre-split as above; Cin % 32 == 0, Cout % 32 == 0): the same arithmetic as y4_conv2d_dgrad_f32 / y4_conv2d_wgrad_f32 in
 * conv mode 3 (conv mode 2: plain bf16 operands, 64-channel multiples, stride 1).  dgrad runs the forward DMA kernel on the
 * mirrored transposed filter (workspace: y4_conv2d_dgrad_workspace()); wgrad stages both operands pixel-major and takes its
 * fragments through the hardware transpose read (split-K slabs in y4_conv2d_wgrad_planes_workspace() bytes, fixed-order
 * reduce: deterministic); H, W are the INPUT dims.  wgrad alone also takes a Cout that is NOT whole K tiles (the head's 255): dy's
 * pixel rows then hold ceil(Cout / 32) tiles (64-channel units in the bf16 mode) whose pad channels are zero, dW has Cout rows.
 * Autograd of the same nn.Conv2d, darknet/darknet.py:31-36. */
int y4_conv2d_dgrad_planes_f32(const void* dy_planes, const float* w /* NULL: workspace prepared by the forward call */, float* dx, int lddx,
                               int B, int H, int W, int Cin, int Cout, int k,
                               int stride /* 1; or 2 (3x3, H and W even, conv mode 3 operands, no residual): four launches, one per
                                  parity class of dx over its 1 / 2 / 2 / 4 taps; the workspace then holds the UN-mirrored
                                  transposed planes (what the forward call prepares for a stride-2 layer) */,
                               void* workspace, size_t workspace_bytes, const unsigned* dy_amax,
                               const float* residual, int ldr, void* stream);
size_t y4_conv2d_wgrad_planes_workspace(int B, int H, int W, int Cin, int Cout, int k, int stride);
int y4_conv2d_wgrad_planes_f32(const void* x_planes, const void* dy_planes, float* dw,
                               int B, int H, int W, int Cin, int Cout, int k, int stride,
                               void* workspace, size_t workspace_bytes, const unsigned* x_amax, const unsigned* dy_amax,
                               void* stream);

/* Any other ConvBNAct shape (darknet/darknet.py:25-36 takes any in_ch / out_ch / odd kernel_size / stride): direct kernels,
 * one thread per output element, plain fp32 fma chains -- written for correctness, YOLOv4 never builds such a layer.
 * Same NHWC / KRSC conventions and epilogue (scale, shift, act, residual) as y4_conv2d_fwd_f32; pad = (k - 1) / 2.
 * y4_conv2d_stem_dgrad_f32: gradient wrt the 3-channel network input (element strides as y4_conv2d_stem_fwd_f32). */
int y4_conv2d_generic_fwd_f32(const float* x, int ldx, const float* w, float* y, int ldy,
                              int B, int H, int W, int Cin, int Cout, int k, int stride,
                              const float* scale, const float* shift, int act, const float* residual, int ldr, void* stream);
int y4_conv2d_generic_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                int B, int H, int W, int Cin, int Cout, int k, int stride,
                                const float* residual, int ldr, void* stream);
int y4_conv2d_generic_wgrad_f32(const float* x, int ldx, const float* dy, int lddy, float* dw,
                                int B, int H, int W, int Cin, int Cout, int k, int stride, void* stream);
int y4_conv2d_stem_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, long long sxb, long long sxc, long long sxh,
                             long long sxw, int B, int H, int W, int Cout, void* stream);

/* Stem conv (Cin = 3): x addressed as x[b*sxb + c*sxc + h*sxh + w*sxw] so both the NCHW
 * tensor the reference feeds (yolo/engine/build.py:60) and NHWC work without a copy.
 * bnstats_partials (nullable): [ceil(B*H*W/256)][2][Cout] column sums of the output, valid when
 * scale/shift are NULL and act is linear (training-mode BatchNorm statistics). */
int y4_conv2d_stem_fwd_f32(const float* x, long long sxb, long long sxc, long long sxh, long long sxw,
                           const float* w, float* y, int ldy, int B, int H, int W, int Cout,
                           const float* scale, const float* shift, int act,
                           float* bnstats_partials, void* stream);

/* dgrad: dx[B,H,W,Cin] = conv_transpose(dy[B,Ho,Wo,Cout], w) (+ residual) -- autograd of nn.Conv2d wrt input.
 * workspace: y4_conv2d_dgrad_workspace() bytes (holds the [Cin][k][k][Cout4] transposed filter).
 * residual (nullable, pitch ldr >= Cin): added in the epilogue -- the gradient arriving over a ResBlock's skip
 * connection (darknet/darknet.py:76-80: x + f(x)), so the fan-in add costs no extra pass. */
size_t y4_conv2d_dgrad_workspace(int Cin, int Cout, int k);
int y4_conv2d_dgrad_f32(const float* dy, int lddy, const float* w /* mode 3: NULL = workspace prepared by the forward call */, float* dx, int lddx,
                        int B, int H, int W, int Cin, int Cout, int k, int stride,
                        void* workspace, size_t workspace_bytes, const unsigned* dy_amax /* mode 3, nullable */,
                        const float* residual, int ldr, void* stream);

/* wgrad: dw[Cout][k][k][Cin] = sum_{b,ho,wo} dy (x) x -- autograd of nn.Conv2d wrt weight.
 * Split-K over pixels into fp32 slabs in `workspace`, reduced in a fixed order (deterministic). */
size_t y4_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int Cout, int k, int stride);
int y4_conv2d_wgrad_f32(const float* x, int ldx, const float* dy, int lddy, float* dw,
                        int B, int H, int W, int Cin, int Cout, int k, int stride,
                        void* workspace, size_t workspace_bytes,
                        const unsigned* x_amax, const unsigned* dy_amax /* mode 3, nullable */, void* stream);
size_t y4_conv2d_stem_wgrad_workspace(int B, int H, int W, int Cout);
int y4_conv2d_stem_wgrad_f32(const float* x, long long sxb, long long sxc, long long sxh, long long sxw,
                             const float* dy, int lddy, float* dw, int B, int H, int W, int Cout,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- BatchNorm + activation
 * Replaces nn.BatchNorm2d (training mode: batch statistics, eps 1e-5, momentum 0.1, biased
 * variance for normalisation, unbiased for running_var) + the activation,
 * darknet/darknet.py:37-51,55-56.  M = B*H*W pixels, C channels.
 *
 * y4_bn_stats_f32: mean[c], invstd[c] = 1/sqrt(var_biased+eps); running stats updated in place
 *   (running = (1-m)*running + m*batch; running_var uses var*M/(M-1)); *num_batches_tracked += 1.
 *   running_mean/running_var/num_batches_tracked may be NULL.  workspace: y4_bn_workspace(M, C)
 *   bytes (fp64 accumulators + per-block fp32 partial rows; no contended atomics).
 * y4_bn_finalize_partials_f32: same outputs from the per-M-tile column sums [nparts][2][C] that
 *   y4_conv2d_fwd_bnstats_f32 / y4_conv2d_stem_fwd_f32 leave behind (statistics fused into the conv
 *   epilogue: the conv output is not read again).  workspace: y4_bn_finalize_workspace(C) bytes.
 * All reductions are two-stage and fixed-order (fp32 partial rows -> <= 64 fp64 rows -> one), i.e. deterministic.
 */
size_t y4_bn_workspace(long long M, int C);
size_t y4_bn_finalize_workspace(int C);
int y4_bn_stats_f32(const float* y, int ldy, long long M, int C, float* mean, float* invstd,
                    float* running_mean, float* running_var, long long* num_batches_tracked,
                    float momentum, float eps, void* workspace, size_t workspace_bytes, void* stream);
int y4_bn_finalize_partials_f32(const float* partials, long long nparts, long long M, int C,
                                float* mean, float* invstd, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float momentum, float eps,
                                void* workspace, size_t workspace_bytes, void* stream);
/* The analytic bound of max|act(BatchNorm(y))| that y4_bn_act_fwd_f32 (z_planes == 2) derives by itself, as a call of its own:
 * *out = max(bound(gamma, beta, M), floor_word ? *floor_word : 0).  For SEVERAL BatchNorm layers that write channel slices of one
 * pre-split tensor (the concat in front of a CSP transition conv, darknet/darknet.py:154-163 in the reference): chain the calls
 * through floor_word, then hand the joint word to every producer's y4_bn_act_fwd_f32 with z_planes == 1, z = its slice and
 * ldz = the pitch of the whole tensor -- one scale for the whole operand, no measuring pass. */
int y4_bn_planes_bound_f32(const float* gamma, const float* beta, int C, long long M, const unsigned* floor_word, unsigned* out,
                           void* stream);
/* z = act(gamma*(y-mean)*invstd + beta) + residual   (residual may be NULL).
 * out_amax (nullable, device word): max|finite z| is folded into it with atomicMax (the caller zeroes it, or it
 * already holds the maximum of other parts of the same concat buffer) -- the operand maximum of conv mode 3.
 * z == NULL: measure only (max|z| into *out_amax, nothing stored).
 * z_planes != 0: z (C % 32 == 0; ldz == C, or z = a channel slice of a wider pre-split tensor: ldz % 32 == 0 and z 128-B aligned --
 * for z_planes == 3 the slice of channels [c, c + C) starts at byte 2 c of the row) receives the tensor PRE-SPLIT for the plane conv kernels -- per pixel and
 * 32-channel K tile [64 B: hi halfs | 64 B: scaled lo halfs], 4 bytes per element -- scaled by the power of two that
 * *out_amax implies.  z_planes == 1: *out_amax must already hold max|z| or an upper bound (e.g. a measure-only call on the
 * same arguments).  z_planes == 2: the call derives the bound itself, without a pass over the data -- |xhat| <= sqrt(M - 1)
 * for any sample of M values, |act(v)| <= |v|, so |z| <= max_c(|gamma_c| sqrt(M - 1) + |beta_c|) + max|residual|
 * (res_amax: the residual's maximum word, required with a residual) -- and leaves it in *out_amax for the consumers.  A
 * loose bound costs the split only headroom (full precision down to 2^-29 of the bound).
 * z_planes + 16: y itself holds plain bf16 values in the first half of each row (y4_conv2d_fwd_planes_f32 y_bf16).
 * z_planes == 3 (conv mode 2, bf16 MFMA conv): z receives plain bf16 values (RN), dense per pixel in the FIRST HALF of the
 * fp32-sized row (ldz == C: the row pitch stays 4 C bytes, the second half is not touched); out_amax is not used.
 * planes_twin (nullable, with z_planes != 0): z stays fp32 (pitch ldz) and planes_twin [M][C] receives the pre-split copy --
 * for a tensor that feeds both a plane-consuming conv and fp32 consumers (a fork in front of a detection head). */
int y4_bn_act_fwd_f32(const float* y, int ldy, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act,
                      const float* residual, int ldr, float* z, int ldz,
                      long long M, int C, unsigned* out_amax, int z_planes, const unsigned* res_amax, float* planes_twin,
                      void* stream);
/* Backward of the two ops above wrt y, gamma, beta given dz (grad wrt z; the residual branch
 * receives dz itself).  dy may alias dz.  workspace: y4_bn_workspace(M, C) bytes. */
int y4_bn_act_bwd_f32(const float* dz, int lddz, const float* y, int ldy,
                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                      int act, float* dy, int lddy, float* dgamma, float* dbeta,
                      long long M, int C, void* workspace, size_t workspace_bytes,
                      unsigned* out_amax /* nullable: max|finite dy|, as above */,
                      unsigned* f16_planes /* nullable, 8 zeroed device words, conv mode 3: dy is written PRE-SPLIT for the
                         plane conv kernels (layout as y4_bn_act_fwd_f32 z_planes; lddy == C, C % 32 == 0), scaled by a
                         bound of max|dy| derived before the sweep; word [5] receives that bound and serves as dy_amax
                         of y4_conv2d_dgrad_planes_f32 / y4_conv2d_wgrad_planes_f32 */,
                      float* planes_twin /* nullable, with f16_planes: dy stays fp32 (pitch lddy) and planes_twin [M][C]
                         receives the pre-split copy -- a layer whose wgrad runs on the plane kernel and whose dgrad does
                         not (3x3 stride 2) */,
                      int frozen_stats /* bit 0: mean / invstd are constants (eval-mode BatchNorm under autograd: running
                         statistics), so dy = gamma invstd g without the two batch-statistic terms; dgamma / dbeta as usual.
                         bit 2: y holds plain bf16 values in the first half of each row (as y4_bn_act_fwd_f32 z_planes + 16).
                         bit 1 (conv mode 2): dy leaves as plain bf16 in the first half of each fp32-sized row, as
                         y4_bn_act_fwd_f32 z_planes == 3 (lddy == C, C % 32 == 0, f16_planes NULL) */,
                      void* stream);
/* dbias[c] = sum_m dy[m,c]  (bias=True head convs, yolov4.py:237,243,249): two fixed-order stages, no atomics
 * (deterministic).  workspace: y4_bias_grad_workspace(M, C) bytes */
size_t y4_bias_grad_workspace(long long M, int C);
int y4_bias_grad_f32(const float* dy, int lddy, long long M, int C, float* dbias,
                     void* workspace, size_t workspace_bytes, void* stream);
/* eval-mode fold (darknet.py:55 in eval): scale = gamma/sqrt(running_var+eps),
 * shift = beta - running_mean*scale */
int y4_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, float eps, float* scale, float* shift, int C, void* stream);

/* ---------------------------------------------------------------- pointwise glue
 * torch.cat along channels (darknet.py:110,135; yolov4.py:71,139,146,181,186): copy src[M,C]
 * into dst[:, c_off:c_off+C]. */
int y4_copy_channels_f32(const float* src, int lds, float* dst, int ldd, long long M, int C, void* stream);
/* out[M,C] = a[M,C] + b[M,C] (gradient fan-in at forks: residual skip, CSP split, FPN/PAN taps);
 * out may alias a or b. */
int y4_add_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo,
               long long M, int C, void* stream);
/* nn.MaxPool2d(ksize, stride 1, pad ksize//2), yolov4.py:60-62,68-70.  idx (int8, per output
 * element: window offset dy*ksize+dx of the first maximum) may be NULL in inference. */
int y4_maxpool_s1_fwd_f32(const float* x, int ldx, float* y, int ldy, signed char* idx,
                          int B, int H, int W, int C, int ksize, void* stream);
int y4_maxpool_s1_bwd_f32(const float* dy, int lddy, const signed char* idx, float* dx, int lddx,
                          int accumulate, int B, int H, int W, int C, int ksize, void* stream);
/* nearest x2 upsample, yolov4.py:77-90, and its adjoint (2x2 block sum). */
int y4_upsample2x_fwd_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream);
int y4_upsample2x_bwd_f32(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------- YOLO head decode
 * Replaces YOLOLayer.forward, yolo/model/yololayer.py:88-166.  logits: head conv output
 * NHWC [B,F,F,>=A*(5+C)] pitch ldl; anchors_wh: A pairs (w,h) in grid units (host array).
 * train: output [B,A,F,F,5+C] (sigmoid on ch 0,1,4..; raw 2,3) and pred [B,A,F,F,4] (grid units).
 * eval : out[b, box_off + a*F*F + j*F + i, :] of an [B,n_total,5+C] buffer, boxes x stride. */
int y4_yolo_decode_train_f32(const float* logits, int ldl, float* output, float* pred,
                             int B, int F, int A, int n_classes, const float* anchors_wh_host, void* stream);
int y4_yolo_decode_eval_f32(const float* logits, int ldl, float* out, long long n_total, long long box_off,
                            int B, int F, int A, int n_classes, const float* anchors_wh_host,
                            float stride, void* stream);
/* d(logits) (NHWC, pitch ldl, pad channels zeroed) from d(output) and d(pred) (either NULL). */
int y4_yolo_decode_bwd_f32(const float* logits, int ldl, const float* g_output, const float* g_pred,
                           float* g_logits, int B, int F, int A, int n_classes,
                           const float* anchors_wh_host, void* stream);

/* ---------------------------------------------------------------- detection loss
 * Replaces YOLOLoss.build_target + YOLOLoss.forward for one layer,
 * yolo/model/yololoss.py:118-371,385-432.  labels: [B,K,5] fp32 rows (xc,yc,w,h,cls) in input
 * pixels.  all_anchors_host: 9 (w,h) pairs in grid units of this layer; anch_mask_host: the A
 * indices of this layer.  Outputs (all device):
 *   obj_mask [B,A,F,F] fp32, pos_index [B,A,F,F] int32 (-1 or index into pos_rec),
 *   pos_rec [B,K,8+ceil(C/32)] (x,y,w,h targets, scale, cell, class bitmask), npos [B],
 *   loss_parts[4] doubles (xy, wh, obj, cls), deterministic two-stage reduction.
 * workspace: y4_yolo_loss_workspace() bytes.
 */
size_t y4_yolo_loss_workspace(int B, int F, int A, int K, int n_classes);
int y4_yolo_loss_fwd_f32(const float* output, const float* pred, const float* labels, int K,
                         int B, int F, int A, int n_classes, float stride, float ignore_thresh,
                         const float* all_anchors_host, int n_all_anchors, const int* anch_mask_host,
                         float* obj_mask, double* loss_parts,
                         void* workspace, size_t workspace_bytes, void* stream);
/* grad wrt `output` (dense [B,A,F,F,5+C]) scaled by *gscale (DEVICE scalar: the upstream grad of
 * the loss, so no host sync is needed); uses the workspace filled by fwd.  output_is_masked = 1
 * when y4_yolo_loss_mask_output_f32 has already been applied to `output` (the reference's
 * in-place side effect): the w/h channels then hold o*scale. */
int y4_yolo_loss_bwd_f32(const float* output, int output_is_masked, const float* obj_mask,
                         const float* gscale, float* g_output,
                         int B, int F, int A, int K, int n_classes,
                         const void* workspace, size_t workspace_bytes, void* stream);
/* the reference's in-place side effect on outputs[*]['output'] (yololoss.py:402-407) */
int y4_yolo_loss_mask_output_f32(float* output, const float* obj_mask, int B, int F, int A, int K,
                                 int n_classes, const void* workspace, size_t workspace_bytes, void* stream);
/* dense views of the sparse targets for API parity / tests (yololoss.py:173-187 tensors) */
int y4_yolo_loss_dense_targets_f32(float* target, float* tgt_mask, float* tgt_scale,
                                   int B, int F, int A, int K, int n_classes,
                                   const void* workspace, size_t workspace_bytes, void* stream);

/* Device-side prefix sums of postprocess (no host round trip between its stages):
 * y4_post_scan_i32: seg_offsets[0..n_segments] = exclusive scan of the candidate counts, clamped to `cap` (the capacity, in
 *   candidates, of the buffers the caller allocated: a too-small capacity truncates segments, it never overflows);
 *   info[0] = true total, info[1] = 1 if it exceeds cap (the caller then repeats the call sequence with cap >= info[0]).
 * y4_post_compact_f32: out_offsets = exclusive scan of `kept`; rows of every segment copied to out_rows back to back
 *   (class ascending, score descending inside an image, utils.py:200-221); img_offsets[b] = first output row of image b,
 *   img_offsets[B] = number of detections.  One device->host copy of img_offsets (+ info) is the call's only sync. */
int y4_post_scan_i32(const int* counts, int n_segments, long long cap, int* seg_offsets, int* info, void* stream);
int y4_post_compact_f32(const float* det_rows, const int* seg_offsets, const int* kept, int B, int n_classes,
                        float* out_rows, int* out_offsets, int* img_offsets, void* stream);
/* ---------------------------------------------------------------- post-processing
 * Replaces postprocess + nms, yolo/util/utils.py:32-89,92-223.
 * Stage 1: xywh->xyxy in place on prediction [B,N,5+C] and count candidates
 *          (obj*cls >= conf) per (image, class) into counts[B*C] (int32, zeroed inside).
 * Stage 2: (host gives exclusive offsets of counts) fill candidate keys, sort inside each
 *          (image, class) segment by (score desc, box index asc), greedy NMS (IoU >= thr
 *          suppresses), write kept rows [x1,y1,x2,y2,obj,cls_conf,cls] compacted per segment
 *          and kept[B*C] counts.
 */
int y4_post_count_f32(float* prediction, int B, long long N, int n_classes, float conf_thre,
                      int convert_xyxy, int* counts, void* stream);
size_t y4_post_nms_workspace(long long total_candidates, int n_segments);
int y4_post_nms_f32(const float* prediction, int B, long long N, int n_classes, float conf_thre,
                    float nms_thre, const int* seg_offsets /* [B*C+1] device */, long long total_candidates,
                    float* det_rows /* [total,7] */, int* kept /* [B*C] */,
                    void* workspace, size_t workspace_bytes, void* stream);
/* plain greedy NMS on one list (nms(), utils.py:32): boxes [R,4] xyxy, order given by
 * score desc / index asc; writes keep flags in sorted order and the sorted index list. */
size_t y4_nms_workspace(long long R);
int y4_nms_f32(const float* boxes, const float* scores /* may be NULL */, long long R, float thresh,
               int limit, int* keep_idx /* [R] */, int* n_keep /* [1] */,
               void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- stand-alone forms of fused pieces
 * bboxes_iou (yolo/model/yololoss.py:16-91): pairwise IoU [Na,4] x [Nb,4] -> iou[Na*Nb] row-major; xyxy != 0:
 * corner boxes, else (xc, yc, w, h).  Same fp32 operation sequence as the reference (bit-exact on the CPU fixtures);
 * the target-assignment / ignore-mask kernels carry their own inlined copies. */
int y4_bboxes_iou_f32(const float* boxes_a, long long Na, const float* boxes_b, long long Nb, int xyxy, float* iou,
                      void* stream);
/* Mish.forward (darknet/darknet.py:14-20) and the other activations of ConvBNAct as a flat elementwise sweep over a
 * dense tensor of n floats (act = Y4_ACT_*); backward: dx = dy * act'(x). */
int y4_act_fwd_f32(const float* x, float* y, long long n, int act, void* stream);
int y4_act_bwd_f32(const float* x, const float* dy, float* dx, long long n, int act, void* stream);
/* Upsample.forward (yolo/model/yolov4.py:82-90) for any target size, NHWC with pitches.  integer_factor = 0: the
 * train branch, F.interpolate(size=(Ho,Wo), mode='nearest'): src = min(floor(dst * (float)in/out), in-1);
 * integer_factor = 1: the eval branch (view/expand by Ho/H, Wo/W; Y4_ERR_SHAPE unless Ho % H == Wo % W == 0).
 * Backward sums each source pixel's destination block in a fixed order. */
int y4_upsample_nearest_fwd_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int Ho, int Wo, int C,
                                int integer_factor, void* stream);
int y4_upsample_nearest_bwd_f32(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int Ho, int Wo, int C,
                                int integer_factor, void* stream);

/* ---------------------------------------------------------------- optimizer ("next" row 1)
 * Fused Adam step over one dense parameter block; replaces torch.optim.Adam as the reference builds
 * it (yolo/optim/optimizers/adam.py:14-15: betas (0.9, 0.999), eps 1e-8; both param groups of
 * build.py:18-35 have weight_decay 0).  grad is multiplied by grad_scale first (1/accumulation). */
int y4_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     float grad_scale, void* stream);
/* Multi-tensor form: ONE launch steps every parameter of the model.  chunks_dev is a device table of 48-byte
 * records {float* p; const float* g; float* m; float* v; int64 n; int64 hyper_row} (Adam: m = exp_avg,
 * v = exp_avg_sq; SGD: m = momentum_buffer, v unused); a chunk is a contiguous run of one tensor (the host cuts
 * tensors into runs of 64 Ki elements).  hyper_host holds nhyper (<= 16) rows of 4 floats, one per distinct
 * (param group, step count): Adam {lr/bc1, 1/sqrt(bc2), weight_decay, 0} as y4_adam_hyper_f32 computes them in
 * double (the same values y4_adam_step_f32 uses, so both forms are bit-identical); SGD {lr, momentum,
 * weight_decay, first_step(0/1)} with torch.optim.SGD semantics (sgd.py:14-15: dampening 0, no nesterov). */
int y4_adam_hyper_f32(float lr, float beta1, float beta2, float weight_decay, int step, float* out4_host);
int y4_adam_multi_step_f32(const void* chunks_dev, int nchunks, const float* hyper_host, int nhyper,
                           float beta1, float beta2, float eps, float grad_scale, void* stream);
int y4_sgd_multi_step_f32(const void* chunks_dev, int nchunks, const float* hyper_host, int nhyper,
                          float grad_scale, void* stream);

/* ---------------------------------------------------------------- eval input pipeline ("next" row 4)
 * One image: src is uint8 HWC (3 channels, row pitch in bytes), as cv2.imread hands it to the reference.
 * Resizes to SxS with the arithmetic of cv2.resize INTER_LINEAR on 8-bit data (yolo/data/transform.py:173-174),
 * swaps B and R when swap_rb (transform.py:437), divides by 255 in fp32 and writes channel-planar
 * (transform.py:461): dst[c*dst_sc + y*dst_sh + x*dst_sw], strides in elements (NCHW or NHWC slot of a batch). */
int y4_preprocess_u8_f32(const void* src, int src_h, int src_w, long long src_pitch_bytes, int swap_rb,
                         float* dst, long long dst_sc, long long dst_sh, long long dst_sw, int S, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLOV4_AMD_H */
