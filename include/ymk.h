/*
 * libymk — C-ABI boundary of the MI355X-native YOLO-Master detection forward pass.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + a
 * HIP stream (passed as void*), returns 0 on success or a negative YMK_E_*
 * code, never throws, never allocates, never synchronises the stream.
 * Workspaces are sized by the matching *_workspace_bytes() query and are
 * owned by the caller.  Device-side error conditions (non-finite router
 * input, candidate overflow) are reported through caller-provided int32
 * `flags`/`status` words that the host checks once per batch.
 *
 * The reference (Tencent/YOLO-Master) has no native op on this path; each
 * function below states the Python reference interface (file:line under the
 * reference checkout) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes stub a maintainer would add on the reference side.
 *
 * Layout convention: activations are NHWC ("pixel-major"): element (b,y,x,c)
 * of a tensor view lives at base[((b*H + y)*W + x) * ld + c]; `ld` (the pixel
 * stride, in elements) may exceed C so that a view can be a channel slice of
 * a wider concat buffer.  Weights are pre-packed by ymk_pack_* helpers on the
 * host side (python: yolo_master_amd/engine.py) as [Cout][Kpad] with
 * K = (ky, kx, cin) and BatchNorm already folded (fp32 fold, then cast).
 */
#ifndef YMK_H_
#define YMK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YMK_ABI_VERSION 4

/* error codes */
#define YMK_OK 0
#define YMK_E_BADARG (-1)    /* unsupported shape / alignment / dtype      */
#define YMK_E_LAUNCH (-2)    /* hipLaunchKernel reported an error           */
#define YMK_E_WORKSPACE (-3) /* workspace too small                         */

/* dtypes of activations / packed weights */
#define YMK_F32 0
#define YMK_H16 1   /* the library's 16-bit element type: bfloat16 in libymk.so, IEEE binary16 in libymk_f16.so — the SAME sources and
                     * the SAME ABI compiled with -DYMK_H16_F16 (v_mfma_f32_16x16x32_f16, RNE float<->half conversions); the reference's
                     * reduced-precision mode is fp16 (`half=True`: engine/predictor.py:174,415, nn/backends/pytorch.py:67).
                     * ymk_h16_format() tells which one a loaded library is. */
#define YMK_BF16 YMK_H16
#define YMK_H16_FORMAT_BF16 1
#define YMK_H16_FORMAT_F16 2

/* activation codes for fused epilogues */
#define YMK_ACT_NONE 0
#define YMK_ACT_SILU 1
#define YMK_ACT_SIGMOID 2 /* ymk_conv2d (fused in the LDS-DMA core, one in-place pass after the other cores), ymk_activation, ymk_group_norm */
#define YMK_ACT_GELU 3    /* exact erf form (torch.nn.GELU default); same scope */

/* device flag bits (int32 words written with atomicOr by kernels) */
#define YMK_FLAG_NONFINITE_INPUT 1  /* router input contains NaN/Inf  (routers.py:51)  */
#define YMK_FLAG_NONFINITE_LOGITS 2 /* router logits contain NaN/Inf  (routers.py:467) */
#define YMK_FLAG_NMS_OVERFLOW 4     /* reserved (ABI 2 raised it above 2*max_nms candidates; since round 3 any count is selected on device) */

int ymk_abi_version(void);
/* human readable build string (arch, compiler) — static storage */
const char* ymk_build_info(void);
int ymk_h16_format(void);   /* YMK_H16_FORMAT_BF16 | YMK_H16_FORMAT_F16 */

/* ------------------------------------------------------------------------
 * Convolution (implicit GEMM on MFMA) + folded-BN bias + SiLU + residual.
 * Replaces Conv.forward_fuse = act(conv2d(x, W', b'))
 *   ultralytics/nn/modules/conv.py:80-89, BN fold utils/torch_utils.py:315-349,
 * and the residual add of Bottleneck.forward (nn/modules/block.py:484-486),
 * ABlock.forward (block.py:1787-1797).
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t dtype;     /* YMK_F32 | YMK_BF16: type of x, w, residual                 */
    int32_t out_dtype; /* type of y (YMK_F32 allowed with bf16 compute)              */
    int32_t B, H, W;   /* input batch / height / width                               */
    int32_t Cin, Cout; /* Cin % 8 == 0 (bf16) or % 4 == 0 (f32); Cout % 4 == 0       */
    int32_t ksize;     /* 1 or 3 (square), padding = ksize/2 (conv.py:30-36 autopad) */
    int32_t stride;    /* 1 or 2                                                     */
    int32_t ldx, ldy, ldr; /* pixel strides (elements) of x, y, residual             */
    int32_t Kpad;      /* packed row length of w, multiple of 64, >= ksize^2*Cin     */
    int32_t act;       /* YMK_ACT_*                                                  */
} ymk_conv_desc;

int ymk_conv2d(const ymk_conv_desc* d, const void* x, const void* w, const float* bias,
               const void* residual /* may be NULL */, void* y, void* stream);

/* Diagnostic: which kernel family the calling thread's most recent ymk_conv2d dispatched to (same arithmetic,
 * different data movement; profilers and bench.py's roofline leg use it to attribute time). */
#define YMK_CONV_TILED       0  /* tiled implicit GEMM (any shape)                                   */
#define YMK_CONV_STREAM_1X1  1  /* weight-stationary persistent streaming 1x1 (large M, Kpad <= 256) */
#define YMK_CONV_SPATIAL_3X3 2  /* spatial-tile 3x3 with LDS-staged im2col (stride 1, Cin 16/32/64)  */
#define YMK_CONV_GLDS        3  /* LDS-DMA tiled core (ymk_next.h): bf16 3x3 with Cin >= 64; every shape with YMK_ENABLE bit 0;
                                * LDS stages in bits 8-15 and pixel-tile height in bits 16+ of the returned value              */
int32_t ymk_conv2d_last_variant(void);

/* 1x1 convolution over the channel concatenation [x1 | x2] without materialising it; with upsample1 != 0 the
 * first source is a [B][H/2][W/2][C1] map read through a nearest 2x upsample.  Replaces nn.Upsample + Concat
 * (ultralytics/nn/modules/conv.py:629-641) + the consumer's 1x1 Conv (C2f.cv1, block.py:318) of the neck.
 * d describes the virtual input (Cin = C1 + C2, ksize 1, stride 1); x1/x2 have pixel strides ldx1/ldx2. */
int ymk_conv1x1_cat2(const ymk_conv_desc* d, const void* x1, int32_t C1, int32_t ldx1, int32_t upsample1,
                     const void* x2, int32_t ldx2, const void* w, const float* bias, void* y, void* stream);

/* Stem convolution reading the NCHW fp32 network input directly (Cin <= 4) and
 * writing NHWC.  Replaces layer 0 `Conv(3, c, 3, 2)` (cfg yolo-master-*.yaml,
 * conv.py:80-89) together with the NCHW->NHWC layout change.
 * w: fp32 [Cout][ksize*ksize*Cin] (ky,kx,cin); wt_kco: the same weights transposed to
 * [ksize*ksize*Cin][Cout] (enables the pixel-per-thread kernel for Cout in {16,32,64}; may be
 * NULL); bias fp32 [Cout]. */
int ymk_conv2d_stem_nchw(const float* x_nchw, const float* w, const float* wt_kco, const float* bias, void* y,
                         int32_t out_dtype, int32_t B, int32_t Cin, int32_t H, int32_t W,
                         int32_t Cout, int32_t ksize, int32_t stride, int32_t ldy, int32_t act,
                         void* stream);

/* Stem + the convolution after it as ONE kernel (bf16 activations): layer 0 `Conv(3, C0, 3, 2)` + SiLU and layer 1
 * `Conv(C0, C1, 3, 2)` + SiLU of the YOLO-Master YAMLs (conv.py:80-89, walked by nn/tasks.py:182-218), for (C0, C1) = (32, 64)
 * (the S width).  The stem's output — the largest tensor of the network — stays in LDS (csrc/stem2.hip): same arithmetic as
 * ymk_conv2d_stem_nchw (fp32 matrix cores, bf16 rounding of the stem map) followed by ymk_conv2d.
 * x fp32 [B][3][H][W] (16-byte aligned, W % 4 == 0); wt0 = the stem's wt_kco [27][C0] fp32, b0 fp32 [C0];
 * w1 bf16 [C1][k1pad] packed as for ymk_conv2d (K = (ky, kx, c)), b1 fp32 [C1]; y bf16 [B][H2][W2][ldy]. */
int ymk_stem_pair_supported(int32_t dtype, int32_t Cin, int32_t C0, int32_t C1, int32_t k0, int32_t s0, int32_t k1, int32_t s1);
int ymk_stem_pair(const float* x_nchw, int32_t B, int32_t H, int32_t W, const float* wt0, const float* b0, int32_t C0,
                  const void* w1, int32_t k1pad, const float* b1, int32_t C1, void* y, int32_t ldy, void* stream);

/* C3k2 block with one plain Bottleneck as ONE kernel (bf16; c1 = 64, c2 = 128, hidden c = 32: YOLO-Master-S row 2 on the 160 x 160
 * map): y = cv2([a | b | b + m.cv2(m.cv1(b))]) with [a | b] = cv1(x), every Conv = convolution + folded BN + SiLU
 * (C2f.forward / C3k2, nn/modules/block.py:293-325, 1074-1111; Bottleneck :462-486).  The three intermediates stay in LDS
 * (csrc/c3k2f.hip).  Weights packed as for ymk_conv2d: w1 [64][k1pad] (1x1), wa [16][kapad] (3x3, 32 -> 16), wb [32][kbpad]
 * (3x3, 16 -> 32), w2 [128][k2pad] (1x1 over 96), fp32 biases.  x [B][H][W][ldx] (64 channels), y [B][H][W][ldy] (128). */
int ymk_c3k2_fused_supported(int32_t dtype, int32_t c1, int32_t c2, int32_t c, int32_t n, int32_t c3k, int32_t shortcut);
int ymk_c3k2_fused(const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, const void* w1, int32_t k1pad, const float* b1,
                   const void* wa, int32_t kapad, const float* ba, const void* wb, int32_t kbpad, const float* bb, const void* w2,
                   int32_t k2pad, const float* b2, void* y, int32_t ldy, void* stream);
/* ... and, for a consumer that starts with a global average pool of y (the ES-MoE router of the next YAML row,
 * moe/routers.py:458-527): gap_part fp32 [B][ymk_c3k2_fused_pool_chunks(H, W)][128] receives the per-tile channel sums of the
 * stored (bf16) values, in a fixed order — feed it to ymk_esmoe_route_pooled instead of re-reading y.  flags (may be NULL):
 * YMK_FLAG_NONFINITE_INPUT when a sum is not finite. */
int32_t ymk_c3k2_fused_pool_chunks(int32_t H, int32_t W);
int ymk_c3k2_fused_pooled(const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, const void* w1, int32_t k1pad, const float* b1,
                          const void* wa, int32_t kapad, const float* ba, const void* wb, int32_t kbpad, const float* bb, const void* w2,
                          int32_t k2pad, const float* b2, void* y, int32_t ldy, float* gap_part, int32_t* flags, void* stream);

/* Streaming 1x1 convolution that ALSO leaves the per-tile channel sums of its output (round 5): the layer that produces an ES-MoE
 * layer's input (a C3k2's last 1x1: block.py:293-351 -> moe/routers.py:458-527) hands the router its global average pool.  Arguments as
 * ymk_conv2d + pool_part fp32 [B][ymk_conv1x1_pool_chunks(d)][Cout]: sums over the 128-pixel tiles of each image of the values AS STORED
 * (rounded to the 16-bit type), in a fixed order — feed it to ymk_esmoe_route_pooled.  ymk_conv1x1_pool_chunks returns 0 for a shape the
 * pooled kernel does not take (then ymk_conv1x1_pooled returns YMK_E_BADARG): 16-bit in and out, 1x1 stride 1, SiLU or no activation,
 * Cout a multiple of 64 and > 64, H * W a multiple of 128, Kpad <= 256, at least the streaming kernel's minimum of pixel tiles. */
int32_t ymk_conv1x1_pool_chunks(const ymk_conv_desc* d);
int ymk_conv1x1_pooled(const ymk_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y,
                       float* pool_part, void* stream);
/* The same sums, in the same order (bit-identical), from a map y [B][HW][ldy] that is already in memory — for producers the pooled kernel
 * does not take (a small batch has too few tiles for the streaming kernel): a router's decision must not depend on which kernel wrote
 * its input.  HW a multiple of 128; pool_part fp32 [B][HW / 128][C]. */
int ymk_pool_tiles128(int32_t dtype, const void* y, int32_t ldy, int32_t B, int32_t HW, int32_t C, float* pool_part, void* stream);

/* Detect class branch of one pyramid level as ONE kernel (bf16 in, fp32 logits out): DWConv3x3 -> Conv1x1 -> DWConv3x3 -> Conv1x1 ->
 * Conv2d 1x1 (+bias) (head.py:111-118, non-legacy `cv3[i]`), c3 = 128, cin = 128 or 256.  The four intermediate maps stay in LDS
 * (csrc/detcls.hip).  dw1 [9][cin] / dw2 [9][128] packed as for ymk_dwconv2d, pw1 [128][k1pad] / pw2 [128][k2pad] / w3 [ncpad][k3pad]
 * as for ymk_conv2d, fp32 biases (BN folded); ncpad = nc rounded up to a multiple of 4 (padded rows zero); y fp32 [B][H][W][ldy]. */
int ymk_detect_cls_fused_supported(int32_t dtype, int32_t cin, int32_t c3, int32_t nc);
int ymk_detect_cls_fused(const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, int32_t cin, const void* dw1, const float* bd1,
                         const void* pw1, int32_t k1pad, const float* bp1, const void* dw2, const float* bd2, const void* pw2,
                         int32_t k2pad, const float* bp2, const void* w3, int32_t k3pad, const float* b3, int32_t ncpad, float* y,
                         int32_t ldy, float* y_out, int32_t nc, int32_t a_off, int32_t A_total, float* best_conf,
                         int32_t* best_cls, void* stream);
/* Fused decode (round 4; head.py:157-171 `cls.sigmoid()`): y_out != NULL -> sigmoid(logits) of the nc classes goes straight to rows
 * 4.. of y_out fp32 [B][4 + nc][A_total] at anchors a_off + oy * W + ox (ymk_detect_decode's values bit for bit), best_conf / best_cls
 * (both or neither, [B][A_total]) as ymk_detect_decode's.  y may then be NULL: the fp32 logits are not materialised.
 *
 * The box branch's tail: Conv2d(64, 4 * reg_max, 1) + bias -> DFL -> dist2bbox -> rows 0..3 of y (head.py:111-112,173-194;
 * csrc/elementwise.hip), reg_max = 16, 16-bit x [B][Hl][Wl][ldx] with 64 channels, w packed [64][kpad]; raw: NULL or fp32
 * [B][Hl][Wl][64] box logits.  With both, ymk_detect_decode and the 0.6 KB per anchor of fp32 logits it re-read are gone. */
int ymk_detect_box_tail_supported(int32_t dtype, int32_t cin, int32_t reg_max, int32_t nc);   /* the pair: box tail + class kernel with y_out (nc <= 96) */
int ymk_detect_box_tail(int32_t dtype, const void* x, int32_t ldx, int32_t B, int32_t Hl, int32_t Wl, const void* w, int32_t kpad,
                        const float* bias, int32_t reg_max, int32_t nc, float stride, int32_t a_off, int32_t A_total, float* y,
                        float* raw, void* stream);

/* ------------------------------------------------------------------------
 * Depthwise k x k convolution (stride 1, pad k/2, k odd <= 15) + bias + act
 * + residual.  Replaces DWConv (conv.py:185-199; Detect cv3 head.py:111-118),
 * AAttn.pe (block.py:1688,1731) and the depthwise stage of
 * DepthwiseSeparableConv (moe/experts.py:283-292).
 * w: packed [k*k][C] in `dtype`; bias fp32 [C] or NULL.
 * ------------------------------------------------------------------------ */
int ymk_dwconv2d(int32_t dtype, const void* x, const void* w, const float* bias,
                 const void* residual, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                 int32_t ksize, int32_t ldx, int32_t ldy, int32_t ldr, int32_t act, void* stream);

/* ------------------------------------------------------------------------
 * ES-MoE (ultralytics/nn/modules/moe/modules.py:410-704)
 * ------------------------------------------------------------------------ */

/* Router: global-average-pool -> 1x1 (C->hidden) -> SiLU -> 1x1 (hidden->E) ->
 * softmax(fp32, logits clamped to +-30) -> top-k -> stable_normalize
 * (DynamicRoutingLayer.forward routers.py:458-496, _hard_top_k :519-527,
 *  stable_normalize _numeric.py:85-90), followed by the sparse-dispatch decision
 * of ES_MOE._sparse_forward (modules.py:665-684): retained = rank0 | (w >=
 * dynamic_threshold), renormalise over the retained set, and the
 * image->expert CSR permutation built with wave ballots / prefix scans.
 *
 * Outputs (all device memory):
 *   route_w   fp32 [B][E]   routing weights before pruning (hard top-k, sums to 1)
 *   gate_w    fp32 [B][E]   retained & renormalised weights (0 where pruned)
 *   sel       int32 [B][top_k] expert id per slot in ascending expert order, -1 = unused
 *   csr_off   int32 [E+1]   expert -> range in csr_pair
 *   csr_pair  int32 [B*top_k] packed (b*top_k + slot), grouped by expert, b ascending
 *   state     fp32 [E+1]    (nullable) the eval-time buffers ES_MOE keeps (modules.py:706-741, moe/loss.py:16-26):
 *                           state[e] = expert_usage_counts[e] = mean over the batch of route_w[:, e],
 *                           state[E] = load_balancing_loss = E * sum_e (u_e / max(sum u, 1e-6))^2
 *   flags     int32 [1]     YMK_FLAG_NONFINITE_* bits (atomicOr)
 * dynamic_threshold > 0: prune the non-leading top-k experts below it; == 0: keep the whole top-k set (renormalised);
 * < 0: the DENSE forward (use_sparse_inference=False, modules.py:648-656) — gate_w = route_w, nothing renormalised.
 * workspace: ymk_esmoe_route_workspace_bytes(B, C, H, W) (partial pooling sums).
 */
size_t ymk_esmoe_route_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W);
int ymk_esmoe_route(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                    int32_t ldx, const float* w1 /*[hidden][C]*/, const float* b1,
                    const float* w2 /*[E][hidden]*/, const float* b2, int32_t hidden, int32_t E,
                    int32_t top_k, float dynamic_threshold, float* route_w, float* gate_w,
                    int32_t* sel, int32_t* csr_off, int32_t* csr_pair, float* state /*[E+1], nullable*/,
                    int32_t* flags, void* workspace, size_t workspace_bytes, void* stream);
/* The same router on per-chunk channel sums a producer kernel already wrote (ymk_c3k2_fused_pooled): part fp32 [B][nchunk][C], any
 * partition of each image's H * W pixels into nchunk chunks.  Skips the read of x; everything else as ymk_esmoe_route. */
int ymk_esmoe_route_pooled(const float* part, int32_t nchunk, int32_t B, int32_t H, int32_t W, int32_t C, const float* w1,
                           const float* b1, const float* w2, const float* b2, int32_t hidden, int32_t E, int32_t top_k,
                           float dynamic_threshold, float* route_w, float* gate_w, int32_t* sel, int32_t* csr_off,
                           int32_t* csr_pair, float* state, int32_t* flags, void* stream);

/* Depthwise stage of the retained experts, dispatched over the CSR pairs
 * (experts.py:283-292, modules.py:690-697).  dw_w is one blob holding every
 * expert's [k_e*k_e][C] filter at element offset dw_off[e]; ksizes[e] odd <= 15.
 * Output: dw_out[pair][H][W][C] (pair = b*top_k + slot), dense (ld = C). */
int ymk_esmoe_dw(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                 int32_t ldx, const void* dw_w, const int32_t* dw_off, const int32_t* ksizes,
                 int32_t E, int32_t top_k, int32_t kmax /* max of ksizes (host copy) */,
                 const int32_t* sel, const int32_t* csr_off,
                 const int32_t* csr_pair, void* dw_out, void* stream);

/* Pointwise stage: grouped GEMM on MFMA over the retained experts of each image,
 * epilogue SiLU(BN_e(.)) * gate_w accumulated in ascending expert order, then the
 * trailing ES_MOE.norm (BN + SiLU) — experts.py:293-296, modules.py:697-702,:581.
 * pw_w: [E][Cout][Kpad] (BN_e folded), pw_b: fp32 [E][Cout];
 * norm_scale/norm_shift: fp32 [Cout] (eval BatchNorm as y*s+t). */
int ymk_esmoe_pw(int32_t dtype, const void* dw_out, int32_t B, int32_t H, int32_t W, int32_t C,
                 int32_t Cout, int32_t Kpad, const void* pw_w, const float* pw_b,
                 const float* norm_scale, const float* norm_shift, int32_t E, int32_t top_k,
                 const int32_t* sel, const float* gate_w, void* y, int32_t ldy, void* stream);

/* ------------------------------------------------------------------------
 * Area attention core: softmax(q^T k / sqrt(d)) v per (image, area, head)
 * (AAttn.forward block.py:1696-1726).  qkv is the NHWC output of the qkv 1x1
 * conv with output channels re-ordered at pack time to [Q | K | V], each
 * [heads][head_dim]; areas are `area` equal contiguous token ranges of H*W.
 * head_dim must be 32.  out: [B][H*W][heads*32] view with pixel stride ldo.
 * ------------------------------------------------------------------------ */
int ymk_area_attn(int32_t dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo, int32_t B,
                  int32_t N /*tokens = H*W*/, int32_t heads, int32_t area, void* stream);

/* The same attention with the qkv projection INSIDE the kernel (16-bit, C = heads * 32 in {64, 128}, H*W / area <= 448: the
 * detector's 40x40 A2C2f row): x is the ABlock's input [B][H*W][C] (pixel stride ldx), w the folded qkv 1x1 convolution + BatchNorm
 * (block.py:1687,1708) packed [3C][Kpad] with rows ordered [Q | K | V], each [heads][32] (nn/modules.py AAttn.qkv.cout_perm), bias
 * fp32 [3C].  K and V^T of a head are produced in LDS and never stored; written: out = the attention output [B][H*W][C] (stride
 * ldo; the kernel also parks q there before overwriting it) and v [B][H*W][C] (stride ldv), the input of AAttn.pe.  Same results as
 * ymk_conv2d + ymk_area_attn up to the rounding of the 16-bit q / k / v.  YMK_E_BADARG outside ymk_area_attn_qkv_supported. */
int ymk_area_attn_qkv_supported(int32_t dtype, int32_t C, int32_t heads, int32_t N /*tokens = H*W*/, int32_t area);
int ymk_area_attn_qkv(int32_t dtype, const void* x, int32_t ldx, const void* w, int32_t Kpad, const float* bias, void* out,
                      int32_t ldo, void* v, int32_t ldv, int32_t B, int32_t N, int32_t C, int32_t heads, int32_t area, void* stream);

/* nearest 2x upsample (nn.Upsample(None, 2, "nearest"), yaml head) into a channel slice */
int ymk_upsample2x(int32_t dtype, const void* x, void* y, int32_t B, int32_t H, int32_t W,
                   int32_t C, int32_t ldx, int32_t ldy, void* stream);
/* channel-slice copy (Concat.forward conv.py:629-641 when a producer could not write in place) */
int ymk_copy_channels(int32_t dtype, const void* x, void* y, int64_t npix, int32_t C, int32_t ldx,
                      int32_t ldy, void* stream);
/* out = residual + gamma[c] * y, per channel: the gamma-residual that closes A2C2f at the l/x scales
 * (ultralytics/nn/modules/block.py:1877-1879; gamma fp32 [C]); out may alias y or residual */
int ymk_scale_residual(int32_t dtype, const void* y, const float* gamma, const void* residual, void* out,
                       int64_t npix, int32_t C, int32_t ldy, int32_t ldr, int32_t ldo, void* stream);
/* NHWC (any ld) -> dense NCHW fp32, for the module-level API and feature taps */
int ymk_nhwc_to_nchw_f32(int32_t dtype, const void* x, float* y, int32_t B, int32_t HW, int32_t C,
                         int32_t ldx, void* stream);

/* ------------------------------------------------------------------------
 * Fused two-layer 1x1 MLP with residual (bf16): y = x + W2 * SiLU(W1 * x + b1) + b2 per token — ABlock's `x + mlp(x)`
 * (nn/modules/block.py:1772-1797) as ONE kernel: the hidden tensor stays in LDS.  Same operands as two ymk_conv2d calls
 * (packed [Cout][Kpad] bf16 weights with the BN folded, fp32 biases).  (C, hidden) in {(64,128), (128,256), (256,512)}.
 * ------------------------------------------------------------------------ */
int ymk_mlp_fused_supported(int32_t dtype, int32_t C, int32_t hidden);
int ymk_mlp_fused(const void* x, int32_t ldx, const void* w1, int32_t k1pad, const float* b1, const void* w2, int32_t k2pad,
                  const float* b2, void* y, int32_t ldy, int64_t M, int32_t C, int32_t hidden, void* stream);


/* ------------------------------------------------------------------------
 * Detect decode: DFL softmax-expectation + dist2bbox(xywh) * stride + sigmoid
 * (Detect._inference head.py:173-194, DFL.forward block.py:81-84,
 *  make_anchors/dist2bbox utils/tal.py:398-423).
 * box_l / cls_l: fp32 NHWC logits of one level: [B][H_l*W_l][4*reg_max] / [..][ldc].
 * ldc: row stride of cls_l in floats: nc, or 4*ceil(nc/4) when the class rows are padded to 16 bytes (nc % 4 != 0).
 * y: fp32 [B][4+nc][A_total]; this call fills anchors [a_off, a_off + H_l*W_l).
 * ------------------------------------------------------------------------ */
int ymk_detect_decode(const float* box_l, const float* cls_l, float* y, int32_t B, int32_t Hl,
                      int32_t Wl, int32_t reg_max, int32_t nc, int32_t ldc, float stride, int32_t a_off,
                      int32_t A_total, float* best_conf /*[B][A_total] or NULL*/, int32_t* best_cls /*[B][A_total] or NULL*/,
                      void* stream);
/* best_conf / best_cls (both or neither): the largest class score of every anchor and its class (first maximum in class order), the
 * values stored in y — the single-label candidate filter of non_max_suppression (utils/nms.py:124-129) precomputed by the producer. */

/* ------------------------------------------------------------------------
 * Batched NMS: non_max_suppression (utils/nms.py:13-171) with TorchNMS.nms
 * greedy semantics (nms.py:245-302): candidates conf > thres (best class, or
 * every class when multi_label), stable score-descending order, cap max_nms,
 * class offset cls*max_wh, suppress IoU > iou_thres, first max_det kept.
 * y: fp32 [B][4+nc+extra][A] as produced by ymk_detect_decode (not modified).  extra: rows behind the class rows that
 * are carried, not scored (utils/nms.py:76-81 `extra = shape[1] - nc - 4`: the mask coefficients of a Segment head); 0 for Detect.
 * out_dets fp32 [B][max_det][6] (x1,y1,x2,y2,conf,cls); out_counts int32 [B];
 * out_idx int32 [B][max_det] anchor index of each kept detection
 * (return_idxs=True of the reference); status int32 [1] YMK_FLAG_NMS_OVERFLOW.
 * ------------------------------------------------------------------------ */
size_t ymk_nms_workspace_bytes(int32_t B, int32_t nc, int32_t A, int32_t multi_label,
                               int32_t max_nms);
int ymk_nms_batched(const float* y, int32_t B, int32_t nc, int32_t extra, int32_t A, float conf_thres,
                    float iou_thres, int32_t multi_label, int32_t agnostic, int32_t max_det,
                    int32_t max_nms, float max_wh, const uint8_t* class_keep /*[nc] or NULL*/,
                    const float* best_conf /*[B][A] or NULL*/, const int32_t* best_cls /*[B][A] or NULL*/,
                    float* out_dets, int32_t* out_counts, int32_t* out_idx, int32_t* status,
                    void* workspace, size_t workspace_bytes, void* stream);
/* best_conf / best_cls: ymk_detect_decode's per-anchor best class of THIS y (must describe it exactly); with them the single-label
 * path does not read the class rows of y at all.  Ignored when multi_label. */
/* class_keep: the `classes=` filter of non_max_suppression (utils/nms.py:63,132): a candidate survives only when
 * class_keep[its class] != 0; applied after the best-class choice of the single-label path, as the reference does. */

/* The carried rows of the kept detections (utils/nms.py:117,122,127: the `mask` columns of each output row):
 * out fp32 [B][max_det][extra], out[b][j][k] = y[b][row0 + k][out_idx[b][j]] for j < out_counts[b], zeros behind the count.
 * rows = 4 + nc + extra of y, row0 = 4 + nc. */
int ymk_nms_gather_rows(const float* y, int32_t B, int32_t rows, int32_t A, int32_t row0, int32_t extra, const int32_t* out_idx,
                        const int32_t* out_counts, int32_t max_det, float* out, void* stream);

/* Cluster-weighted box refinement (CW-NMS).  Not implemented in the reference's
 * Python; algorithm spec = examples/YOLO-Master-Cross-Platform-Edge-Deployment/
 * cpp/src/common.cpp:150-185 (fp64 accumulation, pool = top-3000 candidates).
 * Must be called right after ymk_nms_batched with the same workspace and the same
 * (B, nc, A, multi_label, max_nms) so that the workspace layout matches;
 * rewrites out_dets[..][0:4] in place, keep-set/order/scores/classes unchanged.
 * agnostic != 0 (class-agnostic suppression was used): clusters ignore classes too (the spec defines only the per-class case). */
int ymk_cw_refine(int32_t B, int32_t nc, int32_t A, int32_t multi_label, int32_t agnostic, int32_t max_nms, int32_t max_det,
                  float iou_thres, float sigma, int32_t pool_cap, float* out_dets,
                  const int32_t* out_counts, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YMK_H_ */
