/* libymk — entry points next to ymk.h: the LDS-DMA convolution core under its own names, and the rows right after / beside the hot path.
 *
 * (1) ymk_conv2d_glds / ymk_conv1x1_cat2_glds / ymk_expert_conv_glds: the tiled implicit-GEMM core (csrc/conv_glds.hip): 64-256 couts x
 * 128-512 pixels per 8-wave workgroup, both operands staged into LDS by `buffer_load_dwordx4 ... lds` through raw buffer resources
 * (source-side swizzle; taps outside the image are out-of-range lanes: zeros), 2- or 3-stage k-loop, XCD-aware tile order.  Same
 * arguments and result as ymk_conv2d (ymk.h) plus `two_stage` (bit 0: 0 = three LDS stages with a counted vmcnt, 1 = two stages with
 * a plain barrier; bits 8-11 / 12-21: a forced tile shape for tests and A/B tools); 16-bit only, Cin % 64 == 0, Cout % 64 == 0,
 * operands below 2 GiB, otherwise YMK_E_BADARG.  Since round 2 ymk_conv2d / ymk_conv1x1_cat2 dispatch here by default for every
 * shape in that domain (YMK_DISABLE bits select the older cores); validated on MI355X (tests/test_gpu_next.py) and on the CPU lane
 * emulator (tests/test_hostemu_conv.py); per-shape timings in profiles/r02_glds_tile_ab.txt, profiles/r03_glds_*.txt.
 * (2) The rows right after / beside the hot path: box rescaling, the Segment head's layout kernels and
 * process_mask.  Each is pinned to golden vectors generated from the real reference, verified on the CPU lane emulator
 * (tests/test_hostemu_post.py) and on MI355X (tests/test_gpu_next.py); yolo_master_amd/postprocess.py and the drop-in hooks call them.
 */
#ifndef YMK_NEXT_H_
#define YMK_NEXT_H_
#include "ymk.h"
#ifdef __cplusplus
extern "C" {
#endif
int ymk_conv2d_glds(const ymk_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y,
                    int32_t two_stage, void* stream);

/* Virtual-concatenation form of the same core: arguments and result of ymk_conv1x1_cat2 (ymk.h) + two_stage; bf16, C1 and
 * Cin - C1 multiples of 64, Cout % 64 == 0, otherwise YMK_E_BADARG.  Same dispatch. */
int ymk_conv1x1_cat2_glds(const ymk_conv_desc* d, const void* x1, int32_t C1, int32_t ldx1, int32_t upsample1, const void* x2,
                          int32_t ldx2, const void* w, const float* bias, void* y, int32_t two_stage, void* stream);

/* Routed-expert form of the same core (true sparse dispatch for the gated MoE's expert groups, moe/gated.py:1058-1076,
 * moe/experts.py:235-269): d describes ONE expert's convolution (stride 1, no activation, no bias); w [E][Cout][Kpad]; idx int32
 * [B][K] the routed experts; y slot-major: image j*B + b = conv(x[b], w[idx[b][j]]).  Only the routed filter banks run (the first
 * implementation behind ops.expert_conv convolves with all E banks and gathers).  bf16, Cin % 64 == 0, Cout % 64 == 0. */
int ymk_expert_conv_glds(const ymk_conv_desc* d, const void* x, const void* w, const int32_t* idx, int32_t K, int32_t E, void* y,
                         int32_t two_stage, void* stream);

/* The step after the hot path: scale_boxes + clip_boxes (ultralytics/utils/ops.py:119-205, called
 * per image by models/yolo/detect/predict.py:109-122), batched and in place over padded detections.
 * dets fp32 [B][max_det] rows of `ld` >= 4 floats (x1, y1, x2, y2, ...); counts int32 [B] valid rows per image (NULL = all);
 * params fp32 [B][5] = (gain, pad_x, pad_y, w0, h0) per image, computed by the host exactly as the reference does (Python
 * doubles, round-half-even; gain rounded to fp32).  padding / xywh as in the reference.  Bit-exact against the reference's
 * golden vectors on the CPU lane emulator (tests/test_hostemu_post.py) and on MI355X (tests/test_gpu_next.py). */
int ymk_scale_boxes(float* dets, int32_t ld, const int32_t* counts, const float* params, int32_t B, int32_t max_det,
                    int32_t padding, int32_t xywh, void* stream);
/* Segment head pieces.  ymk_pixel_shuffle2: depth-to-space that turns the 4*C-channel output of a 1x1
 * convolution into Proto's ConvTranspose2d(C, C, 2, 2) result (nn/modules/block.py:101-107): out[b][2y+dy][2x+dx][c] =
 * t[b][y][x][(dy*2+dx)*C + c].  ymk_tokens_to_rows: y[b][row_off + c][a_off + p] = x[b][p][c] into an fp32 [B][rows_total][A_total]
 * tensor — the mask coefficients of the three cv4 branches in the reference's layout (nn/modules/head.py:341-349). */
int ymk_pixel_shuffle2(int32_t dtype, const void* t, int32_t ldt, void* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t C,
                       void* stream);
int ymk_tokens_to_rows(int32_t dtype, const void* x, int32_t ldx, float* y, int32_t B, int32_t HW, int32_t C, int32_t a_off,
                       int32_t A_total, int32_t row_off, int32_t rows_total, void* stream);

/* process_mask of the segmentation predictor (ultralytics/utils/ops.py:477-528), one image per call.
 * ymk_mask_coeff_gather: out[j][k] = mc[b][k][idx[j]] — the mask coefficients (fp32 [B][nm][A], ymk_tokens_to_rows) of the anchors
 * NMS kept (idx int64 [n], the `return_idxs` output).
 * ymk_process_mask: protos NHWC [mh][mw][nm] (pixel stride ldp) of that image, coefs fp32 [n][nm], boxes fp32 rows of ldb >= 4 floats
 * (xyxy in network-input pixels); out uint8 [n][H][W].  upsample = 0: H x W = mh x mw, crop with the boxes scaled by (rw, rh) =
 * (mw / W_in, mh / H_in); upsample != 0: bilinear (align_corners = False) to H x W, then crop with the boxes.  Binarised at 0.
 * lowres_ws: fp32 [n * mh * mw] scratch. */
int ymk_mask_coeff_gather(const float* mc, int32_t nm, int32_t A, int32_t b, const int64_t* idx, int32_t n, float* out, void* stream);
int ymk_process_mask(int32_t dtype, const void* protos, int32_t ldp, int32_t mh, int32_t mw, int32_t nm, const float* coefs,
                     const float* boxes, int32_t ldb, int32_t n, int32_t H, int32_t W, int32_t upsample, float rw, float rh,
                     float* lowres_ws, uint8_t* out, void* stream);

/* Validation matching (the step after the path in val(): ultralytics/models/yolo/detect/val.py `_process_batch`).
 * ymk_box_iou: utils/metrics.py:82-104, out[i][j] = IoU(box1[i], box2[j]) fp32 [N][M]; rows of ld1 / ld2 >= 4 floats (xyxy).
 * ymk_match_predictions: BaseValidator.match_predictions (engine/validator.py:301-336, numpy path) for a whole batch:
 *   dets fp32 [B][max_det][ldd >= 6] (xyxy, conf, cls — the padded output of ymk_nms_batched, in the labels' coordinate frame),
 *   counts int32 [B] or NULL, labels fp32 [total_labels][5] = (cls, x1, y1, x2, y2) grouped by image, label_off int32 [B+1],
 *   iouv fp32 [T] (T <= 16; torch.linspace(0.5, 0.95, 10)), eps of box_iou (1e-7);
 *   correct uint8 [B][max_det][T] (rows past counts[b] are 0).  workspace: ymk_match_predictions_workspace_bytes(total_labels, T). */
int ymk_box_iou(const float* box1, int32_t ld1, int32_t N, const float* box2, int32_t ld2, int32_t M, float eps, float* out, void* stream);
size_t ymk_match_predictions_workspace_bytes(int32_t total_labels, int32_t T);
int ymk_match_predictions(const float* dets, int32_t ldd, const int32_t* counts, int32_t B, int32_t max_det, const float* labels,
                          const int32_t* label_off, int32_t total_labels, const float* iouv, int32_t T, float eps, uint8_t* correct,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Predictor pre-processing on the device (the step before the path): LetterBox resize + pad (ultralytics/data/augment.py:1646-1830)
 * + BGR->RGB + HWC->CHW + /255 (BasePredictor.preprocess, engine/predictor.py:155-178), one launch per batch.
 * src: the batch's uint8 HWC 3-channel images packed back to back, image b at byte offset src_off[b] (device int64 [B]);
 * geom: device int32 [B][6] = (src_h, src_w, new_h, new_w, top, left) as LetterBox.get_params computes them (host side:
 * yolo_master_amd.preprocess.letterbox_params); dst: fp32 [B][3][H][W], padded with pad_value / 255 (114).
 * The bilinear resize restates OpenCV's generic 8-bit INTER_LINEAR path (fixed point, see csrc/preproc.hip). */
int ymk_letterbox_preprocess(const void* src, const int64_t* src_off, const int32_t* geom, float* dst, int32_t B, int32_t H,
                             int32_t W, int32_t pad_value, int32_t swap_rb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YMK_NEXT_H_ */
