// The step right after the hot path: map detections from the letterboxed network input back to
// the original images — scale_boxes + clip_boxes (ultralytics/utils/ops.py:119-205), batched over the padded output of
// ymk_nms_batched.  Opt-in (include/ymk_next.h): verified bit-exact against the reference's golden vectors on the CPU
// lane emulator, not yet run on hardware.  Compiled with -ffp-contract=off: subtract, IEEE divide, clamp, as the reference.
#include "ymk_common.h"

__global__ __launch_bounds__(256) void scale_boxes_kernel(float* dets, int ld, const int32_t* counts, const float* params, int B,
                                                          int max_det, int padding, int xywh) {
    const int64_t total = (int64_t)B * max_det;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / max_det), r = (int)(i % max_det);
        if (counts && r >= counts[b]) continue;
        const float* p = params + 5 * b;   // gain, pad_x, pad_y, w0, h0
        float* d = dets + i * ld;
        float x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
        if (padding) {
            x1 -= p[1]; y1 -= p[2];
            if (!xywh) { x2 -= p[1]; y2 -= p[2]; }
        }
        x1 /= p[0]; y1 /= p[0]; x2 /= p[0]; y2 /= p[0];
        if (!xywh) {
            x1 = fminf(fmaxf(x1, 0.f), p[3]); y1 = fminf(fmaxf(y1, 0.f), p[4]);
            x2 = fminf(fmaxf(x2, 0.f), p[3]); y2 = fminf(fmaxf(y2, 0.f), p[4]);
        }
        d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2;
    }
}

extern "C" int ymk_scale_boxes(float* dets, int32_t ld, const int32_t* counts, const float* params, int32_t B, int32_t max_det,
                               int32_t padding, int32_t xywh, void* stream) {
    if (!dets || !params || ld < 4) return YMK_E_BADARG;
    if (B <= 0 || max_det <= 0) return YMK_OK;
    const int64_t total = (int64_t)B * max_det;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(scale_boxes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dets, ld, counts, params, B, max_det, padding, xywh);
    return ymk_launch_status();
}

// ---- Segment head (nn/modules/block.py:88-107, head.py:317-349) -------------------------------
// ConvTranspose2d(C, C, 2, 2) = a 1x1 convolution to 4*C channels (one C-wide slice per output phase (dy, dx), bias tiled)
// followed by this depth-to-space: out[b][2y+dy][2x+dx][c] = t[b][y][x][(dy*2+dx)*C + c].  16-byte chunks.
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const char* t, int64_t ldt_b, char* out, int64_t ldo_b, int B, int H, int W, int cv) {
    const int64_t total = (int64_t)B * 2 * H * 2 * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        int64_t p = i / cv;
        const int ox = (int)(p % (2 * W));
        p /= 2 * W;
        const int oy = (int)(p % (2 * H));
        const int b = (int)(p / (2 * H));
        const int ph = (oy & 1) * 2 + (ox & 1);
        const u32x4 v = *reinterpret_cast<const u32x4*>(t + (((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * ldt_b + ((int64_t)ph * cv + c) * 16);
        *reinterpret_cast<u32x4*>(out + (((int64_t)b * 2 * H + oy) * 2 * W + ox) * ldo_b + (int64_t)c * 16) = v;
    }
}
extern "C" int ymk_pixel_shuffle2(int32_t dtype, const void* t, int32_t ldt, void* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t C,
                                  void* stream) {
    const int vec = dtype == YMK_BF16 ? 8 : 4, es = dtype == YMK_BF16 ? 2 : 4;
    if (!t || !out || (dtype != YMK_F32 && dtype != YMK_BF16) || C < vec || C % vec || ldt % vec || ldo % vec || ldt < 4 * C || ldo < C)
        return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    const int64_t total = (int64_t)B * 4 * H * W * (C / vec);
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)t, (int64_t)ldt * es, (char*)out,
                       (int64_t)ldo * es, B, H, W, C / vec);
    return ymk_launch_status();
}

// Token-major maps of one pyramid level into rows of a [B][rows_total][A_total] fp32 tensor (the mask coefficients of
// Segment: head.py:341-349): y[b][row_off + c][a_off + p] = x[b][p][c].  64 x 64 LDS transpose tile like ymk_nhwc_to_nchw_f32.
template <typename T>
__global__ __launch_bounds__(256) void tokens_to_rows_kernel(const T* x, int ldx, float* y, int HW, int C, int a_off, int A_total, int row_off,
                                                             int rows_total) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? to_f32(x[((size_t)b * HW + p) * ldx + c]) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) y[((size_t)b * rows_total + row_off + c) * A_total + a_off + p] = tile[tx][r];
    }
}
extern "C" int ymk_tokens_to_rows(int32_t dtype, const void* x, int32_t ldx, float* y, int32_t B, int32_t HW, int32_t C, int32_t a_off,
                                  int32_t A_total, int32_t row_off, int32_t rows_total, void* stream) {
    if (!x || !y || (dtype != YMK_F32 && dtype != YMK_BF16) || C < 1 || ldx < C || a_off < 0 || a_off + HW > A_total || row_off < 0 ||
        row_off + C > rows_total || B > 65535)
        return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    const dim3 grid((HW + 63) / 64, (C + 63) / 64, B);
    if (dtype == YMK_BF16)
        hipLaunchKernelGGL(tokens_to_rows_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, ldx, y, HW, C, a_off, A_total,
                           row_off, rows_total);
    else
        hipLaunchKernelGGL(tokens_to_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, y, HW, C, a_off, A_total,
                           row_off, rows_total);
    return ymk_launch_status();
}

// ---- process_mask (ultralytics/utils/ops.py:477-528), one image per call ---------------------------------------------------
// coefficients of the kept anchors: out[j][k] = mc[b][k][idx[j]]
__global__ __launch_bounds__(256) void mask_coeff_gather_kernel(const float* mc, int nm, int A, int b, const int64_t* idx, int n, float* out) {
    const int total = n * nm;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i / nm, k = i % nm;
        out[i] = mc[((size_t)b * nm + k) * A + idx[j]];
    }
}
// prototype-resolution masks: L[d][y][x] = sum_k coef[d][k] * proto[y][x][k] (prototypes NHWC, as the Proto module leaves them)
template <typename T>
__global__ __launch_bounds__(256) void mask_lowres_kernel(const T* proto, int ldp, const float* coef, int nm, int n, int mh, int mw, float* L) {
    const int64_t total = (int64_t)n * mh * mw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % ((int64_t)mh * mw);
        const int d = (int)(i / ((int64_t)mh * mw));
        const T* pr = proto + pix * ldp;
        const float* c = coef + (size_t)d * nm;
        float s = 0.f;
        for (int k = 0; k < nm; ++k) s += c[k] * to_f32(pr[k]);
        L[i] = s;
    }
}
// binarised output: upsample == 0: crop at prototype resolution with the boxes scaled by (rw, rh); upsample != 0: bilinear
// (align_corners = False, PyTorch's source-index rule) to H x W, then crop with the boxes as they are
__global__ __launch_bounds__(256) void mask_finish_kernel(const float* L, const float* boxes, int ldb, int n, int mh, int mw, int H, int W,
                                                          int upsample, float rw, float rh, uint8_t* out) {
    const int64_t total = (int64_t)n * H * W;
    const float sh = (float)mh / (float)H, sw = (float)mw / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % W);
        const int Y = (int)((i / W) % H);
        const int d = (int)(i / ((int64_t)H * W));
        const float* bx = boxes + (size_t)d * ldb;
        float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
        if (!upsample) { x1 *= rw; y1 *= rh; x2 *= rw; y2 *= rh; }
        const float fx = (float)X, fy = (float)Y;
        uint8_t o = 0;
        if (fx >= x1 && fx < x2 && fy >= y1 && fy < y2) {
            const float* Ld = L + (size_t)d * mh * mw;
            float v;
            if (!upsample) {
                v = Ld[Y * mw + X];
            } else {
                float ry = sh * (fy + 0.5f) - 0.5f, rx = sw * (fx + 0.5f) - 0.5f;
                ry = ry < 0.f ? 0.f : ry;
                rx = rx < 0.f ? 0.f : rx;
                const int y0 = (int)ry, x0 = (int)rx;
                const int y1i = y0 + (y0 < mh - 1 ? 1 : 0), x1i = x0 + (x0 < mw - 1 ? 1 : 0);
                const float ly1 = ry - (float)y0, lx1 = rx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                v = ly0 * (lx0 * Ld[y0 * mw + x0] + lx1 * Ld[y0 * mw + x1i]) + ly1 * (lx0 * Ld[y1i * mw + x0] + lx1 * Ld[y1i * mw + x1i]);
            }
            o = v > 0.f ? 1 : 0;
        }
        out[i] = o;
    }
}

extern "C" int ymk_mask_coeff_gather(const float* mc, int32_t nm, int32_t A, int32_t b, const int64_t* idx, int32_t n, float* out, void* stream) {
    if (!mc || !idx || !out || nm < 1 || A < 1 || b < 0) return YMK_E_BADARG;
    if (n <= 0) return YMK_OK;
    hipLaunchKernelGGL(mask_coeff_gather_kernel, dim3((n * nm + 255) / 256), dim3(256), 0, (hipStream_t)stream, mc, nm, A, b, idx, n, out);
    return ymk_launch_status();
}
extern "C" int ymk_process_mask(int32_t dtype, const void* protos, int32_t ldp, int32_t mh, int32_t mw, int32_t nm, const float* coefs,
                                const float* boxes, int32_t ldb, int32_t n, int32_t H, int32_t W, int32_t upsample, float rw, float rh,
                                float* lowres_ws, uint8_t* out, void* stream) {
    if (!protos || !coefs || !boxes || !lowres_ws || !out || (dtype != YMK_F32 && dtype != YMK_BF16) || nm < 1 || ldp < nm || ldb < 4 ||
        mh < 1 || mw < 1 || H < 1 || W < 1 || (!upsample && (H != mh || W != mw)))
        return YMK_E_BADARG;
    if (n <= 0) return YMK_OK;
    const int64_t tl = (int64_t)n * mh * mw, tf = (int64_t)n * H * W;
    const int bl = (int)((tl + 255) / 256 > 16384 ? 16384 : (tl + 255) / 256), bf = (int)((tf + 255) / 256 > 32768 ? 32768 : (tf + 255) / 256);
    if (dtype == YMK_BF16)
        hipLaunchKernelGGL(mask_lowres_kernel<h16_t>, dim3(bl), dim3(256), 0, (hipStream_t)stream, (const h16_t*)protos, ldp, coefs, nm, n, mh, mw, lowres_ws);
    else
        hipLaunchKernelGGL(mask_lowres_kernel<float>, dim3(bl), dim3(256), 0, (hipStream_t)stream, (const float*)protos, ldp, coefs, nm, n, mh, mw, lowres_ws);
    hipLaunchKernelGGL(mask_finish_kernel, dim3(bf), dim3(256), 0, (hipStream_t)stream, (const float*)lowres_ws, boxes, ldb, n, mh, mw, H, W, upsample, rw,
                       rh, out);
    return ymk_launch_status();
}

// ---- Validation matching (the remainder): box_iou + match_predictions ---------------------------------------
// box_iou (ultralytics/utils/metrics.py:82-104): iou[i][j] = inter / (area1_i + area2_j - inter + eps), fp32, the reference's
// operation order (no FMA contraction in this file).  One thread per pair, rows of box1 as the slow index.
__global__ __launch_bounds__(256) void box_iou_kernel(const float* __restrict__ b1, int ld1, const float* __restrict__ b2, int ld2, int N,
                                                      int M, float eps, float* __restrict__ out) {
    const int64_t total = (int64_t)N * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float* a = b1 + (i / M) * ld1;
        const float* b = b2 + (i % M) * ld2;
        const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f), h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
        const float inter = w * h;
        out[i] = inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + eps);
    }
}
extern "C" int ymk_box_iou(const float* box1, int32_t ld1, int32_t N, const float* box2, int32_t ld2, int32_t M, float eps, float* out,
                           void* stream) {
    if (!out || ld1 < 4 || ld2 < 4 || N < 0 || M < 0) return YMK_E_BADARG;
    if (N == 0 || M == 0) return YMK_OK;
    if (!box1 || !box2) return YMK_E_BADARG;
    const int64_t total = (int64_t)N * M;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(box_iou_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, box1, ld1, box2, ld2, N, M, eps, out);
    return ymk_launch_status();
}

// match_predictions (ultralytics/engine/validator.py:301-336, the numpy path) batched over images: one workgroup per image.
// The reference, per IoU threshold t: pairs (label l, detection d) with iou[l][d] * (cls_l == cls_d) >= t, sorted by IoU descending;
// np.unique over the detection column keeps each detection's best label (result ordered by detection index); np.unique over the label
// column then keeps, per label, the FIRST of those — the lowest detection index, i.e. the most confident detection that chose it.
// A detection's best label does not depend on t (it is the arg-max of its IoU column; t only decides whether the pair exists), so:
//   best[d] = argmax_l iou[l][d] (class-matched);  first[l][t] = min { d : best[d] == l, iou_best[d] >= t };  correct[d][t] = first[best[d]][t] == d.
// Equal IoUs of one detection with two labels (identical ground-truth boxes) are ordered by numpy's unstable argsort in the
// reference — implementation-defined; here the higher label index wins (what a stable ascending sort, reversed, gives).
#define MP_MAXT 16
__global__ __launch_bounds__(256) void match_predictions_kernel(const float* __restrict__ dets, int ldd, const int* __restrict__ counts,
                                                                int max_det, const float* __restrict__ labels,
                                                                const int* __restrict__ label_off, const float* __restrict__ iouv,
                                                                int T, float eps, unsigned char* __restrict__ correct, int* __restrict__ ws) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int D = counts ? min(counts[b], max_det) : max_det;
    const int l0 = label_off[b], L = label_off[b + 1] - l0;
    int* first = ws + (size_t)l0 * T;                       // [L][T], this image's slice of the workspace
    for (int i = t; i < L * T; i += 256) first[i] = 0x7fffffff;
    for (int i = t; i < max_det * T; i += 256) correct[((size_t)b * max_det) * T + i] = 0;
    __syncthreads();
    float thr[MP_MAXT];
    for (int i = 0; i < T; ++i) thr[i] = iouv[i];
    for (int d0 = 0; d0 < D; d0 += 256) {
        const int d = d0 + t;
        int best = -1;
        float bi = 0.f;
        if (d < D) {
            const float* p = dets + ((size_t)b * max_det + d) * ldd;
            const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], pc = p[5];
            const float ap = (x2 - x1) * (y2 - y1);
            for (int l = 0; l < L; ++l) {
                const float* g = labels + (size_t)(l0 + l) * 5;      // cls, x1, y1, x2, y2
                if (g[0] != pc) continue;                            // iou * correct_class: wrong classes are exact zeros
                const float w = fmaxf(fminf(g[3], x2) - fmaxf(g[1], x1), 0.f), h = fmaxf(fminf(g[4], y2) - fmaxf(g[2], y1), 0.f);
                const float inter = w * h;
                const float iou = inter / ((g[3] - g[1]) * (g[4] - g[2]) + ap - inter + eps);
                if (iou >= bi && iou > 0.f) { bi = iou; best = l; }  // >=: the higher label index wins a tie
            }
            if (best >= 0)
                for (int i = 0; i < T; ++i)
                    if (bi >= thr[i]) atomicMin(&first[best * T + i], d);
        }
    }
    __syncthreads();
    // second pass: recompute nothing — a detection is correct at threshold i iff it is the first taker of its best label there.
    for (int d0 = 0; d0 < D; d0 += 256) {
        const int d = d0 + t;
        if (d >= D) continue;
        const float* p = dets + ((size_t)b * max_det + d) * ldd;
        const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3], pc = p[5];
        const float ap = (x2 - x1) * (y2 - y1);
        int best = -1;
        float bi = 0.f;
        for (int l = 0; l < L; ++l) {
            const float* g = labels + (size_t)(l0 + l) * 5;
            if (g[0] != pc) continue;
            const float w = fmaxf(fminf(g[3], x2) - fmaxf(g[1], x1), 0.f), h = fmaxf(fminf(g[4], y2) - fmaxf(g[2], y1), 0.f);
            const float inter = w * h;
            const float iou = inter / ((g[3] - g[1]) * (g[4] - g[2]) + ap - inter + eps);
            if (iou >= bi && iou > 0.f) { bi = iou; best = l; }
        }
        if (best >= 0)
            for (int i = 0; i < T; ++i)
                correct[((size_t)b * max_det + d) * T + i] = (bi >= thr[i] && first[best * T + i] == d) ? 1 : 0;
    }
}
extern "C" size_t ymk_match_predictions_workspace_bytes(int32_t total_labels, int32_t T) {
    return (size_t)(total_labels > 0 ? total_labels : 1) * (size_t)(T > 0 ? T : 1) * sizeof(int);
}
extern "C" int ymk_match_predictions(const float* dets, int32_t ldd, const int32_t* counts, int32_t B, int32_t max_det, const float* labels,
                                     const int32_t* label_off, int32_t total_labels, const float* iouv, int32_t T, float eps,
                                     uint8_t* correct, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dets || !label_off || !iouv || !correct || !workspace || ldd < 6 || T < 1 || T > MP_MAXT || max_det < 1) return YMK_E_BADARG;
    if (total_labels > 0 && !labels) return YMK_E_BADARG;
    if (workspace_bytes < ymk_match_predictions_workspace_bytes(total_labels, T)) return YMK_E_WORKSPACE;
    if (B <= 0) return YMK_OK;
    hipLaunchKernelGGL(match_predictions_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dets, ldd, counts, max_det, labels, label_off,
                       iouv, T, eps, correct, (int*)workspace);
    return ymk_launch_status();
}
