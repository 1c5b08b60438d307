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
        hipLaunchKernelGGL(tokens_to_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, y, HW, C, a_off, A_total,
                           row_off, rows_total);
    else
        hipLaunchKernelGGL(tokens_to_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, y, HW, C, a_off, A_total,
                           row_off, rows_total);
    return ymk_launch_status();
}
