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
