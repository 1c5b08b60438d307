// Device-side pre-processing of the predictor: LetterBox (resize + pad) + BGR->RGB + HWC->CHW + /255, one launch per batch.
// Reference: ultralytics/data/augment.py:1646-1830 (LetterBox.get_params / apply_image; the geometry — new_unpad, top, left —
// is computed on the host exactly as get_params does, Python round() included) and engine/predictor.py:155-205
// (BasePredictor.preprocess: stack, [..., ::-1], transpose(0, 3, 1, 2), float(), /= 255).
//
// The reference resizes with cv2.resize(..., INTER_LINEAR) on uint8 — a third-party dependency that is not vendored in the
// reference tree (opencv-python >= 4.6, pyproject.toml).  This kernel restates OpenCV's generic (non-IPP) 8-bit bilinear path
// (modules/imgproc/src/resize.cpp): 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS), horizontal pass to int, vertical
// pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, source coordinate (d + 0.5) * scale - 0.5 evaluated in
// double and narrowed to float, and the exact-2x-downscale special case that OpenCV routes to its 2x2 area average.
// Bytes: reads every source pixel about once (cached taps), writes B * 3 * H * W fp32: HBM-bound, no arithmetic to speak of.
// Compile with -ffp-contract=off and IEEE division (the /255 must round like torch's).
#include "ymk_common.h"

struct PreArgs {
    const uint8_t* src;
    const long long* off;   // [B] byte offset of image b in src (HWC, 3 channels, rows contiguous)
    const int* geom;        // [B][6]: src_h, src_w, new_h, new_w, top, left
    float* dst;             // [B][3][H][W]
    int B, H, W, pad, swap_rb;
};

// OpenCV's per-destination-index source position and 11-bit coefficients of the linear kernel
// Horizontal taps: at the borders OpenCV moves the tap inside and zeroes the fraction (resize.cpp: `if (sx < 0) fx = 0, sx = 0;
// if (sx >= ssize.width - 1) fx = 0, sx = ssize.width - 1`).  Vertical taps keep the fraction and the generic invoker clamps the ROW
// INDICES instead (`clip(yofs[dy] + k, 0, ssize.height)`): with both rows equal the two products round separately, so the first /
// last rows of an upscaled image can come out one LSB lower than with a zeroed fraction.
__device__ __forceinline__ void pre_coef(int d, double scale, int ssize, bool vertical, int& s0, int& s1, int& a0, int& a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (vertical) {
        s0 = min(max(s, 0), ssize - 1);
        s1 = min(max(s + 1, 0), ssize - 1);
    } else {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
        s0 = s;
        s1 = min(s + 1, ssize - 1);
    }
    a0 = (int)rintf((1.f - f) * 2048.f);   // saturate_cast<short>(cvRound(x)): round half to even, never out of range here
    a1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void letterbox_kernel(PreArgs a) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= a.W) return;
    const int* g = a.geom + b * 6;
    const int sh = g[0], sw = g[1], nh = g[2], nw = g[3], top = g[4], left = g[5];
    const uint8_t* s = a.src + a.off[b];
    int v[3] = {a.pad, a.pad, a.pad};
    const int dy = y - top, dx = x - left;
    if (dy >= 0 && dy < nh && dx >= 0 && dx < nw) {
        if (nh == sh && nw == sw) {                       // no resize (shape[::-1] == new_unpad)
            const uint8_t* p = s + ((size_t)dy * sw + dx) * 3;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        } else if (sh == 2 * nh && sw == 2 * nw) {        // exact 2x downscale: OpenCV's INTER_LINEAR takes the 2x2 area path
            const uint8_t* p0 = s + ((size_t)(2 * dy) * sw + 2 * dx) * 3;
            const uint8_t* p1 = p0 + (size_t)sw * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
        } else {
            const double scale_x = 1.0 / ((double)nw / (double)sw), scale_y = 1.0 / ((double)nh / (double)sh);
            int sx, sx1, ax0, ax1, sy, sy1, by0, by1;
            pre_coef(dx, scale_x, sw, false, sx, sx1, ax0, ax1);
            pre_coef(dy, scale_y, sh, true, sy, sy1, by0, by1);
            const uint8_t* r0 = s + (size_t)sy * sw * 3;
            const uint8_t* r1 = s + (size_t)sy1 * sw * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;     // horizontal pass, scale 2^11
                const int h1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
                v[c] = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
                v[c] = min(max(v[c], 0), 255);
            }
        }
    }
    float* o = a.dst + (size_t)b * 3 * a.H * a.W + (size_t)y * a.W + x;
    const size_t plane = (size_t)a.H * a.W;
    const int c0 = a.swap_rb ? 2 : 0, c2 = a.swap_rb ? 0 : 2;
    o[0] = (float)v[c0] / 255.0f;
    o[plane] = (float)v[1] / 255.0f;
    o[2 * plane] = (float)v[c2] / 255.0f;
}

extern "C" int ymk_letterbox_preprocess(const void* src, const int64_t* src_off, const int32_t* geom, float* dst, int32_t B,
                                        int32_t H, int32_t W, int32_t pad_value, int32_t swap_rb, void* stream) {
    if (!src || !src_off || !geom || !dst || pad_value < 0 || pad_value > 255) return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    if (B > 65535 || H > 65535) return YMK_E_BADARG;
    PreArgs a{(const uint8_t*)src, (const long long*)src_off, geom, dst, B, H, W, pad_value, swap_rb ? 1 : 0};
    hipLaunchKernelGGL(letterbox_kernel, dim3((W + 255) / 256, H, B), dim3(256), 0, (hipStream_t)stream, a);
    return ymk_launch_status();
}
