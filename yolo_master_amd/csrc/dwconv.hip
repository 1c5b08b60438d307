// Depthwise k x k convolution, NHWC, stride 1, pad k/2 — register-blocked
// sliding-window stencil: each thread owns 4 channels x R consecutive output
// pixels of one row; per input row it streams R+K-1 pixel vectors once and keeps
// the K filter taps of that row in registers (loads per FMA ~ 1/4 of a naive
// per-tap gather).  HBM-bound op: no MFMA (the stencil has no shared contraction).
//
// Reference semantics: DWConv (ultralytics/nn/modules/conv.py:185-199),
// AAttn.pe (nn/modules/block.py:1688,1731), DepthwiseSeparableConv.depthwise
// (nn/modules/moe/experts.py:283-292) dispatched per retained (image, expert)
// pair as in ES_MOE._sparse_forward (nn/modules/moe/modules.py:690-697).
#include "ymk_common.h"

#define DW_R 8

__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
    u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = bf16lo(t.x); v[1] = bf16hi(t.x); v[2] = bf16lo(t.y); v[3] = bf16hi(t.y);
}

// xb: image base (pixel (0,0), channel 0 of the view), w: [K*K][C] filter
template <typename T, int K>
__device__ __forceinline__ void dw_strip(const T* __restrict__ xb, int H, int W, int C, int ldx,
                                         const T* __restrict__ w, int y, int x0, int c4,
                                         float (&acc)[DW_R][4]) {
    constexpr int P = K / 2;
#pragma unroll
    for (int r = 0; r < DW_R; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
    for (int ky = 0; ky < K; ++ky) {
        const int iy = y + ky - P;
        if ((unsigned)iy >= (unsigned)H) continue;
        float wr[K][4];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) ld4(w + (size_t)(ky * K + kx) * C + c4, wr[kx]);
        const T* row = xb + (size_t)iy * W * ldx + c4;
#pragma unroll
        for (int j = 0; j < DW_R + K - 1; ++j) {
            const int ix = x0 - P + j;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)ix < (unsigned)W) ld4(row + (size_t)ix * ldx, v);
#pragma unroll
            for (int r = 0; r < DW_R; ++r) {
                const int kx = j - r;  // compile-time after unrolling
                if (kx >= 0 && kx < K) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[r][q] = fmaf(v[q], wr[kx][q], acc[r][q]);
                }
            }
        }
    }
}

struct DwArgs {
    const void* x;
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int B, H, W, C, ldx, ldy, ldr, act;
};

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
    const int nc4 = a.C / 4;
    const int nstrip = (a.W + DW_R - 1) / DW_R;
    const int64_t total = (int64_t)a.B * a.H * nstrip * nc4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % nc4);
    int64_t rest = idx / nc4;
    const int st = (int)(rest % nstrip); rest /= nstrip;
    const int y = (int)(rest % a.H);
    const int b = (int)(rest / a.H);
    const int c4 = cv * 4, x0 = st * DW_R;
    const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.ldx;
    float acc[DW_R][4];
    dw_strip<T, K>(xb, a.H, a.W, a.C, a.ldx, reinterpret_cast<const T*>(a.w), y, x0, c4, acc);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) ld4(a.bias + c4, bv);
    constexpr bool PRECISE = sizeof(T) == 4;
#pragma unroll
    for (int r = 0; r < DW_R; ++r) {
        const int x = x0 + r;
        if (x >= a.W) break;
        const size_t pix = ((size_t)b * a.H + y) * a.W + x;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = acc[r][q] + bv[q];
            if (a.act == YMK_ACT_SILU) v[q] = PRECISE ? silu_exact(v[q]) : silu_f(v[q]);
        }
        if (a.res) {
            float rv[4];
            ld4(reinterpret_cast<const T*>(a.res) + pix * a.ldr + c4, rv);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = rv[q] + v[q];
        }
        store4(reinterpret_cast<T*>(a.y) + pix * a.ldy + c4, v[0], v[1], v[2], v[3]);
    }
}

template <typename T>
static int launch_dw(const DwArgs& a, int k, hipStream_t s) {
    const int64_t total = (int64_t)a.B * a.H * ((a.W + DW_R - 1) / DW_R) * (a.C / 4);
    if (total <= 0) return YMK_OK;
    dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    switch (k) {
        case 1: hipLaunchKernelGGL((dwconv_kernel<T, 1>), grid, blk, 0, s, a); break;
        case 3: hipLaunchKernelGGL((dwconv_kernel<T, 3>), grid, blk, 0, s, a); break;
        case 5: hipLaunchKernelGGL((dwconv_kernel<T, 5>), grid, blk, 0, s, a); break;
        case 7: hipLaunchKernelGGL((dwconv_kernel<T, 7>), grid, blk, 0, s, a); break;
        case 9: hipLaunchKernelGGL((dwconv_kernel<T, 9>), grid, blk, 0, s, a); break;
        case 11: hipLaunchKernelGGL((dwconv_kernel<T, 11>), grid, blk, 0, s, a); break;
        case 13: hipLaunchKernelGGL((dwconv_kernel<T, 13>), grid, blk, 0, s, a); break;
        case 15: hipLaunchKernelGGL((dwconv_kernel<T, 15>), grid, blk, 0, s, a); break;
        default: return YMK_E_BADARG;
    }
    return ymk_launch_status();
}

extern "C" int ymk_dwconv2d(int32_t dtype, const void* x, const void* w, const float* bias,
                            const void* residual, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t ksize, int32_t ldx, int32_t ldy, int32_t ldr, int32_t act,
                            void* stream) {
    if (!x || !w || !y || C % 4 || ldx % 4 || ldy % 4 || (residual && ldr % 4)) return YMK_E_BADARG;
    DwArgs a{x, w, bias, residual, y, B, H, W, C, ldx, ldy, ldr, act};
    if (dtype == YMK_F32) return launch_dw<float>(a, ksize, (hipStream_t)stream);
    if (dtype == YMK_BF16) return launch_dw<bf16_t>(a, ksize, (hipStream_t)stream);
    return YMK_E_BADARG;
}

// ---------------------------------------------------------------------------
// ES-MoE depthwise stage over the image->expert CSR.  blockIdx.y walks the CSR
// pair list (grouped by expert so neighbouring workgroups share a filter and a
// stencil size); every workgroup of a pair takes the same switch arm.
// ---------------------------------------------------------------------------
struct MoeDwArgs {
    const void* x;
    const void* dw_w;
    const int* dw_off;
    const int* ksizes;
    const int* sel;
    const int* csr_off;
    const int* csr_pair;
    void* out;
    int B, H, W, C, ldx, E, top_k;
};

template <typename T, int K>
__device__ __forceinline__ void moe_dw_body(const MoeDwArgs& a, int pair, int e, int64_t idx) {
    const int nc4 = a.C / 4;
    const int nstrip = (a.W + DW_R - 1) / DW_R;
    const int cv = (int)(idx % nc4);
    int64_t rest = idx / nc4;
    const int st = (int)(rest % nstrip);
    const int y = (int)(rest / nstrip);
    const int b = pair / a.top_k;
    const int c4 = cv * 4, x0 = st * DW_R;
    const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.ldx;
    const T* w = reinterpret_cast<const T*>(a.dw_w) + a.dw_off[e];
    float acc[DW_R][4];
    dw_strip<T, K>(xb, a.H, a.W, a.C, a.ldx, w, y, x0, c4, acc);
    T* ob = reinterpret_cast<T*>(a.out) + (size_t)pair * a.H * a.W * a.C;
#pragma unroll
    for (int r = 0; r < DW_R; ++r) {
        const int x = x0 + r;
        if (x >= a.W) break;
        store4(ob + ((size_t)y * a.W + x) * a.C + c4, acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void moe_dw_kernel(MoeDwArgs a) {
    const int p = blockIdx.y;
    if (p >= a.csr_off[a.E]) return;
    const int pair = a.csr_pair[p];
    const int e = a.sel[pair];
    if (e < 0) return;
    const int64_t per = (int64_t)a.H * ((a.W + DW_R - 1) / DW_R) * (a.C / 4);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    switch (a.ksizes[e]) {
        case 3: moe_dw_body<T, 3>(a, pair, e, idx); break;
        case 5: moe_dw_body<T, 5>(a, pair, e, idx); break;
        case 7: moe_dw_body<T, 7>(a, pair, e, idx); break;
        case 9: moe_dw_body<T, 9>(a, pair, e, idx); break;
        case 11: moe_dw_body<T, 11>(a, pair, e, idx); break;
        case 13: moe_dw_body<T, 13>(a, pair, e, idx); break;
        case 15: moe_dw_body<T, 15>(a, pair, e, idx); break;
        default: break;
    }
}

extern "C" int ymk_esmoe_dw(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t ldx, const void* dw_w, const int32_t* dw_off, const int32_t* ksizes,
                            int32_t E, int32_t top_k, const int32_t* sel, const int32_t* csr_off,
                            const int32_t* csr_pair, void* dw_out, void* stream) {
    if (!x || !dw_w || !dw_off || !ksizes || !sel || !csr_off || !csr_pair || !dw_out) return YMK_E_BADARG;
    if (C % 4 || ldx % 4 || E < 1 || top_k < 1) return YMK_E_BADARG;
    MoeDwArgs a{x, dw_w, dw_off, ksizes, sel, csr_off, csr_pair, dw_out, B, H, W, C, ldx, E, top_k};
    const int64_t per = (int64_t)H * ((W + DW_R - 1) / DW_R) * (C / 4);
    if (per <= 0 || B <= 0) return YMK_OK;
    dim3 grid((unsigned)((per + 255) / 256), (unsigned)(B * top_k)), blk(256);
    if (dtype == YMK_F32)
        hipLaunchKernelGGL(moe_dw_kernel<float>, grid, blk, 0, (hipStream_t)stream, a);
    else if (dtype == YMK_BF16)
        hipLaunchKernelGGL(moe_dw_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, a);
    else
        return YMK_E_BADARG;
    return ymk_launch_status();
}
