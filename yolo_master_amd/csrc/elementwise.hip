// Layout / copy kernels (HBM-bound): nearest 2x upsample into a concat slice, channel-slice
// copy, NHWC -> NCHW fp32 export, and the Detect decode.
// -ffp-contract=off is set for this file: the decode must follow the reference's
// op-by-op fp32 arithmetic (no fused multiply-add contraction).
#include "ymk_common.h"

// ---- nn.Upsample(None, 2, "nearest") (yaml head rows 13/16) -------------------------
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H,
                                                        int W, int C, int ldx, int ldy) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = (int64_t)B * (2 * H) * (2 * W) * ncv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % ncv);
        int64_t p = i / ncv;
        const int ox = (int)(p % (2 * W)); p /= (2 * W);
        const int oy = (int)(p % (2 * H));
        const int b = (int)(p / (2 * H));
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * ldx + cv * VEC);
        *reinterpret_cast<u32x4*>(y + (((int64_t)b * 2 * H + oy) * 2 * W + ox) * ldy + cv * VEC) = v;
    }
}

extern "C" int ymk_upsample2x(int32_t dtype, const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                              int32_t ldx, int32_t ldy, void* stream) {
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (!x || !y || (dtype != YMK_F32 && dtype != YMK_BF16) || C % vec || ldx % vec || ldy % vec) return YMK_E_BADARG;
    const int64_t total = (int64_t)B * 4 * H * W * (C / vec);
    if (total <= 0) return YMK_OK;
    const int blocks = (int)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
    if (dtype == YMK_F32)
        hipLaunchKernelGGL(upsample2x_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (float*)y, B, H, W, C, ldx, ldy);
    else
        hipLaunchKernelGGL(upsample2x_kernel<h16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const h16_t*)x, (h16_t*)y, B, H, W, C, ldx, ldy);
    return ymk_launch_status();
}

// ---- Concat by channel-slice copy (conv.py:629-641) ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t npix,
                                                           int C, int ldx, int ldy) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = npix * ncv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % ncv);
        const int64_t p = i / ncv;
        *reinterpret_cast<u32x4*>(y + p * ldy + cv * VEC) = *reinterpret_cast<const u32x4*>(x + p * ldx + cv * VEC);
    }
}

extern "C" int ymk_copy_channels(int32_t dtype, const void* x, void* y, int64_t npix, int32_t C, int32_t ldx,
                                 int32_t ldy, void* stream) {
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (!x || !y || (dtype != YMK_F32 && dtype != YMK_BF16) || C % vec || ldx % vec || ldy % vec) return YMK_E_BADARG;
    const int64_t total = npix * (C / vec);
    if (total <= 0) return YMK_OK;
    const int blocks = (int)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
    if (dtype == YMK_F32)
        hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, npix, C, ldx, ldy);
    else
        hipLaunchKernelGGL(copy_channels_kernel<h16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const h16_t*)x, (h16_t*)y, npix, C, ldx, ldy);
    return ymk_launch_status();
}

// ---- gamma-residual of A2C2f at the l/x scales (block.py:1877-1879): out = res + gamma[c] * y --------------
// Reference op order: the product is rounded, then the sum (this file is compiled with -ffp-contract=off).
template <typename T>
__global__ __launch_bounds__(256) void scale_residual_kernel(const T* __restrict__ yv, const float* __restrict__ gamma,
                                                            const T* __restrict__ res, T* __restrict__ out, int64_t npix,
                                                            int C, int ldy, int ldr, int ldo) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = npix * ncv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % ncv);
        const int64_t p = i / ncv;
        float a[VEC], r[VEC];
        load_vec_f32(yv + p * ldy + cv * VEC, a);
        load_vec_f32(res + p * ldr + cv * VEC, r);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const float prod = gamma[cv * VEC + q] * a[q];
            a[q] = r[q] + prod;
        }
        store_vec_f32(out + p * ldo + cv * VEC, a);
    }
}

extern "C" int ymk_scale_residual(int32_t dtype, const void* y, const float* gamma, const void* residual, void* out,
                                  int64_t npix, int32_t C, int32_t ldy, int32_t ldr, int32_t ldo, void* stream) {
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (!y || !gamma || !residual || !out || (dtype != YMK_F32 && dtype != YMK_BF16) || C % vec || ldy % vec || ldr % vec ||
        ldo % vec)
        return YMK_E_BADARG;
    const int64_t total = npix * (C / vec);
    if (total <= 0) return YMK_OK;
    const int blocks = (int)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
    if (dtype == YMK_F32)
        hipLaunchKernelGGL(scale_residual_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)y,
                           gamma, (const float*)residual, (float*)out, npix, C, ldy, ldr, ldo);
    else
        hipLaunchKernelGGL(scale_residual_kernel<h16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const h16_t*)y,
                           gamma, (const h16_t*)residual, (h16_t*)out, npix, C, ldy, ldr, ldo);
    return ymk_launch_status();
}

// ---- NHWC -> NCHW fp32 via a 64x64 LDS transpose tile -------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int HW, int C,
                                                          int ldx) {
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
        if (c < C && p < HW) y[((size_t)b * C + c) * HW + p] = tile[tx][r];
    }
}

extern "C" int ymk_nhwc_to_nchw_f32(int32_t dtype, const void* x, float* y, int32_t B, int32_t HW, int32_t C,
                                    int32_t ldx, void* stream) {
    if (!x || !y || (dtype != YMK_F32 && dtype != YMK_BF16)) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0 || C <= 0) return YMK_OK;
    if (B > 65535) return YMK_E_BADARG;
    dim3 grid((HW + 63) / 64, (C + 63) / 64, B), blk(256);
    if (dtype == YMK_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)x, y, HW, C, ldx);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<h16_t>, grid, blk, 0, (hipStream_t)stream, (const h16_t*)x, y, HW, C,
                           ldx);
    return ymk_launch_status();
}

// ---- Detect decode (head.py:173-194, block.py:81-84, tal.py:398-423) ------------------
// One workgroup = 64 anchors of one image.  The [64][4*reg_max] box logits and [64][nc]
// class logits are read coalesced (anchor-major rows), transposed through LDS, and written
// as y[b][channel][anchor] (anchor-contiguous).  Arithmetic, in the reference's order:
//   p = softmax_16(logits); dist = sum_i i*p_i; x1y1 = anchor - lt; x2y2 = anchor + rb;
//   c = (x1y1 + x2y2)/2; wh = x2y2 - x1y1; box = (c, wh) * stride; score = sigmoid(cls).
// best_conf / best_cls (nullable, [B][A]): the largest class score of every anchor and its class (first maximum in class order) — what
// the single-label candidate filter of non_max_suppression computes from y (utils/nms.py:124-129); handed to ymk_nms_batched they
// save its pass over the nc class rows.  The value is the float stored in y.
__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ box, const float* __restrict__ cls,
                                                           float* __restrict__ y, int Hl, int Wl, int reg_max, int nc,
                                                           int ldc, float stride, int a_off, int A, float* __restrict__ best_conf,
                                                           int* __restrict__ best_cls) {
    extern __shared__ float sm[];  // cls tile [64][nc+1], dist [64][4], box tile [256][17] (reg_max == 16)
    const int HW = Hl * Wl;
    const int b = blockIdx.y;
    const int a0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    float* scls = sm;
    float* sdist = sm + 64 * (nc + 1);
    float* sbox = sdist + 64 * 4;
    const int na = min(64, HW - a0);
    // class logits: coalesced read of na*nc floats (16-byte loads when the row length allows)
    const float* cb = cls + ((size_t)b * HW + a0) * ldc;
    if (ldc != nc) {
        // class rows padded to a 16-byte multiple (nc % 4 != 0: the tail conv writes ldc = 4*ceil(nc/4) channels)
        const int q = ldc >> 2;
        for (int i = t; i < na * q; i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(cb)[i];
            const int r = i / q, c = 4 * (i - r * q);
            float* d = scls + r * (nc + 1) + c;
            d[0] = v.x;   // c < nc always (ldc - nc < 4)
            if (c + 1 < nc) d[1] = v.y;
            if (c + 2 < nc) d[2] = v.z;
            if (c + 3 < nc) d[3] = v.w;
        }
    } else if ((nc & 3) == 0) {
        for (int i = t; i < na * nc / 4; i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(cb)[i];
            const int r = (4 * i) / nc, c = 4 * i - r * nc;  // nc % 4 == 0: the four values share a row
            float* d = scls + r * (nc + 1) + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int i = t; i < na * nc; i += 256) {
            const int r = i / nc, c = i - r * nc;
            scls[r * (nc + 1) + c] = cb[i];
        }
    }
    // DFL: thread (anchor r = t/4, side s = t%4) reduces reg_max bins
    if (reg_max == 16) {
        // the 64 x 64 box logits arrive coalesced and are re-read per (anchor, side) from LDS (row stride 17)
        const float* bb = box + ((size_t)b * HW + a0) * 64;
        for (int i = t; i < na * 16; i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(bb)[i];
            float* d = sbox + (i >> 2) * 17 + (i & 3) * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        if ((t >> 2) < na) {
            float e[16];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) { e[i] = sbox[t * 17 + i]; mx = fmaxf(mx, e[i]); }
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { e[i] = expf(e[i] - mx); den += e[i]; }
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) d += (float)i * (e[i] / den);
            sdist[t] = d;
        }
    } else {
        const int r = t >> 2, s = t & 3;
        if (r < na) {
            const float* p = box + ((size_t)b * HW + a0 + r) * (4 * reg_max) + s * reg_max;
            float mx = -INFINITY;
            for (int i = 0; i < reg_max; ++i) mx = fmaxf(mx, p[i]);
            float den = 0.f;
            for (int i = 0; i < reg_max; ++i) den += expf(p[i] - mx);
            float d = 0.f;
            for (int i = 0; i < reg_max; ++i) d += (float)i * (expf(p[i] - mx) / den);
            sdist[r * 4 + s] = d;
        }
    }
    __syncthreads();
    float* yb = y + (size_t)b * (4 + nc) * A + a_off + a0;
    if (t < 64 && t < na) {
        const int a = a0 + t;
        const float ax = (float)(a % Wl) + 0.5f, ay = (float)(a / Wl) + 0.5f;
        const float l = sdist[t * 4 + 0], tp = sdist[t * 4 + 1], r = sdist[t * 4 + 2], bt = sdist[t * 4 + 3];
        const float x1 = ax - l, y1 = ay - tp, x2 = ax + r, y2 = ay + bt;
        yb[0 * (size_t)A + t] = ((x1 + x2) / 2.0f) * stride;
        yb[1 * (size_t)A + t] = ((y1 + y2) / 2.0f) * stride;
        yb[2 * (size_t)A + t] = (x2 - x1) * stride;
        yb[3 * (size_t)A + t] = (y2 - y1) * stride;
    }
    for (int i = t; i < 64 * nc; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r < na) {
            const float v = scls[r * (nc + 1) + c];
            const float p = 1.0f / (1.0f + expf(-v));
            yb[(size_t)(4 + c) * A + r] = p;
            if (best_conf) scls[r * (nc + 1) + c] = p;     // (each element is read and rewritten by the same thread)
        }
    }
    if (best_conf) {
        __syncthreads();
        if (t < na) {
            float best = scls[t * (nc + 1)];
            int bi = 0;
            for (int c = 1; c < nc; ++c) {
                const float v = scls[t * (nc + 1) + c];
                if (v > best) { best = v; bi = c; }        // the first maximum wins, as the reference's max / argmax over the class axis
            }
            best_conf[(size_t)b * A + a_off + a0 + t] = best;
            best_cls[(size_t)b * A + a_off + a0 + t] = bi;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Detect box branch tail as ONE kernel (16-bit operands): Conv2d(c2 = 64, 4 * reg_max = 64, 1) + bias (head.py:111-112: the last
// module of cv2[i]) -> DFL softmax expectation (block.py:57-79) -> dist2bbox xywh * stride (head.py:173-194, tal.py:388-399) -> rows
// 0..3 of y.  The fp32 box logits live in registers / LDS only (raw == nullptr) — as separate kernels they made a 256-byte-per-anchor
// round trip through HBM between the 1x1 convolution and detect_decode_kernel.  Arithmetic: the 1x1 is the two MFMA k-steps + bias of
// the tiled convolution cores, the DFL / box code is detect_decode_kernel's line for line, so y is bit-identical to the unfused path.
// Workgroup = 64 anchors of one image (4 waves x one 16-anchor MFMA fragment x the four 16-bin sides).
#define DBT_GROUPS 4   // 64-anchor groups per workgroup
__global__ __launch_bounds__(256) void detect_box_tail_kernel(const h16_t* __restrict__ x, int ldx, const h16_t* __restrict__ w, int kpad,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             float* __restrict__ raw, int Hl, int Wl, int nc, float stride, int a_off,
                                                             int A) {
    __shared__ float sbox[256 * 17];
    __shared__ float sdist[256];
    const int HW = Hl * Wl, b = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fc = lane >> 4;
    const int al = wave * 16 + fr;                                       // this lane's anchor of a 64-anchor group (MFMA column)
    // the 64 x 64 weights stay in registers (MFMA A fragments) for the DBT_GROUPS groups of 64 anchors this workgroup walks
    u32x4 w0[4], w1[4];
    f32x4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                        // side i = output channels 16 i .. 16 i + 15
        const h16_t* wr = w + (size_t)(i * 16 + fr) * kpad;
        w0[i] = *reinterpret_cast<const u32x4*>(wr + fc * 8);
        w1[i] = *reinterpret_cast<const u32x4*>(wr + 32 + fc * 8);
        bv[i] = *reinterpret_cast<const f32x4*>(bias + i * 16 + fc * 4);
    }
    auto xload = [&](int a0, u32x4& b0, u32x4& b1) {
        const h16_t* xp = x + ((size_t)b * HW + min(a0 + al, HW - 1)) * ldx;
        b0 = *reinterpret_cast<const u32x4*>(xp + fc * 8);
        b1 = *reinterpret_cast<const u32x4*>(xp + 32 + fc * 8);
    };
    u32x4 b0, b1;
    int a0 = blockIdx.x * (64 * DBT_GROUPS);
    xload(a0, b0, b1);
    for (int g = 0; g < DBT_GROUPS && a0 < HW; ++g, a0 += 64) {
        const int na = min(64, HW - a0);
        u32x4 n0 = b0, n1 = b1;
        if (g + 1 < DBT_GROUPS && a0 + 64 < HW) xload(a0 + 64, n0, n1);   // the next group's operands travel under this group's decode
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mfma16x16x32_h16(w0[i], b0, acc);
            acc = mfma16x16x32_h16(w1[i], b1, acc);
            const float v0 = acc.x + bv[i].x, v1 = acc.y + bv[i].y, v2 = acc.z + bv[i].z, v3 = acc.w + bv[i].w;   // bins 4 fc .. 4 fc + 3 of side i
            float* d = sbox + (al * 4 + i) * 17 + fc * 4;
            d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
            if (raw && al < na) store4(raw + ((size_t)b * HW + a0 + al) * 64 + i * 16 + fc * 4, v0, v1, v2, v3);
        }
        __syncthreads();
        if ((t >> 2) < na) {   // thread (anchor t / 4, side t % 4): detect_decode_kernel's DFL
            float e[16];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) { e[i] = sbox[t * 17 + i]; mx = fmaxf(mx, e[i]); }
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { e[i] = expf(e[i] - mx); den += e[i]; }
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) d += (float)i * (e[i] / den);
            sdist[t] = d;
        }
        __syncthreads();   // (also: every thread is past its reads of sbox before the next group overwrites it)
        if (t < na) {
            float* yb = y + (size_t)b * (4 + nc) * A + a_off + a0;
            const int a = a0 + t;
            const float ax = (float)(a % Wl) + 0.5f, ay = (float)(a / Wl) + 0.5f;
            const float l = sdist[t * 4 + 0], tp = sdist[t * 4 + 1], r = sdist[t * 4 + 2], bt = sdist[t * 4 + 3];
            const float x1 = ax - l, y1 = ay - tp, x2 = ax + r, y2 = ay + bt;
            yb[0 * (size_t)A + t] = ((x1 + x2) / 2.0f) * stride;
            yb[1 * (size_t)A + t] = ((y1 + y2) / 2.0f) * stride;
            yb[2 * (size_t)A + t] = (x2 - x1) * stride;
            yb[3 * (size_t)A + t] = (y2 - y1) * stride;
        }
        b0 = n0; b1 = n1;
        // (sdist is rewritten behind the next group's first barrier, which every thread reaches after these reads)
    }
}

extern "C" int ymk_detect_box_tail_supported(int32_t dtype, int32_t cin, int32_t reg_max, int32_t nc) {
    return dtype == YMK_BF16 && cin == 64 && reg_max == 16 && nc >= 1 && nc <= 96;
}

// x [B][Hl][Wl][ldx] (64 channels, 16-bit), w packed [64][kpad] as for ymk_conv2d, bias fp32 [64]; y fp32 [B][4 + nc][A_total], rows 0..3 of
// the level's anchor range written; raw: nullptr or fp32 [B][Hl][Wl][64], the box logits (what the 1x1 convolution alone would return).
extern "C" int ymk_detect_box_tail(int32_t dtype, const void* x, int32_t ldx, int32_t B, int32_t Hl, int32_t Wl, const void* w, int32_t kpad,
                                   const float* bias, int32_t reg_max, int32_t nc, float stride, int32_t a_off, int32_t A_total, float* y,
                                   float* raw, void* stream) {
    if (!x || !w || !bias || !y || nc < 1) return YMK_E_BADARG;
    if (!ymk_detect_box_tail_supported(dtype, 64, reg_max, 1) || ldx < 64 || ldx % 8 || kpad < 64 || kpad % 8) return YMK_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || (raw && ((uintptr_t)raw & 15))) return YMK_E_BADARG;
    const int HW = Hl * Wl;
    if (B <= 0 || HW <= 0) return YMK_OK;
    if (B > 65535 || a_off < 0 || a_off + HW > A_total) return YMK_E_BADARG;
    hipLaunchKernelGGL(detect_box_tail_kernel, dim3((HW + 64 * DBT_GROUPS - 1) / (64 * DBT_GROUPS), B), dim3(256), 0, (hipStream_t)stream, static_cast<const h16_t*>(x), ldx,
                       static_cast<const h16_t*>(w), kpad, bias, y, raw, Hl, Wl, nc, stride, a_off, A_total);
    return ymk_launch_status();
}

extern "C" int ymk_detect_decode(const float* box_l, const float* cls_l, float* y, int32_t B, int32_t Hl, int32_t Wl,
                                 int32_t reg_max, int32_t nc, int32_t ldc, float stride, int32_t a_off, int32_t A_total,
                                 float* best_conf, int32_t* best_cls, void* stream) {
    if (!box_l || !cls_l || !y || reg_max < 1 || nc < 1 || (best_conf == nullptr) != (best_cls == nullptr)) return YMK_E_BADARG;
    if (ldc != nc && (ldc < nc || (ldc & 3) || ldc - nc >= 4)) return YMK_E_BADARG;
    const int HW = Hl * Wl;
    if (B <= 0 || HW <= 0) return YMK_OK;
    if (B > 65535 || a_off + HW > A_total) return YMK_E_BADARG;
    const size_t shm = (size_t)(64 * (nc + 1) + 64 * 4 + 256 * 17) * sizeof(float);
    if (shm > 64 * 1024) return YMK_E_BADARG;
    dim3 grid((HW + 63) / 64, B), blk(256);
    hipLaunchKernelGGL(detect_decode_kernel, grid, blk, shm, (hipStream_t)stream, box_l, cls_l, y, Hl, Wl, reg_max, nc,
                       ldc, stride, a_off, A_total, best_conf, best_cls);
    return ymk_launch_status();
}
