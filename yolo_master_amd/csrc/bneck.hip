// Bottleneck with two 3x3 convolutions of 64 channels as ONE kernel (16-bit):   y = [x +] SiLU(cv2(SiLU(cv1 x)))
// Reference: Bottleneck.forward (ultralytics/nn/modules/block.py:462-486) with k = (3, 3), e = 1.0, c1 = c2 = 64 — the two blocks inside
// every C3k of the detector's head (C3k2 [.., True] rows at 40 x 40 and 20 x 20: block.py:1074-1132), each convolution = conv + folded BN +
// SiLU (conv.py:80-89).
//
// Why.  On the 40^2 / 20^2 maps these are sixteen launches of 13-22 us for 7.5 / 1.9 GFLOP and 26 / 6.6 MB each: latency of their own short
// k-loops, and the 64-channel intermediate makes a round trip through HBM in between.  Here a persistent 8-wave workgroup owns an
// 8 x 16 pixel tile: x on the tile + 2 (240 pixels, zero outside the map = the padding of cv1) is staged in LDS once and is also the
// residual; h = SiLU(cv1 x) on the tile + 1 (180 pixels, ZERO outside the map = the padding of cv2) lives in LDS only; both weight sets
// stream from L2 straight into MFMA A fragments, one filter tap (64 x 64) at a time, the next tap's requested while this one multiplies.
//   conv1: wave (cb, pg) = cout block cb of 16, mid-pixel fragments pg, pg + 2, ... (6 of 12);  conv2: fragments pg, pg + 2, ... (4 of 8)
// Arithmetic per stage as ymk_conv2d's LDS-DMA core (16-bit operands, fp32 accumulation over K in (ky, kx, cin) order, bias, SiLU, one
// rounding to 16 bits, residual added in fp32 before the second rounding).
#include "ymk_common.h"

#define BN_TH 8
#define BN_TW 16
#define BN_XR (BN_TH + 4)
#define BN_XC (BN_TW + 4)
#define BN_NX (BN_XR * BN_XC)       // 240 staged input pixels
#define BN_MR (BN_TH + 2)
#define BN_MC (BN_TW + 2)
#define BN_NM (BN_MR * BN_MC)       // 180 pixels of h
#define BN_NF1 ((BN_NM + 15) / 16)  // 12 fragments of h
#define BN_NP (BN_TH * BN_TW)       // 128 tile pixels = 8 fragments (fragment j = tile row j)
#define BN_PITCH 160                // LDS pitch of a 64-channel pixel (bytes): 128 + 32 = 8 dwords modulo 16, conflict-free b128 fragment reads
#define BN_NT 512
#ifndef BN_WD
#define BN_WD 4                     // filter taps of weights in flight per wave (registers)
#endif
#ifndef BN_WPE
#define BN_WPE 2                    // waves per SIMD the register allocation is held to (2: one workgroup per CU)
#endif
#define BN_X_BYTES (BN_NX * BN_PITCH)
#define BN_H_BYTES (BN_NF1 * 16 * BN_PITCH)
#define BN_LDS_BYTES (BN_X_BYTES + BN_H_BYTES)

struct BneckArgs {
    const h16_t* x;      // [B][H][W][ldx], 64 channels
    const h16_t *w1, *w2;   // packed [64][kpad], K = (ky, kx, cin)
    const float *b1, *b2;
    h16_t* y;            // [B][H][W][ldy], 64 channels
    int B, H, W, ldx, ldy, k1pad, k2pad, add, tiles_x, tiles_y;
};

__device__ __forceinline__ u32x2 bn_pack(float a, float b, float c, float d) {
    u32x2 o;
    o.x = pack_h16x2(a, b);
    o.y = pack_h16x2(c, d);
    return o;
}

__global__ __launch_bounds__(BN_NT) __attribute__((amdgpu_waves_per_eu(BN_WPE))) void bottleneck_fused_kernel(BneckArgs a) {
    extern __shared__ u32x4 bn_smem[];
    char* sX = reinterpret_cast<char*>(bn_smem);
    char* sH = sX + BN_X_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int cb = wave & 3, pg = wave >> 2;
    const int ntile = a.B * a.tiles_y * a.tiles_x;

    const h16_t* w1row = a.w1 + (size_t)(cb * 16 + fr) * a.k1pad + fc * 8;
    const h16_t* w2row = a.w2 + (size_t)(cb * 16 + fr) * a.k2pad + fc * 8;
    const f32x4 bv1 = *reinterpret_cast<const f32x4*>(a.b1 + cb * 16 + fc * 4);
    const f32x4 bv2 = *reinterpret_cast<const f32x4*>(a.b2 + cb * 16 + fc * 4);

    // conv1: this lane's mid pixels (one per fragment) as byte offsets of their tap-(0,0) input pixel in the x tile
    int boff1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int p = (pg + 2 * i) * 16 + fr;
        p = p < BN_NM ? p : 0;
        const int u = p / BN_MC, v = p - u * BN_MC;
        boff1[i] = (u * BN_XC + v) * BN_PITCH + fc * 16;
    }

    // staging of the x tile: 240 pixels x 8 chunks of 16 bytes, four passes of 512 threads
    constexpr int NL = (BN_NX * 8 + BN_NT - 1) / BN_NT;
    u32x4 stg[NL];
    auto gload = [&](int tile) {
        const int txi = tile % a.tiles_x, r0 = tile / a.tiles_x;
        const int tyi = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const h16_t* xb = a.x + (size_t)b * a.H * a.W * a.ldx;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t + l * BN_NT;
            const int px = i >> 3, q = i & 7;
            const int u = px / BN_XC, v = px - u * BN_XC;
            const int iy = tyi * BN_TH - 2 + u, ix = txi * BN_TW - 2 + v;
            stg[l] = u32x4{0u, 0u, 0u, 0u};
            if (px < BN_NX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                stg[l] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * a.W + ix) * a.ldx + q * 8);
        }
    };
    // (no register prefetch of the next tile: its 16 registers would push the kernel over the 128 that two workgroups per CU allow, and
    // the other workgroup's convolutions cover this one's staging)
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int txi = tile % a.tiles_x, r0 = tile / a.tiles_x;
        const int tyi = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const int oy0 = tyi * BN_TH, ox0 = txi * BN_TW;
        gload(tile);
        __syncthreads();   // every wave is past the previous tile's reads of the x tile (residual) and of h
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t + l * BN_NT;
            if ((i >> 3) < BN_NX) *reinterpret_cast<u32x4*>(sX + (i >> 3) * BN_PITCH + (i & 7) * 16) = stg[l];
        }
        __syncthreads();

        // ---- cv1 on the tile + 1 -> h ----------------------------------------------------------------------------------------------------
        {
            f32x4 acc[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            // weights: BN_WD filter taps ahead in registers (a tap is 12 MFMAs per wave, ~100 ns; an L2 round trip is several times that)
            u32x4 wq[BN_WD][2];
#pragma unroll
            for (int d = 0; d < BN_WD; ++d) {
                wq[d][0] = *reinterpret_cast<const u32x4*>(w1row + d * 64);
                wq[d][1] = *reinterpret_cast<const u32x4*>(w1row + d * 64 + 32);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const u32x4 wa = wq[tap % BN_WD][0], wb = wq[tap % BN_WD][1];
                if (tap + BN_WD < 9) {
                    wq[tap % BN_WD][0] = *reinterpret_cast<const u32x4*>(w1row + (tap + BN_WD) * 64);
                    wq[tap % BN_WD][1] = *reinterpret_cast<const u32x4*>(w1row + (tap + BN_WD) * 64 + 32);
                }
                const int toff = ((tap / 3) * BN_XC + tap % 3) * BN_PITCH;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {   // three fragments at a time
                    u32x4 b0[3], b1[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        b0[i] = *reinterpret_cast<const u32x4*>(sX + boff1[hf * 3 + i] + toff);
                        b1[i] = *reinterpret_cast<const u32x4*>(sX + boff1[hf * 3 + i] + toff + 64);
                    }
#ifndef YMK_HOST_EMU
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[hf * 3 + i] = mfma16x16x32_h16(wa, b0[i], acc[hf * 3 + i]);
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[hf * 3 + i] = mfma16x16x32_h16(wb, b1[i], acc[hf * 3 + i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int p = (pg + 2 * i) * 16 + fr;
                if (p < BN_NM) {
                    const int u = p / BN_MC, v = p - u * BN_MC;
                    const int my = oy0 - 1 + u, mx = ox0 - 1 + v;
                    const bool inside = (unsigned)my < (unsigned)a.H && (unsigned)mx < (unsigned)a.W;
                    u32x2 o = {0u, 0u};
                    if (inside) o = bn_pack(silu_f(acc[i].x + bv1.x), silu_f(acc[i].y + bv1.y), silu_f(acc[i].z + bv1.z), silu_f(acc[i].w + bv1.w));
                    *reinterpret_cast<u32x2*>(sH + p * BN_PITCH + (cb * 16 + fc * 4) * 2) = o;
                }
            }
        }
        __syncthreads();
        // ---- cv2 on the tile, + x -> y -----------------------------------------------------------------------------------------------------
        {
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x4 wq[BN_WD][2];
#pragma unroll
            for (int d = 0; d < BN_WD; ++d) {
                wq[d][0] = *reinterpret_cast<const u32x4*>(w2row + d * 64);
                wq[d][1] = *reinterpret_cast<const u32x4*>(w2row + d * 64 + 32);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const u32x4 wa = wq[tap % BN_WD][0], wb = wq[tap % BN_WD][1];
                if (tap + BN_WD < 9) {
                    wq[tap % BN_WD][0] = *reinterpret_cast<const u32x4*>(w2row + (tap + BN_WD) * 64);
                    wq[tap % BN_WD][1] = *reinterpret_cast<const u32x4*>(w2row + (tap + BN_WD) * 64 + 32);
                }
                const int toff = ((tap / 3) * BN_MC + tap % 3) * BN_PITCH + fc * 16;
                u32x4 b0[4], b1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int base = ((pg + 2 * i) * BN_MC + fr) * BN_PITCH + toff;   // tile row pg + 2 i, column fr
                    b0[i] = *reinterpret_cast<const u32x4*>(sH + base);
                    b1[i] = *reinterpret_cast<const u32x4*>(sH + base + 64);
                }
#ifndef YMK_HOST_EMU
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma16x16x32_h16(wa, b0[i], acc[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma16x16x32_h16(wb, b1[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = pg + 2 * i;
                const int oy = oy0 + j, ox = ox0 + fr;
                float v0 = silu_f(acc[i].x + bv2.x), v1 = silu_f(acc[i].y + bv2.y), v2 = silu_f(acc[i].z + bv2.z), v3 = silu_f(acc[i].w + bv2.w);
                if (a.add) {   // the residual is the x tile itself (tile pixel (j, fr) = staged pixel (j + 2, fr + 2))
                    const u32x2 rx = *reinterpret_cast<const u32x2*>(sX + ((j + 2) * BN_XC + fr + 2) * BN_PITCH + (cb * 16 + fc * 4) * 2);
                    float r0, r1, r2, r3;
                    unpack_raw4(rx, r0, r1, r2, r3);
                    v0 = r0 + v0; v1 = r1 + v1; v2 = r2 + v2; v3 = r3 + v3;
                }
                if (oy < a.H && ox < a.W) store4(a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.ldy + cb * 16 + fc * 4, v0, v1, v2, v3);
            }
        }
    }
}

extern "C" int ymk_bottleneck_fused_supported(int32_t dtype, int32_t c1, int32_t c_mid, int32_t c2) {
    return dtype == YMK_BF16 && c1 == 64 && c_mid == 64 && c2 == 64;
}

extern "C" int ymk_bottleneck_fused(int32_t dtype, const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, const void* w1, int32_t k1pad,
                                    const float* b1, const void* w2, int32_t k2pad, const float* b2, int32_t add, void* y, int32_t ldy,
                                    void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y) return YMK_E_BADARG;
    if (!ymk_bottleneck_fused_supported(dtype, 64, 64, 64) || ldx < 64 || ldx % 8 || ldy < 64 || ldy % 4 || k1pad < 576 || k2pad < 576 ||
        (k1pad | k2pad) % 8)
        return YMK_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)y & 7)) return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    BneckArgs a;
    a.x = static_cast<const h16_t*>(x); a.w1 = static_cast<const h16_t*>(w1); a.w2 = static_cast<const h16_t*>(w2); a.b1 = b1; a.b2 = b2;
    a.y = static_cast<h16_t*>(y);
    a.B = B; a.H = H; a.W = W; a.ldx = ldx; a.ldy = ldy; a.k1pad = k1pad; a.k2pad = k2pad; a.add = add ? 1 : 0;
    a.tiles_x = (W + BN_TW - 1) / BN_TW; a.tiles_y = (H + BN_TH - 1) / BN_TH;
    const int64_t ntile = (int64_t)B * a.tiles_x * a.tiles_y;
    if (ntile >= (1ll << 31) || (int64_t)B * H * W * (ldx > ldy ? ldx : ldy) >= (1ll << 31)) return YMK_E_BADARG;
#ifdef YMK_MAX_BLOCKS
    const unsigned grid = (unsigned)(ntile < YMK_MAX_BLOCKS ? ntile : YMK_MAX_BLOCKS);
#else
    const int64_t slots = 256 * (BN_WPE >= 4 ? 2 : 1);   // persistent workgroups: one per CU, two where the registers allow (69 KB of LDS each)
    const unsigned grid = (unsigned)(ntile < slots ? ntile : slots);
#endif
    static YmkOncePerDevice attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BN_LDS_BYTES);
        attr_once.done();
    }
    hipLaunchKernelGGL(bottleneck_fused_kernel, dim3(grid), dim3(BN_NT), BN_LDS_BYTES, (hipStream_t)stream, a);
    return ymk_launch_status();
}
