// Fused network entry: stem Conv(3 -> C0, 3x3, stride 2) + SiLU followed by Conv(C0 -> C1, 3x3, stride 2) + SiLU in ONE kernel
// (YOLO-Master YAML rows 0 and 1, e.g. yolo-master.yaml backbone: `Conv [64, 3, 2]`, `Conv [128, 3, 2]` at the S width 32 / 64;
// reference: Conv.forward_fuse, ultralytics/nn/modules/conv.py:80-89, walked by BaseModel._predict_once, nn/tasks.py:182-218).
//
// Why: the stem's output is the largest tensor of the network (B x H/2 x W/2 x C0: 419 MB at 64 x 640 x 640 in bf16).  As two
// kernels it is written once and gathered back with 64-byte, stride-2 requests (the 3x3 stride-2 window of a 32-channel NHWC
// map); fused, it lives in LDS only: HBM sees the fp32 image once (315 MB) and row 1's output once (210 MB).
//
// One persistent workgroup (8 waves) per CU walks output tiles of 4 x 32 row-1 pixels of one image:
//   stage   the (4*4+3) x (4*32+4) x 3 fp32 input window -> LDS (zero outside the image = the stem's padding); the NEXT tile's
//           window is already in flight in registers while the current tile is computed;
//   stem    the (2*4+1) x (2*32+1) stem pixels the tile's 3x3/s2 windows cover, 16 pixels per wave step on the fp32 matrix cores
//           (same operand gathers, same MFMA sequence, same SiLU as stem_rows_kernel: the values are bit-identical to the unfused
//           stem's), rounded to bf16 into an LDS tile [pixel][C0] (zero outside the stem map = row 1's padding);
//   row 1   implicit GEMM from that tile: one 3x3 tap = one 32-deep MFMA step (C0 = 32), the wave's weight fragments resident in
//           registers for the whole kernel; bias + SiLU, one NHWC bf16 store.
// Halo cost: 585 stem pixels per 512 consumed (1.14x stem arithmetic); the input window overlap is served by L2.
//
// Stem arithmetic, two variants (SPLIT template parameter):
//   false  fp32 matrix cores (v_mfma_f32_16x16x4_f32), exactly stem_rows_kernel's sequence: bit-identical to the unfused pair, but
//          16 x 32-cycle MFMAs per 16 pixels make the stem phase the longest of the kernel (stage ablation: 25 % of its time);
//   true   (default) each fp32 operand is split into two bf16 parts, v = hi + lo with hi = bf16(v), lo = bf16(v - hi), and the product
//          is accumulated in fp32 as hi*hi + hi*lo + lo*hi on the bf16 matrix cores (3 x 16-cycle MFMAs per 16 pixels x 16 couts;
//          K = 27 fits one 32-deep step).  Dropped: lo*lo and the parts' own rounding, <= 2^-16 relative per product — 100x below
//          the bf16 rounding the stem map gets anyway (about 0.3 % of its values move by one bf16 ulp against the fp32 variant).
//          The image is split once, when the window is staged: an LDS word is (hi << 16) | lo.
// YMK_DISABLE bit 4096 selects the fp32 variant.
#include "ymk_common.h"

#define S2_TH 4
#define S2_TW 32
#define S2_SR (2 * S2_TH + 1)      // stem rows of a tile
#define S2_SC (2 * S2_TW + 1)      // stem columns of a tile
#define S2_NP (S2_SR * S2_SC)      // stem pixels of a tile (585)
#define S2_IR (4 * S2_TH + 3)      // input rows of a tile
#define S2_IC4 (S2_TW + 1)         // staged 16-byte column chunks per input row (window [4*ox0 - 4, 4*ox0 + 4*TW))
#define S2_IP 136                  // LDS pitch of an input row (floats)
#define S2_PP 80                   // LDS pitch of a stem pixel (bytes): 32 bf16 + 16 bytes
#define S2_NT 512
#define S2_IN_BYTES (3 * S2_IR * S2_IP * 4)
#define S2_LDS_BYTES (S2_IN_BYTES + S2_NP * S2_PP + 16)

typedef __bf16 s2_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void s2_mma_bf16(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = mfma16x16x32_h16(a, b, acc);
}
// 16 values of K per call: component v of lane group fc multiplies k = fc * 4 + v (csrc/igemm.h mma16<float>)
__device__ __forceinline__ void s2_mma_f32(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

struct Stem2Args {
    const float* x;      // [B][3][H][W]
    const float* wt0;    // [27][C0]  (k = (ky * 3 + kx) * 3 + c)
    const float* b0;     // [C0]
    const h16_t* w1;    // [C1][k1pad]  (k = (ky * 3 + kx) * C0 + c)
    const float* b1;     // [C1]
    h16_t* y;           // [B][H2][W2][ldy]
    int B, H, W, H1, W1, H2, W2, k1pad, ldy, tiles_x, tiles_y;
};

// fp32 -> (hi << 16) | lo, both parts in the build's 16-bit format (round to nearest even): v = hi + lo up to 2^-16 (bf16) / 2^-22 (f16)
__device__ __forceinline__ uint32_t s2_split(float v) {
    const uint32_t hi = pack_h16x2(v, 0.f) & 0xffffu;
    const float r = v - h16_to_f32((h16_t)hi);
    return (hi << 16) | (pack_h16x2(r, 0.f) & 0xffffu);
}

template <int C0, int C1, bool SPLIT>
__global__ __launch_bounds__(S2_NT) void stem_pair_kernel(Stem2Args a) {
    static_assert(C0 == 32 && C1 == 64, "tile shapes are written for the S width (32 -> 64)");
    constexpr int TM0 = C0 / 16;            // stem cout fragments
    constexpr int NLD = (3 * S2_IR * S2_IC4 + S2_NT - 1) / S2_NT;   // staging loads per thread (4)
    constexpr int NG = (S2_NP + 15) / 16;   // 16-pixel stem groups per tile (37)
    extern __shared__ u32x4 s2_smem[];   // [input window: 3 x IR x IP floats][stem tile: NP x PP bytes]
    float* sIn = reinterpret_cast<float*>(s2_smem);
    char* sStem = reinterpret_cast<char*>(s2_smem) + S2_IN_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int ntile = a.B * a.tiles_y * a.tiles_x;

    // ---- resident operands ----------------------------------------------------------------------------------------------------
    // stem weights and the LDS offset of each tap relative to the lane's pixel base.  fp32 variant: k = kk * 16 + fc * 4 + v
    // (four 16x16x4 steps per u32x4); split variant: k = fc * 8 + e, af0[i][0] = hi parts, af0[i][1] = lo parts (8 bf16 each)
    u32x4 af0[TM0][2];
    int off0[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = SPLIT ? fc * 8 + q : (q >> 2) * 16 + fc * 4 + (q & 3);
        const int kc = k < 27 ? k : 0;
        const int tap = kc / 3, c = kc - tap * 3;
        const int ky = tap / 3, kx = tap - ky * 3;
        off0[q] = (c * S2_IR + ky) * S2_IP + kx + 1;
#pragma unroll
        for (int i = 0; i < TM0; ++i) {
            const float wv = k < 27 ? a.wt0[k * C0 + i * 16 + fr] : 0.f;
            if (SPLIT) {
                const uint32_t w2 = s2_split(wv);
                uint32_t* h = reinterpret_cast<uint32_t*>(&af0[i][0]) + (q >> 1);
                uint32_t* l = reinterpret_cast<uint32_t*>(&af0[i][1]) + (q >> 1);
                if (q & 1) { *h |= w2 & 0xffff0000u; *l |= w2 << 16; }
                else { *h = w2 >> 16; *l = w2 & 0xffffu; }
            } else {
                reinterpret_cast<float*>(&af0[i][q >> 2])[q & 3] = wv;
            }
        }
    }
    f32x4 bv0[TM0];
#pragma unroll
    for (int i = 0; i < TM0; ++i) bv0[i] = *reinterpret_cast<const f32x4*>(a.b0 + i * 16 + fc * 4);
    const int ch = wave >> 2, pq = wave & 3;   // row-1 phase: cout half (2 fragments) x tile row
    u32x4 af1[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            // MFMA row fr of block i holds cout (fr >> 2) * 8 + i * 4 + (fr & 3) of the wave's 32: after the MFMAs a lane owns the EIGHT
            // consecutive couts fc * 8 ... + 7 of its pixel (one 16-byte store instead of two 8-byte ones)
            af1[i][tap] = *reinterpret_cast<const u32x4*>(a.w1 + (size_t)(ch * 32 + (fr >> 2) * 8 + i * 4 + (fr & 3)) * a.k1pad + tap * C0 + fc * 8);
    f32x4 bv1[2];
    const bool wide = (a.ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;   // 16-byte output stores possible
#pragma unroll
    for (int i = 0; i < 2; ++i) bv1[i] = *reinterpret_cast<const f32x4*>(a.b1 + ch * 32 + fc * 8 + i * 4);

    // ---- input window: global -> registers -> LDS -----------------------------------------------------------------------------
    u32x4 stg[NLD];
    auto gload = [&](int tile) {
        const int txi = tile % a.tiles_x, r0 = tile / a.tiles_x;
        const int tyi = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const int iy0 = 4 * tyi * S2_TH - 3, ix0 = 4 * txi * S2_TW - 4;
        const float* xb = a.x + (size_t)b * 3 * a.H * a.W;
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            const int i = t + l * S2_NT;
            const int rr = i / S2_IC4, q = i - rr * S2_IC4;   // rr = c * IR + r
            const int c = rr / S2_IR, r = rr - c * S2_IR;
            const int iy = iy0 + r, ix = ix0 + q * 4;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < 3 * S2_IR * S2_IC4 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const u32x4*>(xb + ((size_t)c * a.H + iy) * a.W + ix);
            stg[l] = v;
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntile) return;
    gload(tile);
    for (; tile < ntile; tile += gridDim.x) {
        const int txi = tile % a.tiles_x, r0 = tile / a.tiles_x;
        const int tyi = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const int oy0 = tyi * S2_TH, ox0 = txi * S2_TW;
        // every wave is past the previous tile's stem phase (barrier B below), so the window may be overwritten
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            const int i = t + l * S2_NT;
            if (i < 3 * S2_IR * S2_IC4) {
                const int rr = i / S2_IC4, q = i - rr * S2_IC4;
                u32x4 v = stg[l];
                if (SPLIT) {
                    v.x = s2_split(__uint_as_float(v.x)); v.y = s2_split(__uint_as_float(v.y));
                    v.z = s2_split(__uint_as_float(v.z)); v.w = s2_split(__uint_as_float(v.w));
                }
                *reinterpret_cast<u32x4*>(sIn + rr * S2_IP + q * 4) = v;
            }
        }
        __syncthreads();   // A: window visible; every wave has finished the previous tile's row-1 phase (stem tile free)
        if (tile + (int)gridDim.x < ntile) gload(tile + gridDim.x);   // in flight during both phases

        // ---- stem phase: 16 stem pixels per wave step -------------------------------------------------------------------------
        for (int g = wave; g < NG; g += S2_NT / 64) {
            const int p = g * 16 + fr;
            const int pc = p < S2_NP ? p : 0;
            const int u = pc / S2_SC, s = pc - u * S2_SC;
            const float* base = sIn + (2 * u) * S2_IP + 2 * s;
            u32x4 bf[2];
            if (SPLIT) {
                uint32_t w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = reinterpret_cast<const uint32_t*>(base)[off0[q]];
                uint32_t* h = reinterpret_cast<uint32_t*>(&bf[0]);
                uint32_t* l = reinterpret_cast<uint32_t*>(&bf[1]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h[q] = (w[2 * q] >> 16) | (w[2 * q + 1] & 0xffff0000u);
                    l[q] = (w[2 * q] & 0xffffu) | (w[2 * q + 1] << 16);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) reinterpret_cast<float*>(&bf[q >> 2])[q & 3] = base[off0[q]];
            }
            const int sy = 2 * oy0 - 1 + u, sx = 2 * ox0 - 1 + s;
            const bool inside = (unsigned)sy < (unsigned)a.H1 && (unsigned)sx < (unsigned)a.W1;
#pragma unroll
            for (int i = 0; i < TM0; ++i) {
                f32x4 acc = bv0[i];
                if (SPLIT) {
                    s2_mma_bf16(acc, af0[i][1], bf[0]);   // lo * hi
                    s2_mma_bf16(acc, af0[i][0], bf[1]);   // hi * lo
                    s2_mma_bf16(acc, af0[i][0], bf[0]);   // hi * hi
                } else {
                    s2_mma_f32(acc, af0[i][0], bf[0]);
                    s2_mma_f32(acc, af0[i][1], bf[1]);
                }
                u32x2 o = {0u, 0u};   // outside the stem map: row 1's zero padding
                if (inside) {
                    o.x = pack_h16x2(silu_f(acc.x), silu_f(acc.y));
                    o.y = pack_h16x2(silu_f(acc.z), silu_f(acc.w));
                }
                if (p < S2_NP) *reinterpret_cast<u32x2*>(sStem + p * S2_PP + (i * 16 + fc * 4) * 2) = o;
            }
        }
        __syncthreads();   // B: stem tile complete

        // ---- row-1 phase: this wave = couts [ch * 32, +32) x tile row pq (32 pixels = 2 fragments) ------------------------------
        f32x4 acc1[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) acc1[i][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            u32x4 bf1[2];
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int pp = (2 * pq + ky) * S2_SC + 2 * (pf * 16 + fr) + kx;
                bf1[pf] = *reinterpret_cast<const u32x4*>(sStem + pp * S2_PP + fc * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) s2_mma_bf16(acc1[i][pf], af1[i][tap], bf1[pf]);
        }
        const int oy = oy0 + pq;
        if (oy < a.H2) {
            h16_t* yrow = a.y + ((size_t)b * a.H2 + oy) * a.W2 * a.ldy;
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int ox = ox0 + pf * 16 + fr;
                if (ox < a.W2) {
                    float v8[8];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 v = acc1[i][pf] + bv1[i];
                        v8[i * 4 + 0] = silu_f(v.x); v8[i * 4 + 1] = silu_f(v.y); v8[i * 4 + 2] = silu_f(v.z); v8[i * 4 + 3] = silu_f(v.w);
                    }
                    h16_t* yo = yrow + (size_t)ox * a.ldy + ch * 32 + fc * 8;
                    if (wide) {
                        store_vec_f32(yo, v8);
                    } else {
                        store4(yo, v8[0], v8[1], v8[2], v8[3]);
                        store4(yo + 4, v8[4], v8[5], v8[6], v8[7]);
                    }
                }
            }
        }
    }
}

extern "C" int ymk_stem_pair_supported(int32_t dtype, int32_t Cin, int32_t C0, int32_t C1, int32_t k0, int32_t s0, int32_t k1, int32_t s1) {
    return dtype == YMK_BF16 && Cin == 3 && C0 == 32 && C1 == 64 && k0 == 3 && s0 == 2 && k1 == 3 && s1 == 2;
}

extern "C" int ymk_stem_pair(const float* x, int32_t B, int32_t H, int32_t W, const float* wt0, const float* b0, int32_t C0,
                             const void* w1, int32_t k1pad, const float* b1, int32_t C1, void* y, int32_t ldy, void* stream) {
    if (!x || !wt0 || !b0 || !w1 || !b1 || !y) return YMK_E_BADARG;
    if (!ymk_stem_pair_supported(YMK_BF16, 3, C0, C1, 3, 2, 3, 2) || k1pad < 9 * C0 || k1pad % 8 || ldy % 4 || ldy < C1) return YMK_E_BADARG;
    if ((W & 3) || ((uintptr_t)x & 15) || H < 1 || W < 4) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    Stem2Args a;
    a.x = x; a.wt0 = wt0; a.b0 = b0; a.w1 = (const h16_t*)w1; a.b1 = b1; a.y = (h16_t*)y;
    a.B = B; a.H = H; a.W = W;
    a.H1 = (H - 1) / 2 + 1; a.W1 = (W - 1) / 2 + 1;
    a.H2 = (a.H1 - 1) / 2 + 1; a.W2 = (a.W1 - 1) / 2 + 1;
    a.k1pad = k1pad; a.ldy = ldy;
    a.tiles_x = (a.W2 + S2_TW - 1) / S2_TW; a.tiles_y = (a.H2 + S2_TH - 1) / S2_TH;
    const int64_t ntile = (int64_t)B * a.tiles_x * a.tiles_y;
    if (ntile >= (1ll << 31)) return YMK_E_BADARG;
#ifdef YMK_MAX_BLOCKS
    const unsigned grid = (unsigned)(ntile < YMK_MAX_BLOCKS ? ntile : YMK_MAX_BLOCKS);
#else
    const unsigned grid = (unsigned)(ntile < 256 ? ntile : 256);   // one persistent workgroup per CU (109 KB of LDS)
#endif
    static YmkOncePerDevice attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pair_kernel<32, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)S2_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pair_kernel<32, 64, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)S2_LDS_BYTES);
        attr_once.done();
    }
    if (ymk_disabled() & 4096u)
        hipLaunchKernelGGL((stem_pair_kernel<32, 64, false>), dim3(grid), dim3(S2_NT), S2_LDS_BYTES, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((stem_pair_kernel<32, 64, true>), dim3(grid), dim3(S2_NT), S2_LDS_BYTES, (hipStream_t)stream, a);
    return ymk_launch_status();
}
