// C3k2 block with one plain Bottleneck (c3k = False, n = 1, hidden width c = 32) as ONE kernel — YOLO-Master-S row 2
// (`C3k2 [256, False, 0.25]` at width 0.5: 64 -> 128 channels on the 160 x 160 map; reference: C3k2 / C2f.forward,
// ultralytics/nn/modules/block.py:293-325, 1074-1111, Bottleneck :462-486):
//
//     y1 = SiLU(cv1 x)            1x1, 64 -> 64      (a = y1[:, :32], b = y1[:, 32:])
//     h  = SiLU(m.cv1 b)          3x3, 32 -> 16
//     m  = b + SiLU(m.cv2 h)      3x3, 16 -> 32
//     y  = SiLU(cv2 [a | b | m])  1x1, 96 -> 128
//
// As four convolutions the block moves 1.57 GB at 64 x 160 x 160 (every intermediate written and read back, y1 twice); read-x-once /
// write-y-once is 0.63 GB.  Here a persistent workgroup (8 waves) owns an 8 x 32 output tile: y1 on the tile + 2 halo pixels, h on the
// tile + 1, m and y on the tile live in LDS only (105 KB); all four weight sets are resident in registers as MFMA A fragments
// (100 registers per lane).  Halo recompute: y1 on 432 pixels per 256 (1x1, cheap), h on 340.
//   phase 1  y1: B fragments straight from global (a pixel's 64 channels are K-contiguous), 16 pixels per wave step, all 4 cout fragments
//   phase 2  h : 3x3 over b from the y1 tile, one tap = one 32-deep MFMA step
//   phase 3  m : 3x3 over h, two taps per 32-deep step (16 channels each); residual b from the y1 tile
//   phase 4  y : K = [a | b | m] from the y1 and m tiles, one cout fragment per wave; bias + SiLU, NHWC store
// Zero padding of both 3x3 convolutions = zeros written for y1 / h pixels outside the map.  Arithmetic per stage as ymk_conv2d's
// (bf16 operands, fp32 accumulation in ascending K, bias, SiLU, [+ residual], one rounding to bf16).
#include "ymk_common.h"

#define CF_TH 8
#define CF_TW 32
#define CF_YR (CF_TH + 4)
#define CF_YC (CF_TW + 4)
#define CF_NY (CF_YR * CF_YC)      // 432 y1 pixels
#define CF_HR (CF_TH + 2)
#define CF_HC (CF_TW + 2)
#define CF_NH (CF_HR * CF_HC)      // 340 h pixels
#define CF_NM (CF_TH * CF_TW)      // 256 tile pixels
#define CF_YP 160                  // LDS pitch of a y1 pixel (bytes): 64 bf16 + 32 (b128 reads of 16 consecutive pixels are conflict-free)
#define CF_HP 32                   // ... of an h pixel: 16 bf16, dense
#define CF_MP 96                   // ... of an m pixel: 32 bf16 + 32
#define CF_NT 512
#define CF_Y_BYTES (CF_NY * CF_YP)
#define CF_H_BYTES (CF_NH * CF_HP + 64)
#define CF_M_BYTES (CF_NM * CF_MP)
#define CF_LDS_BYTES (CF_Y_BYTES + CF_H_BYTES + CF_M_BYTES)

typedef __bf16 cf_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void cf_mma(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = mfma16x16x32_h16(a, b, acc);
}
__device__ __forceinline__ u32x2 cf_pack_silu(const f32x4& v) {
    u32x2 o;
    o.x = pack_h16x2(silu_f(v.x), silu_f(v.y));
    o.y = pack_h16x2(silu_f(v.z), silu_f(v.w));
    return o;
}

struct C3k2fArgs {
    const h16_t* x;                       // [B][H][W][ldx], 64 channels
    const h16_t *w1, *wa, *wb, *w2;       // packed [Cout][Kpad] (K = (ky, kx, cin)): 64 x 64, 16 x 288, 32 x 144, 128 x 96
    const float *b1, *ba, *bb, *b2;
    h16_t* y;                             // [B][H][W][ldy], 128 channels
    int B, H, W, ldx, ldy, k1pad, kapad, kbpad, k2pad, tiles_x, tiles_y;
    float* gap_part;                       // [B][tiles_y * tiles_x][128] per-tile channel sums of y (or null)
    int* flags;                            // YMK_FLAG_NONFINITE_INPUT is raised when a sum is not finite (or null)
};

__global__ __launch_bounds__(CF_NT) void c3k2_fused_kernel(C3k2fArgs a) {
    extern __shared__ u32x4 cf_smem[];
    char* sY = reinterpret_cast<char*>(cf_smem);
    char* sH = sY + CF_Y_BYTES;
    char* sM = sH + CF_H_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int ntile = a.B * a.tiles_y * a.tiles_x;

    // ---- resident weights (MFMA A fragments: row = cout, 8 consecutive K per lane) ------------------------------------------------------
    u32x4 af1[4][2];
    f32x4 bv1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int s = 0; s < 2; ++s) af1[i][s] = *reinterpret_cast<const u32x4*>(a.w1 + (size_t)(i * 16 + fr) * a.k1pad + s * 32 + fc * 8);
        bv1[i] = *reinterpret_cast<const f32x4*>(a.b1 + i * 16 + fc * 4);
    }
    u32x4 afa[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) afa[tap] = *reinterpret_cast<const u32x4*>(a.wa + (size_t)fr * a.kapad + tap * 32 + fc * 8);
    const f32x4 bva = *reinterpret_cast<const f32x4*>(a.ba + fc * 4);
    const int cfb = wave & 1, rq = wave >> 1;   // phase 3: cout fragment x pair of tile rows
    u32x4 afb[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        // K = 9 taps x 16 channels = 144; the fifth step's upper half (taps 9) is beyond it: zero weights
        u32x4 v = {0u, 0u, 0u, 0u};
        if (s * 32 + fc * 8 < 144) v = *reinterpret_cast<const u32x4*>(a.wb + (size_t)(cfb * 16 + fr) * a.kbpad + s * 32 + fc * 8);
        afb[s] = v;
    }
    const f32x4 bvb = *reinterpret_cast<const f32x4*>(a.bb + cfb * 16 + fc * 4);
    u32x4 af2[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) af2[s] = *reinterpret_cast<const u32x4*>(a.w2 + (size_t)(wave * 16 + fr) * a.k2pad + s * 32 + fc * 8);
    const f32x4 bv2 = *reinterpret_cast<const f32x4*>(a.b2 + wave * 16 + fc * 4);

    // phase-1 operands of a tile: the wave's pixel groups' x fragments, straight from global into registers.  Loaded one tile ahead:
    // the next tile's requests are issued as soon as this tile's phase 1 has consumed the registers and land during phases 2-4.
    constexpr int NG1 = CF_NY / 16;                              // 27 pixel groups
    constexpr int NR1 = (NG1 + CF_NT / 64 - 1) / (CF_NT / 64);   // rounds per wave (4)
    u32x4 bx[NR1][2];
    unsigned inmask = 0;
    auto xload = [&](int tile) {
        const int txi = tile % a.tiles_x, r0 = tile / a.tiles_x;
        const int tyi = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const h16_t* xb = a.x + (size_t)b * a.H * a.W * a.ldx;
        inmask = 0;
#pragma unroll
        for (int r = 0; r < NR1; ++r) {
            const int g = wave + r * (CF_NT / 64);
            const int p = (g < NG1 ? g : 0) * 16 + fr;
            const int u = p / CF_YC, s = p - u * CF_YC;
            const int iy = tyi * CF_TH - 2 + u, ix = txi * CF_TW - 2 + s;
            const bool in = g < NG1 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            bx[r][0] = bx[r][1] = u32x4{0u, 0u, 0u, 0u};
            if (in) {
                const h16_t* px = xb + ((size_t)iy * a.W + ix) * a.ldx + fc * 8;
                bx[r][0] = *reinterpret_cast<const u32x4*>(px);
                bx[r][1] = *reinterpret_cast<const u32x4*>(px + 32);
                inmask |= 1u << r;
            }
        }
    };
    if ((int)blockIdx.x < ntile) xload(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int txi = tile % a.tiles_x, r0 = tile / a.tiles_x;
        const int tyi = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const int oy0 = tyi * CF_TH, ox0 = txi * CF_TW;

        // ---- phase 1: y1 = SiLU(cv1 x) on the tile + 2 ------------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < NR1; ++r) {
            const int g = wave + r * (CF_NT / 64);
            if (g >= NG1) break;
            const int p = g * 16 + fr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 acc = bv1[i];
                cf_mma(acc, af1[i][0], bx[r][0]);
                cf_mma(acc, af1[i][1], bx[r][1]);
                const u32x2 o = ((inmask >> r) & 1u) ? cf_pack_silu(acc) : u32x2{0u, 0u};   // outside the map: the 3x3's zero padding
                *reinterpret_cast<u32x2*>(sY + p * CF_YP + (i * 16 + fc * 4) * 2) = o;
            }
        }
        if (tile + (int)gridDim.x < ntile) xload(tile + gridDim.x);
        __syncthreads();

        // ---- phase 2: h = SiLU(3x3 over b) on the tile + 1 ---------------------------------------------------------------------------------
        for (int g = wave; g < (CF_NH + 15) / 16; g += CF_NT / 64) {
            const int p = g * 16 + fr;
            const int pc = p < CF_NH ? p : 0;
            const int u = pc / CF_HC, s = pc - u * CF_HC;
            const char* base = sY + (u * CF_YC + s) * CF_YP + 64 + fc * 16;
            f32x4 acc = bva;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                cf_mma(acc, afa[tap], *reinterpret_cast<const u32x4*>(base + (ky * CF_YC + kx) * CF_YP));
            }
            const int hy = oy0 - 1 + u, hx = ox0 - 1 + s;
            const bool inside = (unsigned)hy < (unsigned)a.H && (unsigned)hx < (unsigned)a.W;
            if (p < CF_NH) *reinterpret_cast<u32x2*>(sH + p * CF_HP + fc * 8) = inside ? cf_pack_silu(acc) : u32x2{0u, 0u};
        }
        __syncthreads();

        // ---- phase 3: m = b + SiLU(3x3 over h): this wave = couts [cfb * 16, +16) x tile rows 2 * rq, 2 * rq + 1 -----------------------------
        {
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = bvb;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                int tap = 2 * s + (fc >> 1);
                tap = tap > 8 ? 8 : tap;                              // (weights of the absent tenth tap are zero)
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 2 * rq + (j >> 1), xx = (j & 1) * 16 + fr;
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(sH + ((r + ky) * CF_HC + xx + kx) * CF_HP + (fc & 1) * 16);
                    cf_mma(acc[j], afb[s], bh);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 2 * rq + (j >> 1), xx = (j & 1) * 16 + fr;
                const u32x2 rb = *reinterpret_cast<const u32x2*>(sY + ((r + 2) * CF_YC + xx + 2) * CF_YP + 64 + (cfb * 16 + fc * 4) * 2);
                u32x2 o;
                o.x = pack_h16x2(h16lo(rb.x) + silu_f(acc[j].x), h16hi(rb.x) + silu_f(acc[j].y));
                o.y = pack_h16x2(h16lo(rb.y) + silu_f(acc[j].z), h16hi(rb.y) + silu_f(acc[j].w));
                *reinterpret_cast<u32x2*>(sM + (r * CF_TW + xx) * CF_MP + (cfb * 16 + fc * 4) * 2) = o;
            }
        }
        __syncthreads();

        // ---- phase 4: y = SiLU(cv2 [a | b | m]): this wave = couts [wave * 16, +16), every pixel of the tile ---------------------------------
        f32x4 gsum = {0.f, 0.f, 0.f, 0.f};   // channel sums of the STORED (bf16) values over this lane's pixels: the consumer's average pool
#pragma unroll 1
        for (int j0 = 0; j0 < 16; j0 += 4) {
            f32x4 acc[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj, r = j >> 1, xx = (j & 1) * 16 + fr;
                const char* py = sY + ((r + 2) * CF_YC + xx + 2) * CF_YP + fc * 16;
                acc[jj] = bv2;
                cf_mma(acc[jj], af2[0], *reinterpret_cast<const u32x4*>(py));
                cf_mma(acc[jj], af2[1], *reinterpret_cast<const u32x4*>(py + 64));
                cf_mma(acc[jj], af2[2], *reinterpret_cast<const u32x4*>(sM + (r * CF_TW + xx) * CF_MP + fc * 16));
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj, r = j >> 1, xx = (j & 1) * 16 + fr;
                const int oy = oy0 + r, ox = ox0 + xx;
                if (oy < a.H && ox < a.W) {
                    const u32x2 o = cf_pack_silu(acc[jj]);
                    *reinterpret_cast<u32x2*>(a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.ldy + wave * 16 + fc * 4) = o;
                    gsum.x += h16lo(o.x); gsum.y += h16hi(o.x); gsum.z += h16lo(o.y); gsum.w += h16hi(o.y);
                }
            }
        }
        if (a.gap_part) {   // fixed order: the lane's 16 pixels in tile order, then the 16 lanes of a channel quadruple pairwise
#pragma unroll
            for (int msk = 1; msk < 16; msk <<= 1) {
                gsum.x += __shfl_xor(gsum.x, msk); gsum.y += __shfl_xor(gsum.y, msk);
                gsum.z += __shfl_xor(gsum.z, msk); gsum.w += __shfl_xor(gsum.w, msk);
            }
            if (fr == 0) {
                *reinterpret_cast<f32x4*>(a.gap_part + ((size_t)b * a.tiles_y * a.tiles_x + tyi * a.tiles_x + txi) * 128 + wave * 16 + fc * 4) = gsum;
                if (a.flags && !(isfinite(gsum.x) && isfinite(gsum.y) && isfinite(gsum.z) && isfinite(gsum.w))) atomicOr(a.flags, YMK_FLAG_NONFINITE_INPUT);
            }
        }
        __syncthreads();   // the tiles are free for the next iteration
    }
}

extern "C" int ymk_c3k2_fused_supported(int32_t dtype, int32_t c1, int32_t c2, int32_t c, int32_t n, int32_t c3k, int32_t shortcut) {
    return dtype == YMK_BF16 && c1 == 64 && c2 == 128 && c == 32 && n == 1 && !c3k && shortcut;
}

extern "C" int32_t ymk_c3k2_fused_pool_chunks(int32_t H, int32_t W) { return ((H + CF_TH - 1) / CF_TH) * ((W + CF_TW - 1) / CF_TW); }

extern "C" int ymk_c3k2_fused_pooled(const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, const void* w1, int32_t k1pad, const float* b1,
                                     const void* wa, int32_t kapad, const float* ba, const void* wb, int32_t kbpad, const float* bb,
                                     const void* w2, int32_t k2pad, const float* b2, void* y, int32_t ldy, float* gap_part, int32_t* flags,
                                     void* stream) {
    if (!x || !w1 || !b1 || !wa || !ba || !wb || !bb || !w2 || !b2 || !y) return YMK_E_BADARG;
    if (ldx % 8 || ldx < 64 || ldy % 4 || ldy < 128 || k1pad < 64 || kapad < 288 || kbpad < 160 || k2pad < 96 || (k1pad | kapad | kbpad | k2pad) % 8)
        return YMK_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7) || ((uintptr_t)gap_part & 15)) return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    C3k2fArgs a;
    a.x = (const h16_t*)x; a.w1 = (const h16_t*)w1; a.wa = (const h16_t*)wa; a.wb = (const h16_t*)wb; a.w2 = (const h16_t*)w2;
    a.b1 = b1; a.ba = ba; a.bb = bb; a.b2 = b2; a.y = (h16_t*)y;
    a.B = B; a.H = H; a.W = W; a.ldx = ldx; a.ldy = ldy; a.k1pad = k1pad; a.kapad = kapad; a.kbpad = kbpad; a.k2pad = k2pad;
    a.tiles_x = (W + CF_TW - 1) / CF_TW; a.tiles_y = (H + CF_TH - 1) / CF_TH;
    a.gap_part = gap_part; a.flags = flags;
    const int64_t ntile = (int64_t)B * a.tiles_x * a.tiles_y;
    if (ntile >= (1ll << 31) || (int64_t)B * H * W * (ldx > ldy ? ldx : ldy) >= (1ll << 40)) return YMK_E_BADARG;
#ifdef YMK_MAX_BLOCKS
    const unsigned grid = (unsigned)(ntile < YMK_MAX_BLOCKS ? ntile : YMK_MAX_BLOCKS);
#else
    const unsigned grid = (unsigned)(ntile < 256 ? ntile : 256);   // one persistent workgroup per CU
#endif
    static YmkOncePerDevice attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&c3k2_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_LDS_BYTES);
        attr_once.done();
    }
    hipLaunchKernelGGL(c3k2_fused_kernel, dim3(grid), dim3(CF_NT), CF_LDS_BYTES, (hipStream_t)stream, a);
    return ymk_launch_status();
}

extern "C" int ymk_c3k2_fused(const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, const void* w1, int32_t k1pad, const float* b1,
                              const void* wa, int32_t kapad, const float* ba, const void* wb, int32_t kbpad, const float* bb, const void* w2,
                              int32_t k2pad, const float* b2, void* y, int32_t ldy, void* stream) {
    return ymk_c3k2_fused_pooled(x, ldx, B, H, W, w1, k1pad, b1, wa, kapad, ba, wb, kbpad, bb, w2, k2pad, b2, y, ldy, nullptr, nullptr, stream);
}
