// Depthwise k x k convolution, NHWC, stride 1, pad k/2 — LDS-tiled sliding-window stencil.
//
// Workgroup = 16 x 32 output pixels x 16 channels.
// The (16+k-1) x (32+k-1) input halo is staged once in LDS with coalesced 16-byte loads (pixel stride
// padded to 40 (bf16) / 80 (fp32) bytes so the four x-strips of a wave land on disjoint bank groups), the k*k filter
// taps of the channel block are staged as fp32.  Each thread owns 4 channels x 8 consecutive output
// pixels of a row: per filter row it streams 8+k-1 LDS vectors once and keeps the k taps in registers,
// accumulating with packed fp32 FMAs (v_pk_fma_f32 on two channel pairs).  HBM-bound by design (the
// stencil has no contraction shared between channels, so MFMA does not apply); the halo re-read
// ((16+k-1)(32+k-1)/512 = 1.9x at k=9) is served by L2.
//
// Reference semantics: DWConv (ultralytics/nn/modules/conv.py:185-199), AAttn.pe
// (nn/modules/block.py:1688,1731), DepthwiseSeparableConv.depthwise (nn/modules/moe/experts.py:283-292)
// dispatched per retained (image, expert) pair as in ES_MOE._sparse_forward (moe/modules.py:690-697).
#include "ymk_common.h"

#define DW_R 8
#define DW_TH 16
#define DW_TW 32

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct DwTile {
    static constexpr int CB = 16;                             // channels per workgroup
    static constexpr int CPP = CB * (int)sizeof(T) / 16;      // 16-byte chunks per staged pixel (2 bf16 / 4 fp32)
    static constexpr int PSB = CB * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16);  // padded LDS pixel stride (bytes)
    static constexpr int NCG = CB / 4;                        // 4-channel groups
    static constexpr int ROWL = 256 / (NCG * (DW_TW / DW_R)); // row lanes
    static constexpr int RPT = DW_TH / ROWL;                  // rows per thread
    static constexpr size_t lds_bytes(int K) {
        return (size_t)(DW_TH + K - 1) * (DW_TW + K - 1) * PSB + (size_t)K * K * CB * sizeof(float);
    }
};

// lds_ld4 returns channels as two register pairs: fp32 (c0,c1),(c2,c3); bf16 (c0,c2),(c1,c3).
template <typename T> struct DwPair;  // position of channel j (0..3) inside the (a.x, a.y, b.x, b.y) quadruple
template <> struct DwPair<float> { static constexpr int pos[4] = {0, 1, 2, 3}; };
template <> struct DwPair<bf16_t> { static constexpr int pos[4] = {0, 2, 1, 3}; };
__device__ __forceinline__ void lds_ld4(const char* p, f32x2& a, f32x2& b, float) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    a = f32x2{t.x, t.y}; b = f32x2{t.z, t.w};
}
// bf16: the two packed words (c0,c1),(c2,c3) are widened with VECTOR shifts/masks so that the results land in
// register pairs directly: a = (c0, c2), b = (c1, c3) (channel pairing differs from fp32: see DW_PAIR)
__device__ __forceinline__ void lds_ld4(const char* p, f32x2& a, f32x2& b, bf16_t) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    a = __builtin_bit_cast(f32x2, t << 16);
    b = __builtin_bit_cast(f32x2, t & 0xffff0000u);
}

// consecutive logical tiles (the channel blocks of one pixel tile, then the neighbouring pixel tile) run on the
// same XCD back to back, so the 128-byte lines shared by channel blocks / halos hit in that XCD's L2
__device__ __forceinline__ unsigned xcd_remap_dw(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

struct DwEpi {  // epilogue description shared by the plain and the ES-MoE launchers
    const float* bias;  // [C] or null
    const void* res;    // residual view or null
    int ldr, act;
};

// xb/ob: base of this image's input / output view; w: [K*K][C] filter; tile = blockIdx-derived
template <typename T, int K>
__device__ __forceinline__ void dw_tile(const T* __restrict__ xb, int H, int W, int C, int ldx,
                                        const T* __restrict__ w, T* __restrict__ ob, int ldy, int tile,
                                        const DwEpi& ep, const T* __restrict__ rb, char* smem) {
    using D = DwTile<T>;
    constexpr int P = K / 2, HT = DW_TH + K - 1, WT = DW_TW + K - 1;
    constexpr int VEC = 16 / (int)sizeof(T);
    const int t = threadIdx.x;
    const int ncb = (C + D::CB - 1) / D::CB;
    const int tiles_x = (W + DW_TW - 1) / DW_TW;
    const int cb = tile % ncb;
    const int tx = (tile / ncb) % tiles_x, ty = tile / (ncb * tiles_x);
    const int c0 = cb * D::CB, ty0 = ty * DW_TH, tx0 = tx * DW_TW;
    float* wsm = reinterpret_cast<float*>(smem + (size_t)HT * WT * D::PSB);

    // ---- stage the halo tile (coalesced: 4 lanes x 16 B per pixel) and the filter block.
    // All global loads of a thread are issued back to back into registers BEFORE the first LDS write:
    // a load->write->load chain would serialise ~15 HBM/L2 round trips per workgroup.
    constexpr int CPP = D::CPP;
    constexpr int NL = (HT * WT * CPP + 255) / 256;
    u32x4 stg[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int i = t + l * 256;
        const int pix = i / CPP, q = i % CPP;
        const int hy = pix / WT, hx = pix - hy * WT;
        const int iy = ty0 - P + hy, ix = tx0 - P + hx;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (i < HT * WT * CPP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && c0 + q * VEC < C)
            v = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * W + ix) * ldx + c0 + q * VEC);
        stg[l] = v;
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int i = t + l * 256;
        if (i < HT * WT * CPP) {
            u32x2* d = reinterpret_cast<u32x2*>(smem + (size_t)(i / CPP) * D::PSB + (i % CPP) * 16);
            d[0] = u32x2{stg[l].x, stg[l].y};
            d[1] = u32x2{stg[l].z, stg[l].w};
        }
    }
    for (int i = t; i < K * K * D::CB; i += 256) {
        const int tap = i / D::CB, c = i - tap * D::CB;
        // store channel j of each 4-group at its register-quadruple position (see lds_ld4)
        wsm[tap * D::CB + (c & ~3) + DwPair<T>::pos[c & 3]] = (c0 + c) < C ? to_f32(w[(size_t)tap * C + c0 + c]) : 0.f;
    }
    __syncthreads();

    const int cg = t % D::NCG;
    const int strip = (t / D::NCG) % (DW_TW / DW_R);
    const int rl = t / (D::NCG * (DW_TW / DW_R));
    const int x0 = strip * DW_R;
    if (c0 + cg * 4 >= C) return;  // partial last channel block (C need only be a multiple of 16 bytes)
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(ep.bias + c0 + cg * 4);
        bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
    }
    constexpr bool PRECISE = sizeof(T) == 4;
#pragma unroll 1
    for (int rr = 0; rr < D::RPT; ++rr) {
        const int y = rl * D::RPT + rr;
        const int gy = ty0 + y;
        if (gy >= H) break;
        f32x2 acc[DW_R][2];
#pragma unroll
        for (int r = 0; r < DW_R; ++r) { acc[r][0] = f32x2{0.f, 0.f}; acc[r][1] = f32x2{0.f, 0.f}; }
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            f32x2 wr[K][2];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(wsm + (ky * K + kx) * D::CB + cg * 4);
                wr[kx][0] = f32x2{t4.x, t4.y}; wr[kx][1] = f32x2{t4.z, t4.w};
            }
            const char* row = smem + ((size_t)(y + ky) * WT + x0) * D::PSB + cg * 4 * sizeof(T);
#pragma unroll
            for (int j = 0; j < DW_R + K - 1; ++j) {
                f32x2 va, vb;
                lds_ld4(row + (size_t)j * D::PSB, va, vb, T{});
#pragma unroll
                for (int r = 0; r < DW_R; ++r) {
                    const int kx = j - r;  // compile-time after unrolling
                    if (kx >= 0 && kx < K) {
                        acc[r][0] = __builtin_elementwise_fma(va, wr[kx][0], acc[r][0]);
                        acc[r][1] = __builtin_elementwise_fma(vb, wr[kx][1], acc[r][1]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < DW_R; ++r) {
            const int gx = tx0 + x0 + r;
            if (gx >= W) break;
            const size_t pix = (size_t)gy * W + gx;
            const float q4[4] = {acc[r][0].x, acc[r][0].y, acc[r][1].x, acc[r][1].y};
            float v[4] = {q4[DwPair<T>::pos[0]] + bv[0], q4[DwPair<T>::pos[1]] + bv[1], q4[DwPair<T>::pos[2]] + bv[2],
                          q4[DwPair<T>::pos[3]] + bv[3]};
            if (ep.act == YMK_ACT_SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = PRECISE ? silu_exact(v[q]) : silu_f(v[q]);
            }
            if (rb) {
                float r0, r1, r2, r3;
                load4(rb + pix * ep.ldr + c0 + cg * 4, r0, r1, r2, r3);
                v[0] = r0 + v[0]; v[1] = r1 + v[1]; v[2] = r2 + v[2]; v[3] = r3 + v[3];
            }
            store4(ob + pix * ldy + c0 + cg * 4, v[0], v[1], v[2], v[3]);
        }
    }
}

struct DwArgs {
    const void* x;
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int B, H, W, C, ldx, ldy, ldr, act, tiles;  // tiles per image
};

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const size_t img = (size_t)b * a.H * a.W;
    DwEpi ep{a.bias, a.res, a.ldr, a.act};
    const T* rb = a.res ? reinterpret_cast<const T*>(a.res) + img * a.ldr : nullptr;
    dw_tile<T, K>(reinterpret_cast<const T*>(a.x) + img * a.ldx, a.H, a.W, a.C, a.ldx,
                  reinterpret_cast<const T*>(a.w), reinterpret_cast<T*>(a.y) + img * a.ldy, a.ldy,
                  (int)xcd_remap_dw(blockIdx.x, gridDim.x), ep, rb, smem);
}

template <typename T, int K>
static void launch_dw_k(const DwArgs& a, hipStream_t s) {
    const size_t shm = DwTile<T>::lds_bytes(K);
    static bool attr_set = false;
    if (!attr_set && shm > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_kernel<T, K>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv_kernel<T, K>), dim3(a.tiles, a.B), dim3(256), shm, s, a);
}

template <typename T>
static int launch_dw(DwArgs a, int k, hipStream_t s) {
    using D = DwTile<T>;
    constexpr int VEC = 16 / (int)sizeof(T);
    if (a.C % VEC) return YMK_E_BADARG;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) return YMK_OK;
    if (a.B > 65535) return YMK_E_BADARG;
    a.tiles = ((a.H + DW_TH - 1) / DW_TH) * ((a.W + DW_TW - 1) / DW_TW) * ((a.C + D::CB - 1) / D::CB);
    switch (k) {
        case 1: launch_dw_k<T, 1>(a, s); break;
        case 3: launch_dw_k<T, 3>(a, s); break;
        case 5: launch_dw_k<T, 5>(a, s); break;
        case 7: launch_dw_k<T, 7>(a, s); break;
        case 9: launch_dw_k<T, 9>(a, s); break;
        case 11: launch_dw_k<T, 11>(a, s); break;
        case 13: launch_dw_k<T, 13>(a, s); break;
        case 15: launch_dw_k<T, 15>(a, s); break;
        default: return YMK_E_BADARG;
    }
    return ymk_launch_status();
}

extern "C" int ymk_dwconv2d(int32_t dtype, const void* x, const void* w, const float* bias,
                            const void* residual, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t ksize, int32_t ldx, int32_t ldy, int32_t ldr, int32_t act,
                            void* stream) {
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (!x || !w || !y || ldx % vec || ldy % 4 || (residual && ldr % 4)) return YMK_E_BADARG;
    DwArgs a{x, w, bias, residual, y, B, H, W, C, ldx, ldy, ldr, act, 0};
    if (dtype == YMK_F32) return launch_dw<float>(a, ksize, (hipStream_t)stream);
    if (dtype == YMK_BF16) return launch_dw<bf16_t>(a, ksize, (hipStream_t)stream);
    return YMK_E_BADARG;
}

// ---------------------------------------------------------------------------
// ES-MoE depthwise stage over the image->expert CSR.  blockIdx.y walks the CSR
// pair list (grouped by expert, so neighbouring workgroups share a filter and a
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

template <typename T>
__global__ __launch_bounds__(256) void moe_dw_kernel(MoeDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int p = blockIdx.y;
    if (p >= a.csr_off[a.E]) return;
    const int pair = a.csr_pair[p];
    const int e = a.sel[pair];
    if (e < 0) return;
    const int b = pair / a.top_k;
    const size_t hw = (size_t)a.H * a.W;
    const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * hw * a.ldx;
    const T* w = reinterpret_cast<const T*>(a.dw_w) + a.dw_off[e];
    T* ob = reinterpret_cast<T*>(a.out) + (size_t)pair * hw * a.C;
    const DwEpi ep{nullptr, nullptr, 0, YMK_ACT_NONE};
    const int tile = (int)xcd_remap_dw(blockIdx.x, gridDim.x);
    switch (a.ksizes[e]) {
        case 3: dw_tile<T, 3>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, tile, ep, nullptr, smem); break;
        case 5: dw_tile<T, 5>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, tile, ep, nullptr, smem); break;
        case 7: dw_tile<T, 7>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, tile, ep, nullptr, smem); break;
        case 9: dw_tile<T, 9>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, tile, ep, nullptr, smem); break;
        case 11: dw_tile<T, 11>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, tile, ep, nullptr, smem); break;
        case 13: dw_tile<T, 13>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, tile, ep, nullptr, smem); break;
        case 15: dw_tile<T, 15>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, tile, ep, nullptr, smem); break;
        default: break;
    }
}

template <typename T>
static int launch_moe_dw(const MoeDwArgs& a, int kmax, hipStream_t s) {
    using D = DwTile<T>;
    if (a.C % (16 / (int)sizeof(T))) return YMK_E_BADARG;
    const size_t shm = D::lds_bytes(kmax);
    static size_t attr = 0;
    if (shm > 64 * 1024 && shm > attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_dw_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shm);
        attr = shm;
    }
    const int tiles = ((a.H + DW_TH - 1) / DW_TH) * ((a.W + DW_TW - 1) / DW_TW) * ((a.C + D::CB - 1) / D::CB);
    hipLaunchKernelGGL(moe_dw_kernel<T>, dim3(tiles, a.B * a.top_k), dim3(256), shm, s, a);
    return ymk_launch_status();
}

extern "C" int ymk_esmoe_dw(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t ldx, const void* dw_w, const int32_t* dw_off, const int32_t* ksizes,
                            int32_t E, int32_t top_k, int32_t kmax, const int32_t* sel, const int32_t* csr_off,
                            const int32_t* csr_pair, void* dw_out, void* stream) {
    if (!x || !dw_w || !dw_off || !ksizes || !sel || !csr_off || !csr_pair || !dw_out) return YMK_E_BADARG;
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (ldx % vec || E < 1 || top_k < 1 || kmax < 1 || kmax > 15 || (kmax & 1) == 0) return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    if ((int64_t)B * top_k > 65535) return YMK_E_BADARG;
    MoeDwArgs a{x, dw_w, dw_off, ksizes, sel, csr_off, csr_pair, dw_out, B, H, W, C, ldx, E, top_k};
    if (dtype == YMK_F32) return launch_moe_dw<float>(a, kmax, (hipStream_t)stream);
    if (dtype == YMK_BF16) return launch_moe_dw<bf16_t>(a, kmax, (hipStream_t)stream);
    return YMK_E_BADARG;
}
