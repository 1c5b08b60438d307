// Fused depthwise(k x k) -> pointwise(1 x 1) block: the expert body of ES-MoE
// (DepthwiseSeparableConv, ultralytics/nn/modules/moe/experts.py:280-296, dispatched per retained expert as in
// ES_MOE._sparse_forward, moe/modules.py:659-704, with the trailing ES_MOE.norm :581) and the
// DWConv3x3 -> Conv1x1 pairs of the Detect class branch (nn/modules/head.py:111-118).
//
// One workgroup owns an 8 x 16 pixel tile of one image and ALL input channels:
//   phase 1 (VALU): for each 32-channel chunk, stage the (8+k-1) x (16+k-1) halo in LDS and run the
//            sliding-window stencil (4 channels x 4 pixels per thread, packed fp32 FMAs); the result tile
//            [128 px][C] stays in LDS as the K-contiguous "pixel" operand of the GEMM — it never touches HBM;
//   phase 2 (MFMA): [Cout x C] x [C x 128 px] with the weight fragments read straight from L1/L2 into registers
//            (no LDS staging, no barriers in the k-loop), epilogue SiLU(BN_e(.)) * gate accumulated over the
//            image's retained experts, then the final affine + SiLU and one NHWC store.
// Compared with the two-kernel form this removes the dw_out write + read (2 x B*k_ret*H*W*C elements per layer)
// and lets the stencil of one workgroup overlap the MFMA phase of its neighbour on the same CU.
#include "igemm.h"

#define FP_TH 8
#define FP_TW 16
#define FP_R 4
#define FP_CC 32  // channels per stencil chunk

typedef float fp_f32x2 __attribute__((ext_vector_type(2)));

struct DwPwArgs {
    const void* x;
    const void* dw_w;
    const int* dw_off;     // [E] element offsets into dw_w (null: single filter at offset 0)
    const int* ksizes;     // [E] (null: k_plain)
    const float* dw_bias;  // [C] or null (Detect DWConv: folded BN), with dw_act
    const void* pw_w;      // [E][Cout][Kpad]
    const float* pw_b;     // [E][Cout]
    const float* nscale;   // [Cout] or null
    const float* nshift;
    const int* sel;        // [B][top_k] or null (plain: one slot, expert 0, gate 1)
    const float* gate;     // [B][E]
    void* y;
    int B, H, W, C, Cout, Kpad, E, top_k, k_plain, dw_act, final_act, ldx, ldy, tiles_x, tiles_y, kmax;
};

template <typename T>
struct FpGeo {
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr int PSB = FP_CC * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16);  // halo pixel stride (bytes)
    static constexpr int CPP = FP_CC * (int)sizeof(T) / 16;                         // 16-B chunks per halo pixel
    __host__ __device__ static size_t adw_bytes(int C) { return (size_t)128 * C * sizeof(T); }
    __host__ __device__ static size_t halo_bytes(int k) { return (size_t)(FP_TH + k - 1) * (FP_TW + k - 1) * PSB; }
    __host__ __device__ static size_t lds_bytes(int C, int k) {
        return adw_bytes(C) + halo_bytes(k) + (size_t)k * k * FP_CC * sizeof(float);
    }
};

__device__ __forceinline__ void fp_ld4(const char* p, fp_f32x2& a, fp_f32x2& b, float) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    a = fp_f32x2{t.x, t.y}; b = fp_f32x2{t.z, t.w};
}
__device__ __forceinline__ void fp_ld4(const char* p, fp_f32x2& a, fp_f32x2& b, h16_t) {  // (c0,c2),(c1,c3)
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    f32x2 a2, b2;
    h16x4_widen(t, a2, b2);
    a = fp_f32x2{a2.x, a2.y}; b = fp_f32x2{b2.x, b2.y};
}
template <typename T> struct FpPair;
template <> struct FpPair<float> { static constexpr int pos[4] = {0, 1, 2, 3}; };
template <> struct FpPair<h16_t> { static constexpr int pos[4] = {0, 2, 1, 3}; };

// byte offset of element (px, c) inside the swizzled [128][C] GEMM operand tile
template <typename T>
__device__ __forceinline__ int adw_off(int px, int c, int C) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ch16 = c / VEC;                      // 16-byte chunk index along the row
    const int phys = (ch16 & ~7) | ((ch16 & 7) ^ (px & 7));
    return px * C * (int)sizeof(T) + phys * 16 + (c % VEC) * (int)sizeof(T);
}

// phase 1 for one expert: depthwise K x K over all C channels of the tile -> Adw (LDS)
template <typename T, int K>
__device__ __forceinline__ void fp_stencil(const DwPwArgs& a, const T* __restrict__ xb, const T* __restrict__ w,
                                           int ty0, int tx0, char* adw, char* halo, float* wsm) {
    using G = FpGeo<T>;
    constexpr int P = K / 2, HT = FP_TH + K - 1, WT = FP_TW + K - 1;
    constexpr int VEC = G::VEC, CPP = G::CPP;
    constexpr int NL = (HT * WT * CPP + 255) / 256;
    constexpr bool PRECISE = sizeof(T) == 4;
    const int t = threadIdx.x;
    const int cg = t & 7, strip = (t >> 3) & 3, row = t >> 5;
    const int x0 = strip * FP_R;
    for (int c0 = 0; c0 < a.C; c0 += FP_CC) {
        u32x4 stg[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {  // all global loads of the chunk first
            const int i = t + l * 256;
            const int pix = i / CPP, q = i % CPP;
            const int hy = pix / WT, hx = pix - hy * WT;
            const int iy = ty0 - P + hy, ix = tx0 - P + hx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < HT * WT * CPP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * a.W + ix) * a.ldx + c0 + q * VEC);
            stg[l] = v;
        }
        __syncthreads();  // previous chunk's stencil reads are done
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t + l * 256;
            if (i < HT * WT * CPP) {
                u32x2* d = reinterpret_cast<u32x2*>(halo + (size_t)(i / CPP) * G::PSB + (i % CPP) * 16);
                d[0] = u32x2{stg[l].x, stg[l].y};
                d[1] = u32x2{stg[l].z, stg[l].w};
            }
        }
        for (int i = t; i < K * K * FP_CC; i += 256) {
            const int tap = i / FP_CC, c = i - tap * FP_CC;
            wsm[tap * FP_CC + (c & ~3) + FpPair<T>::pos[c & 3]] = to_f32(w[(size_t)tap * a.C + c0 + c]);
        }
        __syncthreads();
        fp_f32x2 acc[FP_R][2];
#pragma unroll
        for (int r = 0; r < FP_R; ++r) { acc[r][0] = fp_f32x2{0.f, 0.f}; acc[r][1] = fp_f32x2{0.f, 0.f}; }
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            fp_f32x2 wr[K][2];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(wsm + (ky * K + kx) * FP_CC + cg * 4);
                wr[kx][0] = fp_f32x2{t4.x, t4.y}; wr[kx][1] = fp_f32x2{t4.z, t4.w};
            }
            const char* rp = halo + ((size_t)(row + ky) * WT + x0) * G::PSB + cg * 4 * sizeof(T);
#pragma unroll
            for (int j = 0; j < FP_R + K - 1; ++j) {
                fp_f32x2 va, vb;
                fp_ld4(rp + (size_t)j * G::PSB, va, vb, T{});
#pragma unroll
                for (int r = 0; r < FP_R; ++r) {
                    const int kx = j - r;
                    if (kx >= 0 && kx < K) {
                        acc[r][0] = __builtin_elementwise_fma(va, wr[kx][0], acc[r][0]);
                        acc[r][1] = __builtin_elementwise_fma(vb, wr[kx][1], acc[r][1]);
                    }
                }
            }
        }
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.dw_bias) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.dw_bias + c0 + cg * 4);
            bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        }
#pragma unroll
        for (int r = 0; r < FP_R; ++r) {
            const float q4[4] = {acc[r][0].x, acc[r][0].y, acc[r][1].x, acc[r][1].y};
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = q4[FpPair<T>::pos[q]] + bv[q];
                if (a.dw_act == YMK_ACT_SILU) v[q] = PRECISE ? silu_exact(v[q]) : silu_f(v[q]);
            }
            const int px = row * FP_TW + x0 + r;
            store4(reinterpret_cast<T*>(adw + adw_off<T>(px, c0 + cg * 4, a.C)), v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();  // Adw complete
}

template <typename T>
__global__ __launch_bounds__(256, 2) void dwpw_kernel(DwPwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = FpGeo<T>;
    constexpr int VEC = G::VEC;
    constexpr bool PRECISE = sizeof(T) == 4;
    constexpr int KSUB = 4 * VEC;  // K elements per MFMA sub-step group (64 bytes of a row: 32 bf16 / 16 fp32)
    char* adw = smem;
    char* halo = smem + G::adw_bytes(a.C);
    float* wsm = reinterpret_cast<float*>(halo + G::halo_bytes(a.kmax));

    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tpi = a.tiles_x * a.tiles_y;
    const int b = lid / tpi, tl = lid % tpi;
    const int ty0 = (tl / a.tiles_x) * FP_TH, tx0 = (tl % a.tiles_x) * FP_TW;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const size_t hw = (size_t)a.H * a.W;
    const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * hw * a.ldx;
    const int nslots = a.sel ? a.top_k : 1;
    const int ncot = (a.Cout + 63) / 64;  // GEMM passes of 64 couts x 128 pixels; wave w owns pixels [32w, 32w+32)

    for (int ct = 0; ct < ncot; ++ct) {
        const int co0 = ct * 64;
        f32x4 out[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) out[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nslots; ++s) {
            const int e = a.sel ? a.sel[b * a.top_k + s] : 0;
            if (e < 0) continue;  // workgroup-uniform
            const float g = a.sel ? a.gate[b * a.E + e] : 1.0f;
            if (ct == 0 || nslots > 1) {  // (re)build the stencil tile: kept across cout passes when there is one slot
                const T* w = reinterpret_cast<const T*>(a.dw_w) + (a.dw_off ? a.dw_off[e] : 0);
                switch (a.ksizes ? a.ksizes[e] : a.k_plain) {
                    case 3: fp_stencil<T, 3>(a, xb, w, ty0, tx0, adw, halo, wsm); break;
                    case 5: fp_stencil<T, 5>(a, xb, w, ty0, tx0, adw, halo, wsm); break;
                    case 7: fp_stencil<T, 7>(a, xb, w, ty0, tx0, adw, halo, wsm); break;
                    case 9: fp_stencil<T, 9>(a, xb, w, ty0, tx0, adw, halo, wsm); break;
                    default: break;
                }
            }
            // phase 2: GEMM over K = C, pixel operand from LDS, weight fragments from global (L1/L2 resident)
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const T* wt = reinterpret_cast<const T*>(a.pw_w) + ((size_t)e * a.Cout + co0) * a.Kpad;
#pragma unroll 2
            for (int k0 = 0; k0 < a.C; k0 += KSUB) {
                u32x4 af[4], bfr[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = i * 16 + fr;
                    af[i] = (co0 + r) < a.Cout ? *reinterpret_cast<const u32x4*>(wt + (size_t)r * a.Kpad + k0 + fc * VEC)
                                               : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int px = wave * 32 + j * 16 + fr;
                    bfr[j] = *reinterpret_cast<const u32x4*>(adw + adw_off<T>(px, k0 + fc * VEC, a.C));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma16<T>(acc[i][j], af[i], bfr[j]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + i * 16 + fc * 4;
                const f32x4 bv = co < a.Cout ? *reinterpret_cast<const f32x4*>(a.pw_b + (size_t)e * a.Cout + co)
                                             : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r] + bv[r];
                        if (a.sel || a.final_act == YMK_ACT_SILU) v = PRECISE ? silu_exact(v) : silu_f(v);
                        out[i][j][r] += v * g;
                    }
            }
            if (nslots > 1) __syncthreads();  // every wave finished reading Adw before the next expert rebuilds it
        }
        // final affine (+ SiLU) and store: each lane owns 4 consecutive couts of a pixel
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + i * 16 + fc * 4;
            if (co >= a.Cout) continue;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (a.nscale) {
                sc = *reinterpret_cast<const f32x4*>(a.nscale + co);
                sh = *reinterpret_cast<const f32x4*>(a.nshift + co);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int px = wave * 32 + j * 16 + fr;
                const int gy = ty0 + px / FP_TW, gx = tx0 + px % FP_TW;
                if (gy >= a.H || gx >= a.W) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = out[i][j][r];
                    if (a.nscale) {
                        v[r] = v[r] * sc[r] + sh[r];
                        v[r] = PRECISE ? silu_exact(v[r]) : silu_f(v[r]);
                    }
                }
                store4(reinterpret_cast<T*>(a.y) + ((size_t)b * hw + (size_t)gy * a.W + gx) * a.ldy + co, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <typename T>
static int launch_dwpw(DwPwArgs a, hipStream_t s) {
    using G = FpGeo<T>;
    constexpr int BK = 64 / (int)sizeof(T) * 2;  // C must be a whole number of 128-byte K groups
    if (a.C % BK || a.C % FP_CC || a.Cout % 4 || a.ldy % 4 || a.ldx % G::VEC || a.Kpad < a.C) return YMK_E_BADARG;
    if (a.kmax < 3 || a.kmax > 9 || (a.kmax & 1) == 0) return YMK_E_BADARG;
    const size_t shm = G::lds_bytes(a.C, a.kmax);
    if (shm > 160 * 1024) return YMK_E_BADARG;
    static size_t attr = 0;
    if (shm > 64 * 1024 && shm > attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpw_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr = shm;
    }
    a.tiles_x = (a.W + FP_TW - 1) / FP_TW;
    a.tiles_y = (a.H + FP_TH - 1) / FP_TH;
    const int64_t nblk = (int64_t)a.B * a.tiles_x * a.tiles_y;
    if (nblk <= 0) return YMK_OK;
    if (nblk >= (1ll << 31)) return YMK_E_BADARG;
    hipLaunchKernelGGL(dwpw_kernel<T>, dim3((unsigned)nblk), dim3(256), shm, s, a);
    return ymk_launch_status();
}

extern "C" int ymk_dwpw_supported(int32_t dtype, int32_t C, int32_t kmax) {
    if (kmax < 3 || kmax > 9 || (kmax & 1) == 0) return 0;
    if (dtype == YMK_BF16) return C % 64 == 0 && FpGeo<h16_t>::lds_bytes(C, kmax) <= 160 * 1024;
    if (dtype == YMK_F32) return C % 32 == 0 && FpGeo<float>::lds_bytes(C, kmax) <= 160 * 1024;
    return 0;
}

extern "C" int ymk_esmoe_experts_fused(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                                       int32_t ldx, const void* dw_w, const int32_t* dw_off, const int32_t* ksizes,
                                       int32_t kmax, int32_t Cout, int32_t Kpad, const void* pw_w, const float* pw_b,
                                       const float* norm_scale, const float* norm_shift, int32_t E, int32_t top_k,
                                       const int32_t* sel, const float* gate_w, void* y, int32_t ldy, void* stream) {
    if (!x || !dw_w || !dw_off || !ksizes || !pw_w || !pw_b || !norm_scale || !norm_shift || !sel || !gate_w || !y)
        return YMK_E_BADARG;
    DwPwArgs a{x, dw_w, dw_off, ksizes, nullptr, pw_w, pw_b, norm_scale, norm_shift, sel, gate_w, y,
               B, H, W, C, Cout, Kpad, E, top_k, 0, YMK_ACT_NONE, YMK_ACT_SILU, ldx, ldy, 0, 0, kmax};
    if (dtype == YMK_F32) return launch_dwpw<float>(a, (hipStream_t)stream);
    if (dtype == YMK_BF16) return launch_dwpw<h16_t>(a, (hipStream_t)stream);
    return YMK_E_BADARG;
}

extern "C" int ymk_dwconv_pwconv(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx,
                                 const void* dw_w, const float* dw_bias, int32_t ksize, int32_t dw_act, int32_t Cout,
                                 int32_t Kpad, const void* pw_w, const float* pw_b, int32_t pw_act, void* y, int32_t ldy,
                                 void* stream) {
    if (!x || !dw_w || !pw_w || !pw_b || !y) return YMK_E_BADARG;
    DwPwArgs a{x, dw_w, nullptr, nullptr, dw_bias, pw_w, pw_b, nullptr, nullptr, nullptr, nullptr, y,
               B, H, W, C, Cout, Kpad, 1, 1, ksize, dw_act, pw_act, ldx, ldy, 0, 0, ksize};
    if (dtype == YMK_F32) return launch_dwpw<float>(a, (hipStream_t)stream);
    if (dtype == YMK_BF16) return launch_dwpw<h16_t>(a, (hipStream_t)stream);
    return YMK_E_BADARG;
}
