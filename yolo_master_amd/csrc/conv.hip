// Fused Conv + folded-BN bias + SiLU (+ residual) as implicit GEMM on MFMA.
// Reference semantics: Conv.forward_fuse (ultralytics/nn/modules/conv.py:80-89)
// after fuse_conv_and_bn (ultralytics/utils/torch_utils.py:315-349).
#include "igemm.h"

struct ConvArgs {
    const void* x;
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int B, H, W, Ho, Wo, Cin, Cout, stride, ldx, ldy, ldr, Kpad, act, out_f32;
    const void* x2;  // second K-source for the concat-fused 1x1 (DUAL kernels), else null
    int ldx2, C1, up1, H1, W1;  // C1 channels come from x (optionally 2x-nearest-upsampled from H1 x W1), the rest from x2
    int ablate; // tools/micro ablation switch (always 0 in the library build)
    int ncot;   // cout tiles (fast block index: workgroups sharing a pixel tile run back to back => L2 reuse)
    int64_t M;  // B*Ho*Wo
    float* pool;      // streaming 1x1 only: fp32 [B][pool_chunks][Cout] per-tile channel sums of the STORED values, or null
    int pool_chunks;  // 128-pixel tiles per image (H * W a multiple of 128: no tile straddles two images)
};

static int ymk_use_ws = 1;          // tools/micro can switch the streaming 1x1 kernel off for A/B runs
// smallest pixel-tile count routed to the streaming kernel (YMK_WS_MIN_TILES overrides it for tuning runs)
static int ymk_ws_min_tiles = [] { const char* e = getenv("YMK_WS_MIN_TILES"); return e ? atoi(e) : 1024; }();
extern "C" void ymk_debug_set_ws(int on) { ymk_use_ws = on; }
static thread_local int ymk_last_variant = YMK_CONV_TILED;
static thread_local int ymk_last_glds_stages = 0;
extern "C" int ymk_conv2d_glds(const ymk_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                               void* y, int32_t two_stage, void* stream);   // csrc/conv_glds.hip, include/ymk_next.h
extern "C" int ymk_conv1x1_cat2_glds(const ymk_conv_desc* d, const void* x1, int32_t C1, int32_t ldx1, int32_t upsample1, const void* x2,
                                     int32_t ldx2, const void* w, const float* bias, void* y, int32_t two_stage, void* stream);
int ymk_glds_last_tile();   // csrc/conv_glds.hip: pixel-tile height of the thread's last LDS-DMA launch
int ymk_glds_last_bn();     // ... and its cout-tile width
static int ymk_glds_min_tiles = [] { const char* e = getenv("YMK_GLDS_MIN_TILES"); return e ? atoi(e) : 192; }();
// one workgroup per CU (144 KB of LDS): with fewer than four k-steps there is nothing to pipeline, the current kernels win
static int ymk_glds_min_k = [] { const char* e = getenv("YMK_GLDS_MIN_K"); return e ? atoi(e) : 256; }();
// diagnostic (kernel naming in bench.py / tools): variant code, plus the LDS stage count << 8 for the LDS-DMA core
extern "C" int32_t ymk_conv2d_last_variant(void) {
    return ymk_last_variant == YMK_CONV_GLDS ? (ymk_last_variant | (ymk_last_glds_stages << 8) | (ymk_glds_last_tile() << 16) | ((ymk_glds_last_bn() / 64) << 26)) : ymk_last_variant;
}

template <typename T, bool PRECISE>
__device__ __forceinline__ float act_silu(float v) {
    return PRECISE ? silu_exact(v) : silu_f(v);
}

template <typename T, int BCO, int BPX, int WCO, int WPX, int KS, bool DUAL = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    using G = IGemm<T, BCO, BPX, WCO, WPX, KS, DUAL>;
    __shared__ u32x4 smem[G::SMEM_U4];
    const int t = threadIdx.x;
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t px0 = (int64_t)(lid / a.ncot) * BPX;
    const int co0 = (int)(lid % a.ncot) * BCO;
    const int pad = KS / 2;

    typename G::Rows rows;
    const unsigned HoWo = (unsigned)(a.Ho * a.Wo);
#pragma unroll
    for (int i = 0; i < G::NB; ++i) {
        const int64_t m = px0 + (t >> 3) + i * G::RPP;
        rows.ok[i] = m < a.M;
        const unsigned mm = rows.ok[i] ? (unsigned)m : 0u;  // M < 2^31 (checked on the host)
        if (DUAL) {  // source 2 is pixel-aligned with the output; source 1 optionally at half resolution
            rows.pix2[0 + (DUAL ? i : 0)] = (int)mm;
            rows.iy0[i] = 0; rows.ix0[i] = 0;
            if (a.up1) {
                const unsigned b = mm / HoWo, rem = mm - b * HoWo;
                const unsigned oy = rem / (unsigned)a.Wo, ox = rem - oy * (unsigned)a.Wo;
                rows.pix[i] = (int)(b * (unsigned)(a.H1 * a.W1) + (oy >> 1) * (unsigned)a.W1 + (ox >> 1));
            } else {
                rows.pix[i] = (int)mm;
            }
        } else if (KS == 1 && a.stride == 1) {                // output pixel == input pixel: no index math
            rows.pix[i] = (int)mm; rows.iy0[i] = 0; rows.ix0[i] = 0;
        } else {
            const unsigned b = mm / HoWo;
            const unsigned rem = mm - b * HoWo;
            const unsigned oy = rem / (unsigned)a.Wo, ox = rem - oy * (unsigned)a.Wo;
            if (KS == 1) {
                rows.pix[i] = (int)(b * (unsigned)(a.H * a.W) + oy * a.stride * (unsigned)a.W + ox * a.stride);
                rows.iy0[i] = 0; rows.ix0[i] = 0;
            } else {
                rows.pix[i] = (int)(b * (unsigned)(a.H * a.W));
                rows.iy0[i] = (int)oy * a.stride - pad;
                rows.ix0[i] = (int)ox * a.stride - pad;
            }
        }
    }

    {   // uniform reference pixels (see IGemm::Rows): the tile's first output pixel mapped like the rows above
        const unsigned m0 = (unsigned)(px0 < a.M ? px0 : 0);
        const unsigned b0 = m0 / HoWo, rem0 = m0 - b0 * HoWo;
        const unsigned oy0 = rem0 / (unsigned)a.Wo, ox0 = rem0 - oy0 * (unsigned)a.Wo;
        if (DUAL) {
            rows.ref2 = m0;
            rows.ref = a.up1 ? (int64_t)(b0 * (unsigned)(a.H1 * a.W1) + (oy0 >> 1) * (unsigned)a.W1 + (ox0 >> 1)) : (int64_t)m0;
        } else if (KS == 1 && a.stride == 1) {
            rows.ref = m0;
        } else if (KS == 1) {
            rows.ref = (int64_t)(b0 * (unsigned)(a.H * a.W) + oy0 * a.stride * (unsigned)a.W + ox0 * a.stride);
        } else {
            rows.ref = (int64_t)(b0 * (unsigned)(a.H * a.W));
        }
    }

    f32x4 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* wt = reinterpret_cast<const T*>(a.w) + (size_t)co0 * a.Kpad;
    G::run(acc, reinterpret_cast<const T*>(a.x), a.ldx, a.H, a.W, a.Cin, rows, wt, a.Kpad,
           a.Cout - co0, smem, a.ablate, typename G::Src2{reinterpret_cast<const T*>(a.x2), a.ldx2, a.C1});

    // epilogue: bias + act in registers, then LDS-staged coalesced store (+ residual)
    const int lane = t & 63, wave = t >> 6;
    const int wco = wave / WPX;
    constexpr bool PRECISE = sizeof(T) == 4;
    f32x4 bv[G::TM];
#pragma unroll
    for (int i = 0; i < G::TM; ++i) {
        const int co = co0 + (wco * G::TM + i) * 16 + (lane >> 4) * 4;
        bv[i] = co < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool silu = a.act == YMK_ACT_SILU;
    auto val = [&](int i, int j, int r) {
        const float v = acc[i][j][r] + bv[i][r];
        return silu ? act_silu<T, PRECISE>(v) : v;
    };
    auto emit = [&](int px, int co_l, const f32x4& v) {
        const int64_t m = px0 + px;
        const int co = co0 + co_l;
        if (m >= a.M || co >= a.Cout) return;
#ifdef YMK_ABLATE
        if (a.ablate == 2 && v.x != 12345.f) return;
#endif
        float v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
        if (a.res) {
            float r0, r1, r2, r3;
            load4(reinterpret_cast<const T*>(a.res) + m * a.ldr + co, r0, r1, r2, r3);
            v0 = r0 + v0; v1 = r1 + v1; v2 = r2 + v2; v3 = r3 + v3;
        }
        if (a.out_f32)
            store4(reinterpret_cast<float*>(a.y) + m * a.ldy + co, v0, v1, v2, v3);
        else
            store4(reinterpret_cast<T*>(a.y) + m * a.ldy + co, v0, v1, v2, v3);
    };
    G::epilogue(smem, val, emit);
}

#ifdef YMK_ABLATE
static int ymk_ablate = 0;
#endif

// ---------------------------------------------------------------------------
// Weight-stationary streaming 1x1 convolution (large M, K <= 4 x 128 bytes, Cout tile of 128).
// A persistent workgroup keeps its [128 cout][K] weight tile in LDS for its whole life and walks over pixel
// tiles; the NEXT tile's activations are already in flight (registers) while the current tile is multiplied and
// stored, so the HBM latency of a tile is hidden behind the previous tile's MFMA + epilogue instead of being
// exposed once per tile, and the k-loop has no barriers (both operands are complete in LDS).
// ---------------------------------------------------------------------------
// eight consecutive channels in one 16-byte store (bf16); the fp32 overload is never taken (`wide` is bf16-only) but must compile
__device__ __forceinline__ void ws_store8(h16_t* p, const float (&v)[8]) { store_vec_f32(p, v); }
__device__ __forceinline__ void ws_store8(float* p, const float (&v)[8]) {
    store4(p, v[0], v[1], v[2], v[3]);
    store4(p + 4, v[4], v[5], v[6], v[7]);
}

// Sum over the 16 lanes of a DPP row (lanes 16 i .. 16 i + 15), left in every lane of the row.  Every step pairs lanes by an INVOLUTION
// (mirror of the row, mirror of its halves, neighbours, neighbours' neighbours), so both partners compute the same sum and the
// association order is a property of the lane index alone: pool_tiles128_kernel below restates it with plain loops, bit for bit.
__device__ __forceinline__ int row16_partner(int i, int step) {   // i = lane & 15
    return step == 0 ? 15 - i : step == 1 ? (i < 8 ? 7 - i : 23 - i) : step == 2 ? (i ^ 1) : (i ^ 2);
}
__device__ __forceinline__ float row16_sum(float v) {
#ifndef YMK_HOST_EMU
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm:[1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm:[2,3,0,1]
#else
    const int lane = (int)(threadIdx.x & 63);
    for (int step = 0; step < 4; ++step) v += __shfl(v, (lane & ~15) | row16_partner(lane & 15, step));
#endif
    return v;
}
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<h16_t>(float v) { return h16_to_f32(f32_to_h16(v)); }
__device__ __forceinline__ float to_f32_elem(float v) { return v; }
__device__ __forceinline__ float to_f32_elem(h16_t v) { return h16_to_f32(v); }

template <typename T, int KG, bool PERM, bool POOL = false>
__global__ __launch_bounds__(256) void conv1x1_ws_kernel(ConvArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BK = 8 * VEC;            // elements per 128-byte K group
    // u32x4 per staged row: K bytes + 32.  A pitch of 8 dwords modulo 16 puts the sixteen rows of every ds_read_b128 lane group (rows
    // {0-3, 12-15} at chunk c and {4-11} at chunk c + 1: MI355X_MICROARCH.md, LDS) on the 64 banks exactly once, without a swizzle.  (Round 4:
    // the XOR swizzle alone left rows of 256 / 512 bytes — KG = 2 / 4 — 2-way conflicted, the row's parity no longer selecting a bank half:
    // 37 % of this kernel's LDS cycles were conflict cycles at KG = 2 and none at KG = 3, profiles/r04_sq_summary.txt.)
    constexpr int RS = KG * 8 + 2;
    constexpr bool PRECISE = sizeof(T) == 4;
    __shared__ u32x4 sW[128 * RS];
    __shared__ u32x4 sA[128 * RS];
    __shared__ float sPool[POOL ? 256 : 1];   // [2 pixel halves][128 couts]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wco = wave >> 1, wpx = wave & 1;
    const int srow = t >> 3, cq = t & 7;
    const int fr = lane & 15, fc = lane >> 4;
    // the a.ncot workgroups that walk the SAME pixel tiles (one per 128-cout tile) are consecutive logical ids: given to one XCD back to back
    // (igemm.h xcd_remap), so that x crosses the fabric once — with blockIdx taken as is they sat on different XCDs and every L2 fetched x
    // for itself (128 -> 384 at 40^2: 79.6 MB for 26 MB of input, profiles/r04_step_dispatch_pmc.txt)
    const unsigned lid = a.ncot > 1 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int ct = lid % a.ncot;
    const int co0 = ct * 128;
    const int nblk_px = gridDim.x / a.ncot;            // workgroups sharing this cout tile
    const int64_t ntiles = (a.M + 127) / 128;
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* wt = reinterpret_cast<const T*>(a.w) + (size_t)co0 * a.Kpad;
    auto swz = [](int, int c) { return c; };   // (no swizzle: see RS)

    // weights: once per workgroup.  With whole 64-cout groups, LDS row r = MFMA row block i = (r >> 4) & 3, row fr = r & 15 of a wave's 64
    // couts is filled with cout (i >> 1) * 32 + (fr >> 2) * 8 + (i & 1) * 4 + (fr & 3): a lane then holds EIGHT consecutive couts per block
    // pair after the MFMAs (16-byte stores, 64 contiguous bytes per pixel and wave; csrc/conv_glds.hip, csrc/esmoe.hip do the same)
    constexpr bool perm = PERM;   // launcher: Cout % 64 == 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = srow + i * 32;
        const int rc = perm ? (r & ~63) + ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3) : r;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (co0 + rc < a.Cout) v = *reinterpret_cast<const u32x4*>(wt + (size_t)rc * a.Kpad + g * BK + cq * VEC);
            sW[r * RS + swz(r, g * 8 + cq)] = v;
        }
    }
    auto cout_of = [&](int i) { return perm ? co0 + wco * 64 + (i >> 1) * 32 + fc * 8 + (i & 1) * 4 : co0 + (wco * 4 + i) * 16 + fc * 4; };
    const bool wide = PERM && sizeof(T) == 2 && (a.ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
    u32x4 ra[4][KG];
    auto gload = [&](int64_t tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = tile * 128 + srow + i * 32;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (m < a.M && g * BK + cq * VEC < a.Cin) v = *reinterpret_cast<const u32x4*>(x + m * a.ldx + g * BK + cq * VEC);
                ra[i][g] = v;
            }
        }
    };
    f32x4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = cout_of(i);
        bv[i] = co < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool silu = a.act == YMK_ACT_SILU;

    int64_t tile = lid / a.ncot;
    if (tile < ntiles) gload(tile);
    for (; tile < ntiles; tile += nblk_px) {
        __syncthreads();  // previous tile's fragment reads are finished (first pass: sW is complete)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + i * 32;
#pragma unroll
            for (int g = 0; g < KG; ++g) sA[r * RS + swz(r, g * 8 + cq)] = ra[i][g];
        }
        __syncthreads();
        const int64_t nxt = tile + nblk_px;
        if (nxt < ntiles) gload(nxt);  // in flight during the MFMA + epilogue below
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = (wco * 4 + i) * 16 + fr;
                    af[i] = sW[r * RS + swz(r, g * 8 + kk * 4 + fc)];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = (wpx * 4 + j) * 16 + fr;
                    bfr[j] = sA[r * RS + swz(r, g * 8 + kk * 4 + fc)];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], af[i], bfr[j]);
            }
        float psum[2][8];   // pooled variant: this lane's sums over its four pixel groups, couts cout_of(2h) .. + 7
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 8; ++r) psum[h][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = tile * 128 + (wpx * 4 + j) * 16 + fr;
            if (m >= a.M) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = 2 * h + q, co = cout_of(i);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[q * 4 + r] = acc[i][j][r] + bv[i][r];
                        if (silu) v[q * 4 + r] = act_silu<T, PRECISE>(v[q * 4 + r]);
                    }
                    if (a.res && co < a.Cout) {
                        float r0, r1, r2, r3;
                        load4(reinterpret_cast<const T*>(a.res) + m * a.ldr + co, r0, r1, r2, r3);
                        v[q * 4 + 0] += r0; v[q * 4 + 1] += r1; v[q * 4 + 2] += r2; v[q * 4 + 3] += r3;
                    }
                }
                if (POOL) {   // sums of the values AS STORED (rounded to the activation type): what a global average pool of y would read
#pragma unroll
                    for (int r = 0; r < 8; ++r) psum[h][r] += round_to<T>(v[r]);
                }
                T* yo = reinterpret_cast<T*>(a.y) + m * a.ldy;
                if (wide) {
                    if (cout_of(2 * h) < a.Cout) ws_store8(yo + cout_of(2 * h), v);   // whole 64-cout groups: the lane's eight couts are in or out together
                } else {
                    if (cout_of(2 * h) < a.Cout) store4(yo + cout_of(2 * h), v[0], v[1], v[2], v[3]);
                    if (cout_of(2 * h + 1) < a.Cout) store4(yo + cout_of(2 * h + 1), v[4], v[5], v[6], v[7]);
                }
            }
        }
        if (POOL) {
            // Fixed order: a lane's four pixel groups (above), the sixteen lanes of a cout group (row rotations by 8, 4, 2, 1: DPP operands of
            // the adds, no LDS round trip — __shfl_xor is a ds_bpermute each and made this epilogue 23 % longer), then the two pixel halves of
            // the tile (waves wpx = 0, 1) through 1 KB of LDS.
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 8; ++r) psum[h][r] = row16_sum(psum[h][r]);
            if (fr == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 8; ++r) sPool[wpx * 128 + wco * 64 + h * 32 + fc * 8 + r] = psum[h][r];
            }
            __syncthreads();   // (the next tile writes sPool only behind the two barriers at the top of the loop)
            if (t < 128 && co0 + t < a.Cout) {
                const int64_t b = tile / a.pool_chunks, ch = tile - b * a.pool_chunks;
                a.pool[((size_t)b * a.pool_chunks + ch) * a.Cout + co0 + t] = sPool[t] + sPool[128 + t];
            }
        }
    }
}

template <typename T>
static bool launch_conv1x1_ws(ConvArgs a, hipStream_t s) {
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    const int kg = a.Kpad / BK;
    const int64_t ntiles = (a.M + 127) / 128;
    if (a.Kpad % BK || kg < 1 || kg > 4 || a.Cout <= 64 || a.out_f32 || ntiles < (kg <= 2 ? ymk_ws_min_tiles / 2 : ymk_ws_min_tiles)) return false;
    a.ncot = (a.Cout + 127) / 128;
    a.ablate = 0;
    // 2 workgroups per CU when LDS allows (2 * 2 * 128 * kg * 128 B <= 160 KB  <=>  kg <= 2), else 1
    const int per_cu = kg <= 2 ? 2 : 1;
    int nblk_px = (256 * per_cu) / a.ncot;
    if (nblk_px < 1) nblk_px = 1;
    if (nblk_px > ntiles) nblk_px = (int)ntiles;
    dim3 grid(nblk_px * a.ncot), blk(256);
    if (a.pool) {   // (ymk_conv1x1_pooled checked: 16-bit type, Cout % 64 == 0, H * W % 128 == 0)
        switch (kg) {
            case 1: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 1, true, true>), grid, blk, 0, s, a); break;
            case 2: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 2, true, true>), grid, blk, 0, s, a); break;
            case 3: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 3, true, true>), grid, blk, 0, s, a); break;
            default: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 4, true, true>), grid, blk, 0, s, a); break;
        }
        return true;
    }
    if (a.Cout % 64 == 0) {   // whole 64-cout groups: weight rows permuted for 16-byte stores
        switch (kg) {
            case 1: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 1, true>), grid, blk, 0, s, a); break;
            case 2: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 2, true>), grid, blk, 0, s, a); break;
            case 3: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 3, true>), grid, blk, 0, s, a); break;
            default: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 4, true>), grid, blk, 0, s, a); break;
        }
    } else {
        switch (kg) {
            case 1: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 1, false>), grid, blk, 0, s, a); break;
            case 2: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 2, false>), grid, blk, 0, s, a); break;
            case 3: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 3, false>), grid, blk, 0, s, a); break;
            default: hipLaunchKernelGGL((conv1x1_ws_kernel<T, 4, false>), grid, blk, 0, s, a); break;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------
// Spatial-tile 3x3 convolution (stride 1, Cin in {16, 32, 64}, Cout tile 32/64): LDS-staged im2col.
// A persistent workgroup keeps its [BCO][9*Cin] weight tile in LDS and walks over 16x16-pixel output tiles.
// The 18x18 input halo of a tile is loaded ONCE (coalesced, zero-filled at the borders) into LDS; the nine taps
// of the implicit GEMM are then plain LDS reads at shifted pixel addresses (pixel stride padded by 16 bytes so
// the 16 pixels of a fragment fall on distinct bank groups) — no per-tap global gathers, no per-k-step address
// and bounds arithmetic, no barriers inside the K loop.  The next tile's halo is prefetched into registers
// while the current tile is multiplied and stored.
// ---------------------------------------------------------------------------
// register allocation held to two waves per SIMD wherever the LDS footprint lets two workgroups share a CU
// LDS row pitch for rows that sixteen lanes of a ds_read_b128 group read side by side (MFMA fragments: lane (fr, fc) reads row fr,
// 16-byte chunk fc): the pitch must be 8 dwords modulo 16 for the group to cover the 64 banks once, i.e. row bytes + pad = 32 (mod 64).
// (The 16-byte pad used before gave pitches of 20 / 36 dwords: rows r and r + 4 / r + 8 on the same banks — half of this kernel's
// LDS cycles were conflict cycles, profiles/r03_sq_summary.txt.)
constexpr int lds_row_pitch(int row_bytes) { return row_bytes + (32 - row_bytes % 64 + 64) % 64; }
template <typename T, int CIN, int BCO, bool RES>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu((18 * 18 * lds_row_pitch(CIN * (int)sizeof(T)) + BCO * lds_row_pitch((9 * CIN + 4 * (16 / (int)sizeof(T)) - 1) / (4 * (16 / (int)sizeof(T))) * (4 * (16 / (int)sizeof(T))) * (int)sizeof(T))) <= 80 * 1024 ? 2 : 1)))
void conv3x3_tile_kernel(ConvArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int CPP = CIN / VEC;                      // 16-byte chunks per pixel
    constexpr int PSB = lds_row_pitch(CIN * (int)sizeof(T));      // padded pixel stride (bytes)
    constexpr int HT = 18;
    constexpr int KSUB = 4 * VEC;                       // K elements per MFMA group (64 bytes)
    constexpr bool PAIRED = CIN < KSUB;                 // bf16 Cin = 16: one MFMA group spans two taps
    constexpr int KTOT = (9 * CIN + KSUB - 1) / KSUB * KSUB;  // weight row length held in LDS (zero tail from Kpad)
    constexpr int WRS = lds_row_pitch(KTOT * (int)sizeof(T));     // padded weight row stride (bytes)
    constexpr int TM = BCO / 16;
    constexpr int NL = (HT * HT * CPP + 255) / 256;
    constexpr bool PRECISE = sizeof(T) == 4;
    __shared__ __attribute__((aligned(16))) char sIn[HT * HT * PSB];
    __shared__ __attribute__((aligned(16))) char sW[BCO * WRS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16;
    const int tpi = tiles_x * tiles_y;
    const int64_t ntiles = (int64_t)a.B * tpi;
    const int ct = blockIdx.x % a.ncot;
    const int co0 = ct * BCO;
    const int nblk_px = gridDim.x / a.ncot;
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* wt = reinterpret_cast<const T*>(a.w) + (size_t)co0 * a.Kpad;

    // weights once: [BCO][9*CIN] (K order (ky,kx,cin) = the packed order)
    for (int q = t; q < BCO * (KTOT / VEC); q += 256) {
        const int r = q / (KTOT / VEC), ch = q % (KTOT / VEC);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (co0 + r < a.Cout) v = *reinterpret_cast<const u32x4*>(wt + (size_t)r * a.Kpad + ch * VEC);
        *reinterpret_cast<u32x4*>(sW + (size_t)r * WRS + ch * 16) = v;
    }
    // register staging: two tiles ahead when a halo tile is small (keeps enough bytes in flight per CU)
    constexpr int DEPTH = (HT * HT * CIN * (int)sizeof(T) <= 24 * 1024 && BCO <= 32) ? 2 : 1;
    u32x4 stg[DEPTH][NL];
    auto gload = [&](int64_t tile, u32x4 (&dst)[NL]) {
        const int b = (int)(tile / tpi), tl = (int)(tile % tpi);
        const int ty0 = (tl / tiles_x) * 16, tx0 = (tl % tiles_x) * 16;
        const T* xb = x + (size_t)b * a.H * a.W * a.ldx;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int q = t + l * 256;
            const int pix = q / CPP, ch = q % CPP;
            const int hy = pix / HT, hx = pix - hy * HT;
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < HT * HT * CPP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * a.W + ix) * a.ldx + ch * VEC);
            dst[l] = v;
        }
    };
    f32x4 bv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int co = co0 + i * 16 + fc * 4;
        bv[i] = co < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool silu = a.act == YMK_ACT_SILU;

    auto body = [&](int64_t tile, u32x4 (&buf)[NL]) {
        __syncthreads();  // previous tile's LDS reads are done (first pass: orders the weight stores)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int q = t + l * 256;
            if (q < HT * HT * CPP) *reinterpret_cast<u32x4*>(sIn + (size_t)(q / CPP) * PSB + (q % CPP) * 16) = buf[l];
        }
        __syncthreads();
        const int64_t nxt = tile + (int64_t)DEPTH * nblk_px;
        if (nxt < ntiles) gload(nxt, buf);  // refill the buffer just consumed
        const int b = (int)(tile / tpi), tl = (int)(tile % tpi);
        const int ty0 = (tl / tiles_x) * 16, tx0 = (tl % tiles_x) * 16;
        // residual operands of this tile: requested now, consumed after the MFMA loop
        typename Raw4<T>::type rres[RES ? TM : 1][4];
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = ty0 + wave * 4 + j, gx = tx0 + fr;
                const int64_t m = ((int64_t)b * a.H + gy) * a.W + gx;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int co = co0 + i * 16 + fc * 4;
                    if (gy < a.H && gx < a.W && co < a.Cout) rres[i][j] = load_raw4(reinterpret_cast<const T*>(a.res) + m * a.ldr + co);
                }
            }
        }
        f32x4 acc[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PAIRED) {
            // K index = tap * 16 + c: lanes with k-group fc < 2 read tap 2u, the others tap 2u+1 (the 10th
            // "tap" has zero weights; its pixel operand re-reads tap 8 so that it stays finite)
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int ta = 2 * u, tb = 2 * u + 1 < 9 ? 2 * u + 1 : 8;
                const int offa = ((ta / 3) * HT + ta % 3) * PSB, offb = ((tb / 3) * HT + tb % 3) * PSB;
                const int poff = ((fc >> 1) ? offb : offa) + (fc & 1) * 16;
                u32x4 af[TM], bfr[4];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const u32x4*>(sW + (size_t)(i * 16 + fr) * WRS + u * 64 + fc * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bfr[j] = *reinterpret_cast<const u32x4*>(sIn + (size_t)((wave * 4 + j) * HT + fr) * PSB + poff);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], af[i], bfr[j]);
            }
        } else
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int cs = 0; cs < CIN / KSUB; ++cs) {
                u32x4 af[TM], bfr[4];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const u32x4*>(sW + (size_t)(i * 16 + fr) * WRS +
                                                            (tap * CIN + cs * KSUB) * (int)sizeof(T) + fc * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ly = wave * 4 + j;
                    bfr[j] = *reinterpret_cast<const u32x4*>(sIn + (size_t)((ly + ky) * HT + fr + kx) * PSB +
                                                             cs * 64 + fc * 16);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], af[i], bfr[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = ty0 + wave * 4 + j, gx = tx0 + fr;
            if (gy >= a.H || gx >= a.W) continue;
            const int64_t m = ((int64_t)b * a.H + gy) * a.W + gx;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = co0 + i * 16 + fc * 4;
                if (co >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][r] + bv[i][r];
                    if (silu) v[r] = act_silu<T, PRECISE>(v[r]);
                }
                if constexpr (RES) {
                    float r0, r1, r2, r3;
                    unpack_raw4(rres[i][j], r0, r1, r2, r3);
                    v[0] = r0 + v[0]; v[1] = r1 + v[1]; v[2] = r2 + v[2]; v[3] = r3 + v[3];
                } else if (a.res) {
                    float r0, r1, r2, r3;
                    load4(reinterpret_cast<const T*>(a.res) + m * a.ldr + co, r0, r1, r2, r3);
                    v[0] = r0 + v[0]; v[1] = r1 + v[1]; v[2] = r2 + v[2]; v[3] = r3 + v[3];
                }
                store4(reinterpret_cast<T*>(a.y) + m * a.ldy + co, v[0], v[1], v[2], v[3]);
            }
        }
    };

    int64_t tile = blockIdx.x / a.ncot;
    if (tile < ntiles) gload(tile, stg[0]);
    if (DEPTH == 2 && tile + nblk_px < ntiles) gload(tile + nblk_px, stg[DEPTH - 1]);
    while (tile < ntiles) {
        body(tile, stg[0]);
        tile += nblk_px;
        if (DEPTH == 2) {
            if (tile >= ntiles) break;
            body(tile, stg[DEPTH - 1]);
            tile += nblk_px;
        }
    }
}

template <typename T>
static bool launch_conv3x3_tile(ConvArgs a, hipStream_t s) {
    if (a.stride != 1 || a.out_f32 || (a.Cin != 16 && a.Cin != 32 && a.Cin != 64) || a.Cout > 64 || a.Cout % 16) return false;
    if (a.Cin == 16 && a.Cout > 32) return false;
    const int bco = a.Cout <= 32 ? 32 : 64;
    const int64_t ntiles = (int64_t)a.B * ((a.W + 15) / 16) * ((a.H + 15) / 16);
    if (ntiles < 128) return false;
    a.ncot = 1;
    a.ablate = 0;
    // persistent grid = resident workgroups (registers and LDS both limit it; asked once per instantiation)
    auto go = [&](auto kern) {
        static int per_cu = 0;
        if (per_cu == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, 0) != hipSuccess || n < 1) n = 1;
            per_cu = n;
        }
        int64_t nblk = 256 * (int64_t)per_cu;
        if (nblk > ntiles) nblk = ntiles;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), 0, s, a);
    };
    // residual variant: operands prefetched into registers (costs occupancy)
    const bool r = a.res != nullptr && !(ymk_disabled() & YMK_OFF_RES_PREFETCH);
#define YMK_TILE(CI, BC) (r ? go(conv3x3_tile_kernel<T, CI, BC, true>) : go(conv3x3_tile_kernel<T, CI, BC, false>))
    if (a.Cin == 16) YMK_TILE(16, 32);
    else if (a.Cin == 32 && bco == 32) YMK_TILE(32, 32);
    else if (a.Cin == 32) YMK_TILE(32, 64);
    else if constexpr (sizeof(T) == 2) {  // Cin = 64 tiles fit the 160 KB LDS only in bf16
        if (bco == 32) YMK_TILE(64, 32);
        else YMK_TILE(64, 64);
    } else {
        return false;
    }
#undef YMK_TILE
    return true;
}

template <typename T>
static int launch_conv_dual(ConvArgs a, hipStream_t s) {
    dim3 blk(256);
    a.ncot = 1;
    a.ablate = 0;
    if (a.Cout > 64) {
        a.ncot = (a.Cout + 127) / 128;
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, 1, true>), dim3((unsigned)(ceil_div64(a.M, 128) * a.ncot)), blk, 0, s, a);
    } else if (a.Cout > 32) {
        hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 256, 1, 4, 1, true>), dim3((unsigned)ceil_div64(a.M, 256)), blk, 0, s, a);
    } else {
        hipLaunchKernelGGL((conv_igemm_kernel<T, 32, 256, 1, 4, 1, true>), dim3((unsigned)ceil_div64(a.M, 256)), blk, 0, s, a);
    }
    return ymk_launch_status();
}

template <typename T, int KS>
static int launch_conv(ConvArgs a, hipStream_t s) {
    dim3 blk(256);
    a.ncot = 1;
    a.ablate = 0;
#ifdef YMK_ABLATE
    a.ablate = ymk_ablate;
#endif
    if (a.Cout > 64) {
        a.ncot = (a.Cout + 127) / 128;
        dim3 grid((unsigned)(ceil_div64(a.M, 128) * a.ncot));
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, KS>), grid, blk, 0, s, a);
    } else if (a.Cout > 32) {
        dim3 grid((unsigned)ceil_div64(a.M, 256), 1);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 256, 1, 4, KS>), grid, blk, 0, s, a);
    } else if (a.Cout > 16) {
        dim3 grid((unsigned)ceil_div64(a.M, 256), 1);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 32, 256, 1, 4, KS>), grid, blk, 0, s, a);
    } else {
        dim3 grid((unsigned)ceil_div64(a.M, 256), 1);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 16, 256, 1, 4, KS>), grid, blk, 0, s, a);
    }
    return ymk_launch_status();
}

extern "C" int ymk_activation(int32_t dtype, void* x, int32_t ldx, int64_t npix, int32_t C, int32_t act, void* stream);
static int conv2d_dispatch(const ymk_conv_desc* d, const ymk_conv_desc* dglds, const void* x, const void* w, const float* bias,
                           const void* residual, void* y, void* stream, bool* act_fused);

// act = YMK_ACT_GELU / YMK_ACT_SIGMOID (no residual): the LDS-DMA core applies them in its epilogue; where the shape goes to another
// core that one runs without activation and ONE in-place pass over y follows (what the caller did in two calls before).
extern "C" int ymk_conv2d(const ymk_conv_desc* d, const void* x, const void* w, const float* bias,
                          const void* residual, void* y, void* stream) {
    if (!d) return YMK_E_BADARG;
    bool fused = false;
    if (d->act != YMK_ACT_GELU && d->act != YMK_ACT_SIGMOID) return conv2d_dispatch(d, d, x, w, bias, residual, y, stream, &fused);
    if (residual) return YMK_E_BADARG;
    ymk_conv_desc plain = *d;
    plain.act = YMK_ACT_NONE;
    const int rc = conv2d_dispatch(&plain, d, x, w, bias, residual, y, stream, &fused);
    if (rc != YMK_OK || fused) return rc;
    const int pad = d->ksize / 2;
    const int64_t Ho = (d->H + 2 * pad - d->ksize) / d->stride + 1, Wo = (d->W + 2 * pad - d->ksize) / d->stride + 1;
    return ymk_activation(d->out_dtype, y, d->ldy, (int64_t)d->B * Ho * Wo, d->Cout, d->act, stream);
}

static int conv2d_dispatch(const ymk_conv_desc* d, const ymk_conv_desc* dglds, const void* x, const void* w, const float* bias,
                           const void* residual, void* y, void* stream, bool* act_fused) {
    if (!d || !x || !w || !bias || !y) return YMK_E_BADARG;
    if (d->ksize != 1 && d->ksize != 3) return YMK_E_BADARG;
    if (d->stride != 1 && d->stride != 2) return YMK_E_BADARG;
    const int vec = d->dtype == YMK_BF16 ? 8 : 4;
    if (d->dtype != YMK_F32 && d->dtype != YMK_BF16) return YMK_E_BADARG;
    if (d->Cin % vec || d->ldx % vec || d->Cout % 4 || d->ldy % 4 || d->Kpad % 64) return YMK_E_BADARG;
    if (d->Kpad < d->ksize * d->ksize * d->Cin) return YMK_E_BADARG;
    if (residual && d->ldr % 4) return YMK_E_BADARG;
    if (d->out_dtype != d->dtype && d->out_dtype != YMK_F32) return YMK_E_BADARG;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y;
    a.x2 = nullptr; a.ldx2 = 0; a.C1 = 0; a.up1 = 0; a.H1 = 0; a.W1 = 0; a.pool = nullptr; a.pool_chunks = 0;
    a.B = d->B; a.H = d->H; a.W = d->W;
    const int pad = d->ksize / 2;
    a.Ho = (d->H + 2 * pad - d->ksize) / d->stride + 1;
    a.Wo = (d->W + 2 * pad - d->ksize) / d->stride + 1;
    a.Cin = d->Cin; a.Cout = d->Cout; a.stride = d->stride;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr; a.Kpad = d->Kpad; a.act = d->act;
    a.out_f32 = (d->out_dtype == YMK_F32 && d->dtype != YMK_F32) ? 1 : 0;
    a.M = (int64_t)d->B * a.Ho * a.Wo;
    if (a.M <= 0) return YMK_OK;
    if (a.M >= (1ll << 31) || (int64_t)d->B * d->H * d->W >= (1ll << 31)) return YMK_E_BADARG;
    // in-kernel row offsets are 32-bit relative to a per-tile reference pixel (at most two images away)
    if ((2ll * d->H * d->W + 4096) * d->ldx >= (1ll << 31)) return YMK_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (d->ksize == 1 && d->stride == 1 && ymk_use_ws && !(ymk_disabled() & YMK_OFF_CONV_STREAM)) {  // large-M short-K 1x1: weight-stationary streaming kernel
        const bool done = d->dtype == YMK_F32 ? launch_conv1x1_ws<float>(a, s) : launch_conv1x1_ws<h16_t>(a, s);
        if (done) { ymk_last_variant = YMK_CONV_STREAM_1X1; return ymk_launch_status(); }
    }
    // 64 -> 64 3x3 on the small maps (C3k bottlenecks at 40x40 / 20x20): the LDS-DMA tiled core is 0-22 % faster than the
    // spatial-tile kernel up to ~100k output pixels (gpu_diag.py glds2: 29.6 -> 23.1 us at 64 x 40 x 40), slower above
    const bool small64 = d->dtype == YMK_BF16 && d->Cin == 64 && d->Cout == 64 && d->stride == 1 && a.M <= 102400 &&
                         !(ymk_disabled() & YMK_OFF_CONV_GLDS3);
    if (d->ksize == 3 && ymk_use_ws && !small64 && !(ymk_disabled() & YMK_OFF_CONV_STREAM)) {  // small-Cin stride-1 3x3: spatial-tile kernel (LDS-staged im2col)
        const bool done = d->dtype == YMK_F32 ? launch_conv3x3_tile<float>(a, s) : launch_conv3x3_tile<h16_t>(a, s);
        if (done) { ymk_last_variant = YMK_CONV_SPATIAL_3X3; return ymk_launch_status(); }
    }
    if (d->ksize == 3 && d->dtype == YMK_BF16 && !(ymk_disabled() & YMK_OFF_CONV_GLDS3) && !(ymk_enabled() & YMK_ON_CONV_GLDS)) {
        // 3x3 with Cin >= 64 (the stride-2 down-sampling convs, the Detect / C3k 3x3s): LDS-DMA tiled core.  Measured faster than
        // conv_igemm_kernel on every such shape of the S detector (profiles/r02a_glds_ab.log: 6-35 %), two LDS stages: with the
        // 128-pixel tiles the core now takes (two or three workgroups per CU) the three-stage counted-vmcnt loop no longer wins
        // anywhere (profiles/r02_glds_tile_ab.txt).  YMK_GLDS_THREE_STAGE=1 brings it back for A/B runs.
        static const bool three = [] { const char* e = getenv("YMK_GLDS_THREE_STAGE"); return e && atoi(e) != 0; }();
        if (d->Cout % 64 == 0) {
            const int rc = ymk_conv2d_glds(dglds, x, w, bias, residual, y, three ? 0 : 1, stream);
            if (rc != YMK_E_BADARG) { ymk_last_variant = YMK_CONV_GLDS; ymk_last_glds_stages = three ? 3 : 2; *act_fused = true; return rc; }
        }
    }
    // The tiled 1x1 shapes (K >= 256 and Cout a multiple of 128, or Cout a multiple of 64; at least 192 tiles of 256 pixels): with 128-pixel tiles and the two-stage
    // loop the LDS-DMA core is 0-20 % faster than conv_igemm_kernel on 15 of the 17 such launches of the S detector and equal on the
    // rest (profiles/r02_glds_tile_ab.txt; with 256-pixel tiles it was slower, which is why it was opt-in before).
    // YMK_ENABLE bit 0 still routes EVERY shape the core accepts here (bit 1: with the two-stage loop).
    const bool glds_all = (ymk_enabled() & YMK_ON_CONV_GLDS) != 0;
    // (64-cout tiles: from K = 64 — 64->64 at 80^2 40 -> 32 us, 128->64 at 40^2 16 -> 15 us against the 64 x 256 tiled kernel)
    const bool glds_1x1 = d->ksize == 1 && d->Cout % 64 == 0 && !(ymk_disabled() & YMK_OFF_CONV_GLDS1);
    if ((glds_all || glds_1x1) && d->dtype == YMK_BF16) {
        const int64_t tiles = ceil_div64(a.M, 256) * (d->Cout / (d->Cout % 128 == 0 ? 128 : 64));
        const int min_k = (glds_all || d->Cout % 128 == 0) ? ymk_glds_min_k : 64;
        if (d->Cout % 64 == 0 && tiles >= ymk_glds_min_tiles && d->Kpad >= min_k) {
            const bool two = glds_all ? (ymk_enabled() & YMK_ON_GLDS_TWO_STAGE) != 0 : true;
            const int rc = ymk_conv2d_glds(dglds, x, w, bias, residual, y, two ? 1 : 0, stream);
            if (rc != YMK_E_BADARG) { ymk_last_variant = YMK_CONV_GLDS; ymk_last_glds_stages = two ? 2 : 3; *act_fused = true; return rc; }
        }
    }
    ymk_last_variant = YMK_CONV_TILED;
    if (d->dtype == YMK_F32)
        return d->ksize == 1 ? launch_conv<float, 1>(a, s) : launch_conv<float, 3>(a, s);
    return d->ksize == 1 ? launch_conv<h16_t, 1>(a, s) : launch_conv<h16_t, 3>(a, s);
}

// 1x1 conv over the channel concatenation [x1 (optionally 2x nearest-upsampled) | x2] without materialising it:
// nn.Upsample + Concat + C2f.cv1 of the neck (yaml head rows 13-15 / 16-18; conv.py:629-641).
extern "C" int ymk_conv1x1_cat2(const ymk_conv_desc* d, const void* x1, int32_t C1, int32_t ldx1, int32_t upsample1,
                                const void* x2, int32_t ldx2, const void* w, const float* bias, void* y, void* stream) {
    if (!d || !x1 || !x2 || !w || !bias || !y || d->ksize != 1 || d->stride != 1) return YMK_E_BADARG;
    const int vec = d->dtype == YMK_BF16 ? 8 : 4;
    if (d->dtype != YMK_F32 && d->dtype != YMK_BF16) return YMK_E_BADARG;
    if (C1 <= 0 || C1 >= d->Cin || C1 % vec || d->Cin % vec || ldx1 % vec || ldx2 % vec || d->Cout % 4 || d->ldy % 4 ||
        d->Kpad % 64 || d->Kpad < d->Cin || d->out_dtype != d->dtype)
        return YMK_E_BADARG;
    if (upsample1 && ((d->H & 1) || (d->W & 1))) return YMK_E_BADARG;
    ConvArgs a;
    a.x = x1; a.w = w; a.bias = bias; a.res = nullptr; a.y = y;
    a.x2 = x2; a.ldx2 = ldx2; a.C1 = C1; a.up1 = upsample1 ? 1 : 0; a.H1 = d->H / 2; a.W1 = d->W / 2;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->H; a.Wo = d->W;
    a.Cin = d->Cin; a.Cout = d->Cout; a.stride = 1;
    a.ldx = ldx1; a.ldy = d->ldy; a.ldr = 0; a.Kpad = d->Kpad; a.act = d->act; a.out_f32 = 0;
    a.M = (int64_t)d->B * d->H * d->W;
    if (a.M <= 0) return YMK_OK;
    if (a.M >= (1ll << 31) || (2ll * d->H * d->W + 4096) * (ldx1 > ldx2 ? ldx1 : ldx2) >= (1ll << 31)) return YMK_E_BADARG;
    // same rule as the tiled 1x1 shapes of ymk_conv2d: the LDS-DMA core with 128-pixel tiles and the two-stage loop is 5-8 % faster on
    // the four cat2 launches of the S detector (118 -> 112, 79 -> 72, 57 -> 52, 27 -> 25 us)
    const bool glds_all = (ymk_enabled() & YMK_ON_CONV_GLDS) != 0;
    const bool glds_1x1 = d->Cout % 128 == 0 && !(ymk_disabled() & YMK_OFF_CONV_GLDS1);
    if ((glds_all || glds_1x1) && d->dtype == YMK_BF16 && d->Cout % 64 == 0 &&
        ceil_div64(a.M, 256) * (d->Cout / (d->Cout % 128 == 0 ? 128 : 64)) >= ymk_glds_min_tiles && d->Kpad >= ymk_glds_min_k) {
        const bool two = glds_all ? (ymk_enabled() & YMK_ON_GLDS_TWO_STAGE) != 0 : true;
        const int rc = ymk_conv1x1_cat2_glds(d, x1, C1, ldx1, upsample1, x2, ldx2, w, bias, y, two ? 1 : 0, stream);
        if (rc != YMK_E_BADARG) { ymk_last_variant = YMK_CONV_GLDS; ymk_last_glds_stages = two ? 2 : 3; return rc; }
    }
    ymk_last_variant = YMK_CONV_TILED;
    return d->dtype == YMK_F32 ? launch_conv_dual<float>(a, (hipStream_t)stream) : launch_conv_dual<h16_t>(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// Stem: NCHW fp32 input with tiny Cin (3) -> NHWC.  Direct VALU convolution: the
// layer is output-bandwidth bound (27 MACs per output element), one thread per
// output pixel x 8 couts, weights broadcast from LDS.
// ---------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, TO* __restrict__ y,
                                                    int B, int Cin, int H, int W, int Ho, int Wo, int Cout,
                                                    int ks, int stride, int ldy, int act) {
    extern __shared__ float sw[];  // [Cout][K] then bias[Cout]
    const int K = ks * ks * Cin;
    for (int i = threadIdx.x; i < Cout * K; i += blockDim.x) sw[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[Cout * K + i] = bias[i];
    __syncthreads();
    const int ng = Cout / 8;  // groups of 8 couts
    const int64_t total = (int64_t)B * Ho * Wo * ng;
    const int pad = ks / 2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % ng);
        const int64_t m = idx / ng;
        const int ox = (int)(m % Wo);
        const int oy = (int)((m / Wo) % Ho);
        const int b = (int)(m / ((int64_t)Wo * Ho));
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < ks; ++ky) {
            const int iy = oy * stride - pad + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < ks; ++kx) {
                const int ix = ox * stride - pad + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                for (int c = 0; c < Cin; ++c) {
                    const float xv = x[(((int64_t)b * Cin + c) * H + iy) * W + ix];
                    const int k = (ky * ks + kx) * Cin + c;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, sw[(g * 8 + j) * K + k], acc[j]);
                }
            }
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = acc[j] + sw[Cout * K + g * 8 + j];
            v[j] = act == YMK_ACT_SILU ? silu_exact(t) : t;
        }
        TO* o = y + m * ldy + g * 8;
        store4(o, v[0], v[1], v[2], v[3]);
        store4(o + 4, v[4], v[5], v[6], v[7]);
    }
}

// Specialised stem: one thread = one output pixel x all CO output channels.  The k*k*Cin input
// values of a pixel are loaded once (lanes walk along x, so loads coalesce), the weights are read
// with wave-uniform indices (scalar loads -> SGPR operands of v_fma_f32), the CO results are written
// as one contiguous NHWC row.  VALU-bound only by the 2*K*CO flops per pixel.
template <typename TO, int CO>
__global__ __launch_bounds__(256) void stem_px_kernel(const float* __restrict__ x, const float* __restrict__ wt /*[K][CO]*/,
                                                       const float* __restrict__ bias, TO* __restrict__ y, int B, int Cin,
                                                       int H, int W, int Ho, int Wo, int ks, int stride, int ldy, int act) {
    const int64_t total = (int64_t)B * Ho * Wo;
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= total) return;
    const int ox = (int)(m % Wo);
    const int oy = (int)((m / Wo) % Ho);
    const int b = (int)(m / ((int64_t)Wo * Ho));
    const int pad = ks / 2;
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = bias[j];
    for (int ky = 0; ky < ks; ++ky) {
        const int iy = oy * stride - pad + ky;
        for (int kx = 0; kx < ks; ++kx) {
            const int ix = ox * stride - pad + kx;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            for (int c = 0; c < Cin; ++c) {
                const float xv = ok ? x[(((int64_t)b * Cin + c) * H + iy) * W + ix] : 0.f;
                const float* wk = wt + ((ky * ks + kx) * Cin + c) * CO;  // wave-uniform address
#pragma unroll
                for (int j = 0; j < CO; ++j) acc[j] = fmaf(xv, wk[j], acc[j]);
            }
        }
    }
    TO* o = y + m * ldy;
#pragma unroll
    for (int j = 0; j < CO; j += 4) {
        float v0 = acc[j], v1 = acc[j + 1], v2 = acc[j + 2], v3 = acc[j + 3];
        if (act == YMK_ACT_SILU) { v0 = silu_exact(v0); v1 = silu_exact(v1); v2 = silu_exact(v2); v3 = silu_exact(v3); }
        store4(o + j, v0, v1, v2, v3);
    }
}


// Stem on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): one wave owns one output row (b, oy) and walks over
// groups of 16 output pixels.  The weights are the MFMA row operand, held in registers for the wave's whole life
// (K = ks*ks*Cin <= 32, zero padded); the pixel operand is gathered straight from the NCHW fp32 image (each lane
// fetches the 8 taps of its k-group; neighbouring lanes read neighbouring pixels, the rest hits L1/L2).  Stays
// in fp32 end to end, so the first layer adds no bf16 rounding of the input image; the result leaves as NHWC
// rows in the activation dtype.  ~16 MFMAs + 8 loads per 16 pixels instead of 27*CO scalar FMAs per pixel.
template <typename TO, int CO>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wt /*[K][CO]*/,
                                                         const float* __restrict__ bias, TO* __restrict__ y, int B, int Cin,
                                                         int H, int W, int Ho, int Wo, int ks, int stride, int ldy, int act) {
    constexpr int TM = CO / 16;
    constexpr bool PRECISE = sizeof(TO) == 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // (b, oy)
    if (row >= (int64_t)B * Ho) return;
    const int b = (int)(row / Ho), oy = (int)(row % Ho);
    const int K = ks * ks * Cin, pad = ks / 2;

    // per-lane tap table: k = kk*16 + fc*4 + v  ->  (ky, kx, c)
    u32x4 af[TM][2];
    int off[8];      // element offset of the tap inside the image for ox = 0 (may be negative at the borders)
    int dx[8];       // kx - pad
    bool rok[8];     // tap exists and its input row is inside the image
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = (q >> 2) * 16 + fc * 4 + (q & 3);
        const int tap = k / Cin, c = k - tap * Cin;
        const int ky = tap / ks, kx = tap - ky * ks;
        const int iy = oy * stride - pad + ky;
        rok[q] = k < K && (unsigned)iy < (unsigned)H;
        dx[q] = kx - pad;
        off[q] = (c * H + iy) * W + dx[q];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float wv = k < K ? wt[k * CO + i * 16 + fr] : 0.f;
            reinterpret_cast<float*>(&af[i][q >> 2])[q & 3] = wv;
        }
    }
    f32x4 bv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bv[i] = *reinterpret_cast<const f32x4*>(bias + i * 16 + fc * 4);
    const float* xb = x + (size_t)b * Cin * H * W;
    TO* yrow = y + (size_t)row * Wo * ldy;

    // SG groups of 16 pixels per iteration: all 8*SG gathers are issued before the first MFMA so that enough
    // loads are in flight per wave to cover the HBM latency (one group at a time is latency-bound)
    constexpr int SG = 4;
    for (int ox0 = 0; ox0 < Wo; ox0 += 16 * SG) {
        u32x4 bf[SG][2];
#pragma unroll
        for (int sg = 0; sg < SG; ++sg) {
            const int ox = ox0 + sg * 16 + fr;
            const int ixb = ox * stride;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ix = ixb + dx[q];
                float v = 0.f;
                if (rok[q] && ox < Wo && (unsigned)ix < (unsigned)W) v = xb[off[q] + ixb];
                reinterpret_cast<float*>(&bf[sg][q >> 2])[q & 3] = v;
            }
        }
#pragma unroll
        for (int sg = 0; sg < SG; ++sg) {
            const int ox = ox0 + sg * 16 + fr;
            f32x4 acc[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i] = bv[i];
                mma16<float>(acc[i], af[i][0], bf[sg][0]);
                mma16<float>(acc[i], af[i][1], bf[sg][1]);
            }
            if (ox < Wo) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v0 = acc[i].x, v1 = acc[i].y, v2 = acc[i].z, v3 = acc[i].w;
                    if (act == YMK_ACT_SILU) {
                        if (PRECISE) { v0 = silu_exact(v0); v1 = silu_exact(v1); v2 = silu_exact(v2); v3 = silu_exact(v3); }
                        else { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
                    }
                    store4(yrow + (size_t)ox * ldy + i * 16 + fc * 4, v0, v1, v2, v3);
                }
            }
        }
    }
}


// Stem, LDS-staged: a workgroup produces 4 consecutive output rows of one image (one per wave).  The
// (3*stride + ks) input rows x Cin channel planes they need are copied from the NCHW fp32 image into LDS with
// coalesced 16-byte loads (all of a thread's loads in flight together), with a zero column band left and right and
// zero rows above/below the image, so the per-tap gathers of the MFMA pixel operand are unconditional 4-byte LDS
// reads.  Arithmetic identical to stem_mfma_kernel (fp32 matrix cores, weights resident in registers).
#define STEM_OR 4     // output rows per workgroup
#define STEM_XP 4     // zero columns on each side of a staged row
template <typename TO, int CO>
__global__ __launch_bounds__(256) void stem_rows_kernel(const float* __restrict__ x, const float* __restrict__ wt /*[K][CO]*/,
                                                         const float* __restrict__ bias, TO* __restrict__ y, int B, int Cin,
                                                         int H, int W, int Ho, int Wo, int ks, int stride, int ldy, int act) {
    extern __shared__ __attribute__((aligned(16))) float srow[];  // [Cin][R][W + 2*STEM_XP]
    constexpr int TM = CO / 16;
    constexpr bool PRECISE = sizeof(TO) == 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int rblk = (Ho + STEM_OR - 1) / STEM_OR;
    const int b = blockIdx.x / rblk, oy0 = (blockIdx.x % rblk) * STEM_OR;
    const int K = ks * ks * Cin, pad = ks / 2;
    const int R = (STEM_OR - 1) * stride + ks;
    const int WP = W + 2 * STEM_XP, W4 = W >> 2;
    const float* xb = x + (size_t)b * Cin * H * W;

    // stage: zero bands, then the rows
    for (int i = t; i < Cin * R * 2 * STEM_XP; i += 256) {
        const int rr = i / (2 * STEM_XP), j = i % (2 * STEM_XP);
        srow[rr * WP + (j < STEM_XP ? j : W + j)] = 0.f;
    }
    const int nchunk = Cin * R * W4;
    constexpr int NB = 18;
    for (int q0 = 0; q0 < nchunk; q0 += 256 * NB) {
        f32x4 v[NB];
#pragma unroll
        for (int l = 0; l < NB; ++l) {
            const int q = q0 + l * 256 + t;
            v[l] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < nchunk) {
                const int rr = q / W4, xq = q - rr * W4;
                const int c = rr / R, r = rr - c * R;
                const int iy = oy0 * stride - pad + r;
                if ((unsigned)iy < (unsigned)H) v[l] = *reinterpret_cast<const f32x4*>(xb + ((size_t)c * H + iy) * W + xq * 4);
            }
        }
#pragma unroll
        for (int l = 0; l < NB; ++l) {
            const int q = q0 + l * 256 + t;
            if (q < nchunk) {
                const int rr = q / W4, xq = q - rr * W4;
                *reinterpret_cast<f32x4*>(srow + rr * WP + STEM_XP + xq * 4) = v[l];
            }
        }
    }
    // weights / tap table (see stem_mfma_kernel): k = kk*16 + fc*4 + v -> (ky, kx, c)
    u32x4 af[TM][2];
    int off[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = (q >> 2) * 16 + fc * 4 + (q & 3);
        const int kc = k < K ? k : 0;
        const int tap = kc / Cin, c = kc - tap * Cin;
        const int ky = tap / ks, kx = tap - ky * ks;
        off[q] = (c * R + wave * stride + ky) * WP + kx + STEM_XP - pad;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float wv = k < K ? wt[k * CO + i * 16 + fr] : 0.f;
            reinterpret_cast<float*>(&af[i][q >> 2])[q & 3] = wv;
        }
    }
    f32x4 bv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bv[i] = *reinterpret_cast<const f32x4*>(bias + i * 16 + fc * 4);
    __syncthreads();
    const int oy = oy0 + wave;
    if (oy >= Ho) return;
    TO* yrow = y + ((size_t)b * Ho + oy) * Wo * ldy;
    for (int ox0 = 0; ox0 < Wo; ox0 += 16) {
        const int ox = ox0 + fr;
        const int oxc = ox < Wo ? ox : Wo - 1;   // clamp the read, skip the store
        u32x4 bf[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) reinterpret_cast<float*>(&bf[q >> 2])[q & 3] = srow[off[q] + oxc * stride];
        f32x4 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            acc[i] = bv[i];
            mma16<float>(acc[i], af[i][0], bf[0]);
            mma16<float>(acc[i], af[i][1], bf[1]);
        }
        if (ox < Wo) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float v0 = acc[i].x, v1 = acc[i].y, v2 = acc[i].z, v3 = acc[i].w;
                if (act == YMK_ACT_SILU) {
                    if (PRECISE) { v0 = silu_exact(v0); v1 = silu_exact(v1); v2 = silu_exact(v2); v3 = silu_exact(v3); }
                    else { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
                }
                store4(yrow + (size_t)ox * ldy + i * 16 + fc * 4, v0, v1, v2, v3);
            }
        }
    }
}

template <typename TO, int CO>
static bool launch_stem_rows(const float* x, const float* wt, const float* bias, void* y, int B, int Cin, int H, int W, int Ho,
                             int Wo, int ks, int stride, int ldy, int act, hipStream_t s) {
    const int R = (STEM_OR - 1) * stride + ks;
    const size_t shm = (size_t)Cin * R * (W + 2 * STEM_XP) * sizeof(float);
    // reads reach column (Wo-1)*stride + ks-1 - pad + STEM_XP of a staged row: must stay inside W + 2*STEM_XP
    if ((W & 3) || ks / 2 > STEM_XP || (Wo - 1) * stride + ks - 1 - ks / 2 + STEM_XP >= W + 2 * STEM_XP || shm > 150 * 1024 ||
        ((uintptr_t)x & 15))
        return false;
    static size_t attr = 0;
    if (shm > 64 * 1024 && shm > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_rows_kernel<TO, CO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr = shm;
    }
    const int rblk = (Ho + STEM_OR - 1) / STEM_OR;
    hipLaunchKernelGGL((stem_rows_kernel<TO, CO>), dim3((unsigned)(B * rblk)), dim3(256), shm, s, x, wt, bias, (TO*)y, B, Cin, H,
                       W, Ho, Wo, ks, stride, ldy, act);
    return true;
}

template <typename TO, int CO>
static void launch_stem_mfma(const float* x, const float* wt, const float* bias, void* y, int B, int Cin, int H, int W, int Ho,
                             int Wo, int ks, int stride, int ldy, int act, hipStream_t s) {
    const int64_t rows = (int64_t)B * Ho;
    hipLaunchKernelGGL((stem_mfma_kernel<TO, CO>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, wt, bias, (TO*)y, B, Cin,
                       H, W, Ho, Wo, ks, stride, ldy, act);
}

template <typename TO, int CO>
static void launch_stem_px(const float* x, const float* wt, const float* bias, void* y, int B, int Cin, int H, int W, int Ho,
                           int Wo, int ks, int stride, int ldy, int act, hipStream_t s) {
    const int64_t total = (int64_t)B * Ho * Wo;
    hipLaunchKernelGGL((stem_px_kernel<TO, CO>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, wt, bias, (TO*)y, B,
                       Cin, H, W, Ho, Wo, ks, stride, ldy, act);
}

extern "C" int ymk_conv2d_stem_nchw(const float* x, const float* w, const float* wt_kco, const float* bias, void* y,
                                    int32_t out_dtype, int32_t B, int32_t Cin, int32_t H, int32_t W,
                                    int32_t Cout, int32_t ksize, int32_t stride, int32_t ldy,
                                    int32_t act, void* stream) {
    if (!x || !w || !bias || !y || Cout % 8 || ldy % 8 || Cin < 1 || Cin > 4 || (ksize & 1) == 0)
        return YMK_E_BADARG;
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const int64_t total = (int64_t)B * Ho * Wo * (Cout / 8);
    if (total <= 0) return YMK_OK;
    const size_t shm = ((size_t)Cout * ksize * ksize * Cin + Cout) * sizeof(float);
    if (shm > 64 * 1024) return YMK_E_BADARG;
    const int blocks = (int)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    hipStream_t s = (hipStream_t)stream;
    if (wt_kco && (Cout == 16 || Cout == 32 || Cout == 64) && (out_dtype == YMK_F32 || out_dtype == YMK_BF16)) {
        const bool f = out_dtype == YMK_F32;
        if (ksize * ksize * Cin <= 32 && (int64_t)Cin * H * W < (1ll << 30) && !(ymk_disabled() & YMK_OFF_STEM_FAST)) {
            if (!(ymk_disabled() & YMK_OFF_STEM_ROWS)) {  // LDS-staged rows when the geometry allows
#define YMK_STEM_R(CO)                                                                                              \
    (f ? launch_stem_rows<float, CO>(x, wt_kco, bias, y, B, Cin, H, W, Ho, Wo, ksize, stride, ldy, act, s)         \
       : launch_stem_rows<h16_t, CO>(x, wt_kco, bias, y, B, Cin, H, W, Ho, Wo, ksize, stride, ldy, act, s))
                const bool done = Cout == 16 ? YMK_STEM_R(16) : Cout == 32 ? YMK_STEM_R(32) : YMK_STEM_R(64);
#undef YMK_STEM_R
                if (done) return ymk_launch_status();
            }
#define YMK_STEM_M(CO)                                                                                              \
    (f ? launch_stem_mfma<float, CO>(x, wt_kco, bias, y, B, Cin, H, W, Ho, Wo, ksize, stride, ldy, act, s)         \
       : launch_stem_mfma<h16_t, CO>(x, wt_kco, bias, y, B, Cin, H, W, Ho, Wo, ksize, stride, ldy, act, s))
            if (Cout == 16) YMK_STEM_M(16);
            else if (Cout == 32) YMK_STEM_M(32);
            else YMK_STEM_M(64);
#undef YMK_STEM_M
            return ymk_launch_status();
        }
#define YMK_STEM(CO)                                                                                                \
    (f ? launch_stem_px<float, CO>(x, wt_kco, bias, y, B, Cin, H, W, Ho, Wo, ksize, stride, ldy, act, s)           \
       : launch_stem_px<h16_t, CO>(x, wt_kco, bias, y, B, Cin, H, W, Ho, Wo, ksize, stride, ldy, act, s))
        if (Cout == 16) YMK_STEM(16);
        else if (Cout == 32) YMK_STEM(32);
        else YMK_STEM(64);
#undef YMK_STEM
        return ymk_launch_status();
    }
    if (out_dtype == YMK_F32)
        hipLaunchKernelGGL(stem_kernel<float>, dim3(blocks), dim3(256), shm, s, x, w, bias, (float*)y, B,
                           Cin, H, W, Ho, Wo, Cout, ksize, stride, ldy, act);
    else if (out_dtype == YMK_BF16)
        hipLaunchKernelGGL(stem_kernel<h16_t>, dim3(blocks), dim3(256), shm, s, x, w, bias, (h16_t*)y,
                           B, Cin, H, W, Ho, Wo, Cout, ksize, stride, ldy, act);
    else
        return YMK_E_BADARG;
    return ymk_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Streaming 1x1 convolution that also leaves the per-tile channel sums of its output (round 5): the producer of an ES-MoE layer's input
// hands the router its global average pool (moe/routers.py:458-527) instead of the router re-reading the map (210 MB at 64 x 80 x 80 x 256).
// pool_part fp32 [B][ymk_conv1x1_pool_chunks(...)][Cout]: sums over the 128-pixel tiles of each image of the values AS STORED, fixed order;
// feed it to ymk_esmoe_route_pooled.  Shapes: 16-bit types, 1x1 stride 1, Cout % 64 == 0, H * W % 128 == 0, and what conv1x1_ws_kernel takes
// (Cin <= 256 ..., enough tiles) — ymk_conv1x1_pool_chunks returns 0 otherwise and ymk_conv1x1_pooled YMK_E_BADARG.
// ---------------------------------------------------------------------------------------------------------------------------------
static bool conv1x1_pool_ok(const ymk_conv_desc* d) {
    if (!d || d->dtype != YMK_BF16 || d->out_dtype != YMK_BF16 || d->ksize != 1 || d->stride != 1) return false;
    if (d->Cout % 64 || d->Cout <= 64 || ((int64_t)d->H * d->W) % 128 || d->Cin % 8 || d->ldx % 8 || d->ldy % 4 || d->Kpad % 64) return false;
    if (d->act != YMK_ACT_NONE && d->act != YMK_ACT_SILU) return false;
    const int kg = d->Kpad / 64;
    const int64_t ntiles = (int64_t)d->B * d->H * d->W / 128;
    if (kg < 1 || kg > 4 || d->Kpad < d->Cin || ntiles < (kg <= 2 ? ymk_ws_min_tiles / 2 : ymk_ws_min_tiles)) return false;
    return ymk_use_ws && !(ymk_disabled() & YMK_OFF_CONV_STREAM);
}
extern "C" int32_t ymk_conv1x1_pool_chunks(const ymk_conv_desc* d) { return conv1x1_pool_ok(d) ? (int32_t)((int64_t)d->H * d->W / 128) : 0; }
extern "C" int ymk_conv1x1_pooled(const ymk_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y,
                                  float* pool_part, void* stream) {
    if (!d || !x || !w || !bias || !y || !pool_part || !conv1x1_pool_ok(d) || (residual && d->ldr % 4)) return YMK_E_BADARG;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y;
    a.x2 = nullptr; a.ldx2 = 0; a.C1 = 0; a.up1 = 0; a.H1 = 0; a.W1 = 0;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->H; a.Wo = d->W;
    a.Cin = d->Cin; a.Cout = d->Cout; a.stride = 1;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr; a.Kpad = d->Kpad; a.act = d->act; a.out_f32 = 0;
    a.M = (int64_t)d->B * d->H * d->W;
    if (a.M <= 0) return YMK_OK;
    if (a.M >= (1ll << 31) || (2ll * d->H * d->W + 4096) * d->ldx >= (1ll << 31)) return YMK_E_BADARG;
    a.pool = pool_part;
    a.pool_chunks = (int)((int64_t)d->H * d->W / 128);
    if (!launch_conv1x1_ws<h16_t>(a, (hipStream_t)stream)) return YMK_E_BADARG;
    ymk_last_variant = YMK_CONV_STREAM_1X1;
    return ymk_launch_status();
}

// The same sums from a map that is already in memory (a producer the pooled kernel does not take — too few tiles at a small batch, another
// kernel family): one workgroup per (image, 128-pixel tile), a thread per channel, the summation order of conv1x1_ws_kernel<..., POOL>
// restated — so a router's decision does not depend on which kernel produced its input (tests: batch-independence at the full size).
template <typename T>
__global__ __launch_bounds__(128) void pool_tiles128_kernel(const T* __restrict__ y, int ldy, int HW, int C, float* __restrict__ part) {
    const int chunks = HW / 128;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int c = blockIdx.y * 128 + threadIdx.x;
    if (c >= C) return;
    const T* base = y + ((size_t)b * HW + (size_t)ch * 128) * ldy + c;
    float half[2];
    for (int wpx = 0; wpx < 2; ++wpx) {
        float sfr[16];
#pragma unroll
        for (int fr = 0; fr < 16; ++fr) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += to_f32_elem(base[(size_t)((wpx * 4 + j) * 16 + fr) * ldy]);
            sfr[fr] = acc;
        }
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            float nx[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) nx[i] = sfr[i] + sfr[row16_partner(i, step)];
#pragma unroll
            for (int i = 0; i < 16; ++i) sfr[i] = nx[i];
        }
        half[wpx] = sfr[0];
    }
    part[((size_t)b * chunks + ch) * C + c] = half[0] + half[1];
}

extern "C" int ymk_pool_tiles128(int32_t dtype, const void* y, int32_t ldy, int32_t B, int32_t HW, int32_t C, float* pool_part, void* stream) {
    if (!y || !pool_part || HW <= 0 || HW % 128 || C < 1 || ldy < C || (dtype != YMK_BF16 && dtype != YMK_F32)) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    const int64_t nb = (int64_t)B * (HW / 128);
    if (nb >= (1ll << 31)) return YMK_E_BADARG;
    dim3 grid((unsigned)nb, (unsigned)((C + 127) / 128));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == YMK_F32) hipLaunchKernelGGL(pool_tiles128_kernel<float>, grid, dim3(128), 0, s, (const float*)y, ldy, HW, C, pool_part);
    else hipLaunchKernelGGL(pool_tiles128_kernel<h16_t>, grid, dim3(128), 0, s, (const h16_t*)y, ldy, HW, C, pool_part);
    return ymk_launch_status();
}
