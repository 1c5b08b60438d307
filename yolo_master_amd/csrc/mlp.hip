// Fused two-layer 1x1 MLP with residual on MFMA (bf16):   y = x + W2 * SiLU(W1 * x + b1) + b2      per token.
// Reference: ABlock.forward `x = x + self.mlp(x)` with mlp = Conv(c, h, 1) -> Conv(h, c, 1, act=False), BN folded
// (ultralytics/nn/modules/block.py:1772-1797, conv.py:80-89).
//
// Why.  On the 40x40 / 20x20 maps of the detector the two 1x1 convolutions of every ABlock are separate launches of
// 25-45 us each for 5-10 us of HBM traffic and ~10 us of matrix work (102,400 / 25,600 tokens at batch 64): they are
// bound by the latency of their own short k-loops, and the hidden tensor (2x the block's width) makes a round trip
// through HBM in between.  Fused, a workgroup owns 64 tokens: x tile -> LDS once (it is also the residual), the hidden
// tile lives in LDS as bf16 (exactly the value the unfused pair stores and re-reads), both weight matrices come from
// L2 straight into MFMA A fragments (16-byte loads, a whole phase requested up front), nothing but x and y touches HBM.
//
// Orientation as csrc/igemm.h: D[cout][token] = sum_k W[cout][k] * X[token][k]; A fragment = 16 weight rows, lane l
// holds row (l & 15), k = (l >> 4) * 8 .. + 8; B fragment = 16 tokens, same shape; D: lane l, reg r -> cout (l >> 4) * 4 + r,
// token (l & 15).  NW waves: each takes 1/NW of the hidden units (phase 1) / of the output channels (phase 2) for all 64
// tokens.  HT / CT = 16-row tiles per wave in the two phases: (C, hidden) = (64,128) (128,256) with 4 waves, (256,512) with 8
// (its 100 KB of LDS allow one workgroup per CU: eight waves keep the SIMDs covered).
//
// PROJ (round 5): AAttn's output projection and the ABlock's first skip in front of it, in the same kernel —
//   x1 = x + Wp a + bp   (a = attention output + positional stencil; AAttn.forward `self.proj(x + pp)`, block.py:1727-1732; ABlock.forward
//   `x = x + self.attn(x)`, :1787-1797),   y = x1 + W2 SiLU(W1 x1 + b1) + b2.
// The a tile is what is staged; each wave produces its output channels of x1 for the 64 tokens, adds the residual it reads from HBM, rounds to
// the 16-bit type (the value the unfused projection stores) and writes it into the x tile in LDS — from there on the kernel is the plain one
// (x1 is both the MLP's input and its skip).  One launch and one [B, H, W, C] round trip less per ABlock.
#include "ymk_common.h"

typedef __bf16 mlp_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mlp_mma(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = mfma16x16x32_h16(a, b, acc);
}

#define MLP_BM 64

struct MlpArgs {
    const h16_t* x; const h16_t* w1; const float* b1; const h16_t* w2; const float* b2; h16_t* y;
    int M, C, Hd, ldx, ldy, k1pad, k2pad;
    // PROJ: a = the projection's input [M][lda], wp [C][kppad], bp [C]; x is then the residual of the projection
    const h16_t* a; const h16_t* wp; const float* bp;
    int lda, kppad;
};

// PERSIST: the workgroup walks token tiles blockIdx.x, blockIdx.x + gridDim.x, ... with BOTH weight matrices' fragments resident in
// registers (they are the same for every tile) and the next tile's tokens prefetched into registers during the current tile's
// arithmetic; used where the fragments fit (C = 128: 2 x 64 registers).  !PERSIST: one tile per workgroup, fragments requested per
// phase (C = 256: 2 x 128 registers would not fit next to the accumulators).
template <int HT, int CT, int NW, bool PERSIST, bool PROJ = false>
__global__ __launch_bounds__(NW * 64) void mlp_fused_kernel(MlpArgs a) {
    constexpr int C = CT * 16 * NW, Hd = HT * 16 * NW, NT = NW * 64;
    // LDS row pitches in bytes: row bytes (multiples of 256 here) + 32 = 8 dwords modulo 64, the pitch at which the sixteen rows of a
    // ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) cover the 64 banks exactly once; with a 16-byte pad rows r and r + 8 of a group met
    constexpr int XP = C * 2 + 32, HP = Hd * 2 + 32;
    constexpr int CPR = C / 8;                                 // 16-byte chunks per token row
    constexpr int NX = (MLP_BM * CPR + NT - 1) / NT;           // staging loads per thread
    static_assert((MLP_BM * CPR) % NT == 0, "token tile is a whole number of passes");
    __shared__ __attribute__((aligned(16))) char sX[MLP_BM * XP];
    __shared__ __attribute__((aligned(16))) char sH[MLP_BM * HP];
    __shared__ __attribute__((aligned(16))) char sA[PROJ ? MLP_BM * XP : 16];   // PROJ: the projection's input tile
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int ntiles = (a.M + MLP_BM - 1) / MLP_BM;

    const h16_t* w1row = a.w1 + (size_t)(wave * HT * 16 + fr) * a.k1pad + fc * 8;
    const h16_t* w2row = a.w2 + (size_t)(wave * CT * 16 + fr) * a.k2pad + fc * 8;
    const h16_t* wprow = PROJ ? a.wp + (size_t)(wave * CT * 16 + fr) * a.kppad + fc * 8 : nullptr;
    u32x4 af1[C / 32][HT], af2[PERSIST ? Hd / 32 : 1][PERSIST ? CT : 1], afp[PROJ && PERSIST ? C / 32 : 1][PROJ && PERSIST ? CT : 1];
    f32x4 bias1[HT], bias2[CT], biasp[PROJ ? CT : 1];
    auto load_w1 = [&]() {
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks)
#pragma unroll
            for (int i = 0; i < HT; ++i) af1[ks][i] = *reinterpret_cast<const u32x4*>(w1row + (size_t)i * 16 * a.k1pad + ks * 32);
    };
    u32x4 xr[NX];
    auto load_x = [&](int tile) {
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int i = t + q * NT;
            const int r = i / CPR, c = i - r * CPR;
            const int m = tile * MLP_BM + r;
            xr[q] = u32x4{0u, 0u, 0u, 0u};
            if (m < a.M) xr[q] = PROJ ? *reinterpret_cast<const u32x4*>(a.a + (size_t)m * a.lda + c * 8)
                                      : *reinterpret_cast<const u32x4*>(a.x + (size_t)m * a.ldx + c * 8);
        }
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    load_x(tile);
    load_w1();
#pragma unroll
    for (int i = 0; i < HT; ++i) bias1[i] = *reinterpret_cast<const f32x4*>(a.b1 + wave * HT * 16 + i * 16 + fc * 4);
#pragma unroll
    for (int i = 0; i < CT; ++i) bias2[i] = *reinterpret_cast<const f32x4*>(a.b2 + wave * CT * 16 + i * 16 + fc * 4);
    if constexpr (PERSIST) {
#pragma unroll
        for (int ks = 0; ks < Hd / 32; ++ks)
#pragma unroll
            for (int i = 0; i < CT; ++i) af2[ks][i] = *reinterpret_cast<const u32x4*>(w2row + (size_t)i * 16 * a.k2pad + ks * 32);
    }
    if constexpr (PROJ) {
#pragma unroll
        for (int i = 0; i < CT; ++i) biasp[i] = *reinterpret_cast<const f32x4*>(a.bp + wave * CT * 16 + i * 16 + fc * 4);
        if constexpr (PERSIST) {
#pragma unroll
            for (int ks = 0; ks < C / 32; ++ks)
#pragma unroll
                for (int i = 0; i < CT; ++i) afp[ks][i] = *reinterpret_cast<const u32x4*>(wprow + (size_t)i * 16 * a.kppad + ks * 32);
        }
    }

    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * MLP_BM;
        // ---- token tile (also the residual) registers -> LDS; the previous tile's readers are past their last barrier ----------
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int i = t + q * NT;
            const int r = i / CPR, c = i - r * CPR;
            *reinterpret_cast<u32x4*>((PROJ ? sA : sX) + r * XP + c * 16) = xr[q];
        }
        // PROJ: this lane's residual operands (4 channels of each of its tokens), requested before the barrier
        u32x2 rres[PROJ ? CT : 1][PROJ ? 4 : 1];
        u32x4 afq[PROJ && !PERSIST ? C / 32 : 1][PROJ && !PERSIST ? CT : 1];
        if constexpr (PROJ) {
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = min(m0 + j * 16 + fr, a.M - 1);
                    rres[i][j] = *reinterpret_cast<const u32x2*>(a.x + (size_t)m * a.ldx + wave * CT * 16 + i * 16 + fc * 4);
                }
            if constexpr (!PERSIST) {
#pragma unroll
                for (int ks = 0; ks < C / 32; ++ks)
#pragma unroll
                    for (int i = 0; i < CT; ++i) afq[ks][i] = *reinterpret_cast<const u32x4*>(wprow + (size_t)i * 16 * a.kppad + ks * 32);
            }
        }
        __syncthreads();
        const int next = tile + (int)gridDim.x;
        if (PERSIST && next < ntiles) load_x(next);            // in flight during the GEMMs

        if constexpr (PROJ) {
            // ---- phase 0: X1[cout][token] = Wp A + bp + x, rounded to the 16-bit type into the x tile (this wave: its CT * 16 channels) ----
            f32x4 acc[CT][4];
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C / 32; ++ks) {
                u32x4 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4*>(sA + (j * 16 + fr) * XP + ks * 64 + fc * 16);
#pragma unroll
                for (int i = 0; i < CT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mlp_mma(acc[i][j], PERSIST ? afp[ks][i] : afq[ks][i], bf[j]);
            }
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int c = wave * CT * 16 + i * 16 + fc * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[i][j] + biasp[i];
                    u32x2 o;
                    o.x = pack_h16x2(v.x + h16lo(rres[i][j].x), v.y + h16hi(rres[i][j].x));
                    o.y = pack_h16x2(v.z + h16lo(rres[i][j].y), v.w + h16hi(rres[i][j].y));
                    *reinterpret_cast<u32x2*>(sX + (j * 16 + fr) * XP + c * 2) = o;
                }
            }
            __syncthreads();   // the x1 tile is complete
        }

        // ---- phase 1: H[hidden][token] = SiLU(W1 X + b1); this wave: hidden units [wave * HT * 16, +HT * 16) -----------------
        {
            f32x4 acc[HT][4];
#pragma unroll
            for (int i = 0; i < HT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C / 32; ++ks) {
                u32x4 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4*>(sX + (j * 16 + fr) * XP + ks * 64 + fc * 16);
#pragma unroll
                for (int i = 0; i < HT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mlp_mma(acc[i][j], af1[ks][i], bf[j]);
            }
            // bias + SiLU, bf16, into the hidden tile [token][hidden] (lane: 4 consecutive hidden units of one token)
#pragma unroll
            for (int i = 0; i < HT; ++i) {
                const int h = wave * HT * 16 + i * 16 + fc * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[i][j] + bias1[i];
                    u32x2 o;
                    o.x = pack_h16x2(silu_f(v.x), silu_f(v.y));
                    o.y = pack_h16x2(silu_f(v.z), silu_f(v.w));
                    *reinterpret_cast<u32x2*>(sH + (j * 16 + fr) * HP + h * 2) = o;
                }
            }
        }

        // ---- phase 2: Y[cout][token] = W2 H + b2 + x; this wave: output channels [wave * CT * 16, +CT * 16) ------------------
        {
            f32x4 acc[CT][4];
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x4 afl[PERSIST ? 1 : Hd / 32][PERSIST ? 1 : CT];
            if constexpr (!PERSIST) {   // requested before the barrier: the round trip overlaps the other waves' SiLU epilogue
#pragma unroll
                for (int ks = 0; ks < Hd / 32; ++ks)
#pragma unroll
                    for (int i = 0; i < CT; ++i) afl[ks][i] = *reinterpret_cast<const u32x4*>(w2row + (size_t)i * 16 * a.k2pad + ks * 32);
            }
            __syncthreads();   // the hidden tile is complete
#pragma unroll
            for (int ks = 0; ks < Hd / 32; ++ks) {
                u32x4 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4*>(sH + (j * 16 + fr) * HP + ks * 64 + fc * 16);
#pragma unroll
                for (int i = 0; i < CT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mlp_mma(acc[i][j], PERSIST ? af2[ks][i] : afl[ks][i], bf[j]);
            }
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int c = wave * CT * 16 + i * 16 + fc * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + j * 16 + fr;
                    if (m < a.M) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(sX + (j * 16 + fr) * XP + c * 2);     // residual: the staged x
                        const f32x4 v = acc[i][j] + bias2[i];
                        store4(a.y + (size_t)m * a.ldy + c, v.x + h16lo(r.x), v.y + h16hi(r.x), v.z + h16lo(r.y), v.w + h16hi(r.y));
                    }
                }
            }
        }
        if (!PERSIST) break;
        __syncthreads();   // every wave is done with sX (residual) and sH before the next tile overwrites them
    }
}

extern "C" int ymk_mlp_fused_supported(int32_t dtype, int32_t C, int32_t hidden) {
    return dtype == YMK_BF16 && ((C == 64 && hidden == 128) || (C == 128 && hidden == 256) || (C == 256 && hidden == 512));
}

// x [M][ldx >= C] bf16, w1 [hidden][k1pad] (k1pad >= C), b1 fp32 [hidden], w2 [C][k2pad] (k2pad >= hidden), b2 fp32 [C],
// y [M][ldy >= C] bf16 (may not alias x: other workgroups' tokens are read after this one's are written only through x itself).
extern "C" int ymk_mlp_fused(const void* x, int32_t ldx, const void* w1, int32_t k1pad, const float* b1, const void* w2, int32_t k2pad,
                             const float* b2, void* y, int32_t ldy, int64_t M, int32_t C, int32_t hidden, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || !ymk_mlp_fused_supported(YMK_BF16, C, hidden)) return YMK_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7) || ldx % 8 || ldy % 4 || k1pad < C || k2pad < hidden || k1pad % 8 || k2pad % 8 || ldx < C || ldy < C) return YMK_E_BADARG;
    if (M <= 0) return YMK_OK;
    if (M >= (1ll << 31)) return YMK_E_BADARG;
    MlpArgs a{(const h16_t*)x, (const h16_t*)w1, b1, (const h16_t*)w2, b2, (h16_t*)y, (int)M, C, hidden, ldx, ldy, k1pad, k2pad,
              nullptr, nullptr, nullptr, 0, 0};
    const unsigned ntiles = (unsigned)((M + MLP_BM - 1) / MLP_BM);
    // persistent variants: as many workgroups as are resident at once (C = 128 holds 128 fragment registers: one per CU)
    const unsigned slots = C == 128 ? 256u : 768u;
    const unsigned npers = ntiles < slots ? ntiles : slots;
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) hipLaunchKernelGGL((mlp_fused_kernel<2, 1, 4, true>), dim3(npers), dim3(256), 0, s, a);
    else if (C == 128) hipLaunchKernelGGL((mlp_fused_kernel<4, 2, 4, true>), dim3(npers), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mlp_fused_kernel<4, 2, 8, false>), dim3(ntiles), dim3(512), 0, s, a);
    return ymk_launch_status();
}

// x1 = x + Wp a + bp;  y = x1 + W2 SiLU(W1 x1 + b1) + b2   (AAttn's projection + both skips of an ABlock + its MLP: block.py:1727-1732, 1787-1797).
// a [M][lda >= C] (the projection's input: attention output + positional stencil), wp [C][kppad >= C], bp fp32 [C]; x [M][ldx] the block's
// input (first skip); the rest as ymk_mlp_fused.  y may not alias a or x.
extern "C" int ymk_proj_mlp_fused(const void* a_in, int32_t lda, const void* wp, int32_t kppad, const float* bp, const void* x, int32_t ldx,
                                  const void* w1, int32_t k1pad, const float* b1, const void* w2, int32_t k2pad, const float* b2, void* y,
                                  int32_t ldy, int64_t M, int32_t C, int32_t hidden, void* stream) {
    if (!a_in || !wp || !bp || !x || !w1 || !b1 || !w2 || !b2 || !y || !ymk_mlp_fused_supported(YMK_BF16, C, hidden) || C < 128) return YMK_E_BADARG;
    if (((uintptr_t)a_in & 15) || ((uintptr_t)x & 7) || ((uintptr_t)y & 7) || lda % 8 || ldx % 4 || ldy % 4 || kppad < C || k1pad < C || k2pad < hidden ||
        kppad % 8 || k1pad % 8 || k2pad % 8 || lda < C || ldx < C || ldy < C)
        return YMK_E_BADARG;
    if (M <= 0) return YMK_OK;
    if (M >= (1ll << 31)) return YMK_E_BADARG;
    MlpArgs a{(const h16_t*)x, (const h16_t*)w1, b1, (const h16_t*)w2, b2, (h16_t*)y, (int)M, C, hidden, ldx, ldy, k1pad, k2pad,
              (const h16_t*)a_in, (const h16_t*)wp, bp, lda, kppad};
    const unsigned ntiles = (unsigned)((M + MLP_BM - 1) / MLP_BM);
    hipStream_t s = (hipStream_t)stream;
    if (C == 128) hipLaunchKernelGGL((mlp_fused_kernel<4, 2, 4, true, true>), dim3(ntiles < 256u ? ntiles : 256u), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mlp_fused_kernel<4, 2, 8, false, true>), dim3(ntiles), dim3(512), 0, s, a);
    return ymk_launch_status();
}
