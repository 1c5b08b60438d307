// Implicit-GEMM tile engine on MFMA (gfx950), shared by conv2d and the ES-MoE
// pointwise grouped GEMM.
//
// Orientation: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k]; both operands
// are K-contiguous rows, so every MFMA fragment is one 16-byte LDS row read:
//   bf16  v_mfma_f32_16x16x32_bf16 : lane l holds row (l&15), k = (l>>4)*8 .. +8
//   f32   v_mfma_f32_16x16x4_f32 x4: lane l holds row (l&15), k = (l>>4)*4 .. +4;
//         step s uses element s of both fragments (same k-permutation on A and B,
//         so the dot product is the same set of products in a fixed order).
// Result layout (both dtypes): lane l, reg r -> cout (l>>4)*4 + r, pixel (l&15).
//
// Staging: rows of 128 bytes (one full cache line per pixel/cout row per k-step: 64 bf16 or 32 fp32
// of K), eight 16-byte chunks XOR-swizzled with (row & 7), register-staged (zero fill for the
// convolution halo / ragged edges) and double buffered: one barrier per k-step.
//
// Epilogue: cout tiles >= 64 store straight from registers (each lane owns 4 consecutive couts of a pixel);
// narrow cout tiles (16/32) go through LDS (re-using the pipeline buffer) to form full row segments.
#pragma once
#include "ymk_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);

template <>
__device__ __forceinline__ void mma16<h16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = mfma16x16x32_h16(a, b, acc);   // v_mfma_f32_16x16x32_bf16 / _f16 by the build's 16-bit format
}
template <>
__device__ __forceinline__ void mma16<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// XCD-aware workgroup order: the dispatcher places workgroup b on XCD b % 8 (each XCD has a private L2), so
// consecutive *logical* tiles are made to run on the same XCD, back to back: tiles that share a pixel tile
// (several cout tiles) or convolution halo rows then hit in that XCD's L2.  Bijective for any grid size.
// Speed only — results never depend on placement.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// DUAL (1x1 only): the K dimension is the channel concatenation of two NHWC sources (x: channels [0, C1),
// x2: channels [C1, Cin)), each with its own pixel index per row — this is how Upsample+Concat feeding a 1x1
// conv is executed without materialising the concatenated tensor.
template <typename T, int BCO, int BPX, int WCO, int WPX, int KS, bool DUAL = false>
struct IGemm {
    static constexpr int NT = 256;
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr int CH = 8;         // 16-byte chunks per 128-byte row
    static constexpr int BK = CH * VEC;  // elements of K per step: 64 bf16 / 32 fp32
    static constexpr int RPP = NT / CH;  // rows staged per pass (32)
    static constexpr int NA = (BCO + RPP - 1) / RPP;
    static constexpr int NB = BPX / RPP;
    static constexpr int TM = BCO / WCO / 16;
    static constexpr int TN = BPX / WPX / 16;
    static constexpr int STAGE = (BCO + BPX) * CH;  // u32x4 per stage
    // One LDS stage + one register stage (the next k-step's tile lives in VGPRs while this one is
    // multiplied): half the LDS of a double buffer => twice the resident workgroups per CU, which is what
    // hides HBM latency for these short-K, bandwidth-bound GEMMs.
    static constexpr int NSTAGE = KS == 1 ? 1 : 2;  // long-K (3x3) tiles: double-buffered LDS, one barrier per k-step
    // epilogue tile (fp32, XOR-swizzled 16-byte chunks, no padding) must fit in the pipeline buffer
    static constexpr int ECO = BCO > 64 ? 64 : BCO;
    static constexpr int EPX = ECO >= 64 ? 128 : BPX;
    static constexpr int ELD = ECO;  // floats per staged pixel row
    static constexpr int SMEM_U4 = NSTAGE * STAGE;
    static_assert(WCO * WPX == 4, "4 waves");
    static_assert(BCO % (WCO * 16) == 0 && BPX % (WPX * 16) == 0 && BPX % RPP == 0, "tile");
    static_assert(EPX * ELD * 4 <= SMEM_U4 * 16, "epilogue tile must fit in the pipeline buffers");
    static_assert(BPX % EPX == 0 && BCO % ECO == 0, "epilogue passes");

    struct Rows {     // per-thread description of the NB pixel rows this thread stages
        int pix[NB];  // KS>1: pixel index of (b, 0, 0) i.e. b*H*W;  KS==1: the input pixel index itself
        int iy0[NB], ix0[NB];
        int pix2[DUAL ? NB : 1];  // DUAL: pixel index in the second source
        bool ok[NB];
        // workgroup-uniform reference pixels: every row of the tile lies within a couple of images of them, so a
        // row's element offset relative to (x + ref*ldx) fits 32 bits and the k-loop needs no 64-bit index math
        int64_t ref = 0, ref2 = 0;
    };
    struct Src2 {  // second K-source (DUAL)
        const T* x2;
        int ldx2, C1;
    };

    __device__ static __forceinline__ int swz(int row, int c) { return c ^ (row & 7); }

    // acc must be initialised by the caller.  x: activation base, wt: packed weights offset to the tile's
    // first cout row, co_valid: number of valid cout rows in this tile (others read as zero).
    __device__ static __forceinline__ void run(f32x4 (&acc)[TM][TN], const T* __restrict__ x, int ldx,
                                               int H, int W, int Cin, const Rows& rows,
                                               const T* __restrict__ wt, int Kpad, int co_valid,
                                               u32x4* smem, int ablate = 0, Src2 s2 = Src2{nullptr, 0, 0}) {
#ifndef YMK_ABLATE
        ablate = 0;  // ablation switches exist only in the tools/micro harness build
#endif
        const int t = threadIdx.x;
        const int lane = t & 63, wave = t >> 6;
        const int wco = wave / WPX, wpx = wave % WPX;
        const int srow = t >> 3, cq = t & 7;
        const int nk = Kpad / BK;

        u32x4 ra[NA], rb[NB];
        int tap = 0, c = cq * VEC;  // (tap, channel) of this thread's chunk at k-step 0
        if constexpr (KS != 1)
            while (c >= Cin) { c -= Cin; ++tap; }

        // per-row constants of the k-loop: 32-bit element offsets relative to uniform base pointers and, for 3x3,
        // one validity bit per tap (image border / ragged tile), so that a k-step costs one add and one bit test
        // per staged row instead of 2-D index arithmetic, bounds compares and a 64-bit multiply
        const T* xb = x + rows.ref * ldx;
        const T* x2b = DUAL ? s2.x2 + rows.ref2 * s2.ldx2 : nullptr;
        int roff[NB], roff2[DUAL ? NB : 1], woff[NA];
        unsigned rmask[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if constexpr (KS == 1) {
                roff[i] = rows.ok[i] ? (int)(rows.pix[i] - rows.ref) * ldx : 0;
                rmask[i] = rows.ok[i] ? 1u : 0u;
                if constexpr (DUAL) roff2[i] = rows.ok[i] ? (int)(rows.pix2[i] - rows.ref2) * s2.ldx2 : 0;
            } else {
                roff[i] = rows.ok[i] ? ((int)(rows.pix[i] - rows.ref) + rows.iy0[i] * W + rows.ix0[i]) * ldx : 0;
                unsigned m = 0;
#pragma unroll
                for (int tp = 0; tp < KS * KS; ++tp) {
                    const int iy = rows.iy0[i] + tp / KS, ix = rows.ix0[i] + tp % KS;
                    m |= (rows.ok[i] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (1u << tp) : 0u;
                }
                rmask[i] = m;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = srow + i * RPP;
            woff[i] = (r < BCO && r < co_valid) ? r * Kpad + cq * VEC : -1;
        }

        auto gload = [&](int kt) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (woff[i] >= 0 && ablate != 4) v = *reinterpret_cast<const u32x4*>(wt + (woff[i] + kt * BK));
                ra[i] = v;
            }
            if constexpr (KS == 1) {
                // 1x1: the caller guarantees iy0/ix0 are in range (pad 0); rows.pix holds the input pixel index
                const bool kin = c < Cin;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (rmask[i] && kin && ablate != 3) {
                        if (DUAL && c >= s2.C1)
                            v = *reinterpret_cast<const u32x4*>(x2b + (roff2[DUAL ? i : 0] + (c - s2.C1)));
                        else
                            v = *reinterpret_cast<const u32x4*>(xb + (roff[i] + c));
                    }
                    rb[i] = v;
                }
                c += BK;
            } else {
                const bool kin = tap < KS * KS;
                const int ky = tap / KS, kx = tap - ky * KS;
                const int tapoff = (ky * W + kx) * ldx + c;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (kin && ((rmask[i] >> tap) & 1u) && ablate != 3) v = *reinterpret_cast<const u32x4*>(xb + (roff[i] + tapoff));
                    rb[i] = v;
                }
                c += BK;
                while (c >= Cin) { c -= Cin; ++tap; }
            }
        };
        auto sstore = [&](int buf) {
            u32x4* sA = smem + buf * STAGE;
            u32x4* sB = sA + BCO * CH;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int r = srow + i * RPP;
                if (r < BCO) sA[r * CH + swz(r, cq)] = ra[i];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = srow + i * RPP;
                sB[r * CH + swz(r, cq)] = rb[i];
            }
        };
        auto compute = [&](int buf) {
            const u32x4* sA = smem + buf * STAGE;
            const u32x4* sB = sA + BCO * CH;
            const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = (wco * TM + i) * 16 + fr;
                    af[i] = sA[r * CH + swz(r, kk * 4 + fc)];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = (wpx * TN + j) * 16 + fr;
                    bfr[j] = sB[r * CH + swz(r, kk * 4 + fc)];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mma16<T>(acc[i][j], af[i], bfr[j]);
            }
        };

        gload(0);
        sstore(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1) < nk;
            if (more) gload(kt + 1);  // next tile in flight (registers) while this one is multiplied
            if (NSTAGE == 2) {
                compute(kt & 1);
                if (more) sstore((kt + 1) & 1);
                __syncthreads();
            } else {
                if (ablate != 1) compute(0);
                __syncthreads();  // every wave is done reading the stage
                if (more) {
                    sstore(0);
                    __syncthreads();
                }
            }
        }
    }

    // LDS-staged epilogue.  `val(i, j, r)` returns the finished fp32 value of accumulator element
    // (cout tile i, pixel tile j, reg r) BEFORE the residual add; `emit(px_local, co_local, v4)` is called
    // with 4 consecutive couts of one pixel (tile-local coordinates) for the coalesced global write.
    // Direct epilogue: each lane stores its 4 consecutive couts per (i, j) fragment straight from registers
    // (8-byte bf16 / 16-byte fp32 stores, no LDS round trip, no barriers).
    template <typename FVal, typename FEmit>
    __device__ static __forceinline__ void epilogue_direct(FVal&& val, FEmit&& emit) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wco = wave / WPX, wpx = wave % WPX;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 v;
                v.x = val(i, j, 0); v.y = val(i, j, 1); v.z = val(i, j, 2); v.w = val(i, j, 3);
                emit((wpx * TN + j) * 16 + (lane & 15), (wco * TM + i) * 16 + (lane >> 4) * 4, v);
            }
    }

    template <typename FVal, typename FEmit>
    __device__ static __forceinline__ void epilogue(u32x4* smem, FVal&& val, FEmit&& emit) {
        if constexpr (BCO >= 64) {  // wide cout tiles: 8/16-byte register stores already fill 32..128-byte segments per pixel;
            epilogue_direct(val, emit);  // measured 3-10 % faster than the LDS round trip (tools/micro/convbench.hip)
            return;
        }
        float* ep = reinterpret_cast<float*>(smem);
        const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
        const int wco = wave / WPX, wpx = wave % WPX;
        constexpr int WCO_SPAN = TM * 16, WPX_SPAN = TN * 16;  // couts / pixels per wave
#pragma unroll
        for (int pc = 0; pc < BCO / ECO; ++pc) {
#pragma unroll
            for (int pp = 0; pp < BPX / EPX; ++pp) {
                // (the caller's last k-step barrier / previous pass barrier protects the buffer)
                const int co_lo = pc * ECO, px_lo = pp * EPX;
                const bool mine = (wco * WCO_SPAN >= co_lo) && (wco * WCO_SPAN < co_lo + ECO) &&
                                  (wpx * WPX_SPAN >= px_lo) && (wpx * WPX_SPAN < px_lo + EPX);
                if (mine) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int col = wco * WCO_SPAN + i * 16 + (lane >> 4) * 4 - co_lo;
                            const int px = wpx * WPX_SPAN + j * 16 + (lane & 15) - px_lo;
                            f32x4 v;
                            v.x = val(i, j, 0); v.y = val(i, j, 1); v.z = val(i, j, 2); v.w = val(i, j, 3);
                            *reinterpret_cast<f32x4*>(ep + px * ELD + (((col >> 2) ^ (px & (ECO / 4 - 1))) << 2)) = v;
                        }
                }
                __syncthreads();
                constexpr int CPR = ECO / 4;  // 16-byte fp32 chunks per staged pixel row
                for (int idx = t; idx < EPX * CPR; idx += NT) {
                    const int px = idx / CPR, ch = idx - px * CPR;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ep + px * ELD + ((ch ^ (px & (CPR - 1))) << 2));
                    emit(px_lo + px, co_lo + ch * 4, v);
                }
                __syncthreads();
            }
        }
    }
};
