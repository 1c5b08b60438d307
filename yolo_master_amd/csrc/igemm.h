// Implicit-GEMM tile engine on MFMA (gfx950), shared by conv2d and the ES-MoE
// pointwise grouped GEMM.
//
// Orientation: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k]; both operands
// are K-contiguous rows, so every MFMA fragment is one 16-byte LDS row read:
//   bf16  v_mfma_f32_16x16x32_bf16 : lane l holds row (l&15), k = (l>>4)*8 .. +8
//   f32   v_mfma_f32_16x16x4_f32 x4: lane l holds row (l&15), k = (l>>4)*4 .. +4;
//         step s uses element s of both fragments (same k-permutation on A and B,
//         so the dot product is the same set of products in a fixed order).
// Result layout (both dtypes): lane l, reg r -> cout (l>>4)*4 + r, pixel (l&15):
// four consecutive output channels of one pixel per lane => vector NHWC stores.
//
// LDS tile: rows of 64 bytes (4 chunks of 16 B), double buffered; chunk index is
// XOR-swizzled with ((row>>3)&1)*3 so that the four 16-lane service groups of
// ds_read_b128 hit 16 distinct 16-B slots of the 256-B bank row
// (MI355X_MICROARCH.md "LDS": ds_read_b128 groups {0-3,12-15,20-27} ...).
#pragma once
#include "ymk_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);

template <>
__device__ __forceinline__ void mma16<bf16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

__device__ __forceinline__ int swz_chunk(int row, int c) { return c ^ (((row >> 3) & 1) * 3); }

template <typename T, int BCO, int BPX, int WCO, int WPX, int KS>
struct IGemm {
    static constexpr int NT = 256;
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr int BK = 4 * VEC;  // elements per k-step (64-byte rows)
    static constexpr int NA = (BCO + 63) / 64;
    static constexpr int NB = BPX / 64;
    static constexpr int TM = BCO / WCO / 16;
    static constexpr int TN = BPX / WPX / 16;
    static constexpr int STAGE = (BCO + BPX) * 4;  // u32x4 per stage
    static_assert(WCO * WPX == 4, "4 waves");
    static_assert(BCO % (WCO * 16) == 0 && BPX % (WPX * 16) == 0 && BPX % 64 == 0, "tile");

    // per-thread description of the NB pixel rows this thread stages
    struct Rows {
        int pix[NB];  // pixel index of (b, 0, 0) i.e. b*H*W
        int iy0[NB], ix0[NB];
        bool ok[NB];
    };

    // acc must be zero-initialised (or carry a running sum) by the caller.
    // x: activation base, wt: packed weights already offset to the tile's first cout row,
    // co_valid: number of valid cout rows in this tile (rows >= co_valid read as zero).
    __device__ static __forceinline__ void run(f32x4 (&acc)[TM][TN], const T* __restrict__ x, int ldx,
                                               int H, int W, int Cin, const Rows& rows,
                                               const T* __restrict__ wt, int Kpad, int co_valid,
                                               u32x4* smem) {
        const int t = threadIdx.x;
        const int lane = t & 63, wave = t >> 6;
        const int wco = wave / WPX, wpx = wave % WPX;
        const int srow = t >> 2, cq = t & 3;
        const int nk = Kpad / BK;
        const int K = KS * KS * Cin;

        u32x4 ra[NA], rb[NB];
        int tap = 0, c = cq * VEC;  // (tap, channel) of this thread's chunk at k-step 0
        while (c >= Cin) { c -= Cin; ++tap; }

        auto gload = [&](int kt) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int r = srow + i * 64;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (r < BCO && r < co_valid)
                    v = *reinterpret_cast<const u32x4*>(wt + (size_t)r * Kpad + kt * BK + cq * VEC);
                ra[i] = v;
            }
            const bool kin = (kt * BK + cq * VEC) < K;
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                u32x4 v = {0u, 0u, 0u, 0u};
                const int iy = rows.iy0[i] + ky, ix = rows.ix0[i] + kx;
                if (rows.ok[i] && kin && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const int64_t p = (int64_t)rows.pix[i] + (int64_t)iy * W + ix;
                    v = *reinterpret_cast<const u32x4*>(x + p * ldx + c);
                }
                rb[i] = v;
            }
            c += BK;
            while (c >= Cin) { c -= Cin; ++tap; }
        };
        auto sstore = [&](int buf) {
            u32x4* sA = smem + buf * STAGE;
            u32x4* sB = sA + BCO * 4;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int r = srow + i * 64;
                if (r < BCO) sA[r * 4 + swz_chunk(r, cq)] = ra[i];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = srow + i * 64;
                sB[r * 4 + swz_chunk(r, cq)] = rb[i];
            }
        };
        auto compute = [&](int buf) {
            const u32x4* sA = smem + buf * STAGE;
            const u32x4* sB = sA + BCO * 4;
            const int fr = lane & 15, fc = lane >> 4;
            u32x4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = (wco * TM + i) * 16 + fr;
                af[i] = sA[r * 4 + swz_chunk(r, fc)];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = (wpx * TN + j) * 16 + fr;
                bfr[j] = sB[r * 4 + swz_chunk(r, fc)];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma16<T>(acc[i][j], af[i], bfr[j]);
        };

        gload(0);
        sstore(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1) < nk;
            if (more) gload(kt + 1);
            compute(kt & 1);
            if (more) sstore((kt + 1) & 1);
            __syncthreads();
        }
    }
};
