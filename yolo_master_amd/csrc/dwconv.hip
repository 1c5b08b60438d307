// Depthwise k x k convolution, NHWC, stride 1, pad k/2 — LDS-tiled sliding-window stencil, persistent workgroups.
//
// Workgroup tile = TH x TW output pixels x 16 channels with (TH, TW) = (16, 40) or, for maps at most 20 pixels
// wide, (32, 20): the detector's maps are 160/80/40/20 pixels wide, which these shapes cover without ragged tiles.
// A workgroup owns one channel block and a run of consecutive spatial tiles of one image: the k*k filter taps of
// the block are staged (as fp32) once, and while a tile is computed the NEXT tile's (TH+k-1) x (TW+k-1) halo is
// already in flight into registers (coalesced 16-byte loads), so the HBM/L2 latency of a tile hides behind the
// previous tile's arithmetic.  The LDS pixel stride is padded to 40 (bf16) / 80 (fp32) bytes so the x-strips of a
// wave land on disjoint bank groups.  Each thread owns 4 channels x 10 consecutive output pixels of a row: per
// filter row it streams 10+k-1 LDS vectors once and keeps the k taps in registers, accumulating with packed fp32
// FMAs (v_pk_fma_f32 on two channel pairs).  The stencil has no contraction shared between channels, so MFMA does
// not apply; the halo re-read is served by L2.
//
// Reference semantics: DWConv (ultralytics/nn/modules/conv.py:185-199), AAttn.pe
// (nn/modules/block.py:1688,1731), DepthwiseSeparableConv.depthwise (nn/modules/moe/experts.py:283-292)
// dispatched per retained (image, expert) pair as in ES_MOE._sparse_forward (moe/modules.py:690-697).
#ifndef DW_ABLATE
#define DW_ABLATE 0   // tools/micro/dwv_ablate.sh builds stage-ablated copies of this file (bits: 1 no global loads, 2 no LDS staging writes, 4 no FMA loop, 8 no global stores)
#endif
#include "ymk_common.h"

#define DW_R 10   // consecutive output pixels per thread
// LDS layout / read form of the 16-bit stencil (A/B builds: tools/micro/dw_lds_ab.sh).
//   0  rounds 1-4: [row][pixel][16 channels] at a 32 / 48-byte pixel pitch, a thread's 4 channels of a pixel = one 8-byte read.  The
//      compiler pairs those reads into ds_read2_b64, which the LDS services in 16-lane groups over 32 banks: the four x-strips of a
//      group sit 320 bytes apart, strips 0 / 2 and 1 / 3 on the same banks — every read 2-way conflicted (SQ_LDS_BANK_CONFLICT = 38 % of
//      the LDS cycles of moe_dw_kernel, profiles/r04_sq_summary.txt; the pitch rule below was derived for un-paired ds_read_b64).
//   1  the same layout read with volatile 8-byte loads: the compiler may not pair them (ds_read_b64: 32-lane groups over 64 banks, the
//      access pattern the pitch rule of round 2 makes conflict-free)
//   2  PLANAR (default): [row][4-channel group][pixel], a thread's DW_R + K - 1 pixels of its channel group are CONTIGUOUS — (DW_R + K - 1) / 2
//      16-byte reads per filter row instead of DW_R + K - 1 8-byte ones, plane pitch and row pitch chosen so that every 16-lane
//      ds_read_b128 group covers the 64 banks exactly once; the filter block is stored [ky][group][kx] the same way.
//   3  PAIR-DOT (default since the dot2 form was measured): mode 2's planes with the two pixels of a 16-byte slot INTERLEAVED per channel — a
//      32-bit word = (pixel 2j, pixel 2j + 1) of one channel — and the filter rows stored as tap pairs, so that the arithmetic is
//      v_dot2_f32_bf16 (two multiply-adds per instruction straight from the 16-bit words) instead of widening shifts / masks +
//      v_pk_fma_f32: (K + 1) / 2 dot2 per output and channel and filter row, no widening (K = 9: 200 VALU instructions per thread and
//      row instead of 180 + 108).  Output pixel r even pairs its taps (0,1)(2,3)...; r odd pairs (-,0)(1,2)(3,4)... against the same words.
#ifndef DW_LDS_MODE
#define DW_LDS_MODE 3
#endif
#ifndef DW_EARLY_HALO
#define DW_EARLY_HALO 0   // 1: the first tile's halo loads are issued before the filter block is staged (A/B builds)
#endif
#ifndef DW_PAIRDOT_PF
#define DW_PAIRDOT_PF 0   // 1: the dot2 stencil walks runs of tiles with the next halo prefetched into registers (A/B builds)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T, int TW>
struct DwTile {
    static constexpr int TH = TW == 40 ? 16 : 32;
    static constexpr int CB = 16;                             // channels per workgroup
    static constexpr int CPP = CB * (int)sizeof(T) / 16;      // 16-byte chunks per staged pixel (2 bf16 / 4 fp32)
    static constexpr int NCG = CB / 4;                        // 4-channel groups
    static constexpr int STRIPS = TW / DW_R;                  // x-strips per tile row
    // LDS pixel stride (bytes).  bf16: a thread's 4 channels are one ds_read_b64, serviced 32 lanes at a time over 64 four-byte
    // banks = (4 channel groups) x (x-strips) x (rows of the half wave); with pitch 32 (40-wide tiles: 4 strips, 2 rows) / 48
    // (20-wide: 2 strips, 4 rows) and the row pitch below those 32 addresses cover the 64 banks exactly once.  The 40-byte
    // pitch used before was 2-way conflicted on every stencil size (SQ_LDS_BANK_CONFLICT in profiles/r01_sq_pass2.json).
    static constexpr int PSB = sizeof(T) == 2 ? (TW == 40 ? 32 : 48) : CB * (int)sizeof(T) + 16;
    static constexpr int conflicts(int rp) {   // largest number of distinct addresses on one bank within a 32-lane group
        int worst = 0;
        for (int bank = 0; bank < 64; ++bank) {
            int n = 0;
            for (int l = 0; l < 32; ++l) {
                const int cg = l % NCG, strip = (l / NCG) % STRIPS, y = l / (NCG * STRIPS);
                const int d = (y * rp + strip * DW_R * PSB + cg * 8) / 4;
                n += (d % 64 == bank) || ((d + 1) % 64 == bank);
            }
            worst = n > worst ? n : worst;
        }
        return worst;
    }
    // PLANAR (16-bit, DW_LDS_MODE 2): plane pitch PL and row pitch RP are the smallest pair for which every ds_read_b128 lane group
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS) covers the 64 banks exactly once with
    // lane = group + 4 x strip + 4 x STRIPS x row and strips 80 bytes apart (exhaustive search, tools/micro/dw_planar_pitch.py): in units of
    // 16 bytes the 40-wide tile reads slots 12 x group + {0, 15, RP + 5, RP + 10} / {5, 10, RP, RP + 15} (mod 16) — all sixteen for PL = 28,
    // RP = 112.  Planes hold up to TW + 14 pixels of 8 bytes (k <= 15).
    static constexpr bool PLANAR = DW_LDS_MODE >= 2 && sizeof(T) == 2;
    static constexpr bool PAIRDOT = DW_LDS_MODE == 3 && sizeof(T) == 2;
    static constexpr int PL = TW == 40 ? 448 : 288;
    static constexpr size_t wbytes(int K) {    // filter block in LDS (bf16 stays bf16); planar: [ky][group][K + 1 taps] x 4 channels
        return PAIRDOT ? (size_t)K * NCG * 2 * ((K + 1) / 2) * 16 : PLANAR ? (size_t)K * NCG * (K + 1) * 8 : (size_t)K * K * CB * (sizeof(T) == 2 ? 2 : 4);
    }
    static constexpr int resident_with(int K, int rp) { return (int)(163840 / ((size_t)(TH + K - 1) * rp + wbytes(K))); }
    static constexpr int row_pitch(int K) {    // bytes between staged rows
        if (PLANAR) return TW == 40 ? 1792 : 1152;
        const int base = (TW + K - 1) * PSB;
        if (sizeof(T) != 2) return base;
        int best = base, bc = conflicts(base);
        for (int extra = 8; extra <= 128 && bc > 1; extra += 8)
            if (conflicts(base + extra) < bc) { bc = conflicts(base + extra); best = base + extra; }
        // ... unless the padding costs a resident workgroup (16 x 40 tile, k = 9: four workgroups per CU only without it)
        return resident_with(K, best) < resident_with(K, base) ? base : best;
    }
    static constexpr int resident(int K) { return resident_with(K, row_pitch(K)); }   // workgroups per CU by LDS
    // Register prefetch of the next tile's halo (36 registers held through the arithmetic): only where the LDS leaves fewer workgroups per
    // CU than the register file does.  The 16-bit kernels take 142-158 registers at k = 7 / 9 (three waves per SIMD = three 4-wave
    // workgroups per CU), so the planar layout's three residents by LDS are not the limit and it stages without the prefetch.
    static constexpr bool prefetch(int K) { return resident(K) < (PLANAR ? 3 : 4); }
    static constexpr int ROWL = 256 / (NCG * STRIPS);         // row lanes
    static_assert(TH == ROWL, "one output row per thread");
    static constexpr size_t lds_bytes(int K) {
        return (size_t)(TH + K - 1) * row_pitch(K) + wbytes(K);
    }
};

// lds_ld4 returns channels as two register pairs: fp32 (c0,c1),(c2,c3); bf16 (c0,c2),(c1,c3).
template <typename T> struct DwPair;  // position of channel j (0..3) inside the (a.x, a.y, b.x, b.y) quadruple
template <> struct DwPair<float> { static constexpr int pos[4] = {0, 1, 2, 3}; };
template <> struct DwPair<h16_t> { static constexpr int pos[4] = {0, 2, 1, 3}; };
__device__ __forceinline__ void lds_ld4(const char* p, f32x2& a, f32x2& b, float) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    a = f32x2{t.x, t.y}; b = f32x2{t.z, t.w};
}
// bf16: the two packed words (c0,c1),(c2,c3) are widened with VECTOR shifts/masks so that the results land in
// register pairs directly: a = (c0, c2), b = (c1, c3) (channel pairing differs from fp32: see DwPair)
__device__ __forceinline__ void lds_ld4(const char* p, f32x2& a, f32x2& b, h16_t) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    h16x4_widen(t, a, b);
}

// consecutive logical workgroups (the channel blocks of one run of tiles, then the next run) execute on the same
// XCD back to back, so the 128-byte lines shared by channel blocks / halos hit in that XCD's L2
__device__ __forceinline__ unsigned xcd_remap_dw(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

struct DwEpi {  // epilogue description shared by the plain and the ES-MoE launchers
    const float* bias;  // [C] or null
    const void* res;    // residual view or null
    int ldr, act;
};

// xb/ob: base of this image's input / output view; w: [K*K][C] filter; the workgroup handles channel block cb and
// the spatial tiles [st0, st1) (row-major over the tile grid) of the image
template <typename T, int K, int TW>
__device__ __forceinline__ void dw_run(const T* __restrict__ xb, int H, int W, int C, int ldx,
                                       const T* __restrict__ w, T* __restrict__ ob, int ldy, int cb, int st0, int st1,
                                       const DwEpi& ep, const T* __restrict__ rb, char* smem) {
    using D = DwTile<T, TW>;
    constexpr int TH = D::TH;
    constexpr int P = K / 2, HT = TH + K - 1, WT = TW + K - 1;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int CPP = D::CPP;
    constexpr int NL = (HT * WT * CPP + 255) / 256;
    constexpr bool PRECISE = sizeof(T) == 4;
    const int t = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW;
    const int c0 = cb * D::CB;
    constexpr int RP = D::row_pitch(K);
    float* wsm = reinterpret_cast<float*>(smem + (size_t)HT * RP);
    T* wsb = reinterpret_cast<T*>(smem + (size_t)HT * RP);
    // With four workgroups resident per CU (bf16, k <= 9: 16 waves hide each other's staging) the next tile's halo is NOT prefetched
    // into registers: that prefetch holds 36 registers through the arithmetic and caps the kernel at three waves per SIMD.
#ifdef DW_FORCE_PREFETCH
    constexpr bool PREFETCH = true;
#else
    constexpr bool PREFETCH = D::prefetch(K);
#endif
    constexpr bool PLANAR = D::PLANAR;
    constexpr int KP = K + 1;   // planar filter rows: K taps + one (zero) pad so that a row is a whole number of 16-byte pairs

    // filter block once.  fp32: channel j of each 4-group at its register-quadruple position (see lds_ld4); bf16: as stored, widened
    // at use with the same shifts / masks as the activations
    if constexpr (PLANAR) {
        for (int i = t; i < K * KP * D::CB; i += 256) {   // [ky][group][kx <= K][4 channels]; tap K of every row is the zero pad
            const int c = i % D::CB, r = i / D::CB, kx = r % KP, ky = r / KP;
            wsb[((ky * D::NCG + (c >> 2)) * KP + kx) * 4 + (c & 3)] = (kx < K && (c0 + c) < C) ? w[(size_t)(ky * K + kx) * C + c0 + c] : (T)0;
        }
    } else
    for (int i = t; i < K * K * D::CB; i += 256) {
        const int tap = i / D::CB, c = i - tap * D::CB;
        if constexpr (sizeof(T) == 2) wsb[i] = (c0 + c) < C ? w[(size_t)tap * C + c0 + c] : (T)0;
        else wsm[tap * D::CB + (c & ~3) + DwPair<T>::pos[c & 3]] = (c0 + c) < C ? to_f32(w[(size_t)tap * C + c0 + c]) : 0.f;
    }
    // halo staging: all global loads of a thread are issued back to back into registers
    u32x4 stg[NL];
    auto gload = [&](int st) {
        const int ty0 = (st / tiles_x) * TH, tx0 = (st % tiles_x) * TW;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t + l * 256;
            const int pix = i / CPP, q = i % CPP;
            const int hy = pix / WT, hx = pix - hy * WT;
            const int iy = ty0 - P + hy, ix = tx0 - P + hx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (!(DW_ABLATE & 1) && i < HT * WT * CPP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && c0 + q * VEC < C)
                v = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * W + ix) * ldx + c0 + q * VEC);
            stg[l] = v;
        }
    };
    const int cg = t % D::NCG;
    const int strip = (t / D::NCG) % D::STRIPS;
    const int y = t / (D::NCG * D::STRIPS);   // output row of the tile owned by this thread
    const int x0 = strip * DW_R;
    const bool chan_ok = c0 + cg * 4 < C;      // partial last channel block (C need only be a multiple of 16 bytes)
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias && chan_ok) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(ep.bias + c0 + cg * 4);
        bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
    }

    if (PREFETCH && st0 < st1) gload(st0);
    for (int st = st0; st < st1; ++st) {
        if (!PREFETCH) gload(st);
        __syncthreads();  // the previous tile's LDS reads are finished
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t + l * 256;
            if (!(DW_ABLATE & 2) && i < HT * WT * CPP) {
                const int pix = i / CPP, hy = pix / WT;
                if constexpr (PLANAR) {
                    // a 16-byte piece = channel groups 2q and 2q + 1 of one pixel: 8 bytes into each of the two planes.  A 16-lane store group
                    // holds 8 pixels x 2 q: where planes 0 and 2 share their banks (2 PL = 0 mod 128: the 40-wide tile) the lanes with q = 1
                    // write group 3 first, so that the two halves of the group sit 64 (mod 128) bytes apart in both stores
                    constexpr bool SWZ = (2 * D::PL) % 128 == 0;
                    const int q = i % CPP;
                    char* d = smem + (size_t)hy * RP + (pix - hy * WT) * 8;
                    const u32x2 lo = u32x2{stg[l].x, stg[l].y}, hi = u32x2{stg[l].z, stg[l].w};
                    if constexpr (SWZ) {
                        *reinterpret_cast<u32x2*>(d + (q ? 3 : 0) * D::PL) = q ? hi : lo;
                        *reinterpret_cast<u32x2*>(d + (q ? 2 : 1) * D::PL) = q ? lo : hi;
                    } else {
                        *reinterpret_cast<u32x2*>(d + 2 * q * D::PL) = lo;
                        *reinterpret_cast<u32x2*>(d + (2 * q + 1) * D::PL) = hi;
                    }
                } else {
                    u32x2* d = reinterpret_cast<u32x2*>(smem + (size_t)hy * RP + (pix - hy * WT) * D::PSB + (i % CPP) * 16);
                    d[0] = u32x2{stg[l].x, stg[l].y};
                    d[1] = u32x2{stg[l].z, stg[l].w};
                }
            }
        }
        __syncthreads();  // halo (and, first pass, the filter block) visible
        if (PREFETCH && st + 1 < st1) gload(st + 1);  // in flight during the arithmetic below

        const int ty0 = (st / tiles_x) * TH, tx0 = (st % tiles_x) * TW;
        const int gy = ty0 + y;
        if (!chan_ok || gy >= H || tx0 + x0 >= W) continue;
        f32x2 acc[DW_R][2];
#pragma unroll
        for (int r = 0; r < DW_R; ++r) { acc[r][0] = f32x2{0.f, 0.f}; acc[r][1] = f32x2{0.f, 0.f}; }
#pragma unroll 1
        for (int ky = 0; ky < ((DW_ABLATE & 4) ? 1 : K); ++ky) {
            // ALL LDS reads of this filter row (K taps + DW_R + K - 1 pixels) are issued before the first use.  Left to itself the
            // compiler reads one or two operands at a time and waits for each (`ds_read2_b64; s_waitcnt lgkmcnt(0)`, 14 times per row at
            // k = 9): one LDS round trip per pair of operands beside ~20 packed FMAs — what kept this kernel at 0.4 of the VALU rate
            // with four waves per SIMD (found in the ISA of the round-4 one-kernel form's stencil, tools/micro/parked/esfused.hip.txt).  Same arithmetic, same order.
            typedef typename Raw4<T>::type raw_t;
            raw_t rw[K + 1], rd[DW_R + K - 1];
            if constexpr (PLANAR) {
                static_assert(((DW_R + K - 1) & 1) == 0 && (KP & 1) == 0, "pixel and tap pairs");
                const char* row = smem + (size_t)(y + ky) * RP + cg * D::PL + x0 * 8;
                const T* wrow = wsb + (ky * D::NCG + cg) * KP * 4;
#pragma unroll
                for (int kx = 0; kx < KP; kx += 2) {
                    const u32x4 q = *reinterpret_cast<const u32x4*>(wrow + kx * 4);
                    rw[kx] = raw_t{q.x, q.y}; rw[kx + 1] = raw_t{q.z, q.w};
                }
#pragma unroll
                for (int j = 0; j < DW_R + K - 1; j += 2) {
                    const u32x4 q = *reinterpret_cast<const u32x4*>(row + (size_t)j * 8);
                    rd[j] = raw_t{q.x, q.y}; rd[j + 1] = raw_t{q.z, q.w};
                }
            } else {
#if DW_LDS_MODE == 1 && !defined(YMK_HOST_EMU)
                typedef const volatile __attribute__((address_space(3))) raw_t* rd_ptr;   // (16-bit: volatile LDS loads are not paired into ds_read2_b64)
#else
                typedef const raw_t* rd_ptr;
#endif
                const char* row = smem + (size_t)(y + ky) * RP + x0 * D::PSB + cg * 4 * sizeof(T);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    if constexpr (sizeof(T) == 2) rw[kx] = *reinterpret_cast<const raw_t*>(wsb + (ky * K + kx) * D::CB + cg * 4);
                    else rw[kx] = *reinterpret_cast<const raw_t*>(wsm + (ky * K + kx) * D::CB + cg * 4);
                }
#pragma unroll
                for (int j = 0; j < DW_R + K - 1; ++j) {
                    if constexpr (sizeof(T) == 2) rd[j] = *(rd_ptr)(row + (size_t)j * D::PSB);
                    else rd[j] = *reinterpret_cast<const raw_t*>(row + (size_t)j * D::PSB);
                }
            }
#if !defined(YMK_HOST_EMU) && !defined(DW_SERIAL_READS)   // (-DDW_SERIAL_READS: the compiler's own schedule, for A/B runs: tools/micro/dw_batch_ab.sh)
            __builtin_amdgcn_sched_barrier(0);
#endif
            f32x2 wr[K][2];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                if constexpr (sizeof(T) == 2) h16x4_widen(rw[kx], wr[kx][0], wr[kx][1]);
                else { wr[kx][0] = f32x2{rw[kx].x, rw[kx].y}; wr[kx][1] = f32x2{rw[kx].z, rw[kx].w}; }
            }
#pragma unroll
            for (int j = 0; j < DW_R + K - 1; ++j) {
                f32x2 va, vb;
                if constexpr (sizeof(T) == 2) h16x4_widen(rd[j], va, vb);
                else { va = f32x2{rd[j].x, rd[j].y}; vb = f32x2{rd[j].z, rd[j].w}; }
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
            if (!(DW_ABLATE & 8) || v[0] == 12345.f) store4(ob + pix * ldy + c0 + cg * 4, v[0], v[1], v[2], v[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// DW_LDS_MODE 3: the 16-bit stencil on v_dot2 (see the header).  Same tiling, thread map (4 channels x DW_R consecutive pixels of one row),
// planes and pitches as the planar form; one tile per pass, no register prefetch (three workgroups per CU by registers and LDS).
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T, int K, int TW>
__device__ __forceinline__ void dw_run_pairdot(const T* __restrict__ xb, int H, int W, int C, int ldx, const T* __restrict__ w,
                                               T* __restrict__ ob, int ldy, int cb, int st0, int st1, const DwEpi& ep, const T* __restrict__ rb,
                                               char* smem) {
    using D = DwTile<T, TW>;
    static_assert(sizeof(T) == 2 && (DW_R & 1) == 0, "16-bit elements, an even pixel run");
    constexpr int TH = D::TH;
    constexpr int P = K / 2, HT = TH + K - 1, WT = TW + K - 1;
    static_assert((WT & 1) == 0, "whole pixel pairs per halo row");
    constexpr int NPR = WT / 2;                       // pixel pairs per halo row
    constexpr int NW = (K + 1) / 2;                   // tap pairs per filter row
    constexpr int NPAIR = (DW_R + K - 1) / 2;         // pixel pairs a thread reads per filter row
    constexpr int NPC = HT * NPR * 2;                 // staged pieces: (pixel pair, 8-channel half)
    constexpr int NL = (NPC + 255) / 256;
    constexpr int RP = D::row_pitch(K);
    constexpr bool PF = DW_PAIRDOT_PF != 0;
    const int t = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW;
    const int c0 = cb * D::CB;
    uint32_t* wsw = reinterpret_cast<uint32_t*>(smem + (size_t)HT * RP);

    const int cg = t % D::NCG;
    const int strip = (t / D::NCG) % D::STRIPS;
    const int y = t / (D::NCG * D::STRIPS);
    const int x0 = strip * DW_R;
    const bool chan_ok = c0 + cg * 4 < C;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias && chan_ok) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(ep.bias + c0 + cg * 4);
        bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
    }
    // halo staging: a piece = the same 8 channels of the two pixels of a pair (two 16-byte loads, issued back to back for all pieces)
    u32x4 sa[NL], sb[NL];
    auto gload = [&](int st) {
        const int ty0 = (st / tiles_x) * TH, tx0 = (st % tiles_x) * TW;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t + l * 256;
            const int pr = i >> 1, q = i & 1;
            const int hy = pr / NPR, hp = pr - hy * NPR;
            const int iy = ty0 - P + hy, ix = tx0 - P + 2 * hp;
            u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
            if (!(DW_ABLATE & 1) && i < NPC && (unsigned)iy < (unsigned)H && c0 + q * 8 < C) {
                const T* src = xb + ((size_t)iy * W + ix) * ldx + c0 + q * 8;
                if ((unsigned)ix < (unsigned)W) va = *reinterpret_cast<const u32x4*>(src);
                if ((unsigned)(ix + 1) < (unsigned)W) vb = *reinterpret_cast<const u32x4*>(src + ldx);
            }
            sa[l] = va; sb[l] = vb;
        }
    };
    // PF (DW_PAIRDOT_PF builds): the workgroup walks a RUN of tiles and the next tile's halo is requested right after the current one has been
    // written to LDS — in flight under the stencil and the output stores — instead of at the top of its own pass
    // EARLY: the first halo is requested BEFORE the filter block is staged (two dependent-free round trips in flight together instead of one after the other)
    constexpr bool EARLY = PF || DW_EARLY_HALO != 0;
    if (EARLY && st0 < st1) gload(st0);
    // filter block: word ((ky * 4 + group) * 2 + parity) * NW + m, channel ci = (tap lo, tap hi) with the taps (2m, 2m + 1) for even
    // output pixels and (2m - 1, 2m) for odd ones; taps outside [0, K) and channels past C are zero
    for (int i = t; i < K * D::NCG * 2 * NW * 4; i += 256) {
        const int ci = i & 3, m = (i >> 2) % NW, par = ((i >> 2) / NW) & 1, cg = ((i >> 2) / (2 * NW)) % D::NCG, ky = (i >> 2) / (2 * NW * D::NCG);
        const int c = c0 + cg * 4 + ci, tlo = 2 * m - par, thi = tlo + 1;
        const uint32_t lo = (tlo >= 0 && tlo < K && c < C) ? w[(size_t)(ky * K + tlo) * C + c] : 0u;
        const uint32_t hi = (thi >= 0 && thi < K && c < C) ? w[(size_t)(ky * K + thi) * C + c] : 0u;
        wsw[i] = lo | (hi << 16);
    }
    for (int st = st0; st < st1; ++st) {
        const int ty0 = (st / tiles_x) * TH, tx0 = (st % tiles_x) * TW;
        if (!PF && !(EARLY && st == st0)) gload(st);
        __syncthreads();  // the previous tile's LDS reads are finished (first pass: nothing pending)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t + l * 256;
            if (!(DW_ABLATE & 2) && i < NPC) {
                const int pr = i >> 1, q = i & 1;
                const int hy = pr / NPR, hp = pr - hy * NPR;
                // per channel (even pixel, odd pixel): channels 0-3 of the piece -> group 2q, channels 4-7 -> group 2q + 1
                const u32x4 a = sa[l], b = sb[l];
                const u32x4 g0 = {(a.x & 0xffffu) | (b.x << 16), (a.x >> 16) | (b.x & 0xffff0000u), (a.y & 0xffffu) | (b.y << 16), (a.y >> 16) | (b.y & 0xffff0000u)};
                const u32x4 g1 = {(a.z & 0xffffu) | (b.z << 16), (a.z >> 16) | (b.z & 0xffff0000u), (a.w & 0xffffu) | (b.w << 16), (a.w >> 16) | (b.w & 0xffff0000u)};
                char* d = smem + (size_t)hy * RP + hp * 16;
                constexpr bool SWZ = (2 * D::PL) % 128 == 0;   // planes 0 and 2 on the same banks: the q = 1 lanes store group 3 first (see dw_run)
                if constexpr (SWZ) {
                    *reinterpret_cast<u32x4*>(d + (q ? 3 : 0) * D::PL) = q ? g1 : g0;
                    *reinterpret_cast<u32x4*>(d + (q ? 2 : 1) * D::PL) = q ? g0 : g1;
                } else {
                    *reinterpret_cast<u32x4*>(d + (2 * q) * D::PL) = g0;
                    *reinterpret_cast<u32x4*>(d + (2 * q + 1) * D::PL) = g1;
                }
            }
        }
        __syncthreads();  // halo (and, first pass, the filter block) visible
        if (PF && st + 1 < st1) gload(st + 1);
        const int gy = ty0 + y;
        if (!chan_ok || gy >= H || tx0 + x0 >= W) continue;
        float acc[DW_R][4];
#pragma unroll
        for (int r = 0; r < DW_R; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < ((DW_ABLATE & 4) ? 1 : K); ++ky) {
            u32x4 we[NW], wo[NW], pp[NPAIR];
            const uint32_t* wrow = wsw + (size_t)((ky * D::NCG + cg) * 2) * NW * 4;
            const char* row = smem + (size_t)(y + ky) * RP + cg * D::PL + x0 * 8;
#pragma unroll
            for (int m = 0; m < NW; ++m) {
                we[m] = *reinterpret_cast<const u32x4*>(wrow + m * 4);
                wo[m] = *reinterpret_cast<const u32x4*>(wrow + (NW + m) * 4);
            }
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) pp[j] = *reinterpret_cast<const u32x4*>(row + (size_t)j * 16);
#ifndef YMK_HOST_EMU
            __builtin_amdgcn_sched_barrier(0);     // all reads of the row requested before the first use (see dw_run)
#endif
#pragma unroll
            for (int j = 0; j < NPAIR; ++j)        // pair-major: every word is used while it is hot, by up to 2 x NW outputs
#pragma unroll
                for (int r = 0; r < DW_R; ++r) {
                    const int m = j - (r >> 1);    // compile-time after unrolling
                    if (m >= 0 && m < NW) {
                        const u32x4& wv = (r & 1) ? wo[m] : we[m];
                        acc[r][0] = dot2_h16(pp[j].x, wv.x, acc[r][0]);
                        acc[r][1] = dot2_h16(pp[j].y, wv.y, acc[r][1]);
                        acc[r][2] = dot2_h16(pp[j].z, wv.z, acc[r][2]);
                        acc[r][3] = dot2_h16(pp[j].w, wv.w, acc[r][3]);
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < DW_R; ++r) {
            const int gx = tx0 + x0 + r;
            if (gx >= W) break;
            const size_t pix = (size_t)gy * W + gx;
            float v[4] = {acc[r][0] + bv[0], acc[r][1] + bv[1], acc[r][2] + bv[2], acc[r][3] + bv[3]};
            if (ep.act == YMK_ACT_SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
            }
            if (rb) {
                float r0, r1, r2, r3;
                load4(rb + pix * ep.ldr + c0 + cg * 4, r0, r1, r2, r3);
                v[0] = r0 + v[0]; v[1] = r1 + v[1]; v[2] = r2 + v[2]; v[3] = r3 + v[3];
            }
            if (!(DW_ABLATE & 8) || v[0] == 12345.f) store4(ob + pix * ldy + c0 + cg * 4, v[0], v[1], v[2], v[3]);
        }
    }
}

// the 16-bit stencil of this build: dot2 form where it applies (no register prefetch needed), else dw_run
template <typename T, int K, int TW>
__device__ __forceinline__ void dw_tile_run(const T* __restrict__ xb, int H, int W, int C, int ldx, const T* __restrict__ w, T* __restrict__ ob,
                                            int ldy, int cb, int st0, int st1, const DwEpi& ep, const T* __restrict__ rb, char* smem) {
    if constexpr (DwTile<T, TW>::PAIRDOT && !DwTile<T, TW>::prefetch(K)) dw_run_pairdot<T, K, TW>(xb, H, W, C, ldx, w, ob, ldy, cb, st0, st1, ep, rb, smem);
    else dw_run<T, K, TW>(xb, H, W, C, ldx, w, ob, ldy, cb, st0, st1, ep, rb, smem);
}

// Work decomposition shared by both launchers: blockIdx.x = run * ncb + cb (channel block fastest, XCD-remapped),
// run = `spt` consecutive spatial tiles.
struct DwGeom {
    int ncb, nsp, spt, nrun;
};
template <typename T>
static DwGeom dw_geom(int H, int W, int C, int tw, int64_t planes, bool prefetch) {
    const int th = tw == 40 ? 16 : 32;
    DwGeom g;
    g.ncb = (C + 15) / 16;
    g.nsp = ((H + th - 1) / th) * ((W + tw - 1) / tw);
    // with the register prefetch of the next halo (fewer than four workgroups per CU) a workgroup needs a run of tiles for it to pay: as
    // long as the launch keeps >= ~8k workgroups, at most 8.  Without it ONE tile per workgroup is faster (ES-MoE layers 3 / 4: 471 -> 435,
    // 230 -> 207 us): the dispatcher balances short workgroups better than a static run does.  YMK_DW_WG_TARGET overrides (A/B runs).
    static const int64_t target_env = [] { const char* e = getenv("YMK_DW_WG_TARGET"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
    const int64_t target = target_env > 0 ? target_env : prefetch ? 8192 : ((int64_t)1 << 40);
    int64_t spt = (int64_t)g.nsp * g.ncb * planes / target;
    g.spt = (int)(spt < 1 ? 1 : spt > 8 ? 8 : spt);
    g.nrun = (g.nsp + g.spt - 1) / g.spt;
    return g;
}
static inline int dw_tile_width(int W) { return W <= 20 ? 20 : 40; }

struct DwArgs {
    const void* x;
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int B, H, W, C, ldx, ldy, ldr, act;
    int ncb, nsp, spt;
};

template <typename T, int K, int TW>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ONE-dimensional grid over (image, run of tiles, channel block): the XCD of a workgroup is its LINEAR dispatch index modulo 8, and the
    // remap hands every XCD one contiguous range of logical ids.  (Round 4: with images on blockIdx.y the remap ran over the 16-24
    // workgroups of one image — two or three per XCD — so the four channel blocks that share a 128-byte line sat on two XCDs and the
    // fabric counter showed the 7x7 stencil's input fetched twice: 121 MB for 67 at 40^2.)
    const int per_img = (int)(gridDim.x / a.B);
    const int gid = (int)xcd_remap_dw(blockIdx.x, gridDim.x);
    const int b = gid / per_img, lid = gid - b * per_img;
    const size_t img = (size_t)b * a.H * a.W;
    DwEpi ep{a.bias, a.res, a.ldr, a.act};
    const T* rb = a.res ? reinterpret_cast<const T*>(a.res) + img * a.ldr : nullptr;
    const int cb = lid % a.ncb, st0 = (lid / a.ncb) * a.spt;
    dw_tile_run<T, K, TW>(reinterpret_cast<const T*>(a.x) + img * a.ldx, a.H, a.W, a.C, a.ldx, reinterpret_cast<const T*>(a.w),
                     reinterpret_cast<T*>(a.y) + img * a.ldy, a.ldy, cb, st0, min(a.nsp, st0 + a.spt), ep, rb, smem);
}

template <typename T, int K, int TW>
static void launch_dw_k(const DwArgs& a, int nrun, hipStream_t s) {
    const size_t shm = DwTile<T, TW>::lds_bytes(K);
    static YmkOncePerDevice attr_once;
    if (attr_once.need() && shm > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_kernel<T, K, TW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_once.done();
    }
    hipLaunchKernelGGL((dwconv_kernel<T, K, TW>), dim3((unsigned)((size_t)nrun * a.ncb * a.B)), dim3(256), shm, s, a);
}

template <typename T, int TW>
static int launch_dw_tw(DwArgs a, int k, hipStream_t s) {
    const bool prefetch = k <= 15 && DwTile<T, TW>::prefetch(k);
    const DwGeom g = dw_geom<T>(a.H, a.W, a.C, TW, a.B, prefetch || (DW_PAIRDOT_PF && sizeof(T) == 2));
    a.ncb = g.ncb; a.nsp = g.nsp; a.spt = g.spt;
    switch (k) {
        case 1: launch_dw_k<T, 1, TW>(a, g.nrun, s); break;
        case 3: launch_dw_k<T, 3, TW>(a, g.nrun, s); break;
        case 5: launch_dw_k<T, 5, TW>(a, g.nrun, s); break;
        case 7: launch_dw_k<T, 7, TW>(a, g.nrun, s); break;
        case 9: launch_dw_k<T, 9, TW>(a, g.nrun, s); break;
        case 11: launch_dw_k<T, 11, TW>(a, g.nrun, s); break;
        case 13: launch_dw_k<T, 13, TW>(a, g.nrun, s); break;
        case 15: launch_dw_k<T, 15, TW>(a, g.nrun, s); break;
        default: return YMK_E_BADARG;
    }
    return ymk_launch_status();
}

template <typename T>
static int launch_dw(DwArgs a, int k, hipStream_t s) {
    constexpr int VEC = 16 / (int)sizeof(T);
    if (a.C % VEC) return YMK_E_BADARG;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) return YMK_OK;
    if (a.B > 65535) return YMK_E_BADARG;
    return dw_tile_width(a.W) == 20 ? launch_dw_tw<T, 20>(a, k, s) : launch_dw_tw<T, 40>(a, k, s);
}

extern "C" int ymk_dwconv2d(int32_t dtype, const void* x, const void* w, const float* bias,
                            const void* residual, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t ksize, int32_t ldx, int32_t ldy, int32_t ldr, int32_t act,
                            void* stream) {
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (!x || !w || !y || ldx % vec || ldy % 4 || (residual && ldr % 4)) return YMK_E_BADARG;
    DwArgs a{x, w, bias, residual, y, B, H, W, C, ldx, ldy, ldr, act, 0, 0, 0};
    if (dtype == YMK_F32) return launch_dw<float>(a, ksize, (hipStream_t)stream);
    if (dtype == YMK_BF16) return launch_dw<h16_t>(a, ksize, (hipStream_t)stream);
    return YMK_E_BADARG;
}

// ---------------------------------------------------------------------------
// ES-MoE depthwise stage: blockIdx.y is the (image, slot) pair; dropped slots (sel < 0) exit at once.  The
// expert's stencil size selects the switch arm, the same for every workgroup of a pair.  (The image->expert CSR
// of the router is not needed here: one dependent load — sel — instead of a chain of three.)
// ---------------------------------------------------------------------------
struct MoeDwArgs {
    const void* x;
    const void* dw_w;
    const int* dw_off;
    const int* ksizes;
    const int* sel;
    void* out;
    int B, H, W, C, ldx, E, top_k;
    int ncb, nsp, spt;
};

template <typename T, int TW, int KMAX>
__global__ __launch_bounds__(256) void moe_dw_kernel(MoeDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pair = blockIdx.y;
    const int e = a.sel[pair];
    if (e < 0) return;
    const int ks = a.ksizes[e];
    const int woff = a.dw_off[e];
    const int b = pair / a.top_k;
    const size_t hw = (size_t)a.H * a.W;
    const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * hw * a.ldx;
    const T* w = reinterpret_cast<const T*>(a.dw_w) + woff;
    T* ob = reinterpret_cast<T*>(a.out) + (size_t)pair * hw * a.C;
    const DwEpi ep{nullptr, nullptr, 0, YMK_ACT_NONE};
    const int lid = (int)xcd_remap_dw(blockIdx.x, gridDim.x);
    const int cb = lid % a.ncb, st0 = (lid / a.ncb) * a.spt, st1 = min(a.nsp, st0 + a.spt);
    switch (ks) {
        case 3: if constexpr (KMAX >= 3) dw_tile_run<T, 3, TW>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, cb, st0, st1, ep, nullptr, smem); break;
        case 5: if constexpr (KMAX >= 5) dw_tile_run<T, 5, TW>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, cb, st0, st1, ep, nullptr, smem); break;
        case 7: if constexpr (KMAX >= 7) dw_tile_run<T, 7, TW>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, cb, st0, st1, ep, nullptr, smem); break;
        case 9: if constexpr (KMAX >= 9) dw_tile_run<T, 9, TW>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, cb, st0, st1, ep, nullptr, smem); break;
        case 11: if constexpr (KMAX >= 11) dw_tile_run<T, 11, TW>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, cb, st0, st1, ep, nullptr, smem); break;
        case 13: if constexpr (KMAX >= 13) dw_tile_run<T, 13, TW>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, cb, st0, st1, ep, nullptr, smem); break;
        case 15: if constexpr (KMAX >= 15) dw_tile_run<T, 15, TW>(xb, a.H, a.W, a.C, a.ldx, w, ob, a.C, cb, st0, st1, ep, nullptr, smem); break;
        default: break;
    }
}

// KMAX = largest stencil among the experts: larger switch arms are compiled out (the register allocation of the
// kernel is the maximum over its arms, and it bounds the resident workgroups per CU)
template <typename T, int TW, int KMAX>
static int launch_moe_dw_k(MoeDwArgs a, hipStream_t s) {
    const size_t shm = DwTile<T, TW>::lds_bytes(KMAX);
    static YmkOncePerDevice attr_once;
    if (shm > 64 * 1024 && attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_dw_kernel<T, TW, KMAX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_once.done();
    }
    const DwGeom g = dw_geom<T>(a.H, a.W, a.C, TW, (int64_t)a.B * a.top_k, DwTile<T, TW>::prefetch(KMAX) || (DW_PAIRDOT_PF && sizeof(T) == 2));
    a.ncb = g.ncb; a.nsp = g.nsp; a.spt = g.spt;
    hipLaunchKernelGGL((moe_dw_kernel<T, TW, KMAX>), dim3(g.nrun * g.ncb, a.B * a.top_k), dim3(256), shm, s, a);
    return ymk_launch_status();
}

template <typename T, int TW>
static int launch_moe_dw_tw(const MoeDwArgs& a, int kmax, hipStream_t s) {
    switch (kmax) {
        case 1: case 3: return launch_moe_dw_k<T, TW, 3>(a, s);
        case 5: return launch_moe_dw_k<T, TW, 5>(a, s);
        case 7: return launch_moe_dw_k<T, TW, 7>(a, s);
        case 9: return launch_moe_dw_k<T, TW, 9>(a, s);
        case 11: return launch_moe_dw_k<T, TW, 11>(a, s);
        case 13: return launch_moe_dw_k<T, TW, 13>(a, s);
        default: return launch_moe_dw_k<T, TW, 15>(a, s);
    }
}

template <typename T>
static int launch_moe_dw(const MoeDwArgs& a, int kmax, hipStream_t s) {
    if (a.C % (16 / (int)sizeof(T))) return YMK_E_BADARG;
    return dw_tile_width(a.W) == 20 ? launch_moe_dw_tw<T, 20>(a, kmax, s) : launch_moe_dw_tw<T, 40>(a, kmax, s);
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
    MoeDwArgs a{x, dw_w, dw_off, ksizes, sel, dw_out, B, H, W, C, ldx, E, top_k, 0, 0, 0};
    if (dtype == YMK_F32) return launch_moe_dw<float>(a, kmax, (hipStream_t)stream);
    if (dtype == YMK_BF16) return launch_moe_dw<h16_t>(a, kmax, (hipStream_t)stream);
    return YMK_E_BADARG;
}
