// Next tiled implicit-GEMM core: 256 output pixels x BN couts per workgroup on 8 waves, BK = 64
// channels of one filter tap per k-step, both operands staged straight into LDS by `buffer_load_dwordx4 ... lds` (16 bytes per lane,
// lane-linear LDS image, XOR swizzle applied on the SOURCE side, border taps are out-of-range lanes of the raw buffer: zeros), two LDS stages with a
// plain barrier or three stages with a raw barrier and a counted vmcnt (one k-step of DMA in flight across every
// barrier), XCD-aware tile order.  bf16 only; Cin % 64 == 0, Cout % 64 == 0.
//
// STATUS: the DEFAULT core of ymk_conv2d / ymk_conv1x1_cat2 for every 16-bit 3x3 with Cin >= 64 and the tiled 1x1 shapes since round 2
// (validated and timed on MI355X: tests/test_gpu_next.py, profiles/r02_glds_tile_ab.txt, profiles/r03_*).  Round 3: tile shapes up to
// 128 couts x 512 pixels / 256 x 256 (one workgroup per CU, 160 / 128 KB of LDS in two stages): the loop is paced by the global->LDS
// transfer latency (one k-step in flight per workgroup, ~1.7 us per k-step at two 128 x 128 workgroups per CU = 38 GB/s per CU), so
// at a fixed LDS budget the FLOPs per in-flight byte decide: 2 x (128 x 128) -> 128 x 512 or 256 x 256 doubles them.
#include <stdio.h>
#include <type_traits>

#include "ymk_common.h"
#include "glds.h"

#define GLDS_BM 256   // pixels per tile (default); GLDS_BM_SMALL for launches that would leave CUs without a second workgroup
#define GLDS_BM_SMALL 128
#ifndef GLDS_SMALL_BELOW_DEFAULT
#define GLDS_SMALL_BELOW_DEFAULT (-1)   // no override: the measured rule of glds_launch
#endif
#ifndef GLDS_ABLATE
#define GLDS_ABLATE 0   // tools/micro/glds_ablate.sh: 1 no fragment reads / MFMA, 2 no DMA after the prologue, 4 no workgroup barrier, 8 no MFMA only, 16 no output stores
#endif


struct GldsArgs {
    const h16_t* x;
    const h16_t* w;
    const float* bias;
    const h16_t* res;
    void* y;
    int B, H, W, Ho, Wo, Cin, Cout, ks, stride, ldx, ldy, ldr, Kpad, act, out_f32;
    // virtual concatenation (1x1 only): channels [0, C1) come from x (a [B][H/2][W/2] map read through a nearest 2x upsample
    // when up1), channels [C1, Cin) from x2; x2 == nullptr: single source
    const h16_t* x2;
    int C1, ldx2, up1;
    // routed experts (stride 1): w holds E filter banks [E][Cout][Kpad]; image b convolves with bank eidx[b*K + j] for slot j and
    // writes image j*B + b of a slot-major output.  Tiles never straddle images.  eidx == nullptr: one filter bank.
    const int32_t* eidx;
    int K;
    int tap_outer;   // k-step order of a 3x3: 1 = filter tap outermost (rounds 2-3), 0 = channel chunk outermost (see `issue`)
};

// EXT: the body that also carries the GELU / sigmoid epilogues (conv_glds_ext_kernel).  128 inlined erf / exp expansions compiled into
// every instantiation cost the 256 x 256 tile 20 % (instruction cache) on launches that never use them: they are separate kernels.
template <int BN, int STAGES, int BM, bool EXT>
__device__ __forceinline__ void conv_glds_body(const GldsArgs& a) {
    constexpr int BK = 64;                    // channels of one filter tap per k-step: LDS rows of 128 bytes = 8 chunks of 16
    constexpr int ROWS = BN + BM;             // staged rows per k-step: weights first, then pixels
    constexpr int CPR = BK / 8;               // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;             // rows per wave-instruction
    constexpr int STAGE_U4 = ROWS * CPR;      // 16-byte slots per stage
    constexpr int G = (ROWS + 8 * RPI - 1) / (8 * RPI);   // buffer_load ... lds instructions per wave per k-step (6 or 5)
    constexpr bool GRAG = ROWS % (8 * RPI) != 0;          // ... the last of them only in the waves whose rows exist (BM = 208)
    constexpr int GW = BN / (8 * RPI);        // of which weight rows
    constexpr int WN = BN / 64;               // waves along couts (64 couts per wave)
    constexpr int WM = 8 / WN;                // waves along pixels
    constexpr int FRAGS = BM / 16;            // 16-pixel fragments of the tile
    constexpr int TP = (FRAGS + WM - 1) / WM; // ... per wave (4 or 2; 2 or 1 with 128-pixel tiles); UNEVEN: the last wave row holds fewer (208 pixels = 7 + 6)
    constexpr bool UNEVEN = FRAGS % WM != 0;
    static_assert(STAGES == 2 || STAGES == 3, "two or three LDS stages");
    static_assert(!(GRAG || UNEVEN) || STAGES == 2, "ragged tiles: two-stage loop only (the three-stage loop counts G transfers per wave)");
    static_assert(BM % 16 == 0 && ROWS % RPI == 0, "tile");
    extern __shared__ u32x4 smem[];           // STAGES * STAGE_U4

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int tp_mine = UNEVEN ? min(TP, FRAGS - (wave / (BN / 64)) * TP) : TP;   // pixel fragments of this wave (wave-uniform)
    const bool glast = !GRAG || ((G - 1) * 8 + wave) * RPI < ROWS;                // this wave takes part in the last transfer instruction of a k-step
    const int M = a.B * a.Ho * a.Wo;
    const int nt = a.Cout / BN;
    // bijective XCD remap: workgroup b runs on XCD b % 8, so consecutive LOGICAL tiles (same pixels, next couts; then the
    // neighbouring pixel tile that shares halo rows) are given to one XCD back to back
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n0 = (bid % nt) * BN;
    int m0 = (bid / nt) * BM, Mlim = M, b_img = -1;
    size_t out_base = 0;
    const h16_t* wbase = a.w;
    if (a.eidx) {   // (slot, image, tile-in-image)
        const int HW = a.Ho * a.Wo, tpi = (HW + BM - 1) / BM, r = bid / nt;
        const int bj = r / tpi;
        b_img = bj % a.B;
        m0 = (r % tpi) * BM;
        Mlim = HW;
        out_base = (size_t)bj * HW;
        wbase = a.w + (size_t)a.eidx[b_img * a.K + bj / a.B] * a.Cout * a.Kpad;
    }
    const int cpt = a.Cin / BK;               // k-steps per filter tap
    const int nk = a.ks * a.ks * cpt;
    const int pad = a.ks >> 1;

    // ---- staging map: lane (lr, lc) of wave-instruction q stages chunk lc ^ (row & 7) of row q*8 + lr ----------------
    const int lr = lane / CPR, lc = lane % CPR;
    auto swz = [](int r) { return r & 7; };   // chunk c of row r lives in slot c ^ (r & 7): conflict-free ds_read_b128 fragments
    // Operands are fetched through raw buffer resources: the per-lane offset of a staged row never changes (a VGPR set up once), the
    // k-step's position is a wave-uniform SGPR offset, and taps outside the image are lanes whose offset has bit 31 set — out of range
    // for the hardware's bounds check, which returns zeros without touching memory.  (The flat-pointer form built a 64-bit address per
    // load and selected a zero page for border taps: ~20 VALU instructions per load beside 4 MFMAs per load.)
    // The pixel buffer starts `pad` rows and columns BEFORE the map, so that the offset of tap (0, 0) is never negative; valid taps
    // never address below the map itself.
    unsigned wvoff[GW];
    unsigned pvoff[G - GW];    // byte offset of the tap-(0,0) input pixel (+ swizzled chunk) for this lane's pixel rows, from the shifted base
    unsigned pvoff2[G - GW];   // the same pixel in the second source of a virtual concatenation; bit 31: row past the end
    unsigned pbad[G - GW];     // bit (ky*3+kx): tap OUTSIDE the image (all ones for rows past the end)
    const int64_t shiftB = (int64_t)(pad * a.W + pad) * a.ldx * 2;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int r = (j * 8 + wave) * RPI + lr;
        const int sc = (lc ^ swz(r)) * 8;
        if (j < GW) {
            // LDS row r = MFMA row block i = (r >> 4) & 3, row fr = r & 15 of a wave's 64 couts.  It is filled with cout
            // (i >> 1) * 32 + (fr >> 2) * 8 + (i & 1) * 4 + (fr & 3): after the MFMAs a lane then holds EIGHT consecutive couts per
            // block pair (16-byte stores, 64 contiguous bytes per pixel and wave) instead of four (csrc/esmoe.hip does the same)
            const int rc = (r & ~63) + ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
            wvoff[j] = (unsigned)(((n0 + rc) * a.Kpad + sc) * 2);
        } else {
            const int p = m0 + r - BN;
            unsigned mask = 0, off = 0, off2 = 0x80000000u;
            if (p < Mlim && (!GRAG || r < ROWS)) {
                const int ox = p % a.Wo, oy = (p / a.Wo) % a.Ho, b = b_img >= 0 ? b_img : p / (a.Wo * a.Ho);
                const int iy0 = oy * a.stride - pad, ix0 = ox * a.stride - pad;
                off = (unsigned)((((b * a.H + oy * a.stride) * a.W + ox * a.stride) * a.ldx + sc) * 2);   // = tap (0, 0) from the shifted base
                if (a.x2) {   // 1x1, stride 1: (oy, ox) is the pixel itself
                    off2 = (unsigned)((((b * a.H + oy) * a.W + ox) * a.ldx2 + sc) * 2);
                    if (a.up1) off = (unsigned)((((b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1)) * a.ldx + sc) * 2);
                }
                unsigned ry = 0, rx = 0;   // rows / columns of the filter window that fall inside the image
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (k < a.ks && (unsigned)(iy0 + k) < (unsigned)a.H) ry |= 1u << k;
                    if (k < a.ks && (unsigned)(ix0 + k) < (unsigned)a.W) rx |= 1u << k;
                }
                mask = ((ry & 1u) ? rx : 0u) | ((ry & 2u) ? rx << 3 : 0u) | ((ry & 4u) ? rx << 6 : 0u);
            }
            pvoff[j - GW] = off;
            pvoff2[j - GW] = off2;
            pbad[j - GW] = ~mask;
        }
    }
    const int64_t x1rows = a.up1 ? (int64_t)a.B * (a.H >> 1) * (a.W >> 1) : (int64_t)a.B * a.H * a.W;
    const int c1 = a.x2 ? a.C1 : a.Cin;
    const glds_rsrc rs_w = GLDS_MAKE_RSRC(wbase, (int64_t)a.Cout * a.Kpad * 2);
    const glds_rsrc rs_x = GLDS_MAKE_RSRC(reinterpret_cast<const char*>(a.x) - shiftB, ((x1rows - 1) * a.ldx + c1) * 2 + shiftB);
    const glds_rsrc rs_x2 = GLDS_MAKE_RSRC(a.x2 ? a.x2 : a.x, a.x2 ? (((int64_t)a.B * a.H * a.W - 1) * a.ldx2 + (a.Cin - a.C1)) * 2 : 0);
    int it_tap_bit = 0, it_ky = 0, it_kx = 0, it_c = 0, it_k = 0;   // cursor of the NEXT k-step to issue (uniform)
    const int k1 = a.x2 ? a.C1 / BK : 0;   // k-steps served by the first source of a virtual concatenation
    // Order of the k-steps of a 3x3 with several 64-channel chunks: chunk OUTERMOST, the nine taps inside (a.tap_outer == 0, the default).
    // With the taps outermost every tap walked ALL channels of the tile's pixels before the next tap came back to them: at 256 input
    // channels and ~400 resident 256-pixel tiles that is the whole input map (52 MB at 40^2, 6.5 MB per XCD against 4 MB of L2) between two
    // uses of a line, and the fabric counters showed it — 8.6x the input bytes fetched for 256 -> 64 at 40^2, 1.6-2.2x for the stride-2
    // shapes (tools/micro/conv_shape_pmc.py, profiles/r04_conv_shape_fetch.txt).  Chunk-outermost the nine taps re-read a quarter of that.
    auto issue = [&](int stage) {
        const unsigned tapoffB = (unsigned)(((it_ky * a.W + it_kx) * a.ldx + it_c * BK) * 2);
        const unsigned woffB = a.tap_outer ? (unsigned)(it_k * BK * 2) : (unsigned)((((it_ky * a.ks + it_kx) * cpt + it_c) * BK) * 2);
        const bool second = a.x2 && it_k >= k1;
        u32x4* dst0 = smem + stage * STAGE_U4 + wave * 64;   // wave-uniform; instruction j lands at + j * 512 slots, the lane at + lane * 16 B
#pragma unroll
        for (int j = 0; j < GW; ++j) GLDS_BUFFER_LOAD_LDS(rs_w, dst0 + j * 512, wvoff[j], woffB);
        if (second) {
            const unsigned x2offB = (unsigned)((it_k - k1) * BK * 2);
#pragma unroll
            for (int j = GW; j < G; ++j)
                if (j < G - 1 || glast) GLDS_BUFFER_LOAD_LDS(rs_x2, dst0 + j * 512, pvoff2[j - GW], x2offB);
        } else {
#pragma unroll
            for (int j = GW; j < G; ++j) {
                const int m = ((int)(pbad[j - GW] << (31 - it_tap_bit))) >> 31;   // -1: this tap is outside the image
                if (j < G - 1 || glast) GLDS_BUFFER_LOAD_LDS(rs_x, dst0 + j * 512, pvoff[j - GW] | ((unsigned)m & 0x80000000u), tapoffB);
            }
        }
        ++it_k;
        if (a.tap_outer) {
            if (++it_c == cpt) {
                it_c = 0;
                ++it_kx; ++it_tap_bit;
                if (it_kx == a.ks) { it_kx = 0; ++it_ky; it_tap_bit = it_ky * 3; }
            }
        } else {
            ++it_kx; ++it_tap_bit;
            if (it_kx == a.ks) {
                it_kx = 0; ++it_ky; it_tap_bit = it_ky * 3;
                if (it_ky == a.ks) { it_ky = 0; it_tap_bit = 0; ++it_c; }
            }
        }
    };

    f32x4 acc[4][TP];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // All fragment reads of a k-step (TP <= 4: both 32-channel halves, 24-48 registers) or of one half (the 256-pixel-per-wave tiles) are
    // ISSUED before the first MFMA: left to itself the compiler interleaves reads and MFMAs in the smallest register footprint
    // (4 reads, wait, 2 MFMAs, wait, 2 MFMAs, 2 reads, wait ...: eight exposed LDS round trips per k-step beside 16 MFMAs).
    // tpw: the wave's pixel-fragment count as a compile-time constant (uneven tiles: the two wave rows run two straight-line copies of the
    // k-step — a per-fragment predicate instead broke the read / MFMA clusters: 208-pixel tiles at 90 us per tile against 75 for 256)
    auto compute = [&](int stage, auto tpw) {
        constexpr int TPW = decltype(tpw)::value;
        if (GLDS_ABLATE & 1) return;
        const u32x4* sW = smem + stage * STAGE_U4;
        const u32x4* sX = sW + BN * CPR;
        constexpr int KH = BK / 32;
        constexpr int KG = TP <= 4 ? KH : 1;        // halves read together
#pragma unroll
        for (int k0 = 0; k0 < KH; k0 += KG) {
            u32x4 af[KG][4], bfr[KG][TP];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = ((wave % WN) * 4 + i) * 16 + fr;
                    af[kk][i] = sW[r * CPR + (((k0 + kk) * 4 + fc) ^ swz(r))];
                }
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const int r = ((wave / WN) * TP + j) * 16 + fr;   // BN % 8 == 0: swz(BN + r) == swz(r)
                    bfr[kk][j] = sX[r * CPR + (((k0 + kk) * 4 + fc) ^ swz(r))];
                }
            }
#ifndef YMK_HOST_EMU
            __builtin_amdgcn_sched_barrier(0);
#ifdef GLDS_SETPRIO   // A/B builds (tools/micro/lib_variant.sh): matrix-core cluster at raised wave priority (cdna_hip_programming.md T5)
            __builtin_amdgcn_s_setprio(1);
#endif
#endif
#pragma unroll
            for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TPW; ++j)
                        if (GLDS_ABLATE & 8) acc[i][j].x += __uint_as_float(af[kk][i].x ^ bfr[kk][j].y); else
                        acc[i][j] = mfma16x16x32_h16(af[kk][i], bfr[kk][j], acc[i][j]);
#ifndef YMK_HOST_EMU
#ifdef GLDS_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            if (k0 + KG < KH) __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };

    if constexpr (STAGES == 2) {
        issue(0);
        for (int kt = 0; kt < nk; ++kt) {
            // my DMA pieces of k-step kt have landed and my fragment reads of k-step kt-1 are done: stated explicitly (a
            // workgroup barrier alone is not required to wait for outstanding global->LDS transfers of OTHER waves' making)
            __builtin_amdgcn_s_waitcnt(GLDS_WAITCNT_VM(0));
            GLDS_WAIT_LGKM0();
            if (!(GLDS_ABLATE & 4)) __builtin_amdgcn_s_barrier();   // everyone's pieces landed: stage kt&1 complete, the other one free
            GLDS_COMPILER_FENCE();
            if (kt + 1 < nk && !(GLDS_ABLATE & 2)) issue((kt + 1) & 1);
            if (UNEVEN && tp_mine != TP) compute(kt & 1, std::integral_constant<int, UNEVEN ? FRAGS - (WM - 1) * TP : TP>{});
            else compute(kt & 1, std::integral_constant<int, TP>{});
        }
    } else {
        // STAGES - 1 k-steps of DMA in flight: k-step kt + STAGES - 1 is issued right after the barrier of k-step kt, into the stage that
        // k-step kt - 1 was multiplied from
#pragma unroll
        for (int p = 0; p < STAGES - 1; ++p)
            if (p < nk) issue(p);
        int cur = 0, nxt = STAGES - 1;
        for (int kt = 0; kt < nk; ++kt) {
            // my pieces of k-step kt have landed: one younger k-step of mine may still be travelling
            const int younger = nk - 1 - kt;
            if (younger >= 1) __builtin_amdgcn_s_waitcnt(GLDS_WAITCNT_VM(G));
            else __builtin_amdgcn_s_waitcnt(GLDS_WAITCNT_VM(0));
            GLDS_WAIT_LGKM0();                  // my fragment reads of k-step kt-1 are done (WAR on stage nxt)
            if (!(GLDS_ABLATE & 4)) __builtin_amdgcn_s_barrier();       // everyone's pieces landed, everyone's reads done
            GLDS_COMPILER_FENCE();
            if (kt + STAGES - 1 < nk && !(GLDS_ABLATE & 2)) issue(nxt);
            compute(cur, std::integral_constant<int, TP>{});
            cur = cur == STAGES - 1 ? 0 : cur + 1;
            nxt = nxt == STAGES - 1 ? 0 : nxt + 1;
        }
    }

    // ---- epilogue: bias, activation, residual; 8 consecutive couts per lane and block pair (see the staging map).  All operand
    //      loads are issued first (rows past the end clamped to the last pixel) so that they overlap: one wait instead of one per
    //      fragment -------------------------------------------------------------------------------------------------------------
    auto cout_of = [&](int i) { return n0 + (wave % WN) * 64 + (i >> 1) * 32 + fc * 8 + (i & 1) * 4; };
    f32x4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + cout_of(i)) : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool RES_PREFETCH = TP <= 4;   // 16 x TP more registers next to 16 x TP accumulators: only for the small tiles
    u32x2 rr[4][RES_PREFETCH ? TP : 1];
    if (a.res && RES_PREFETCH) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < (RES_PREFETCH ? TP : 1); ++j) {
                const int p = min(m0 + ((wave / WN) * TP + j) * 16 + fr, Mlim - 1);
                rr[i][j] = load_raw4(a.res + (out_base + p) * a.ldr + cout_of(i));
            }
    }
    // 16-byte stores need the row pitch and the base to allow them (a channel-slice view of a concatenation buffer may not)
    const bool wide = !a.out_f32 && (a.ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int co = cout_of(2 * h);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            if (UNEVEN && j >= tp_mine) continue;
            const int p = m0 + ((wave / WN) * TP + j) * 16 + fr;
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = 2 * h + q;
                float v0 = acc[i][j].x + bv[i].x, v1 = acc[i][j].y + bv[i].y, v2 = acc[i][j].z + bv[i].z, v3 = acc[i][j].w + bv[i].w;
                if (a.act == YMK_ACT_SILU) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
                else if constexpr (EXT) {
                    if (a.act == YMK_ACT_GELU) { v0 = gelu_exact(v0); v1 = gelu_exact(v1); v2 = gelu_exact(v2); v3 = gelu_exact(v3); }
                    else if (a.act == YMK_ACT_SIGMOID) { v0 = sigmoid_exact(v0); v1 = sigmoid_exact(v1); v2 = sigmoid_exact(v2); v3 = sigmoid_exact(v3); }
                }
                if (a.res) {
                    float r0, r1, r2, r3;
                    if constexpr (RES_PREFETCH) unpack_raw4(rr[i][j], r0, r1, r2, r3);
                    else unpack_raw4(load_raw4(a.res + (out_base + min(p, Mlim - 1)) * a.ldr + cout_of(i)), r0, r1, r2, r3);
                    v0 += r0; v1 += r1; v2 += r2; v3 += r3;
                }
                v[q * 4 + 0] = v0; v[q * 4 + 1] = v1; v[q * 4 + 2] = v2; v[q * 4 + 3] = v3;
            }
            if (p >= Mlim) continue;
            if ((GLDS_ABLATE & 16) && v[0] != 12345.678f) continue;   // ablation: the epilogue's arithmetic without its stores
            if (a.out_f32) {
                float* yo = static_cast<float*>(a.y) + (out_base + p) * a.ldy + co;
                store4(yo, v[0], v[1], v[2], v[3]);
                store4(yo + 4, v[4], v[5], v[6], v[7]);
            } else {
                h16_t* yo = static_cast<h16_t*>(a.y) + (out_base + p) * a.ldy + co;
                if (wide) {
                    store_vec_f32(yo, v);
                } else {
                    store4(yo, v[0], v[1], v[2], v[3]);
                    store4(yo + 4, v[4], v[5], v[6], v[7]);
                }
            }
        }
    }
}

template <int BN, int STAGES, int BM = GLDS_BM>
__global__ __launch_bounds__(512) void conv_glds_kernel(GldsArgs a) { conv_glds_body<BN, STAGES, BM, false>(a); }
template <int BN, int STAGES, int BM>
__global__ __launch_bounds__(512) void conv_glds_ext_kernel(GldsArgs a) { conv_glds_body<BN, STAGES, BM, true>(a); }

template <int BN, int STAGES, int BM, bool EXT = false>
static int glds_launch_bm(const GldsArgs& a, hipStream_t s) {
    if constexpr (!EXT) {
        if (a.act == YMK_ACT_GELU || a.act == YMK_ACT_SIGMOID) {   // the two-stage tiles of the default rule carry these epilogues
            if constexpr (STAGES == 2 && BM <= 256) return glds_launch_bm<BN, STAGES, BM, true>(a, s);
            else return YMK_E_BADARG;
        }
    }
    const int M = a.B * a.Ho * a.Wo;
    const int grid = (a.eidx ? a.K * a.B * ((a.Ho * a.Wo + BM - 1) / BM) : (M + BM - 1) / BM) * (a.Cout / BN);
    const size_t lds = (size_t)STAGES * (BN + BM) * 8 * 16;
    static YmkOncePerDevice once;
    if (once.need()) {
        const void* fn;
        if constexpr (EXT) fn = (const void*)conv_glds_ext_kernel<BN, STAGES, BM>;
        else fn = (const void*)conv_glds_kernel<BN, STAGES, BM>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return YMK_E_LAUNCH;
        once.done();
    }
    if constexpr (EXT) hipLaunchKernelGGL((conv_glds_ext_kernel<BN, STAGES, BM>), dim3(grid), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv_glds_kernel<BN, STAGES, BM>), dim3(grid), dim3(512), lds, s, a);
    return ymk_launch_status();
}

// Tile height: 128 pixels when the 256-pixel launch would have fewer than `glds_small_below` workgroups (default: always), else 256.
// A 128-cout x 256-pixel tile of two 49 KB stages leaves room for ONE workgroup per CU: its DMA wait, barrier and MFMA phases run one
// after the other (stage ablation, tools/micro/glds_ablate.sh: the MFMAs are 6 % of 128->128 s2 at 160^2, the DMA 36 %); with
// 128-pixel tiles two (128 couts) or three (64 couts) workgroups share a CU and overlap them.  Same arithmetic, bit-identical output.
// pixel-tile height of the calling thread's last launch (ymk_conv2d_last_variant reports it in bits 16+: profilers' kernel names)
static thread_local int glds_last_tile = 0, glds_last_bn = 0;
int ymk_glds_last_tile() { return glds_last_tile; }
int ymk_glds_last_bn() { return glds_last_bn; }
static int glds_big_min_tiles() {   // YMK_GLDS_BIG_MIN_TILES=<n>: the 256 x 256 tile needs n workgroups (A/B runs); default 192
    static const int v = [] { const char* e = getenv("YMK_GLDS_BIG_MIN_TILES"); return e ? atoi(e) : 192; }();
    return v;
}

static int glds_tile208() {   // YMK_GLDS_TILE208=1: the 256 x 208 tile where it fills the CU rounds better (A/B runs; off: see glds_launch_any)
    static const int v = [] { const char* e = getenv("YMK_GLDS_TILE208"); return e ? atoi(e) : 0; }();
    return v;
}
static int glds_cu_count() {
    static const int v = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return v;
}

static int glds_tap_outer() {   // YMK_GLDS_TAP_OUTER=1: the rounds 2-3 k-step order (A/B runs)
    static const int v = [] { const char* e = getenv("YMK_GLDS_TAP_OUTER"); return e ? atoi(e) : 0; }();
    return v;
}
static int glds_small_below() {   // YMK_GLDS_SMALL_BELOW=<n>: 128-pixel tiles iff the 256-pixel launch has fewer than n workgroups (A/B runs); unset: the rule below
    static const int v = [] { const char* e = getenv("YMK_GLDS_SMALL_BELOW"); return e ? atoi(e) : GLDS_SMALL_BELOW_DEFAULT; }();
    return v;
}
// YMK_GLDS_TILE=<BN>x<BM> forces a tile shape for A/B runs (128x128, 128x256, 128x512, 256x256, 64x128, 64x256); unset: the rule below
static void glds_forced_tile(int& bn, int& bm) {
    static const int v = [] {
        const char* e = getenv("YMK_GLDS_TILE");
        int n = 0, m = 0;
        return (e && sscanf(e, "%dx%d", &n, &m) == 2) ? n * 4096 + m : 0;
    }();
    bn = v / 4096; bm = v % 4096;
}

// flags of the three entry points below (`two_stage` argument): bit 0 = two-stage loop; bits 8-11 = forced cout-tile width / 64 and
// bits 12-21 = forced pixel-tile height (0: the rule / YMK_GLDS_TILE) — tests and A/B tools pick a tile shape per call with them
template <int STAGES>
static int glds_launch_any(const GldsArgs& a, hipStream_t s, int flags) {
    const int64_t M = a.eidx ? (int64_t)a.Ho * a.Wo : (int64_t)a.B * a.Ho * a.Wo;
    const int64_t groups = a.eidx ? (int64_t)a.K * a.B : 1;
    auto tiles = [&](int bn, int bm) { return groups * ((M + bm - 1) / bm) * (a.Cout / bn); };
    int bn = ((flags >> 8) & 15) * 64, bm = (flags >> 12) & 1023;
    if (!bn) glds_forced_tile(bn, bm);
    if (bn && !((bn == 64 || bn == 128 || (bn == 256 && STAGES == 2)) && (bm == 128 || bm == 256 || (bm == 512 && STAGES == 2 && bn == 128) || (bm == 208 && bn == 256)) &&
                !(bn == 256 && bm != 256 && bm != 208)))
        return YMK_E_BADARG;
    if (!bn || a.Cout % bn) {
        const bool c128 = a.Cout % 128 == 0;
        bn = c128 ? 128 : 64;
        const int64_t grid256 = tiles(bn, 256);
        // measured rule of round 2 (profiles/r02_glds_tile_ab.txt): 128-cout tiles always on 128 pixels; 64-cout tiles when the launch is
        // small (under one 256-pixel workgroup per CU) or large (>= 1024), not in between (256 -> 64 at 40^2: 69 vs 78 us)
        // round 4 (tools/micro/glds64_tile_ab.py): a LARGE 64-cout launch stays on 256 pixels when it is a 3x3 — with 128-pixel tiles a wave holds
        // ONE pixel fragment beside the 64 couts and re-reads every weight fragment for it (10 ds_read_b128 per 8 MFMAs: bound by the LDS read
        // rate); 128 -> 64 at 80^2 94 -> 83 us.  The 1x1 (one k-step per 64 channels, nothing to amortise) keeps 128: 26 vs 29 us.
        bm = (c128 || grid256 < 256 || (grid256 >= 1024 && a.ks == 1)) ? 128 : 256;
        if (glds_small_below() >= 0) bm = grid256 < glds_small_below() ? 128 : 256;
        // round 3 (profiles/r03_glds_tile_ab.txt): a 256 x 256 tile (one 8-wave workgroup per CU, 128 KB of LDS, 64 MFMAs per wave and
        // k-step) wins 5-10 % where there are at least ~3/4 of a round of them and the reduction is long (256 -> 256 s2 at 80^2: 149 -> 138
        // us, 256 -> 512 s2 at 40^2: 80 -> 72, 768 -> 256 1x1 at 40^2: 66 -> 62), loses on short reductions / few tiles (256 -> 768 1x1 at
        // 20^2: 26 -> 33).  The 128 x 512 tile never won (reachable through the flags only).
        if (STAGES == 2 && a.Cout % 256 == 0 && a.Kpad >= 384 && tiles(256, 256) >= glds_big_min_tiles()) {
            bn = 256; bm = 256;
            // round 6 (measured, OFF): ONE 256 x 256 workgroup fits a CU, and the detector's maps give 400 tiles (64 images x 40^2:
            // 1.56 rounds of 256 CUs) or 200 (20^2 x 512 couts: 0.78) — 78 % of the CU-rounds paid for.  A 208-pixel tile (13 fragments: seven in the
            // first wave row, six in the second, each row its own straight-line copy of the k-step) makes that 493 / 248 tiles: two rounds / one at 96 %,
            // each tile 13/16 of the MFMAs.  Measured: 256 -> 256 s2 at 40^2 151.8 -> 143.0 us, 256 -> 512 s2 at 20^2 79.0 -> 75.1, the six 1x1 shapes
            // +-2 %; value 12 940 -> 12 804, value_sync 11 252 -> 11 113 (inside the noise).  A tile's time does not follow its MFMA count: the k-step is
            // paced by the transfer round trip + barrier of the ONE workgroup a CU holds, so a shorter tile is not a faster tile and the "half-empty
            // second round" costs what it costs.  Reachable through the flags / YMK_GLDS_TILE208=1; profiles/r06_negative_results.txt item 3.
            const int64_t cus = glds_cu_count();
            const int64_t t256 = tiles(256, 256), t208 = tiles(256, 208);
            const int64_t c256 = ((t256 + cus - 1) / cus) * 256 * 100, c208 = ((t208 + cus - 1) / cus) * 208 * 104;   // (4 % for the tile's poorer staging / MFMA ratio)
            if (glds_tile208() && c208 < c256) bm = 208;
        }
    }
    glds_last_tile = bm;
    glds_last_bn = bn;
    if (STAGES == 2) {
        if (bn == 256 && bm == 256) return glds_launch_bm<256, 2, 256>(a, s);
        if (bn == 256 && bm == 208) return glds_launch_bm<256, 2, 208>(a, s);
        if (bn == 128 && bm == 512) return glds_launch_bm<128, 2, 512>(a, s);
    }
    if (bn == 128) return bm == 256 ? glds_launch_bm<128, STAGES, 256>(a, s) : glds_launch_bm<128, STAGES, 128>(a, s);
    return bm == 256 ? glds_launch_bm<64, STAGES, 256>(a, s) : glds_launch_bm<64, STAGES, 128>(a, s);
}

// Same arguments and result as ymk_conv2d; returns YMK_E_BADARG for shapes outside this kernel's domain (the caller then
// takes the validated path).  two_stage != 0 selects the plain-barrier loop.
extern "C" int ymk_conv2d_glds(const ymk_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                               void* y, int32_t two_stage, void* stream) {
    if (!d || !x || !w || !bias || !y) return YMK_E_BADARG;
    if (d->dtype != YMK_BF16 || (d->out_dtype != YMK_BF16 && d->out_dtype != YMK_F32)) return YMK_E_BADARG;
    if ((d->ksize != 1 && d->ksize != 3) || (d->stride != 1 && d->stride != 2)) return YMK_E_BADARG;
    if (d->Cin < 64 || d->Cin % 64 || d->Cout % 64 || d->ldx % 8 || d->ldy % 4 || (residual && d->ldr % 4)) return YMK_E_BADARG;
    if (d->Kpad != d->ksize * d->ksize * d->Cin) return YMK_E_BADARG;
    if (d->act != YMK_ACT_NONE && d->act != YMK_ACT_SILU && d->act != YMK_ACT_GELU && d->act != YMK_ACT_SIGMOID) return YMK_E_BADARG;
    if (residual && d->act != YMK_ACT_NONE && d->act != YMK_ACT_SILU) return YMK_E_BADARG;
    const int pad = d->ksize / 2;
    GldsArgs a;
    a.x = static_cast<const h16_t*>(x); a.w = static_cast<const h16_t*>(w); a.bias = bias;
    a.res = static_cast<const h16_t*>(residual); a.y = y;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.ks = d->ksize; a.stride = d->stride;
    a.Ho = (d->H + 2 * pad - d->ksize) / d->stride + 1;
    a.Wo = (d->W + 2 * pad - d->ksize) / d->stride + 1;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr; a.Kpad = d->Kpad; a.act = d->act;
    a.out_f32 = d->out_dtype == YMK_F32;
    a.x2 = nullptr; a.C1 = 0; a.ldx2 = 0; a.up1 = 0; a.eidx = nullptr; a.K = 0; a.tap_outer = glds_tap_outer();
    const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
    if (M <= 0) return YMK_OK;
    // 31-bit BYTE offsets of the staged operands (bit 31 of a lane's buffer offset marks a tap outside the image), 32-bit element offsets of the output
    if (M >= (1ll << 31) || ((int64_t)d->B * d->H * d->W + 2 * d->W + 4) * d->ldx >= (1ll << 30) || M * d->ldy >= (1ll << 31) ||
        (int64_t)d->Cout * d->Kpad >= (1ll << 30))
        return YMK_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    return (two_stage & 1) ? glds_launch_any<2>(a, s, two_stage) : glds_launch_any<3>(a, s, two_stage);
}

// Same arguments and result as ymk_conv1x1_cat2 (ymk.h) + two_stage; C1 and Cin - C1 multiples of 64.
extern "C" int ymk_conv1x1_cat2_glds(const ymk_conv_desc* d, const void* x1, int32_t C1, int32_t ldx1, int32_t upsample1, const void* x2,
                                     int32_t ldx2, const void* w, const float* bias, void* y, int32_t two_stage, void* stream) {
    if (!d || !x1 || !x2 || !w || !bias || !y || d->ksize != 1 || d->stride != 1) return YMK_E_BADARG;
    if (d->dtype != YMK_BF16 || d->out_dtype != YMK_BF16) return YMK_E_BADARG;
    if (C1 < 64 || C1 % 64 || d->Cin - C1 < 64 || (d->Cin - C1) % 64 || d->Cout % 64 || ldx1 % 8 || ldx2 % 8 || d->ldy % 4) return YMK_E_BADARG;
    if (d->Kpad != d->Cin || (d->act != YMK_ACT_NONE && d->act != YMK_ACT_SILU)) return YMK_E_BADARG;
    if (upsample1 && ((d->H & 1) || (d->W & 1))) return YMK_E_BADARG;
    GldsArgs a;
    a.x = static_cast<const h16_t*>(x1); a.w = static_cast<const h16_t*>(w); a.bias = bias; a.res = nullptr; a.y = y;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->H; a.Wo = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.ks = 1; a.stride = 1;
    a.ldx = ldx1; a.ldy = d->ldy; a.ldr = 0; a.Kpad = d->Kpad; a.act = d->act; a.out_f32 = 0;
    a.x2 = static_cast<const h16_t*>(x2); a.C1 = C1; a.ldx2 = ldx2; a.up1 = upsample1 ? 1 : 0; a.eidx = nullptr; a.K = 0; a.tap_outer = 1;   // (1x1: one tap, it_k walks the concatenated channels)
    const int64_t M = (int64_t)a.B * a.H * a.W;
    if (M <= 0) return YMK_OK;
    if (M >= (1ll << 31) || (M + 2) * (ldx1 > ldx2 ? ldx1 : ldx2) >= (1ll << 30) || M * d->ldy >= (1ll << 31) || (int64_t)d->Cout * d->Kpad >= (1ll << 30))
        return YMK_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    return (two_stage & 1) ? glds_launch_any<2>(a, s, two_stage) : glds_launch_any<3>(a, s, two_stage);
}

// Routed-expert convolution (FusedExpertGroup moe/gated.py:1058-1076, SharedInvertedExpertGroup moe/experts.py:235-269) with
// true sparse dispatch: only the filter banks the router picked run.  d describes ONE expert's convolution (Cout, Kpad of one
// bank; stride 1; no activation); w [E][Cout][Kpad]; idx int32 [B][K]; y slot-major [K*B][Ho][Wo] rows of d->ldy.
extern "C" int ymk_expert_conv_glds(const ymk_conv_desc* d, const void* x, const void* w, const int32_t* idx, int32_t K, int32_t E,
                                    void* y, int32_t two_stage, void* stream) {
    if (!d || !x || !w || !idx || !y || K < 1 || E < 1) return YMK_E_BADARG;
    if (d->dtype != YMK_BF16 || d->out_dtype != YMK_BF16 || d->stride != 1 || (d->ksize != 1 && d->ksize != 3)) return YMK_E_BADARG;
    if (d->Cin < 64 || d->Cin % 64 || d->Cout % 64 || d->ldx % 8 || d->ldy % 4 || d->Kpad != d->ksize * d->ksize * d->Cin ||
        d->act != YMK_ACT_NONE)
        return YMK_E_BADARG;
    GldsArgs a;
    a.x = static_cast<const h16_t*>(x); a.w = static_cast<const h16_t*>(w); a.bias = nullptr; a.res = nullptr; a.y = y;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->H; a.Wo = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.ks = d->ksize; a.stride = 1;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = 0; a.Kpad = d->Kpad; a.act = YMK_ACT_NONE; a.out_f32 = 0;
    a.x2 = nullptr; a.C1 = 0; a.ldx2 = 0; a.up1 = 0; a.eidx = idx; a.K = K; a.tap_outer = glds_tap_outer();
    const int64_t HW = (int64_t)d->H * d->W;
    if (d->B <= 0 || HW <= 0) return YMK_OK;
    if (((int64_t)d->B * HW + 2 * d->W + 4) * d->ldx >= (1ll << 30) || (int64_t)K * d->B * HW >= (1ll << 31) || (int64_t)d->Cout * d->Kpad >= (1ll << 30))
        return YMK_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    return (two_stage & 1) ? glds_launch_any<2>(a, s, two_stage) : glds_launch_any<3>(a, s, two_stage);
}
