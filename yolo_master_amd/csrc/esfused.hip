// ES-MoE expert body as ONE kernel per layer: depthwise k x k stencil -> pointwise grouped GEMM (matrix cores) -> BN / SiLU / gate /
// accumulate over the image's retained experts -> trailing BN + SiLU.  The depthwise result never leaves the CU.
//
// Reference: ES_MOE._sparse_forward (ultralytics/nn/modules/moe/modules.py:659-704) dispatching DepthwiseSeparableConv.forward
// (nn/modules/moe/experts.py:291-311) per retained (image, expert) pair, ES_MOE.norm (modules.py:581).  Same arithmetic, operand
// rounding and accumulation order as ymk_esmoe_dw + ymk_esmoe_pw (csrc/dwconv.hip, csrc/esmoe.hip): the results are bit-identical.
//
// One persistent workgroup (12 waves, one per CU) walks a contiguous run of (image, 8 x TX pixel tile) items.  An item is processed in
// channel chunks of 64: for chunk c the (8 + 8) x (TX + 8) pixel halo of x[.., c*64 .. c*64+64) is staged ONCE, by LDS-DMA, for both
// retained experts of the image; each expert's stencil runs on it and leaves a bf16 [pixels x 64] tile in LDS, which is the B operand
// of the pointwise product acc[cout][pixel] += W_e[cout][c*64 ..] * tile.  The waves are SPECIALISED: waves 0-3 do nothing but the
// stencil (VALU: v_pk_fma_f32 on LDS operands — no address arithmetic for memory, no loads); waves 4-11 (4 cout groups x 2 pixel
// halves: 64 accumulator registers for the two experts of an image) issue every transfer
// (halo and depthwise-filter slices by `buffer_load ... lds`, pointwise weights straight into A fragments), run the MFMAs of the
// PREVIOUS step and the epilogue.  VALU and matrix pipes issue from different waves of a SIMD, so the product hides under the stencil;
// step s + 1's operands travel while step s computes (two halo buffers, two filter slices, two tile buffers).
//
//   step = (item, chunk, retained expert); iteration s:   stencil waves: stencil(s) -> tile[s & 1]
//                                                        matrix waves:  DMA for s + 1 (and the next chunk's halo), W(s) -> registers,
//                                                                       MFMA(s - 1) on tile[(s - 1) & 1], epilogue after an item's last step
//   one workgroup barrier per iteration.
//
// LDS image of a halo row: (TX + 8) / 8 blocks of 8 pixels x 128 bytes, exactly what one `buffer_load_dwordx4 ... lds` writes
// (lane-linear); the 16-byte chunk s of pixel p holds source chunk s ^ (p & 7), rows are 128 bytes apart from a multiple of 1 KB, so
// that the stencil's ds_read_b64 (16 channel groups x two adjacent rows per half wave) touch every bank once.  The tile buffer is
// [pixel][128 bytes] with the same XOR on its chunks: the MFMA B fragments (ds_read_b128) are conflict-free without padding.
#include "ymk_common.h"
#include "glds.h"

#define ESF_MAXITEM 64    // items of one workgroup (the launcher sizes the grid accordingly)
#define ESF_MAXB 1024     // images per launch (prefix of the retained-expert counts: two images per thread)
#define ESF_MAXE 4
#define ESF_P 4           // halo margin: stencils up to 9 x 9
#ifndef ESF_ABLATE
#define ESF_ABLATE 0      // tools/micro/esf_ablate.sh builds stage-ablated copies: 1 no stencil arithmetic, 2 no halo transfers, 4 no MFMAs, 8 no epilogue, 16 no filter / weight loads
#endif
#define ESF_THREADS 768   // 4 stencil waves + 8 matrix waves: three waves per SIMD, 168 registers each

template <int TX, int NMT>
struct Esf {
    static constexpr int TY = 8, TP = TY * TX, R = TX / 2, NNT = TP / 16, NCB = (TX + 2 * ESF_P) / 8, HR = TY + 2 * ESF_P;
    static constexpr int C = NMT * 64, NCH = NMT;
    static constexpr int ROWP = NCB * 1024 + 128;            // bytes between halo rows: an odd number of 128-byte pixels
    static constexpr int HALO = HR * ROWP;
    static constexpr int DWB = TP * 128;                     // one [pixel][64 channel] tile
    static constexpr int WDW = 11 * 1024;                    // one depthwise-filter slice: 81 taps x 128 bytes, staged 8 taps at a time
    static constexpr int OFF_DW = 2 * HALO, OFF_WDW = OFF_DW + 2 * DWB, DYN = OFF_WDW + 2 * WDW;
    static_assert(R == 8 || R == 4, "two x-strips per tile row");
};

struct EsfArgs {
    const h16_t* x;
    const h16_t* dw_w;
    const int* dw_off;
    const int* ksizes;
    const h16_t* pw_w;
    const float* pw_b;
    const float* nscale;
    const float* nshift;
    const int* sel;
    const float* gate;
    h16_t* y;
    int B, H, W, ldx, ldy, Kpad, E, top_k, tiles_x, tiles;
};

// ---- the stencil of one step: 256 threads = 16 channel groups (4 channels) x 8 tile rows x 2 strips of R pixels ------------------
// Accumulation order per output: ky ascending, kx ascending, fp32 FMA from zero — the order of dw_run (csrc/dwconv.hip).
template <int K, int TX, int NMT, int STRIP>
__device__ __forceinline__ void esf_stencil(const char* halo, const char* wdw, char* tile, int t) {
    using G = Esf<TX, NMT>;
    constexpr int R = G::R, OFF = ESF_P - K / 2, XS = STRIP * R;   // the strip is wave-uniform (waves 0, 1 / 2, 3): every column is compile-time
    const int cg = t & 15, row = (t >> 4) & 7;
    f32x2 acc[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r][0] = f32x2{0.f, 0.f}; acc[r][1] = f32x2{0.f, 0.f}; }
    const char* wrow = wdw + cg * 8;
    const int sub = (cg & 1) * 8, ch = cg >> 1;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        // ALL LDS reads of this filter row are issued before the first use: left to itself the compiler reads one operand at a time into
        // the same register pair and waits for each (`ds_read_b64; s_waitcnt lgkmcnt(0)` 25 times per row: one LDS round trip per
        // operand beside 16 packed FMAs — the stencil then runs at a tenth of the VALU rate with one wave per SIMD)
        u32x2 rw[K], rd[R + K - 1];
        const char* hrow = halo + (row + OFF + ky) * G::ROWP + sub;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) rw[kx] = *reinterpret_cast<const u32x2*>(wrow + (ky * K + kx) * 128);
#pragma unroll
        for (int j = 0; j < R + K - 1; ++j) {
            const int col = XS + OFF + j;     // compile-time after unrolling
            rd[j] = *reinterpret_cast<const u32x2*>(hrow + (col >> 3) * 1024 + (col & 7) * 128 + ((ch ^ (col & 7)) << 4));
        }
#ifndef YMK_HOST_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        f32x2 wr[K][2];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) h16x4_widen(rw[kx], wr[kx][0], wr[kx][1]);
#pragma unroll
        for (int j = 0; j < R + K - 1; ++j) {
            f32x2 va, vb;
            h16x4_widen(rd[j], va, vb);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kx = j - r;
                if (kx >= 0 && kx < K) {
                    acc[r][0] = __builtin_elementwise_fma(va, wr[kx][0], acc[r][0]);
                    acc[r][1] = __builtin_elementwise_fma(vb, wr[kx][1], acc[r][1]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {   // (c0, c2), (c1, c3) register pairs -> packed (c0, c1), (c2, c3): the rounding of store4 in dw_run
        const int pix = row * TX + XS + r;
        u32x2 o;
        o.x = pack_h16x2(acc[r][0].x, acc[r][1].x);
        o.y = pack_h16x2(acc[r][0].y, acc[r][1].y);
        *reinterpret_cast<u32x2*>(tile + pix * 128 + ((ch ^ (pix & 7)) << 4) + sub) = o;
    }
}
template <int K, int TX, int NMT>
__device__ __forceinline__ void esf_stencil_k(const char* halo, const char* wdw, char* tile, int t) {
    if ((t >> 7) == 0) esf_stencil<K, TX, NMT, 0>(halo, wdw, tile, t); else esf_stencil<K, TX, NMT, 1>(halo, wdw, tile, t);
}

template <int TX, int NMT>
__global__ __launch_bounds__(ESF_THREADS) void esf_kernel(EsfArgs a) {
    using G = Esf<TX, NMT>;
    constexpr int C = G::C, NCH = G::NCH, NNT = G::NNT, NCB = G::NCB, HR = G::HR;
    extern __shared__ u32x4 esf_dyn[];
    char* const lds = reinterpret_cast<char*>(esf_dyn);
    __shared__ int s_item[ESF_MAXITEM][8];     // b, y0, x0, e0 | e1 << 8 | nexp << 16, gate0, gate1, -, -
    __shared__ float s_bias[ESF_MAXE * C];
    __shared__ float s_norm[2 * C];
    __shared__ int s_wsum[ESF_THREADS / 64];
    __shared__ int s_rng[4];
    int* const s_sel = reinterpret_cast<int*>(lds);                  // prologue tables live in the (not yet used) halo buffers
    int* const s_pref = s_sel + ESF_MAXB * 2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles = a.tiles;

    // ---- prologue: this workgroup's run of items ---------------------------------------------------------------------------------
    for (int i = t; i < a.B * a.top_k; i += ESF_THREADS) s_sel[i] = a.sel[i];
    for (int i = t; i < a.E * C; i += ESF_THREADS) s_bias[i] = a.pw_b[i];
    for (int i = t; i < 2 * C; i += ESF_THREADS) s_norm[i] = i < C ? a.nscale[i] : a.nshift[i - C];
    __syncthreads();
    {   // exclusive prefix of the retained experts per image (an image without one still owns one, empty, unit): two images per thread
        int c[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = 2 * t + q;
            int nv = 0;
            if (b < a.B) {
                for (int k = 0; k < a.top_k; ++k) nv += s_sel[b * a.top_k + k] >= 0;
                nv = nv > 0 ? nv : 1;
            }
            c[q] = nv;
        }
        const int own = c[0] + c[1];
        int inc = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(inc, d, 64);
            if (lane >= d) inc += v;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int ex = inc - own;
        for (int w = 0; w < wave; ++w) ex += s_wsum[w];
        if (2 * t < a.B) s_pref[2 * t] = ex;
        if (2 * t + 1 < a.B) s_pref[2 * t + 1] = ex + c[0];
        if (2 * t == a.B - 1 || 2 * t + 1 == a.B - 1) s_pref[a.B] = ex + own;
    }
    __syncthreads();
    {   // range of units (tile x retained expert) of this workgroup, cut at whole items; as item ids (image * tiles + tile)
        const int total = s_pref[a.B] * tiles;
        const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int target = ((int)blockIdx.x + q) * per;
            if (target >= total) {
                if (t == 0) s_rng[q] = a.B * tiles;
            } else {
                for (int b = t; b < a.B; b += ESF_THREADS) {
                    const int lo = s_pref[b] * tiles, hi = s_pref[b + 1] * tiles;
                    if (lo <= target && target < hi) {
                        const int c = s_pref[b + 1] - s_pref[b];
                        s_rng[q] = b * tiles + (target - lo + c - 1) / c;     // first item that starts at or after the target
                    }
                }
            }
        }
    }
    __syncthreads();
    const int it0 = s_rng[0];
    const int n_items = s_rng[1] - it0;
    if (n_items <= 0) return;
    for (int i = t; i < n_items; i += ESF_THREADS) {   // n_items <= ESF_MAXITEM (launcher)
        const int id = it0 + i, b = id / tiles, tile = id - b * tiles;
        int e[2] = {0, 0}, n = 0;
        float g[2] = {0.f, 0.f};
        for (int k = 0; k < a.top_k; ++k) {
            const int ee = s_sel[b * a.top_k + k];
            if (ee >= 0 && n < 2) { e[n] = ee; g[n] = a.gate[b * a.E + ee]; ++n; }
        }
        s_item[i][0] = b;
        s_item[i][1] = (tile / a.tiles_x) * G::TY;
        s_item[i][2] = (tile % a.tiles_x) * TX;
        s_item[i][3] = e[0] | (e[1] << 8) | (n << 16);
        s_item[i][4] = __float_as_int(g[0]);
        s_item[i][5] = __float_as_int(g[1]);
    }
    __syncthreads();   // s_sel / s_pref are dead from here on: the halo buffers may be written

    // ---- roles -------------------------------------------------------------------------------------------------------------------
    const bool matrix_wave = wave >= 4;
    const int mi = (wave - 4) & 7;                 // matrix wave 0..7 = (pixel half, cout group)
    const int mw = mi & 3, ph = mi >> 2;           // couts [mw * NMT * 16, + NMT * 16); n-tiles [ph * NNTW, + NNTW)
    constexpr int NNTW = NNT / 2;
    const int fr = lane & 15, fc = lane >> 4;
    const glds_rsrc rs_x = GLDS_MAKE_RSRC(reinterpret_cast<const char*>(a.x) - (int64_t)(ESF_P * a.W + ESF_P) * a.ldx * 2,
                                          ((int64_t)a.B * a.H * a.W - 1) * a.ldx * 2 + C * 2 + (int64_t)(ESF_P * a.W + ESF_P) * a.ldx * 2);
    const glds_rsrc rs_w = GLDS_MAKE_RSRC(a.dw_w, 0x7fffffff);
    const int pl = lane >> 3, sl = lane & 7;       // DMA lane = (pixel or tap of the block, 16-byte slot)

    // cursors over the step sequence (item, chunk, slot): all wave-uniform
    struct Cur { int item, chunk, slot, nexp; };
    auto nexp_of = [&](int item) { const int n = s_item[item][3] >> 16; return n > 0 ? n : 1; };
    auto advance = [&](Cur& c) {
        if (++c.slot == c.nexp) {
            c.slot = 0;
            if (++c.chunk == NCH) {
                c.chunk = 0;
                ++c.item;
                c.nexp = c.item < n_items ? nexp_of(c.item) : 1;
            }
        }
    };
    auto expert_of = [&](const Cur& c) { const int w = s_item[c.item][3]; return (w >> 16) == 0 ? -1 : (c.slot ? (w >> 8) & 0xff : w & 0xff); };

    // halo of (item, chunk) -> buffer hb: HR x NCB blocks of 1 KB shared by the four matrix waves
    auto dma_halo = [&](int item, int chunk, int hb) {
        if (ESF_ABLATE & 2) return;
        const int b = s_item[item][0], y0 = s_item[item][1], x0 = s_item[item][2];
        unsigned voff[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int ix = x0 - ESF_P + cb * 8 + pl;
            voff[cb] = (unsigned)(((x0 + cb * 8 + pl) * a.ldx) * 2 + ((sl ^ pl) << 4)) | ((unsigned)ix < (unsigned)a.W ? 0u : 0x80000000u);
        }
        char* dst = lds + hb * G::HALO;
        for (int i = mi; i < HR * NCB; i += 8) {
            const int hr = i / NCB, cb = i - hr * NCB;
            const int iy = y0 - ESF_P + hr;
            const unsigned rowbad = (unsigned)iy < (unsigned)a.H ? 0u : 0x80000000u;
            const unsigned soff = (unsigned)(((b * a.H + y0 + hr) * a.W * a.ldx + chunk * 64) * 2);   // from the shifted base
            unsigned v = voff[0];
#pragma unroll
            for (int q = 1; q < NCB; ++q) v = cb == q ? voff[q] : v;
            GLDS_BUFFER_LOAD_LDS(rs_x, dst + hr * G::ROWP + cb * 1024, v | rowbad, soff);
        }
    };
    // depthwise filter slice of (expert e, chunk): taps x 64 channels, [tap][128 bytes]
    auto dma_wdw = [&](int e, int chunk, int wb) {
        if (e < 0 || (ESF_ABLATE & 16)) return;
        const int k = a.ksizes[e], taps = k * k;
        const unsigned base = (unsigned)((a.dw_off[e] + chunk * 64) * 2);
        char* dst = lds + G::OFF_WDW + wb * G::WDW;
        for (int i = mi; i * 8 < taps; i += 8) {
            const int tap = i * 8 + pl;
            GLDS_BUFFER_LOAD_LDS(rs_w, dst + i * 1024, (unsigned)(tap * C * 2 + (sl << 4)) | (tap < taps ? 0u : 0x80000000u), base);
        }
    };

    // matrix waves: accumulators of the two slots; pointwise-weight fragments of ONE step (those of step s are requested right after
    // the MFMAs of step s - 1 have consumed their predecessors, and arrive while the wave waits for the stencil at the barrier)
    f32x4 acc0[NMT][NNTW], acc1[NMT][NNTW];
    u32x4 wcur[NMT][2];
    // cout of MFMA row fr of m-tile mt: pair q = mt >> 1 covers 32 couts; a lane ends up with 8 consecutive couts per pair (16-byte stores)
    auto cout_row = [&](int mt, int r) { return mw * NMT * 16 + (mt >> 1) * 32 + (r >> 2) * 8 + (mt & 1) * 4 + (r & 3); };
    auto load_w = [&](const Cur& c) {
        const int e = expert_of(c);
        if (e < 0 || (ESF_ABLATE & 16)) return;
        const h16_t* base = a.pw_w + (size_t)e * C * a.Kpad + c.chunk * 64 + fc * 8;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wcur[mt][ks] = *reinterpret_cast<const u32x4*>(base + (size_t)cout_row(mt, fr) * a.Kpad + ks * 32);
    };
    auto gemm = [&](f32x4 (&acc)[NMT][NNTW], const Cur& c, int tb) {
        const int e = expert_of(c);
        if (e < 0 || (ESF_ABLATE & 4)) return;
        if (c.chunk == 0) {   // accumulators start at the expert's folded BN bias (as moe_pw_lean_kernel does)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(s_bias + e * C + cout_row(mt, fc * 4));
#pragma unroll
                for (int nt = 0; nt < NNTW; ++nt) acc[mt][nt] = bv;
            }
        }
        const char* tile = lds + G::OFF_DW + tb * G::DWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 bfr[NNTW];
#pragma unroll
            for (int nt = 0; nt < NNTW; ++nt) {
                const int pix = (ph * NNTW + nt) * 16 + fr;
                bfr[nt] = *reinterpret_cast<const u32x4*>(tile + pix * 128 + (((ks * 4 + fc) ^ (pix & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NNTW; ++nt) acc[mt][nt] = mfma16x16x32_h16(wcur[mt][ks], bfr[nt], acc[mt][nt]);
        }
    };
    auto epilogue = [&](int item) {
        if (ESF_ABLATE & 8) return;
        const int b = s_item[item][0], y0 = s_item[item][1], x0 = s_item[item][2], w = s_item[item][3];
        const int nexp = w >> 16;
        const float g0 = __int_as_float(s_item[item][4]), g1 = __int_as_float(s_item[item][5]);
#pragma unroll
        for (int q = 0; q < NMT / 2; ++q) {
            const int co = mw * NMT * 16 + q * 32 + fc * 8;
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sc[i] = *reinterpret_cast<const f32x4*>(s_norm + co + i * 4);
                sh[i] = *reinterpret_cast<const f32x4*>(s_norm + C + co + i * 4);
            }
#pragma unroll
            for (int nt = 0; nt < NNTW; ++nt) {
                const int pix = (ph * NNTW + nt) * 16 + fr;
                const int yy = y0 + pix / TX, xx = x0 + pix % TX;
                float v[8];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // weighted expert outputs are rounded products, then added (modules.py:697-702: `expert_out * w`, index_add_):
                        // no FMA contraction, so that the sum does not depend on the order the two experts are visited in
                        float p = 0.f;
                        if (nexp >= 1) p = ymk_mul_rn(silu_f(acc0[2 * q + i][nt][r]), g0);
                        if (nexp >= 2) p = ymk_add_rn(p, ymk_mul_rn(silu_f(acc1[2 * q + i][nt][r]), g1));
                        v[i * 4 + r] = silu_f(p * sc[i][r] + sh[i][r]);     // trailing ES_MOE.norm: BatchNorm(eval) + SiLU
                    }
                if (yy < a.H && xx < a.W) store_vec_f32(a.y + ((size_t)(b * a.H + yy) * a.W + xx) * a.ldy + co, v);
            }
        }
    };

    // ---- pipeline ----------------------------------------------------------------------------------------------------------------
    // The two roles run SEPARATE loops with the same trip count and one workgroup barrier per iteration each (the accumulators of
    // the matrix waves are then not live in the stencil waves' code: one loop with a role branch inside spills ~100 registers).
    auto sync = [&]() {
        GLDS_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        GLDS_COMPILER_FENCE();
    };
    Cur cs{0, 0, 0, nexp_of(0)};      // step s
    int pidx = 0;                     // index of the (item, chunk) pair step s belongs to: its halo is in buffer pidx & 1
    auto next_step = [&]() {
        const int pi = cs.item, pc = cs.chunk;
        advance(cs);
        if (cs.item != pi || cs.chunk != pc) ++pidx;
    };
    if (!matrix_wave) {
        sync();                                               // (P) the prologue transfers of the matrix waves have landed
        for (int s = 0; cs.item < n_items; ++s) {
            const int e = expert_of(cs);
            if (e >= 0 && !(ESF_ABLATE & 1)) {
                const char* halo = lds + (pidx & 1) * G::HALO;
                const char* wdw = lds + G::OFF_WDW + (s & 1) * G::WDW;
                char* tile = lds + G::OFF_DW + (s & 1) * G::DWB;
                switch (a.ksizes[e]) {
                    case 3: esf_stencil_k<3, TX, NMT>(halo, wdw, tile, t); break;
                    case 5: esf_stencil_k<5, TX, NMT>(halo, wdw, tile, t); break;
                    case 7: esf_stencil_k<7, TX, NMT>(halo, wdw, tile, t); break;
                    case 9: esf_stencil_k<9, TX, NMT>(halo, wdw, tile, t); break;
                    default: break;
                }
            }
            next_step();
            sync();                                           // (s) tile[s & 1] complete; the operands of step s + 1 have landed
        }
        return;
    }
    dma_halo(0, 0, 0);
    dma_wdw(expert_of(cs), 0, 0);
    load_w(cs);
    __builtin_amdgcn_s_waitcnt(GLDS_WAITCNT_VM(0));
    sync();                                                   // (P)
    Cur cm = cs;                                              // step s - 1
    for (int s = 0; cs.item < n_items; ++s) {
        {
            Cur cn = cs;
            advance(cn);                                      // step s + 1: its filter slice travels during step s
            if (cn.item < n_items) dma_wdw(expert_of(cn), cn.chunk, (s + 1) & 1);
            if (cs.slot == 0) {                               // the stencil starts pair pidx now: the other halo buffer is free for pair pidx + 1
                int ni = cs.item, nc = cs.chunk + 1;
                if (nc == NCH) { nc = 0; ++ni; }
                if (ni < n_items) dma_halo(ni, nc, (pidx + 1) & 1);
            }
        }
        if (s >= 1) {                                         // MFMAs of step s - 1 (its weights arrived before the last barrier)
            if (cm.slot == 0) gemm(acc0, cm, (s - 1) & 1); else gemm(acc1, cm, (s - 1) & 1);
            if (cm.chunk == NCH - 1 && cm.slot == cm.nexp - 1) epilogue(cm.item);
            advance(cm);
        }
        load_w(cs);                                           // weights of step s: consumed in the next iteration
        next_step();
        __builtin_amdgcn_s_waitcnt(GLDS_WAITCNT_VM(0));
        sync();                                               // (s)
    }
    // drain: the last step's MFMAs and the last item's epilogue (cm is the last step; its tile is in buffer (steps - 1) & 1)
    {
        int steps = 0;
        for (int i = 0; i < n_items; ++i) steps += NCH * nexp_of(i);
        if (cm.slot == 0) gemm(acc0, cm, (steps - 1) & 1); else gemm(acc1, cm, (steps - 1) & 1);
        epilogue(cm.item);
    }
}

template <int TX, int NMT>
static int esf_launch(EsfArgs a, hipStream_t s) {
    using G = Esf<TX, NMT>;
    a.tiles_x = (a.W + TX - 1) / TX;
    a.tiles = a.tiles_x * ((a.H + G::TY - 1) / G::TY);
    static YmkOncePerDevice attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&esf_kernel<TX, NMT>), hipFuncAttributeMaxDynamicSharedMemorySize, G::DYN);
        attr_once.done();
    }
    // one workgroup per CU; more when a run would exceed the item table (at most top_k units per item, at least one)
#ifdef YMK_HOST_EMU
    static const int ncu = 3;    // a few runs of items per launch on the emulator: run boundaries inside and between images
#else
    static const int ncu = [] {
        int d = 0, n = 256;
        if (hipGetDevice(&d) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
        }
        return n;
    }();
#endif
    const int64_t items = (int64_t)a.B * a.tiles;
    int64_t nwg = ncu;
    const int64_t need = (items * 2 + ESF_MAXITEM - 2) / (ESF_MAXITEM - 1);   // units <= 2 * items; a run holds <= units / nwg + 1 items
    if (nwg < need) nwg = need;
    if (nwg > items) nwg = items;
    hipLaunchKernelGGL((esf_kernel<TX, NMT>), dim3((unsigned)nwg), dim3(ESF_THREADS), G::DYN, s, a);
    return ymk_launch_status();
}

// Which (C, map) shapes the fused kernel takes: 16-bit, Cin == Cout == Kpad in {128, 256}, stencils 3 ... 9, at most two retained experts
// per image (top_k <= 2), E <= 4.  C = 128: 8 x 16 pixel tiles; C = 256: 8 x 8 (64 accumulator registers per expert and matrix wave).
extern "C" int ymk_esmoe_fused_supported(int32_t dtype, int32_t C, int32_t Cout, int32_t H, int32_t W, int32_t kmax, int32_t E, int32_t top_k) {
    if (dtype != YMK_BF16 || C != Cout || (C != 128 && C != 256) || kmax < 3 || kmax > 9 || (kmax & 1) == 0 || E < 1 || E > ESF_MAXE ||
        top_k < 1 || top_k > 2 || H < 1 || W < 1)
        return 0;
    return 1;
}

extern "C" int ymk_esmoe_fused(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx, const void* dw_w,
                               const int32_t* dw_off, const int32_t* ksizes, int32_t kmax, int32_t Cout, int32_t Kpad, const void* pw_w,
                               const float* pw_b, const float* norm_scale, const float* norm_shift, int32_t E, int32_t top_k,
                               const int32_t* sel, const float* gate_w, void* y, int32_t ldy, void* stream) {
    if (!x || !dw_w || !dw_off || !ksizes || !pw_w || !pw_b || !norm_scale || !norm_shift || !sel || !gate_w || !y) return YMK_E_BADARG;
    if (!ymk_esmoe_fused_supported(dtype, C, Cout, H, W, kmax, E, top_k)) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    if (B > ESF_MAXB || Kpad != C || ldx % 8 || ldy % 8 || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        ((int64_t)B * H * W + (int64_t)ESF_P * W + ESF_P) * ldx * 2 >= (1ll << 31))   // 16-byte accesses, 32-bit buffer offsets
        return YMK_E_BADARG;
    EsfArgs a{static_cast<const h16_t*>(x), static_cast<const h16_t*>(dw_w), dw_off, ksizes, static_cast<const h16_t*>(pw_w), pw_b,
              norm_scale, norm_shift, sel, gate_w, static_cast<h16_t*>(y), B, H, W, ldx, ldy, Kpad, E, top_k, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    return C == 128 ? esf_launch<16, 2>(a, s) : esf_launch<8, 4>(a, s);
}
