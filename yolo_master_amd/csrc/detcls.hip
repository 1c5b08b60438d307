// Detect class branch as ONE kernel (bf16): DWConv3x3 -> Conv1x1 -> DWConv3x3 -> Conv1x1 -> Conv2d 1x1 (+bias) -> fp32 class logits and / or
// (round 4) their sigmoid straight into the class rows of y plus every anchor's best class: the decode's class half in the epilogue —
// WITHIN TOLERANCE of detect_decode_kernel's scores, not bit-identical to them: the epilogue's sigmoid is v_exp_f32 + v_rcp_f32 (<= 2 ulp
// of the libm expression; dc_sigmoid below), bounded on the GPU by tests/test_gpu_model.py::test_fused_decode_scores_within_ulps_and_same_nms
// (ultralytics/nn/modules/head.py:111-118: `cv3[i] = Sequential(Sequential(DWConv(x, x, 3), Conv(x, c3, 1)),
// Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)), Conv2d(c3, nc, 1))`, every Conv / DWConv = convolution + folded BN + SiLU), for
// c3 = 128 and x = 128 or 256 input channels (the three pyramid levels of YOLO-Master-S / -N ...: c3 = max(ch[0], min(nc, 100))).
//
// As five kernels a level writes and re-reads four [B, H, W, 128] maps; here a persistent 8-wave workgroup owns an 8 x 16 pixel tile
// and keeps them in LDS (121 KB): HBM sees the level's input once (with a 2-pixel halo, served by L2) and the logits once.
//   stage   x on the tile + 2, in 128-channel chunks (zero outside the map = DWConv's padding)
//   dw1     3x3 stencil on the tile + 1 (VALU, 4 channels x 11-12 pixels per thread), bias, SiLU -> LDS
//   pw1     1x1 (K = x channels, accumulated over the chunks) on the tile + 1: each wave owns 16 output channels, its weight rows
//           resident as MFMA A fragments; bias, SiLU, ZERO outside the map (the second DWConv's padding) -> LDS
//   dw2     3x3 on the tile -> LDS;  pw2  1x1 128 -> 128 -> LDS;  out  1x1 128 -> nc (+ bias), fp32 store
// Arithmetic per stage as the unfused kernels' (bf16 operands, fp32 accumulation, one rounding to bf16 per stage).
#include "ymk_common.h"

#ifndef DC_ABLATE
#define DC_ABLATE 0   // stage ablation for timing runs (tools/micro/lib_variant.sh): 1 no global staging loads, 2 no 3x3 stencil arithmetic, 4 no MFMA,
#endif                //   8 SiLU / sigmoid -> identity, 16 no y stores, 32 no best-class pass
#define DC_TH 8
#define DC_TW 16
#define DC_XR (DC_TH + 4)
#define DC_XC (DC_TW + 4)
#define DC_NX (DC_XR * DC_XC)        // 240 input pixels of a tile
#define DC_MR (DC_TH + 2)
#define DC_MC (DC_TW + 2)
#define DC_NMID (DC_MR * DC_MC)      // 180 pixels of the first pair's maps
#define DC_NP (DC_TH * DC_TW)        // 128 tile pixels
#define DC_PITCH 288                 // LDS pitch of a 128-channel pixel (bytes): 256 + 32, conflict-free b128 fragment reads
#define DC_NT 512
#define DC_A_BYTES (DC_NX * DC_PITCH)        // region A: x chunk, then pw1's map, then pw2's map
#define DC_B_BYTES (DC_NMID * DC_PITCH)      // region B: dw1's map, then dw2's map
#define DC_LDS_BYTES (DC_A_BYTES + DC_B_BYTES)
#define DC_BEST_PITCH (DC_NP + 4)            // floats per class of the decode's score tile in region B: 16 classes x 16 bytes cover the 64 banks
#define DC_BEST_MAXNC (DC_B_BYTES / (DC_BEST_PITCH * 4))   // 98 classes fit

typedef __bf16 dc_bf16x8 __attribute__((ext_vector_type(8)));
#ifndef YMK_HOST_EMU
#define DC_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define DC_SCHED_BARRIER() ((void)0)
#endif
__device__ __forceinline__ void dc_mma(f32x4& acc, const u32x4& a, const u32x4& b) {
    if (DC_ABLATE & 4) { acc.x += __uint_as_float(a.x ^ b.y); return; }
    acc = mfma16x16x32_h16(a, b, acc);
}
__device__ __forceinline__ float dc_silu(float v) { return (DC_ABLATE & 8) ? v : silu_f(v); }
__device__ __forceinline__ u32x2 dc_pack_silu(const f32x4& v) {
    u32x2 o;
    o.x = pack_h16x2(dc_silu(v.x), dc_silu(v.y));
    o.y = pack_h16x2(dc_silu(v.z), dc_silu(v.w));
    return o;
}

// sigmoid of the fused decode.  Default: v_exp_f32 + v_rcp_f32 (about 2 ulp: 2e-7 on a score, against the 1e-4 the scores are held to).
// The exact form (libm expf + IEEE division = detect_decode_kernel's bits) costs this 1.25-wave-per-SIMD epilogue 6 us per tile of 128
// anchors (measured: 216 -> 297 us at P3, more than the decode kernel it replaces); -DDC_EXACT_SIGMOID keeps it for A/B runs and tests.
__device__ __forceinline__ float dc_sigmoid(float v) {
    if (DC_ABLATE & 8) return v;
#ifdef DC_EXACT_SIGMOID
    return 1.0f / (1.0f + expf(-v));
#else
    return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
#endif
}

struct DetClsArgs {
    const h16_t* x;                         // [B][H][W][ldx], CIN channels
    const h16_t *dw1, *pw1, *dw2, *pw2, *w3;   // dw: [9][C]; pw1 [128][k1pad]; pw2 [128][k2pad]; w3 [ncpad][k3pad]
    const float *bd1, *bp1, *bd2, *bp2, *b3;
    float* y;                                // [B][H][W][ldy] fp32 logits (ncpad channels written); nullptr: not materialised
    int B, H, W, ldx, ldy, k1pad, k2pad, k3pad, ncpad, tiles_x, tiles_y;
    // fused decode (all optional): sigmoid(logits) -> rows 4.. of yo [B][4 + nc][A] at anchor a_off + oy * W + ox, and every anchor's
    // largest class score / its class (first maximum in class order) -> bconf / bcls [B][A]
    float* yo;
    float* bconf;
    int* bcls;
    int nc, A, a_off;
};

// Stencil inputs (round 5) live in LDS as PIXEL PAIRS: a 32-bit word = (pixel 2p, pixel 2p + 1) of ONE channel, a pair's 128 channels =
// 512 bytes, pairs DC_PP bytes apart, rows of the map an even number of pixels long (20 / 18), so horizontally adjacent pixels share their
// words.  The 3x3 stencil is then v_dot2 straight from the 16-bit words (csrc/dwconv.hip dw_run_pairdot has the idea): a thread computes the
// TWO outputs of a pair from the pairs (v, v + 1) and (v + 2, v + 3) of three rows — the even output pairs its taps (0, 1)(2, -), the odd
// one (-, 0)(1, 2): 48 v_dot2 and six 16-byte LDS reads per output pair and 4 channels, where the form before read nine 8-byte pixels per
// OUTPUT and spent 36 widening shifts / masks beside 36 FMAs on them.  (The instruction rounds its two-term sum once: an output may differ
// from the FMA chain in its last bf16 place.)  Outputs go to `out` in the [pixel][channel] layout the pointwise stages read as MFMA B rows.
#define DC_PP 576   // bytes between pixel pairs: 512 + 64 (the pointwise epilogue's 8-byte stores of eight pairs then spread over the banks)

// the 8-channel chunk q of pixels (2 pr, 2 pr + 1) -> words of channels 8q .. 8q + 7 of pair pr
__device__ __forceinline__ void dc_store_pair(char* base, int pr, int q, const u32x4& a, const u32x4& b) {
    const u32x4 g0 = {(a.x & 0xffffu) | (b.x << 16), (a.x >> 16) | (b.x & 0xffff0000u), (a.y & 0xffffu) | (b.y << 16), (a.y >> 16) | (b.y & 0xffff0000u)};
    const u32x4 g1 = {(a.z & 0xffffu) | (b.z << 16), (a.z >> 16) | (b.z & 0xffff0000u), (a.w & 0xffffu) | (b.w << 16), (a.w >> 16) | (b.w & 0xffff0000u)};
    *reinterpret_cast<u32x4*>(base + pr * DC_PP + q * 32) = g0;
    *reinterpret_cast<u32x4*>(base + pr * DC_PP + q * 32 + 16) = g1;
}

// 3x3 depthwise stencil + bias + SiLU for this thread's 4 channels: output pairs s, s + 16, ... < npair of a map `oprow` pairs wide, read
// from a pair-layout input tile `iprow` pairs wide whose origin is one pixel up / left of the output map's
// LEAN (the first stencil of the 256-channel levels, which runs with the first pointwise stage's twelve accumulators live): the odd output's
// (-, 0) tap words are shifted out of the even output's (0, 1) words instead of being held, and the pairs are read row by row: 28 registers less
template <int CTOT, bool LEAN>
__device__ __forceinline__ void dc_dw3(const char* in, int iprow, char* out, int npair, int oprow, const h16_t* w, int coff, const float* bias,
                                       int cg, int s) {
    u32x4 we0[3], we1[3], wo0[LEAN ? 1 : 3], wo1[3];   // per filter row: tap pairs (0,1) (2,-) for the even output, (-,0) (1,2) for the odd one; .xyzw = channels
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const u32x2 t0 = *reinterpret_cast<const u32x2*>(w + (size_t)(ky * 3 + 0) * CTOT + coff + cg * 4);
        const u32x2 t1 = *reinterpret_cast<const u32x2*>(w + (size_t)(ky * 3 + 1) * CTOT + coff + cg * 4);
        const u32x2 t2 = *reinterpret_cast<const u32x2*>(w + (size_t)(ky * 3 + 2) * CTOT + coff + cg * 4);
        const uint32_t a0[4] = {t0.x & 0xffffu, t0.x >> 16, t0.y & 0xffffu, t0.y >> 16};
        const uint32_t a1[4] = {t1.x & 0xffffu, t1.x >> 16, t1.y & 0xffffu, t1.y >> 16};
        const uint32_t a2[4] = {t2.x & 0xffffu, t2.x >> 16, t2.y & 0xffffu, t2.y >> 16};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            we0[ky][c] = a0[c] | (a1[c] << 16);
            we1[ky][c] = a2[c];
            if (!LEAN) wo0[ky][c] = a0[c] << 16;
            wo1[ky][c] = a1[c] | (a2[c] << 16);
        }
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + coff + cg * 4);
    const int ocol = 2 * oprow;
    for (int p = s; p < npair; p += DC_NT / 32) {
        const int u = p / oprow, vp = p - u * oprow;
        f32x4 ae = bv, ao = bv;
        if constexpr (LEAN) {
#pragma unroll
            for (int ky = 0; ky < ((DC_ABLATE & 2) ? 1 : 3); ++ky) {
                const char* r = in + ((u + ky) * iprow + vp) * DC_PP + cg * 16;
                const u32x4 pa = *reinterpret_cast<const u32x4*>(r), pb = *reinterpret_cast<const u32x4*>(r + DC_PP);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ae[c] = dot2_h16(pa[c], we0[ky][c], ae[c]);
                    ae[c] = dot2_h16(pb[c], we1[ky][c], ae[c]);
                    ao[c] = dot2_h16(pa[c], we0[ky][c] << 16, ao[c]);
                    ao[c] = dot2_h16(pb[c], wo1[ky][c], ao[c]);
                }
            }
        } else {
            u32x4 pa[3], pb[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const char* r = in + ((u + ky) * iprow + vp) * DC_PP + cg * 16;
                pa[ky] = *reinterpret_cast<const u32x4*>(r);
                pb[ky] = *reinterpret_cast<const u32x4*>(r + DC_PP);
            }
            DC_SCHED_BARRIER();
#pragma unroll
            for (int ky = 0; ky < ((DC_ABLATE & 2) ? 1 : 3); ++ky)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ae[c] = dot2_h16(pa[ky][c], we0[ky][c], ae[c]);
                    ae[c] = dot2_h16(pb[ky][c], we1[ky][c], ae[c]);
                    ao[c] = dot2_h16(pa[ky][c], wo0[ky][c], ao[c]);
                    ao[c] = dot2_h16(pb[ky][c], wo1[ky][c], ao[c]);
                }
        }
        char* o = out + (u * ocol + 2 * vp) * DC_PITCH + cg * 8;
        *reinterpret_cast<u32x2*>(o) = dc_pack_silu(ae);
        *reinterpret_cast<u32x2*>(o + DC_PITCH) = dc_pack_silu(ao);
    }
}

// DEC: the fused-decode variant (a.yo set): its own instantiation, so that neither variant carries the other's epilogue registers
template <int CIN, bool DEC>
__global__ __launch_bounds__(DC_NT) void detect_cls_kernel(DetClsArgs a) {
    constexpr int NCH = CIN / 128;            // 128-channel chunks of the input
    constexpr int NF1 = (DC_NMID + 15) / 16;   // 12 pixel fragments of the first pair's maps
    extern __shared__ u32x4 dc_smem[];
    char* sA = reinterpret_cast<char*>(dc_smem);
    char* sB = sA + DC_A_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int cg = t & 31, sl = t >> 5;        // stencil stages: 4-channel group x pixel slot
    const int ntile = a.B * a.tiles_y * a.tiles_x;

    // resident pointwise weights: this wave's 16 output channels
    u32x4 af1[NCH * 4], af2[4], af3[4];
#pragma unroll
    for (int s = 0; s < NCH * 4; ++s) af1[s] = *reinterpret_cast<const u32x4*>(a.pw1 + (size_t)(wave * 16 + fr) * a.k1pad + s * 32 + fc * 8);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        af2[s] = *reinterpret_cast<const u32x4*>(a.pw2 + (size_t)(wave * 16 + fr) * a.k2pad + s * 32 + fc * 8);
        af3[s] = u32x4{0u, 0u, 0u, 0u};
        if (wave * 16 + fr < a.ncpad) af3[s] = *reinterpret_cast<const u32x4*>(a.w3 + (size_t)(wave * 16 + fr) * a.k3pad + s * 32 + fc * 8);
    }
    const f32x4 bv1 = *reinterpret_cast<const f32x4*>(a.bp1 + wave * 16 + fc * 4);
    const f32x4 bv2 = *reinterpret_cast<const f32x4*>(a.bp2 + wave * 16 + fc * 4);
    f32x4 bv3 = {0.f, 0.f, 0.f, 0.f};
    if (!DEC && wave * 16 + fc * 4 < a.ncpad) bv3 = *reinterpret_cast<const f32x4*>(a.b3 + wave * 16 + fc * 4);
    const float bvt = (DEC && wave * 16 + fr < a.ncpad) ? a.b3[wave * 16 + fr] : 0.f;   // transposed product: one class per lane

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int txi = tile % a.tiles_x, r0 = tile / a.tiles_x;
        const int tyi = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const int oy0 = tyi * DC_TH, ox0 = txi * DC_TW;
        const h16_t* xb = a.x + (size_t)b * a.H * a.W * a.ldx;

        f32x4 acc1[NF1];
#pragma unroll
        for (int j = 0; j < NF1; ++j) acc1[j] = bv1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {   // (unrolled: the fragment array is indexed by ch)
            // ---- stage: x chunk on the tile + 2 -> region A -----------------------------------------------------------------------------
            {
                constexpr int NPR = DC_NX / 2;                          // 120 pixel pairs (rows of DC_XC = 20 pixels)
                constexpr int NL = (NPR * 16 + DC_NT - 1) / DC_NT;      // (pair, 8-channel chunk) pieces per thread (4)
                u32x4 va[NL], vb[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int i = t + l * DC_NT;
                    const int pr = i >> 4, q = i & 15;
                    const int u = pr / (DC_XC / 2), w = 2 * (pr - u * (DC_XC / 2));
                    const int iy = oy0 - 2 + u, ix = ox0 - 2 + w;
                    va[l] = u32x4{0u, 0u, 0u, 0u};
                    vb[l] = u32x4{0u, 0u, 0u, 0u};
                    if (!(DC_ABLATE & 1) && pr < NPR && (unsigned)iy < (unsigned)a.H) {
                        const h16_t* src = xb + ((size_t)iy * a.W + ix) * a.ldx + ch * 128 + q * 8;
                        if ((unsigned)ix < (unsigned)a.W) va[l] = *reinterpret_cast<const u32x4*>(src);
                        if ((unsigned)(ix + 1) < (unsigned)a.W) vb[l] = *reinterpret_cast<const u32x4*>(src + a.ldx);
                    }
                }
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int i = t + l * DC_NT;
                    if ((i >> 4) < NPR) dc_store_pair(sA, i >> 4, i & 15, va[l], vb[l]);
                }
            }
            __syncthreads();
            // ---- dw1: region A -> region B (tile + 1) -------------------------------------------------------------------------------------
            dc_dw3<CIN, (CIN > 128)>(sA, DC_XC / 2, sB, DC_NMID / 2, DC_MC / 2, a.dw1, ch * 128, a.bd1, cg, sl);
            __syncthreads();
            // ---- pw1: K chunk ch ----------------------------------------------------------------------------------------------------------
            // k-step outermost: the MFMAs that follow one another then belong to DIFFERENT accumulators (with the pixel fragment outermost
            // every accumulator was a chain of four dependent MFMAs, each behind its own `ds_read_b128; s_waitcnt`), and the B fragments of
            // a k-step are all requested before the first MFMA.  Per accumulator the order of the four k-steps is unchanged.
            constexpr int NH = NF1 / 2;   // six fragments at a time (twelve would need 48 registers beside the resident weight sets)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 bfr[NH];
#pragma unroll
                    for (int j = 0; j < NH; ++j) {
                        int px = (h * NH + j) * 16 + fr;
                        px = px < DC_NMID ? px : 0;
                        bfr[j] = *reinterpret_cast<const u32x4*>(sB + px * DC_PITCH + fc * 16 + s * 64);
                    }
                    DC_SCHED_BARRIER();
#pragma unroll
                    for (int j = 0; j < NH; ++j) dc_mma(acc1[h * NH + j], af1[ch * 4 + s], bfr[j]);
                }
            if (ch + 1 < NCH) __syncthreads();   // region A / B are restaged for the next chunk
        }
        // pw1 epilogue -> region A in the stencil's pair layout (the x chunk is dead: every wave is past its dw1).  Lanes fr and fr ^ 1 hold the
        // two pixels of a pair (DC_MC is even) for the same four channels: the even lane hands over channels 2, 3 and assembles the words
        // of channels 0, 1, the odd lane the other way round — one 8-byte store per lane and fragment as before.
#pragma unroll
        for (int j = 0; j < NF1; ++j) {
            const int px = j * 16 + fr;
            const int u = px / DC_MC, v = px - u * DC_MC;
            const int my = oy0 - 1 + u, mx = ox0 - 1 + v;
            const bool inside = px < DC_NMID && (unsigned)my < (unsigned)a.H && (unsigned)mx < (unsigned)a.W;
            const u32x2 mine = inside ? dc_pack_silu(acc1[j]) : u32x2{0u, 0u};     // (c0 | c1 << 16, c2 | c3 << 16) of this lane's pixel
            const bool odd = fr & 1;
            const uint32_t give = odd ? mine.x : mine.y, keep = odd ? mine.y : mine.x;
            const uint32_t got = (uint32_t)__shfl_xor((int)give, 1);               // the partner pixel's half this lane assembles
            const uint32_t ev = odd ? got : keep, od = odd ? keep : got;           // (even pixel, odd pixel) of this lane's two channels
            const u32x2 wds = {(ev & 0xffffu) | (od << 16), (ev >> 16) | (od & 0xffff0000u)};
            if (px < DC_NMID) *reinterpret_cast<u32x2*>(sA + (px >> 1) * DC_PP + (wave * 4 + fc) * 16 + (odd ? 8 : 0)) = wds;
        }
        __syncthreads();
        // ---- dw2: region A (tile + 1) -> region B (tile) -------------------------------------------------------------------------------------
        dc_dw3<128, false>(sA, DC_MC / 2, sB, DC_NP / 2, DC_TW / 2, a.dw2, 0, a.bd2, cg, sl);
        __syncthreads();
        // ---- pw2: region B -> region A ----------------------------------------------------------------------------------------------------------
        {
            f32x4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = bv2;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 bfr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(sB + (j * 16 + fr) * DC_PITCH + fc * 16 + s * 64);
                DC_SCHED_BARRIER();
#pragma unroll
                for (int j = 0; j < 8; ++j) dc_mma(acc[j], af2[s], bfr[j]);
            }
            // region A still holds pw1's map, which dw2 of OTHER waves may be reading: they are past the barrier above, so it is free
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<u32x2*>(sA + (j * 16 + fr) * DC_PITCH + (wave * 16 + fc * 4) * 2) = dc_pack_silu(acc[j]);
        }
        __syncthreads();
        // ---- out: 1x1 128 -> nc, fp32 logits -------------------------------------------------------------------------------------------------------
        if (!DEC && wave * 16 < a.ncpad) {
            f32x4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = bv3;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 bfr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(sA + (j * 16 + fr) * DC_PITCH + fc * 16 + s * 64);
                DC_SCHED_BARRIER();
#pragma unroll
                for (int j = 0; j < 8; ++j) dc_mma(acc[j], af3[s], bfr[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int oy = oy0 + j, ox = ox0 + fr;   // fragment j = tile row j (16 pixels)
                if (oy < a.H && ox < a.W && wave * 16 + fc * 4 < a.ncpad)
                    *reinterpret_cast<f32x4*>(a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.ldy + wave * 16 + fc * 4) = acc[j];
            }
        } else if (DEC && wave * 16 < a.ncpad) {
            // Fused decode: the product is taken TRANSPOSED (pixels as the A operand, this wave's 16 classes as B): a lane then holds FOUR
            // CONSECUTIVE ANCHORS of one class, i.e. 16 contiguous bytes of a class row of y (with the classes along the registers every
            // score was a 4-byte store: 32 store instructions per lane and tile, +50 us at P3).  Same operands, same k order per output.
            // Region B (dw2's map, dead since pw2) collects the tile's scores [class][pixel] for the best-class pass below.
            float* sbest = reinterpret_cast<float*>(sB);
            const int cls = wave * 16 + fr;
            const bool vec = ((a.W | a.A | a.a_off) & 3) == 0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {   // (two halves: eight accumulators + the decode's temporaries beside three resident weight sets spill)
                f32x4 acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = f32x4{bvt, bvt, bvt, bvt};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    u32x4 bfr[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(sA + ((hh * 4 + j) * 16 + fr) * DC_PITCH + fc * 16 + s * 64);
                    DC_SCHED_BARRIER();
#pragma unroll
                    for (int j = 0; j < 4; ++j) dc_mma(acc[j], bfr[j], af3[s]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jj = hh * 4 + j;
                    const int oy = oy0 + jj, ox = ox0 + fc * 4;       // this lane: anchors ox .. ox + 3 of tile row jj, class cls
                    if (a.y && cls < a.ncpad && oy < a.H) {           // (logits wanted too: strided 4-byte stores, the rare path)
                        float* rp = a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.ldy + cls;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ox + r < a.W) rp[(size_t)r * a.ldy] = acc[j][r];
                    }
                    const f32x4 p = {dc_sigmoid(acc[j].x), dc_sigmoid(acc[j].y), dc_sigmoid(acc[j].z), dc_sigmoid(acc[j].w)};
                    if (cls < a.nc) {
                        *reinterpret_cast<f32x4*>(sbest + cls * DC_BEST_PITCH + jj * 16 + fc * 4) = p;
                        if (oy < a.H && ox < a.W && (!(DC_ABLATE & 16) || p.x == 12345.f)) {
                            float* yp = a.yo + ((size_t)b * (4 + a.nc) + 4 + cls) * a.A + a.a_off + oy * a.W + ox;
                            if (vec) {
                                *reinterpret_cast<f32x4*>(yp) = p;    // W % 4 == 0: the four anchors are inside the map together
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (ox + r < a.W) yp[r] = p[r];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();   // region A is restaged by the next tile
#ifdef DC_BEST_SERIAL   // rounds 4's form (A/B builds): 128 threads, one serial scan of the nc classes each
        if (DEC && a.bconf && t < DC_NP) {   // (region B is next written by dw1, behind the staging barrier of the next tile)
            const float* sbest = reinterpret_cast<const float*>(sB);
            const int oy = oy0 + (t >> 4), ox = ox0 + (t & 15);
            if (oy < a.H && ox < a.W) {
                float bv = sbest[t];
                int bc = 0;
                for (int cc = 1; cc < a.nc; ++cc) {   // ascending classes, strict >: the first maximum wins (utils/nms.py:124-129)
                    const float ov = sbest[cc * DC_BEST_PITCH + t];
                    if (ov > bv) { bv = ov; bc = cc; }
                }
                const size_t o = (size_t)b * a.A + a.a_off + oy * a.W + ox;
                a.bconf[o] = bv;
                a.bcls[o] = bc;
            }
        }
#else
        // Every anchor's best class, all 512 threads: thread (anchor p, quarter q) scans classes q, q + 4, q + 8 ... in ascending order
        // (strict >: its FIRST maximum), eight LDS reads in flight at a time; the four quarters of an anchor are four lanes apart by 128 and
        // meet through region A's first rows (dead until the next tile's staging, which waits at the barrier below); the combine keeps the
        // larger score and, between equal scores, the smaller class — the first maximum in class order (utils/nms.py:124-129).  The serial
        // form this replaces (128 threads x nc dependent compare-selects, six waves idle) was ~4 of a tile's ~19 us.
        if (DEC && a.bconf && !(DC_ABLATE & 32)) {
            const float* sbest = reinterpret_cast<const float*>(sB);
            const int p = t & (DC_NP - 1), q = t >> 7;
            float bv = -1.f;      // scores are sigmoids: > 0
            int bc = 0x7fffffff;
            for (int c0 = q; c0 < a.nc; c0 += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = c0 + 4 * u < a.nc ? sbest[(c0 + 4 * u) * DC_BEST_PITCH + p] : -1.f;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (v[u] > bv) { bv = v[u]; bc = c0 + 4 * u; }
            }
            float* comb = reinterpret_cast<float*>(sA);                 // [4][128] scores, then [4][128] classes
            comb[q * DC_NP + p] = bv;
            reinterpret_cast<int*>(comb)[(4 + q) * DC_NP + p] = bc;
        }
        __syncthreads();
        if (DEC && a.bconf && t < DC_NP) {
            const float* comb = reinterpret_cast<const float*>(sA);
            const int oy = oy0 + (t >> 4), ox = ox0 + (t & 15);
            if (oy < a.H && ox < a.W) {
                float bv = comb[t];
                int bc = reinterpret_cast<const int*>(comb)[4 * DC_NP + t];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float ov = comb[q * DC_NP + t];
                    const int oc = reinterpret_cast<const int*>(comb)[(4 + q) * DC_NP + t];
                    if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
                }
                const size_t o = (size_t)b * a.A + a.a_off + oy * a.W + ox;
                a.bconf[o] = bv;
                a.bcls[o] = bc;
            }
        }
        if (DEC && a.bconf) __syncthreads();   // the combine buffer (region A) is read before the next tile's staging writes it
#endif
    }
}

extern "C" int ymk_detect_cls_fused_supported(int32_t dtype, int32_t cin, int32_t c3, int32_t nc) {
    return dtype == YMK_BF16 && (cin == 128 || cin == 256) && c3 == 128 && nc >= 1 && nc <= 128;
}

extern "C" int ymk_detect_cls_fused(const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, int32_t cin, const void* dw1, const float* bd1,
                                    const void* pw1, int32_t k1pad, const float* bp1, const void* dw2, const float* bd2, const void* pw2,
                                    int32_t k2pad, const float* bp2, const void* w3, int32_t k3pad, const float* b3, int32_t ncpad, float* y,
                                    int32_t ldy, float* y_out, int32_t nc, int32_t a_off, int32_t A_total, float* best_conf,
                                    int32_t* best_cls, void* stream) {
    if (!x || !dw1 || !bd1 || !pw1 || !bp1 || !dw2 || !bd2 || !pw2 || !bp2 || !w3 || !b3 || (!y && !y_out)) return YMK_E_BADARG;
    if (!ymk_detect_cls_fused_supported(YMK_BF16, cin, 128, ncpad) || ncpad % 4 || ldx % 8 || ldx < cin || (y && (ldy % 4 || ldy < ncpad)) ||
        k1pad < cin || k2pad < 128 || k3pad < 128 || (k1pad | k2pad | k3pad) % 8)
        return YMK_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return YMK_E_BADARG;
    if ((best_conf == nullptr) != (best_cls == nullptr) || (best_conf && !y_out)) return YMK_E_BADARG;
    if (y_out && (nc < 1 || nc > ncpad || ncpad - nc >= 4 || nc > DC_BEST_MAXNC || a_off < 0 || (int64_t)a_off + (int64_t)H * W > A_total)) return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    DetClsArgs a;
    a.x = (const h16_t*)x; a.dw1 = (const h16_t*)dw1; a.pw1 = (const h16_t*)pw1; a.dw2 = (const h16_t*)dw2; a.pw2 = (const h16_t*)pw2;
    a.w3 = (const h16_t*)w3; a.bd1 = bd1; a.bp1 = bp1; a.bd2 = bd2; a.bp2 = bp2; a.b3 = b3; a.y = y;
    a.yo = y_out; a.bconf = best_conf; a.bcls = best_cls; a.nc = nc; a.A = A_total; a.a_off = a_off;
    a.B = B; a.H = H; a.W = W; a.ldx = ldx; a.ldy = ldy; a.k1pad = k1pad; a.k2pad = k2pad; a.k3pad = k3pad; a.ncpad = ncpad;
    a.tiles_x = (W + DC_TW - 1) / DC_TW; a.tiles_y = (H + DC_TH - 1) / DC_TH;
    const int64_t ntile = (int64_t)B * a.tiles_x * a.tiles_y;
    if (ntile >= (1ll << 31)) return YMK_E_BADARG;
#ifdef YMK_MAX_BLOCKS
    const unsigned grid = (unsigned)(ntile < YMK_MAX_BLOCKS ? ntile : YMK_MAX_BLOCKS);
#else
    const unsigned grid = (unsigned)(ntile < 256 ? ntile : 256);   // one persistent workgroup per CU (121 KB of LDS)
#endif
    static YmkOncePerDevice attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&detect_cls_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DC_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&detect_cls_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DC_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&detect_cls_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DC_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&detect_cls_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DC_LDS_BYTES);
        attr_once.done();
    }
    hipStream_t st = (hipStream_t)stream;
    if (cin == 128 && !y_out) hipLaunchKernelGGL((detect_cls_kernel<128, false>), dim3(grid), dim3(DC_NT), DC_LDS_BYTES, st, a);
    else if (cin == 128) hipLaunchKernelGGL((detect_cls_kernel<128, true>), dim3(grid), dim3(DC_NT), DC_LDS_BYTES, st, a);
    else if (!y_out) hipLaunchKernelGGL((detect_cls_kernel<256, false>), dim3(grid), dim3(DC_NT), DC_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((detect_cls_kernel<256, true>), dim3(grid), dim3(DC_NT), DC_LDS_BYTES, st, a);
    return ymk_launch_status();
}
