// Depthwise k x k convolution (bf16, k = 3/5/7/9, stride 1, pad k/2) on the MATRIX CORES, for gfx950.
//
// Why.  The VALU stencil (csrc/dwconv.hip) is arithmetic-bound, not bandwidth-bound: profiles/r01_sq_* show the VALU
// pipe ~85 % busy at 52 TFLOP/s, i.e. 0.46 ms for the 9x9/7x7/5x5/3x3 experts of the 160x160 ES-MoE layer against
// 0.1 ms of HBM time.  A depthwise filter has no contraction shared between channels, but along x it is a banded
// (Toeplitz) matrix: for one channel c and one filter row ky
//       out[x, y] += sum_xin T_{c,ky}[x, xin] * in[xin, y + ky],      T_{c,ky}[x, xin] = w[c][ky][xin - x + pad]
// which is a GEMM with M = 16 output columns, K = 32 input columns (16 + k - 1 <= 32 for k <= 17) and N = 16 image rows,
// all of ONE channel: one v_mfma_f32_16x16x32_bf16 per (channel, ky, 16x16 output tile), k of them per tile instead of
// 4*k*k wave-wide FMAs.  Only k/32 of the MACs are useful, and it is still 2-9x the VALU rate (the matrix pipe is
// 16x the vector pipe); the kernel then runs at the speed of its LDS operand reads.  Products are exact in fp32 (bf16
// inputs), accumulation is fp32: the result differs from the VALU stencil only in summation order.
//
// Structure (one workgroup = 4 waves = 16 channels; tile = 16 rows x 32 columns = two M tiles):
//   * A operand = the Toeplitz fragments of the wave's 4 channels, k per channel, 4 VGPRs each, loaded ONCE per
//     (expert, channel block) from a table built at pack time (ymk_dw_toeplitz_pack) and kept in registers while
//     the workgroup walks its run of (image/pair, tile) items — the filter never touches LDS.
//   * B operand = the input tile, transposed on the way into LDS from NHWC to [channel][row][column] so that the
//     8 consecutive columns of one channel a lane needs are one 16-byte ds_read_b128; the 96-byte row pitch makes
//     that access pattern bank-conflict free for the instruction's fixed 16-lane groups.  Staging: every thread
//     loads two x-adjacent pixels x 8 channels (2 x 16 bytes, coalesced), re-pairs them per channel (8 v_perm) and
//     writes 8 dwords.  LDS read traffic is 1 KB per MFMA = the LDS's 256 B/clk against 4 SIMDs x 16-cycle MFMAs.
//   * D: lane (row n, column group j) ends up with 4 consecutive output columns of its row for each of its 4
//     channels -> bias / SiLU / residual -> 8-byte NHWC stores through an fp32 LDS tile so that the workgroup writes 32 contiguous bytes per pixel.
//   * ES-MoE: work = the image->expert CSR of the router; one launch per filter size present (register allocation = that
//     filter's), a workgroup takes a contiguous range of the item list of the experts with that size.
//
// Reference semantics: DWConv (ultralytics/nn/modules/conv.py:185-199), AAttn.pe (nn/modules/block.py:1688,1731),
// DepthwiseSeparableConv.depthwise (nn/modules/moe/experts.py:283-292) per retained (image, expert) pair as in
// ES_MOE._sparse_forward (moe/modules.py:690-697).
#include "ymk_common.h"

typedef __bf16 dwm_bf16x8 __attribute__((ext_vector_type(8)));

// Diagnostic build switch (tools/micro/dw_ablate.sh): -DDWM_ABLATE=<bits> removes one stage of the item loop so that its cost can
// be read off a timing difference: 1 global loads, 2 LDS staging writes, 4 MFMA phase, 8 global stores, 16 output-tile writes.
#ifndef DWM_ABLATE
#define DWM_ABLATE 0
#endif

#define DWM_TH 16
#define DWM_TW 32
#define DWM_CC 16                                 // channels per workgroup
#define DWM_CPW 4                                 // channels per wave
#define DWM_PAIRS 24                              // x pairs per staged row: columns x0-8 .. x0+39
#define DWM_PITCH 96                              // bytes per LDS row = 48 columns (conflict-free B reads, see above)
#define DWM_KMAX 9
#define DWM_ROWS (DWM_TH + DWM_KMAX - 1)
#define DWM_PLANE (DWM_ROWS * DWM_PITCH)          // one channel plane
#define DWM_LDS_IN (DWM_CC * DWM_PLANE)           // 36,864 bytes: input tile, [channel][row][column] bf16
#define DWM_OPITCH 80                             // output tile: 16 fp32 channels per pixel + 16 bytes (b128 writes conflict-free)
#define DWM_LDS (DWM_LDS_IN + DWM_TH * DWM_TW * DWM_OPITCH)   // + 40,960 bytes

extern "C" size_t ymk_dw_toeplitz_elems(int32_t C, int32_t k) {
    if (C <= 0 || k < 1 || k > DWM_KMAX || !(k & 1)) return 0;
    return (size_t)C * k * 512;   // C * k fragments of 64 lanes x 8 bf16
}

// fragment (c, ky), lane l = (m = l & 15, j = l >> 4), element i: A[m][kk = 8j + i] = w[ky][kx = kk - 8 - m + pad][c]
// (LDS column kk of an M tile is image column x_tile - 8 + kk; output column m is x_tile + m)
__global__ __launch_bounds__(256) void dw_toeplitz_pack_kernel(const h16_t* __restrict__ w, int C, int k,
                                                              h16_t* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)C * k * 64) return;
    const int lane = (int)(idx & 63);
    const int ky = (int)((idx >> 6) % k), c = (int)((idx >> 6) / k);
    const int m = lane & 15, j = lane >> 4, pad = k / 2;
    h16_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int kx = 8 * j + i - 8 - m + pad;
        v[i] = (kx >= 0 && kx < k) ? w[(size_t)(ky * k + kx) * C + c] : (h16_t)0;
    }
    u32x4 o;
    o.x = v[0] | ((uint32_t)v[1] << 16); o.y = v[2] | ((uint32_t)v[3] << 16);
    o.z = v[4] | ((uint32_t)v[5] << 16); o.w = v[6] | ((uint32_t)v[7] << 16);
    reinterpret_cast<u32x4*>(out)[idx] = o;
}

extern "C" int ymk_dw_toeplitz_pack(const void* w_packed, int32_t C, int32_t k, void* out, void* stream) {
    if (!w_packed || !out || ymk_dw_toeplitz_elems(C, k) == 0) return YMK_E_BADARG;
    const int64_t n = (int64_t)C * k * 64;
    hipLaunchKernelGGL(dw_toeplitz_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const h16_t*)w_packed, C, k, (h16_t*)out);
    return ymk_launch_status();
}

struct DwmArgs {
    const h16_t* x;
    const h16_t* toep;      // Toeplitz fragments: expert-major, then [c][ky][64 lanes][8]
    const int* ksizes;       // [E] (device) or null: single filter of size k_single
    const int* csr_off;      // [E+1] (device) or null: every image once
    const int* csr_pair;     // [..] pair = b * top_k + slot
    h16_t* out;
    const float* bias;       // [C] or null
    const h16_t* res;       // residual view or null
    int B, H, W, C, ldx, ldo, ldr, act, E, top_k, tiles_x, tiles_y, k_single, G;
};

__device__ __forceinline__ unsigned xcd_remap_dwm(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// One run of items [i0, i1) of one filter: item = (pair index within the run's CSR segment, spatial tile).
// tile width by filter size: the 7- and 9-tap filters keep 112 / 144 A registers per lane, so they get one M tile (16 columns) where the
// others take two — that is what keeps the kernel at two waves per SIMD without spilling
__host__ __device__ __forceinline__ int dwm_tile_w(int k) { return k >= 7 ? 16 : 32; }

#ifdef YMK_HOST_EMU
static inline uint32_t dwm_perm(uint32_t s0, uint32_t s1, uint32_t sel) {   // v_perm_b32: byte i of the result = byte sel[i] of {s0:s1}
    const uint64_t v = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
#else
#define dwm_perm(s0, s1, sel) __builtin_amdgcn_perm(s0, s1, sel)
#endif

// One run of items [i0, i1) of one filter: item = (pair index within the run's CSR segment, spatial tile).
// Everything that does not depend on the item is computed once per run and kept in registers (per-thread byte offsets of
// the staging loads / LDS writes / output stores relative to the tile origin): the item loop itself is a handful of VALU
// operations per memory instruction — the first version recomputed its 64-bit addresses per item and was VALU-bound at
// 24 VALU instructions per MFMA (profiles/r02_dwmfma_v1_pmc.txt).
template <int K>
__device__ __forceinline__ void dwm_segment(const DwmArgs& a, const h16_t* __restrict__ toep_e, int csr_base, int i0,
                                            int i1, int chunk, char* smem) {
    constexpr int P = K / 2, R = DWM_TH + K - 1;
    constexpr int NMT = K >= 7 ? 1 : 2, TW = 16 * NMT, PAIRS = (TW + 16) / 2;
    constexpr int NL = (R * PAIRS * 2 + 255) / 256;          // staging iterations (two 16-byte loads each)
    constexpr int NO = (TW * DWM_TH * 2) / 256;              // output iterations (one 16-byte store each)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = lane & 15, j = lane >> 4;
    const int H = a.H, W = a.W, ldx = a.ldx, ldo = a.ldo, ldr = a.ldr;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = a.tiles_y;
    const int T = tiles_x * tiles_y;
    const int c0 = chunk * DWM_CC + wave * DWM_CPW;
    char* otile = smem + DWM_LDS_IN;

    // A operand: the Toeplitz fragments of this wave's channels, resident in registers for the whole run
    u32x4 A[DWM_CPW][K];
    {
        const u32x4* tp = reinterpret_cast<const u32x4*>(toep_e) + (size_t)c0 * K * 64 + lane;
#pragma unroll
        for (int cc = 0; cc < DWM_CPW; ++cc)
#pragma unroll
            for (int ky = 0; ky < K; ++ky) A[cc][ky] = tp[(cc * K + ky) * 64];
    }
    // staging constants: thread item q = t + 256 it -> (channel octet g8, row r, x pair pr) of the staged region
    uint32_t sgoff[NL], sldst[NL];
#pragma unroll
    for (int it = 0; it < NL; ++it) {
        const int q = t + it * 256;
        const int g8 = q & 1, rp = q >> 1;
        const int r = rp / PAIRS, pr = rp - r * PAIRS;
        sgoff[it] = (uint32_t)((r * W + 2 * pr) * ldx * 2 + g8 * 16);     // bytes from pixel (y0 - P, x0 - 8)
        sldst[it] = (uint32_t)((g8 * 8) * DWM_PLANE + (r < R ? r : 0) * DWM_PITCH + pr * 4);
    }
    const char* bp = smem + (wave * DWM_CPW) * DWM_PLANE + n * DWM_PITCH + j * 16;        // B fragments: immediates from here
    char* ot1 = otile + ((4 * j) * 16 + n) * DWM_OPITCH + wave * 16;                      // epilogue stage 1
    // output constants: q = t + 256 it -> (half, column (t >> 5) + 8 it, row (t >> 1) & 15)
    const int ocol = t >> 5, orow = (t >> 1) & 15, ohalf = t & 1;
    const char* ot2 = otile + (t >> 1) * DWM_OPITCH + ohalf * 32;
    const uint32_t ooff = (uint32_t)((orow * W + ocol) * ldo * 2 + ohalf * 16);           // bytes from pixel (y0, x0)
    const uint32_t roff = (uint32_t)((orow * W + ocol) * ldr * 2 + ohalf * 16);
    float bs[DWM_CPW];
#pragma unroll
    for (int cc = 0; cc < DWM_CPW; ++cc) bs[cc] = a.bias ? a.bias[c0 + cc] : 0.f;

#ifndef YMK_HOST_EMU
    // The A fragments (and the bias) must have LANDED before the item loop: stated explicitly, because otherwise the compiler
    // keeps counted vmcnt waits for them inside the loop's MFMA sequence, and in steady state those waits drain the NEXT item's
    // prefetch loads instead — serialising the prefetch with the MFMA phase (first version: ~5 us per item whatever the filter).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
#endif
    u32x4 v0[NL], v1[NL];
    auto issue_loads = [&](int b, int y0, int x0) {
        // rows y0-P .. y0+15+P, columns x0-8 .. x0+TW+7, 16 channels; every load of the thread is in flight at once
        const char* gb = reinterpret_cast<const char*>(a.x) + ((((int64_t)b * H + (y0 - P)) * W + (x0 - 8)) * ldx + chunk * DWM_CC) * 2;
        const bool interior = y0 - P >= 0 && y0 + DWM_TH + P <= H && x0 - 8 >= 0 && x0 + TW + 8 <= W;
        if (DWM_ABLATE & 1) {
#pragma unroll
            for (int it = 0; it < NL; ++it) { v0[it] = u32x4{(uint32_t)y0, 1u, 2u, 3u}; v1[it] = u32x4{(uint32_t)x0, 5u, 6u, 7u}; }
        } else if (interior) {
#pragma unroll
            for (int it = 0; it < NL; ++it) {
                if (NL * 256 <= R * PAIRS * 2 || t + it * 256 < R * PAIRS * 2) {
                    v0[it] = *reinterpret_cast<const u32x4*>(gb + sgoff[it]);
                    v1[it] = *reinterpret_cast<const u32x4*>(gb + sgoff[it] + ldx * 2);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < NL; ++it) {
                const int rp = (t + it * 256) >> 1;          // edge tiles only: the (row, pair) of this load, recomputed
                const int r = rp / PAIRS, pr = rp - r * PAIRS;
                const int y = y0 - P + r, x = x0 + 2 * pr - 8;
                const bool oky = r < R && (unsigned)y < (unsigned)H;
                v0[it] = u32x4{0u, 0u, 0u, 0u};
                v1[it] = u32x4{0u, 0u, 0u, 0u};
                if (oky && (unsigned)x < (unsigned)W) v0[it] = *reinterpret_cast<const u32x4*>(gb + sgoff[it]);
                if (oky && (unsigned)(x + 1) < (unsigned)W) v1[it] = *reinterpret_cast<const u32x4*>(gb + sgoff[it] + ldx * 2);
            }
        }
    };
    // item -> (pair index, tile row, tile column), advanced incrementally
    int pidx = i0 / T, ty = (i0 - pidx * T) / tiles_x, tx = i0 - pidx * T - ty * tiles_x;
    int pair = a.csr_pair ? a.csr_pair[csr_base + pidx] : pidx;
    int b = a.csr_pair ? pair / a.top_k : pair;
    issue_loads(b, ty * DWM_TH, tx * TW);

    __syncthreads();   // a previous segment's LDS reads are done
    for (int item = i0; item < i1; ++item) {
        // ---- NHWC registers -> LDS [channel][row][column]: re-pair two x-adjacent pixels per channel (v_perm_b32) ------
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            if (!(DWM_ABLATE & 2) && (NL * 256 <= R * PAIRS * 2 || t + it * 256 < R * PAIRS * 2)) {
                char* dst = smem + sldst[it];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<uint32_t*>(dst + (2 * i) * DWM_PLANE) = dwm_perm(v1[it][i], v0[it][i], 0x05040100u);      // channel 2i:   (x, x+1)
                    *reinterpret_cast<uint32_t*>(dst + (2 * i + 1) * DWM_PLANE) = dwm_perm(v1[it][i], v0[it][i], 0x07060302u);  // channel 2i+1: (x, x+1)
                }
            }
        }
        __syncthreads();
        const int cpair = pair, cb = b, cy0 = ty * DWM_TH, cx0 = tx * TW;

        // the next item's loads are issued as soon as the staging registers are free: their round trip overlaps this item's
        // MFMA phase and epilogue, and they are queued ahead of this item's stores
        if (item + 1 < i1) {
            if (++tx == tiles_x) {
                tx = 0;
                if (++ty == tiles_y) {
                    ty = 0;
                    ++pidx;
                    pair = a.csr_pair ? a.csr_pair[csr_base + pidx] : pidx;
                    b = a.csr_pair ? pair / a.top_k : pair;
                }
            }
            issue_loads(b, ty * DWM_TH, tx * TW);
        }

        // ---- k MFMAs per (channel, M tile): B fragment = 8 consecutive columns of row n + ky --------------------------
        f32x4 acc[DWM_CPW][NMT];
#pragma unroll
        for (int cc = 0; cc < DWM_CPW; ++cc)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[cc][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(DWM_ABLATE & 4))
#pragma unroll
        for (int cc = 0; cc < DWM_CPW; ++cc)
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const u32x4 bf = *reinterpret_cast<const u32x4*>(bp + cc * DWM_PLANE + ky * DWM_PITCH + mt * 32);
                    acc[cc][mt] = mfma16x16x32_h16(A[cc][ky], bf, acc[cc][mt]);
                }

        // ---- epilogue, stage 1: lane (n, j) holds output columns 16 mt + 4 j + i of row n for its 4 channels: bias / SiLU in
        //      fp32, then into the output tile [column][row][16 channels] (fp32, 80-byte pitch: conflict-free b128 writes) ----
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int cc = 0; cc < DWM_CPW; ++cc) {
                    float u = acc[cc][mt][i] + bs[cc];
                    if (a.act == YMK_ACT_SILU) u = silu_f(u);
                    v[cc] = u;
                }
                if (!(DWM_ABLATE & 16)) *reinterpret_cast<f32x4*>(ot1 + (mt * 16 + i) * 16 * DWM_OPITCH) = v;
            }
        __syncthreads();
        // ---- stage 2: the workgroup writes 32 contiguous bytes per pixel (lane pairs), residual added in fp32 first ---------
        {
            const size_t plane = a.csr_pair ? (size_t)cpair : (size_t)cb;
            char* ob = reinterpret_cast<char*>(a.out) + (((plane * H + cy0) * W + cx0) * (size_t)ldo + chunk * DWM_CC) * 2;
            const char* rb = a.res ? reinterpret_cast<const char*>(a.res) + ((((size_t)cb * H + cy0) * W + cx0) * (size_t)ldr + chunk * DWM_CC) * 2
                                   : nullptr;
            const bool rowok = cy0 + orow < H;
#pragma unroll
            for (int it = 0; it < NO; ++it) {
                if (rowok && cx0 + ocol + 8 * it < W) {
                    const float* src = reinterpret_cast<const float*>(ot2 + it * 128 * DWM_OPITCH);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    if (rb) {
                        float r[8];
                        load_vec_f32(reinterpret_cast<const h16_t*>(rb + roff + it * 8 * ldr * 2), r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    if (!(DWM_ABLATE & 8) || v[0] == 1234.5f) store_vec_f32(reinterpret_cast<h16_t*>(ob + ooff + it * 8 * ldo * 2), v);
                }
            }
        }
    }
}

#ifdef YMK_HOST_EMU
#define DWM_OCCUPANCY
#else
#define DWM_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2)))   // 2 waves/SIMD (the 9-tap instantiation sits at 242 registers)
#endif
// One kernel per filter size (register allocation = that filter's, not the maximum over a switch).  The workgroup takes a
// cost-balanced contiguous range of the item list of the experts whose filter size is K (for ES-MoE that is normally one
// expert; for a plain depthwise convolution the single filter).
template <int K>
__global__ __launch_bounds__(256) DWM_OCCUPANCY void dw_mfma_kernel(DwmArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[DWM_LDS];
    const int nchunk = a.C / DWM_CC;
    const unsigned lid = xcd_remap_dwm(blockIdx.x, gridDim.x);   // the channel blocks of one run share an XCD (L2 lines)
    const int chunk = (int)(lid % nchunk), g = (int)(lid / nchunk);
    const int T = ((a.W + dwm_tile_w(K) - 1) / dwm_tile_w(K)) * a.tiles_y;
    long long tot = 0;
    for (int e = 0; e < a.E; ++e) {
        const int k = a.ksizes ? a.ksizes[e] : a.k_single;
        if (k == K) tot += (long long)(a.csr_off ? a.csr_off[e + 1] - a.csr_off[e] : a.B) * T;
    }
    const long long lo = tot * g / a.G, hi = tot * (g + 1) / a.G;
    long long S = 0;
    size_t frag = 0;   // fragment index of expert e's table
    for (int e = 0; e < a.E; ++e) {
        const int k = a.ksizes ? a.ksizes[e] : a.k_single;
        if (k == K) {
            const long long nT = (long long)(a.csr_off ? a.csr_off[e + 1] - a.csr_off[e] : a.B) * T;
            const long long d0 = lo - S, d1 = hi - S;
            const int i0 = (int)(d0 <= 0 ? 0 : min(d0, nT));
            const int i1 = (int)(d1 <= 0 ? 0 : min(d1, nT));
            if (i1 > i0) dwm_segment<K>(a, a.toep + frag * 512, a.csr_off ? a.csr_off[e] : 0, i0, i1, chunk, smem);
            S += nT;
        }
        frag += (size_t)a.C * k;
    }
}

template <int K>
static void dwm_launch_k(DwmArgs a, int64_t max_items, hipStream_t s) {
    const int nchunk = a.C / DWM_CC;
    // ~3 workgroups per CU, but at least ~6 items per workgroup so that the A registers are amortised
    int64_t G = (768 + nchunk - 1) / nchunk;
    const int64_t items = max_items * ((a.W + dwm_tile_w(K) - 1) / dwm_tile_w(K)) * a.tiles_y;
    if (G > (items + 5) / 6) G = (items + 5) / 6;
    if (G < 1) G = 1;
    a.G = (int)G;
    hipLaunchKernelGGL(dw_mfma_kernel<K>, dim3((unsigned)(nchunk * G)), dim3(256), 0, s, a);
}

// kmask: bit (k - 1) / 2 set for every filter size k present (host knowledge: the sizes are module hyper-parameters)
static int dwm_launch(DwmArgs a, int64_t max_items, int kmask, hipStream_t s) {
    if (a.C % DWM_CC || a.ldx % 8 || a.ldo % 8 || (a.res && a.ldr % 8)) return YMK_E_BADARG;
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) return YMK_OK;
    a.tiles_x = 0;
    a.tiles_y = (a.H + DWM_TH - 1) / DWM_TH;
    if (kmask & 2) dwm_launch_k<3>(a, max_items, s);
    if (kmask & 4) dwm_launch_k<5>(a, max_items, s);
    if (kmask & 8) dwm_launch_k<7>(a, max_items, s);
    if (kmask & 16) dwm_launch_k<9>(a, max_items, s);
    return ymk_launch_status();
}

extern "C" int ymk_dw_mfma_supported(int32_t dtype, int32_t C, int32_t k) {
    return dtype == YMK_BF16 && C > 0 && C % DWM_CC == 0 && k >= 3 && k <= DWM_KMAX && (k & 1);
}

extern "C" int ymk_dwconv2d_mfma(const void* x, const void* toep, const float* bias, const void* residual, void* y,
                                 int32_t B, int32_t H, int32_t W, int32_t C, int32_t ksize, int32_t ldx, int32_t ldy,
                                 int32_t ldr, int32_t act, void* stream) {
    if (!x || !toep || !y || !ymk_dw_mfma_supported(YMK_BF16, C, ksize)) return YMK_E_BADARG;
    if (act != YMK_ACT_NONE && act != YMK_ACT_SILU) return YMK_E_BADARG;
    DwmArgs a{(const h16_t*)x, (const h16_t*)toep, nullptr, nullptr, nullptr, (h16_t*)y, bias, (const h16_t*)residual,
              B, H, W, C, ldx, ldy, ldr, act, 1, 1, 0, 0, ksize, 1};
    return dwm_launch(a, B, 1 << (ksize / 2), (hipStream_t)stream);
}

extern "C" int ymk_esmoe_dw_mfma(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx, const void* toep,
                                 const int32_t* ksizes, int32_t kmask, int32_t E, int32_t top_k, const int32_t* csr_off,
                                 const int32_t* csr_pair, void* dw_out, void* stream) {
    if (!x || !toep || !ksizes || !csr_off || !csr_pair || !dw_out || E < 1 || E > 16 || top_k < 1) return YMK_E_BADARG;
    if (kmask <= 0 || (kmask & ~30)) return YMK_E_BADARG;   // filter sizes 3, 5, 7, 9 only
    DwmArgs a{(const h16_t*)x, (const h16_t*)toep, ksizes, csr_off, csr_pair, (h16_t*)dw_out, nullptr, nullptr,
              B, H, W, C, ldx, C, 0, YMK_ACT_NONE, E, top_k, 0, 0, 0, 1};
    return dwm_launch(a, (int64_t)B * top_k, kmask, (hipStream_t)stream);
}
