// Design probe #2 for the next tiled implicit-GEMM core: the gemm256 staging scheme carried over to
// real convolutions (1x1 and 3x3, stride 1 / 2, zero padding through a zero page), with the k-loop in three flavours
// so that ONE run on the GPU box ranks them next to the library's current kernels:
//
//   STAGES = 2, plain __syncthreads()          (what gemm256.hip measured: DMA drained at every barrier)
//   STAGES = 3, raw s_barrier + counted vmcnt  (one k-step of DMA stays in flight across each barrier)
//   both with and without the XCD-aware tile order (tiles that share activation rows land on one XCD's L2)
//
// Tile: 256 output pixels x BN couts (BN = 128 or 64) on 8 waves, BK = 64 channels of one filter tap per k-step
// (Cin % 64 == 0), operands staged by global_load_lds (16 B per lane, lane-linear LDS image, XOR swizzle applied to the
// SOURCE chunk).  Standalone and self-checking (naive kernel on sampled pixels); NOT part of libymk.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/conv256.hip -o tools/micro/conv256.bin
//   tools/micro/conv256.bin [nshapes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifndef YMK_HOST_EMU
#include "../../yolo_master_amd/csrc/conv.hip"   // library kernels for the side-by-side number + mma16 / store4 / silu_f

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
// Host emulation (tests/hostemu): the kernel below runs lane by lane on the CPU — matrix core, LDS-DMA and barriers are
// emulated, DMA completes at once — so that its addressing (taps, masks, swizzle, tail tiles, XCD order) is checked
// against the naive convolution before the first GPU call.  No library kernels, no timing.
#include <cmath>
#include "../../yolo_master_amd/csrc/ymk_common.h"
typedef const void* gptr_t;
typedef void* lptr_t;
#define WAIT_LGKM0() ((void)0)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <typename T> inline void mma16(f32x4& acc, const u32x4& a, const u32x4& b);
template <> inline void mma16<h16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
#endif

#define C_BM 256

struct ConvGeom {
    int B, H, W, Cin, Cout, ks, stride, Ho, Wo, Kpad;
    int ldx, ldy;
};

// s_waitcnt immediates for gfx9-family encodings: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
#define WAITCNT_VM(n) (0x0F70 | ((n) & 15) | ((((n) >> 4) & 3) << 14))

template <int BN, int STAGES, bool XCD>
__global__ __launch_bounds__(512) void conv256_kernel(const h16_t* __restrict__ X, const h16_t* __restrict__ Wt,
                                                       const float* __restrict__ bias, h16_t* __restrict__ Y,
                                                       const h16_t* __restrict__ zero_page, ConvGeom g) {
    constexpr int ROWS = BN + C_BM;          // staged rows per k-step: weights first, then pixels
    constexpr int STAGE_U4 = ROWS * 8;       // 16-byte slots per stage
    constexpr int G = ROWS / 64;             // global_load_lds instructions per wave per k-step (6 or 5)
    constexpr int GW = BN / 64;              // of which weight rows
    constexpr int WN = BN / 64;              // waves along couts (64 couts per wave)
    constexpr int WM = 8 / WN;               // waves along pixels
    constexpr int TP = C_BM / WM / 16;       // 16-pixel fragments per wave (4 or 2)
    extern __shared__ u32x4 smem[];          // STAGES * STAGE_U4

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int M = g.B * g.Ho * g.Wo;
    const int nt = g.Cout / BN;
    int bid = blockIdx.x;
    if (XCD) {   // bijective remap: consecutive logical tiles (same pixel rows, different couts) share an XCD
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = (bid / nt) * C_BM, n0 = (bid % nt) * BN;
    const int cpt = g.Cin >> 6;              // k-steps per filter tap
    const int nk = g.ks * g.ks * cpt;
    const int pad = g.ks >> 1;

    // ---- staging map --------------------------------------------------------------------------------------
    const int lr = lane >> 3, lc = lane & 7;
    const h16_t* wsrc[GW];
    int poff[G - GW];        // element offset of the tap-(0,0) input pixel (+ swizzled chunk) for this lane's pixel rows
    unsigned pmask[G - GW];  // bit (ky*3+kx): tap inside the image
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int r = (j * 8 + wave) * 8 + lr;       // staged row
        const int sc = (lc ^ (r & 7)) * 8;            // source chunk (elements)
        if (j < GW) {
            wsrc[j] = Wt + (size_t)(n0 + r) * g.Kpad + sc;
        } else {
            const int p = m0 + r - BN;
            unsigned mask = 0;
            int off = 0;
            if (p < M) {
                const int ox = p % g.Wo, oy = (p / g.Wo) % g.Ho, b = p / (g.Wo * g.Ho);
                const int iy0 = oy * g.stride - pad, ix0 = ox * g.stride - pad;
                off = ((b * g.H + iy0) * g.W + ix0) * g.ldx + sc;
                unsigned ry = 0, rx = 0;   // rows / columns of the filter window that fall inside the image
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (k < g.ks && (unsigned)(iy0 + k) < (unsigned)g.H) ry |= 1u << k;
                    if (k < g.ks && (unsigned)(ix0 + k) < (unsigned)g.W) rx |= 1u << k;
                }
                mask = ((ry & 1u) ? rx : 0u) | ((ry & 2u) ? rx << 3 : 0u) | ((ry & 4u) ? rx << 6 : 0u);
            }
            poff[j - GW] = off;
            pmask[j - GW] = mask;
        }
    }
    const h16_t* zsrc = zero_page + lc * 8;
    // k-step cursor of the NEXT tile to issue (uniform)
    int it_tap_bit = 0, it_ky = 0, it_kx = 0, it_c = 0, it_k = 0;
    auto issue = [&](int stage) {
        const int tapoff = (it_ky * g.W + it_kx) * g.ldx + it_c * 64;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            u32x4* dst = smem + stage * STAGE_U4 + (j * 8 + wave) * 64;   // wave-uniform; lane lands at + lane * 16 B
            const h16_t* s;
            if (j < GW) s = wsrc[j] + it_k * 64;
            else s = ((pmask[j - GW] >> it_tap_bit) & 1u) ? X + (poff[j - GW] + tapoff) : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)dst, 16, 0, 0);
        }
        ++it_k;
        if (++it_c == cpt) {
            it_c = 0;
            ++it_kx; ++it_tap_bit;
            if (it_kx == g.ks) { it_kx = 0; ++it_ky; it_tap_bit = it_ky * 3; }
        }
    };

    f32x4 acc[4][TP];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int stage) {
        const u32x4* sW = smem + stage * STAGE_U4;
        const u32x4* sX = sW + BN * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 af[4], bfr[TP];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = ((wave % WN) * 4 + i) * 16 + fr;
                af[i] = sW[r * 8 + ((kk * 4 + fc) ^ (r & 7))];
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int r = ((wave / WN) * TP + j) * 16 + fr;   // BN % 8 == 0: (BN + r) & 7 == r & 7
                bfr[j] = sX[r * 8 + ((kk * 4 + fc) ^ (r & 7))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma16<h16_t>(acc[i][j], af[i], bfr[j]);
        }
    };

    if constexpr (STAGES == 2) {
        issue(0);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();   // drains the DMA (vmcnt(0)): stage kt&1 complete, the other one free
            if (kt + 1 < nk) issue((kt + 1) & 1);
            compute(kt & 1);
        }
    } else {
        // three stages: tile kt+1 stays in flight across the barrier that publishes tile kt
        issue(0);
        if (nk > 1) issue(1);
        int cur = 0, nxt = 2;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(WAITCNT_VM(G));   // my pieces of tile kt have landed
            else __builtin_amdgcn_s_waitcnt(WAITCNT_VM(0));
            WAIT_LGKM0();                                                 // my reads of tile kt-1 are done (WAR on stage nxt)
            __builtin_amdgcn_s_barrier();                                 // everyone's pieces landed / reads done
#ifndef YMK_HOST_EMU
            asm volatile("" ::: "memory");
#endif
            if (kt + 2 < nk) issue(nxt);
            compute(cur);
            cur = cur == 2 ? 0 : cur + 1;
            nxt = nxt == 2 ? 0 : nxt + 1;
        }
    }

    // ---- epilogue: bias + SiLU, 4 consecutive couts per lane ---------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = n0 + ((wave % WN) * 4 + i) * 16 + fc * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int p = m0 + ((wave / WN) * TP + j) * 16 + fr;
            if (p < M)
                store4(Y + (size_t)p * g.ldy + co, silu_f(acc[i][j].x + bv.x), silu_f(acc[i][j].y + bv.y),
                       silu_f(acc[i][j].z + bv.z), silu_f(acc[i][j].w + bv.w));
        }
    }
}

__global__ void naive_conv_kernel(const h16_t* X, const h16_t* Wt, const float* bias, float* Yr, ConvGeom g, const int* pix,
                                  int npix) {
    const int i = blockIdx.x, n = threadIdx.x + blockIdx.y * blockDim.x;
    if (i >= npix || n >= g.Cout) return;
    const int p = pix[i];
    const int ox = p % g.Wo, oy = (p / g.Wo) % g.Ho, b = p / (g.Wo * g.Ho), pad = g.ks / 2;
    float s = 0.f;
    for (int ky = 0; ky < g.ks; ++ky)
        for (int kx = 0; kx < g.ks; ++kx) {
            const int iy = oy * g.stride - pad + ky, ix = ox * g.stride - pad + kx;
            if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
            const h16_t* xr = X + ((size_t)(b * g.H + iy) * g.W + ix) * g.ldx;
            const h16_t* wr = Wt + (size_t)n * g.Kpad + (ky * g.ks + kx) * g.Cin;
            for (int c = 0; c < g.Cin; ++c) s += h16_to_f32(xr[c]) * h16_to_f32(wr[c]);
        }
    s += bias[n];
    Yr[(size_t)i * g.Cout + n] = s / (1.0f + expf(-s));
}

template <typename F>
static float timeit(F&& f, int reps = 20) {
#ifdef YMK_HOST_EMU
    return 1.0f;   // no timing on the emulator
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < 3; ++i) f();
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ts[ts.size() / 2];
}

template <int BN, int STAGES, bool XCD>
static int launch(const h16_t* x, const h16_t* w, const float* b, h16_t* y, const h16_t* zp, const ConvGeom& g) {
    const int M = g.B * g.Ho * g.Wo;
    const int grid = ((M + C_BM - 1) / C_BM) * (g.Cout / BN);
    const size_t lds = (size_t)STAGES * (BN + C_BM) * 8 * 16;
    static bool once = false;
    if (!once) {
        once = true;
        if (hipFuncSetAttribute((const void*)conv256_kernel<BN, STAGES, XCD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL((conv256_kernel<BN, STAGES, XCD>), dim3(grid), dim3(512), lds, 0, x, w, b, y, zp, g);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

typedef int (*launch_fn)(const h16_t*, const h16_t*, const float*, h16_t*, const h16_t*, const ConvGeom&);
struct Variant { const char* name; int bn; launch_fn fn; };

int main(int argc, char** argv) {
    // the tiled-GEMM launches that dominate the S model at batch 64 (gpurun_out calls log), largest first
    struct Shape { int B, H, W, Cin, Cout, ks, stride; } shapes[] = {
#ifdef YMK_HOST_EMU
        {2, 23, 19, 64, 128, 3, 2},       // one full + one tail tile, borders on every side
        {1, 18, 17, 128, 64, 3, 1},       // two k-steps per tap, BN = 64 only
        {3, 9, 11, 192, 128, 1, 1},       // 1x1: three k-steps
        {1, 40, 13, 64, 256, 3, 1},       // several cout tiles per pixel tile (XCD order matters)
#else
        {64, 160, 160, 128, 128, 3, 2},   // 128->128 k3 s2 @80x80 out     (246 us today)
        {64, 80, 80, 256, 256, 3, 2},     // 256->256 k3 s2 @40x40 out     (221 us)
        {64, 80, 80, 128, 64, 3, 1},      // 128->64  k3 s1 @80x80         (165 us)
        {64, 40, 40, 256, 512, 3, 2},     // 256->512 k3 s2 @20x20 out     (121 us)
        {64, 80, 80, 512, 128, 1, 1},     // 512->128 k1 @80x80 (cat2-like, 124 us)
        {64, 40, 40, 768, 256, 1, 1},     // 768->256 k1 @40x40 (cat2-like, 102 us)
        {64, 40, 40, 256, 64, 3, 1},      // 256->64  k3 s1 @40x40         (92 us)
        {64, 40, 40, 384, 256, 1, 1},     // 384->256 k1 @40x40            (70-76 us)
        {64, 20, 20, 256, 64, 3, 1},      // 256->64  k3 s1 @20x20         (70 us: 100 tiles, latency-bound)
        {3, 37, 29, 64, 128, 3, 2},       // ragged: tail tile, odd sizes, borders everywhere (correctness only)
#endif
    };
    const int total = (int)(sizeof(shapes) / sizeof(shapes[0]));
    const int nshape = argc > 1 ? std::min(atoi(argv[1]), total) : total;
    const Variant variants[] = {
        {"bn128 s2      ", 128, launch<128, 2, false>}, {"bn128 s3      ", 128, launch<128, 3, false>},
        {"bn128 s3 xcd  ", 128, launch<128, 3, true>},  {"bn64  s2      ", 64, launch<64, 2, false>},
        {"bn64  s3      ", 64, launch<64, 3, false>},   {"bn64  s3 xcd  ", 64, launch<64, 3, true>},
    };
    h16_t* zp;
    hipMalloc(&zp, 256); hipMemset(zp, 0, 256);
    int bad = 0;
    for (int si = 0; si < nshape; ++si) {
        const Shape sh = shapes[si];
        ConvGeom g{sh.B, sh.H, sh.W, sh.Cin, sh.Cout, sh.ks, sh.stride, 0, 0, 0, sh.Cin, sh.Cout};
        const int pad = sh.ks / 2;
        g.Ho = (sh.H + 2 * pad - sh.ks) / sh.stride + 1;
        g.Wo = (sh.W + 2 * pad - sh.ks) / sh.stride + 1;
        g.Kpad = sh.ks * sh.ks * sh.Cin;   // multiple of 64 by construction
        const int M = g.B * g.Ho * g.Wo;
        const size_t nx = (size_t)sh.B * sh.H * sh.W * sh.Cin, nw = (size_t)sh.Cout * g.Kpad, ny = (size_t)M * sh.Cout;
        std::vector<h16_t> hx(nx), hw(nw);
        std::vector<float> hb(sh.Cout);
        uint32_t s = 12345u + si;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hx) { float f = rnd(); v = (h16_t)(__builtin_bit_cast(uint32_t, f) >> 16); }
        const float wscale = 2.0f / sqrtf((float)g.Kpad);
        for (auto& v : hw) { float f = rnd() * wscale; v = (h16_t)(__builtin_bit_cast(uint32_t, f) >> 16); }
        for (auto& v : hb) v = rnd() * 0.2f;
        h16_t *x, *w, *y, *y2; float *b, *yr; int* pix;
        hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2); hipMalloc(&y2, ny * 2); hipMalloc(&b, sh.Cout * 4);
        hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemcpy(b, hb.data(), sh.Cout * 4, hipMemcpyHostToDevice);
        const int npix = 768;
        std::vector<int> hp(npix);
        for (int i = 0; i < npix; ++i) hp[i] = (int)(((size_t)i * 1040543u + (size_t)(i % 5) * (M - 1) / 4) % M);
        hp[0] = 0; hp[1] = M - 1; hp[2] = g.Wo - 1; hp[3] = g.Wo * (g.Ho - 1); hp[4] = std::min(M - 1, 255); hp[5] = std::min(M - 1, 256);
        hipMalloc(&pix, npix * 4); hipMemcpy(pix, hp.data(), npix * 4, hipMemcpyHostToDevice);
        hipMalloc(&yr, (size_t)npix * sh.Cout * 4);
        hipLaunchKernelGGL(naive_conv_kernel, dim3(npix, (sh.Cout + 127) / 128), dim3(128), 0, 0, x, w, b, yr, g, pix, npix);
        if (hipDeviceSynchronize() != hipSuccess) { printf("naive launch failed\n"); return 1; }
        std::vector<float> hyr((size_t)npix * sh.Cout);
        hipMemcpy(hyr.data(), yr, hyr.size() * 4, hipMemcpyDeviceToHost);
        double maxref = 0;
        for (float v : hyr) maxref = std::max(maxref, (double)fabsf(v));
        const double flops = 2.0 * M * sh.Cout * g.Kpad, bytes = 2.0 * (nx / (double)(sh.stride == 2 && sh.ks == 1 ? 4 : 1) + nw + ny);
        printf("shape %d: %d->%d k%d s%d in %dx%d out %dx%d  M %d K %d  (%.2f GFLOP, %.1f MB)\n", si, sh.Cin, sh.Cout, sh.ks, sh.stride,
               sh.H, sh.W, g.Ho, g.Wo, M, g.Kpad, flops / 1e9, bytes / 1e6);
        std::vector<h16_t> hy(ny);
        auto check = [&](const h16_t* dev) {
            hipMemcpy(hy.data(), dev, ny * 2, hipMemcpyDeviceToHost);
            double maxerr = 0;
            for (int i = 0; i < npix; ++i)
                for (int n = 0; n < sh.Cout; ++n) {
                    const float got = __builtin_bit_cast(float, (uint32_t)hy[(size_t)hp[i] * sh.Cout + n] << 16);
                    maxerr = std::max(maxerr, (double)fabsf(got - hyr[(size_t)i * sh.Cout + n]));
                }
            return maxerr;
        };
        for (const Variant& v : variants) {
            if (sh.Cout % v.bn) continue;
            hipMemset(y, 0xff, ny * 2);
            if (v.fn(x, w, b, y, zp, g) != 0 || hipDeviceSynchronize() != hipSuccess) {
                printf("  %s launch failed: %s\n", v.name, hipGetErrorString(hipGetLastError()));
                ++bad;
                continue;
            }
            const double err = check(y);
            const bool ok = err <= 0.02 * std::max(1.0, maxref);
            bad += !ok;
            const float ms = timeit([&] { v.fn(x, w, b, y, zp, g); });
            printf("  %s %8.1f us %7.1f TF/s %5.2f TB/s   max |err| %.3e (max |ref| %.2f) %s\n", v.name, ms * 1e3, flops / ms / 1e9,
                   bytes / ms / 1e9, err, maxref, ok ? "OK" : "MISMATCH");
        }
#ifndef YMK_HOST_EMU
        ymk_conv_desc d{YMK_BF16, YMK_BF16, sh.B, sh.H, sh.W, sh.Cin, sh.Cout, sh.ks, sh.stride, sh.Cin, sh.Cout, 0, g.Kpad, YMK_ACT_SILU};
        hipMemset(y2, 0xff, ny * 2);
        if (ymk_conv2d(&d, x, w, b, nullptr, y2, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess) {
            const double err = check(y2);
            const float ms = timeit([&] { ymk_conv2d(&d, x, w, b, nullptr, y2, nullptr); });
            printf("  library (v%d)    %8.1f us %7.1f TF/s %5.2f TB/s   max |err| %.3e\n", ymk_conv2d_last_variant(), ms * 1e3,
                   flops / ms / 1e9, bytes / ms / 1e9, err);
        } else {
            printf("  library: rejected this shape\n");
        }
#endif
        hipFree(x); hipFree(w); hipFree(y); hipFree(y2); hipFree(b); hipFree(yr); hipFree(pix);
    }
    printf(bad ? "RESULT: %d variant(s) FAILED\n" : "RESULT: all variants match the naive convolution\n", bad);
    return bad ? 1 : 0;
}
