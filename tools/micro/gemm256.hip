// Design probe for the next tiled-GEMM core: 256-pixel x 128-cout tiles on 8 waves, operands
// staged with global_load_lds (16 bytes per lane, source-side XOR swizzle, lane-linear LDS image), two LDS stages,
// one barrier per k-step.  Standalone: builds and runs on the GPU box, checks itself against a naive kernel and
// prints TFLOP/s next to the library's tiled 1x1 kernel on the same problem.  NOT part of libymk.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I yolo_master_amd/csrc tools/micro/gemm256.hip -o tools/micro/gemm256.bin
//   tools/micro/gemm256.bin
//
// Problem: Y[M][N] = SiLU(X[M][K] . W[N][K]^T + b), bf16 operands, fp32 accumulation (a 1x1 convolution in NHWC).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../yolo_master_amd/csrc/conv.hip"   // the library's kernels for the side-by-side number (mma16, helpers)

#define G_BM 256   // pixels per tile
#define G_BN 128   // couts per tile
#define G_ROWS (G_BN + G_BM)          // staged rows per k-step (weights first, then pixels)
#define G_STAGE_U4 (G_ROWS * 8)       // u32x4 per stage: rows x 8 chunks of 16 bytes (BK = 64 bf16)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(512) void gemm256_glds_kernel(const h16_t* __restrict__ X, const h16_t* __restrict__ Wt,
                                                            const float* __restrict__ bias, h16_t* __restrict__ Y, int M,
                                                            int N, int K) {
    __shared__ u32x4 smem[2 * G_STAGE_U4];   // 96 KB
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fc = lane >> 4;
    const int nt = N / G_BN;
    const int m0 = (blockIdx.x / nt) * G_BM, n0 = (blockIdx.x % nt) * G_BN;
    const int nk = K / 64;
    const int wm = wave >> 1, wn = wave & 1;   // 4 (pixels) x 2 (couts) waves, 64 x 64 each

    // staging map: k-step = 48 wave-instructions of 8 rows; wave w issues row blocks rb = j*8 + w, j = 0..5
    const int lr = lane >> 3, lc = lane & 7;
    const h16_t* src[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int r = (j * 8 + wave) * 8 + lr;                 // staged row 0..383
        const int sc = (lc ^ (r & 7)) * 8;                      // source chunk: XOR swizzle on the GLOBAL side
        src[j] = r < G_BN ? Wt + (size_t)(n0 + r) * K + sc : X + (size_t)(m0 + r - G_BN) * K + sc;
    }
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            u32x4* dst = smem + stage * G_STAGE_U4 + (j * 8 + wave) * 64;   // wave-uniform; the lane lands at +lane*16 B
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kt * 64), (lptr_t)dst, 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();   // hipcc drains the DMA (vmcnt(0)) before the barrier: stage kt&1 complete, stage (kt+1)&1 free
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const u32x4* sW = smem + (kt & 1) * G_STAGE_U4;
        const u32x4* sX = sW + G_BN * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (wn * 4 + i) * 16 + fr;
                af[i] = sW[r * 8 + ((kk * 4 + fc) ^ (r & 7))];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (wm * 4 + j) * 16 + fr;   // rows of sX start at staged row 128: (128 + r) & 7 == r & 7
                bfr[j] = sX[r * 8 + ((kk * 4 + fc) ^ (r & 7))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<h16_t>(acc[i][j], af[i], bfr[j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = n0 + (wn * 4 + i) * 16 + fc * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + (wm * 4 + j) * 16 + fr;
            store4(Y + (size_t)m * N + co, silu_f(acc[i][j].x + bv.x), silu_f(acc[i][j].y + bv.y), silu_f(acc[i][j].z + bv.z),
                   silu_f(acc[i][j].w + bv.w));
        }
    }
}

__global__ void naive_kernel(const h16_t* X, const h16_t* Wt, const float* bias, float* Yr, int N, int K, const int* rows,
                             int nrows) {
    const int i = blockIdx.x, n = threadIdx.x + blockIdx.y * blockDim.x;
    if (i >= nrows || n >= N) return;
    const size_t m = rows[i];
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += h16_to_f32(X[m * K + k]) * h16_to_f32(Wt[(size_t)n * K + k]);
    s += bias[n];
    Yr[(size_t)i * N + n] = s / (1.0f + expf(-s));
}

template <typename F>
static float timeit(F&& f, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < 3; ++i) f();
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    struct Shape { int M, N, K; } shapes[] = {{409600, 128, 128}, {102400, 256, 384}, {409600, 128, 1152}, {102400, 256, 2304},
                                              {25600, 512, 768}, {1638400, 128, 128}};
    const int nshape = argc > 1 ? atoi(argv[1]) : 6;   // `gemm256.bin 2` = the two quick shapes
    for (int si = 0; si < nshape && si < 6; ++si) {
        const Shape sh = shapes[si];
        const size_t nx = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, ny = (size_t)sh.M * sh.N;
        std::vector<h16_t> hx(nx), hw(nw);
        std::vector<float> hb(sh.N);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hx) { float f = rnd(); v = (h16_t)(__builtin_bit_cast(uint32_t, f) >> 16); }
        const float wscale = 2.0f / sqrtf((float)sh.K);
        for (auto& v : hw) { float f = rnd() * wscale; v = (h16_t)(__builtin_bit_cast(uint32_t, f) >> 16); }
        for (auto& v : hb) v = rnd() * 0.2f;
        h16_t *x, *w, *y, *y2; float *b, *yr; int* rows;
        hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2); hipMalloc(&y2, ny * 2); hipMalloc(&b, sh.N * 4);
        hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemcpy(b, hb.data(), sh.N * 4, hipMemcpyHostToDevice);
        hipMemset(y, 0xff, ny * 2);
        const int nrows = 512;
        std::vector<int> hr(nrows);
        for (int i = 0; i < nrows; ++i) hr[i] = (int)(((size_t)i * 7919u * 131u + (i % 3) * (sh.M - 1) / 2) % sh.M);
        hr[0] = 0; hr[1] = sh.M - 1;
        hipMalloc(&rows, nrows * 4); hipMemcpy(rows, hr.data(), nrows * 4, hipMemcpyHostToDevice);
        hipMalloc(&yr, (size_t)nrows * sh.N * 4);
        const dim3 grid((sh.M / G_BM) * (sh.N / G_BN));
        hipLaunchKernelGGL(gemm256_glds_kernel, grid, dim3(512), 0, 0, x, w, b, y, sh.M, sh.N, sh.K);
        hipLaunchKernelGGL(naive_kernel, dim3(nrows, (sh.N + 127) / 128), dim3(128), 0, 0, x, w, b, yr, sh.N, sh.K, rows, nrows);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<h16_t> hy(ny);
        std::vector<float> hyr((size_t)nrows * sh.N);
        hipMemcpy(hy.data(), y, ny * 2, hipMemcpyDeviceToHost);
        hipMemcpy(hyr.data(), yr, hyr.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < nrows; ++i)
            for (int n = 0; n < sh.N; ++n) {
                const float got = __builtin_bit_cast(float, (uint32_t)hy[(size_t)hr[i] * sh.N + n] << 16);
                const float ref = hyr[(size_t)i * sh.N + n];
                maxerr = std::max(maxerr, (double)fabsf(got - ref));
                maxref = std::max(maxref, (double)fabsf(ref));
            }
        const double flops = 2.0 * sh.M * sh.N * sh.K, bytes = 2.0 * (nx + nw + ny);
        const float ms = timeit([&] { hipLaunchKernelGGL(gemm256_glds_kernel, grid, dim3(512), 0, 0, x, w, b, y, sh.M, sh.N, sh.K); });
        // the library's tiled 1x1 kernel on the same problem (B x H x W = 1 x 1 x M pixels)
        ymk_conv_desc d{YMK_BF16, YMK_BF16, 1, 1, sh.M, sh.K, sh.N, 1, 1, sh.K, sh.N, 0, sh.K, YMK_ACT_SILU};
        ymk_use_ws = 0;
        const float ms_lib = timeit([&] { ymk_conv2d(&d, x, w, b, nullptr, y2, nullptr); });
        ymk_use_ws = 1;
        const float ms_ws = timeit([&] { ymk_conv2d(&d, x, w, b, nullptr, y2, nullptr); });
        printf("M %7d N %4d K %5d | glds256 %8.1f us %7.1f TF/s %5.2f TB/s | lib tiled %8.1f us %7.1f TF/s | lib default %8.1f us | "
               "max |err| %.3e (max |ref| %.2f) %s\n",
               sh.M, sh.N, sh.K, ms * 1e3, flops / ms / 1e9, bytes / ms / 1e9, ms_lib * 1e3, flops / ms_lib / 1e9, ms_ws * 1e3, maxerr,
               maxref, maxerr <= 0.02 * std::max(1.0, maxref) ? "OK" : "MISMATCH");
        hipFree(x); hipFree(w); hipFree(y); hipFree(y2); hipFree(b); hipFree(yr); hipFree(rows);
    }
    return 0;
}
