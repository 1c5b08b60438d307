// Standalone ablation harness for the igemm conv kernel (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I yolo_master_amd/csrc tools/micro/convbench.hip -o /tmp/convbench && /tmp/convbench
// Times (HIP events, median of 20) a 1x1 conv M x Cin -> Cout in bf16 plus a same-bytes copy kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define YMK_ABLATE 1
#include "../../yolo_master_amd/csrc/conv.hip"

__global__ void copy_kernel(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_kernel(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) b[0] = acc;
}

template <typename F>
static float timeit(F&& f, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    for (int i = 0; i < 3; ++i) f();
    for (int i = 0; i < reps; ++i) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    struct Shape { int B, H, W, Cin, Cout, k, s; } shapes[] = {
        {64, 160, 160, 128, 128, 3, 2}, {64, 80, 80, 256, 256, 3, 2}, {64, 320, 320, 32, 64, 3, 2}, {64, 80, 80, 128, 64, 3, 1},
        {64, 40, 40, 384, 256, 1, 1}, {64, 20, 20, 256, 768, 1, 1}, {64, 40, 40, 256, 128, 1, 1}};
    ymk_ws_min_tiles = 1;
    for (auto sh : shapes) {
        const size_t nin = (size_t)sh.B * sh.H * sh.W * sh.Cin;
        const int Ho = (sh.H + 2 * (sh.k / 2) - sh.k) / sh.s + 1, Wo = (sh.W + 2 * (sh.k / 2) - sh.k) / sh.s + 1;
        const size_t nout = (size_t)sh.B * Ho * Wo * sh.Cout;
        const int K = sh.k * sh.k * sh.Cin, Kpad = (K + 63) / 64 * 64;
        h16_t *x, *w, *y; float* bias;
        hipMalloc(&x, nin * 2); hipMalloc(&y, nout * 2); hipMalloc(&w, (size_t)sh.Cout * Kpad * 2); hipMalloc(&bias, sh.Cout * 4);
        hipMemset(x, 0x3c, nin * 2); hipMemset(w, 0x3c, (size_t)sh.Cout * Kpad * 2); hipMemset(bias, 0, sh.Cout * 4);
        ymk_conv_desc d{YMK_BF16, YMK_BF16, sh.B, sh.H, sh.W, sh.Cin, sh.Cout, sh.k, sh.s, sh.Cin, sh.Cout, 0, Kpad, YMK_ACT_SILU};
        const double bytes = (double)(nin + nout + (size_t)sh.Cout * K) * 2, flops = 2.0 * sh.B * Ho * Wo * sh.Cout * K;
        ymk_use_ws = 0;
        float ms0 = timeit([&] { ymk_conv2d(&d, x, w, bias, nullptr, y, nullptr); });
        ymk_use_ws = 1;
        float ms = timeit([&] { ymk_conv2d(&d, x, w, bias, nullptr, y, nullptr); });
        printf("[tiled %.1f us -> streaming %.1f us] ", ms0 * 1e3, ms * 1e3);
        const size_t n16 = std::min(nin, nout) * 2 / 16;
        float mc = timeit([&] { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)y, n16); });
        float mr = timeit([&] { hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)y, nin * 2 / 16); });
        printf("conv B%d %dx%d %d->%d k%d s%d: %.1f us  %.2f TB/s alg  %.0f TF/s | copy(%zu MB x2) %.1f us %.2f TB/s | read(%zu MB) %.1f us %.2f TB/s\n",
               sh.B, sh.H, sh.W, sh.Cin, sh.Cout, sh.k, sh.s, ms * 1e3, bytes / ms / 1e9, flops / ms / 1e9,
               n16 * 16 >> 20, mc * 1e3, 2.0 * n16 * 16 / mc / 1e9, nin * 2 >> 20, mr * 1e3, nin * 2.0 / mr / 1e9);
        ymk_use_ws = 0;
        for (int ab = 1; ab <= 4; ++ab) {
            ymk_ablate = ab;
            float m2 = timeit([&] { ymk_conv2d(&d, x, w, bias, nullptr, y, nullptr); });
            printf("     ablate %d (%s): %.1f us\n", ab, ab == 1 ? "no MFMA" : ab == 2 ? "no global store" : ab == 3 ? "no activation loads (zeros)" : "no weight loads", m2 * 1e3);
        }
        ymk_ablate = 0;
        hipFree(x); hipFree(y); hipFree(w); hipFree(bias);
    }
    return 0;
}
