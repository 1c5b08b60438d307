// VALU issue-rate probe: v_fma_f32, v_pk_fma_f32, v_dot2c_f32_bf16, the transcendentals and a whole SiLU — lane-results per second with
// every SIMD saturated (8 workgroups of 4 waves per CU, 8 independent chains per lane).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate.bin && tools/micro/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned seed) {
    float a[8];
    f2 p[8];
    unsigned u = seed + threadIdx.x;
    const bf2 x = __builtin_bit_cast(bf2, u * 2654435761u | 0x3f803f80u), y = __builtin_bit_cast(bf2, 0x3f003f00u);
    const float fx = 1.0001f, fy = 0.5f;
    const f2 px = {1.0001f, 0.9999f}, py = {0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (float)i; p[i] = f2{(float)i, 1.f}; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], fx, fy);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], px, py);
            if (MODE == 2) a[i] = __builtin_amdgcn_fdot2_f32_bf16(x, y, a[i], false);
            if (MODE == 3) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (MODE == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (MODE == 5) a[i] = a[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a[i] * -1.4426950408889634f));   // silu_f
            if (MODE == 6) a[i] = a[i] + fy;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, int macs_per_instr) {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;   // 8 workgroups of 4 waves per CU
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)grid * 4 /*waves*/ * ITERS * 8;
    const double lane_macs = instr * 64 * macs_per_instr;
    printf("%-20s %8.3f ms  %6.2f T MAC/s  (%.1f wave-instr / us / CU)\n", name, ms, lane_macs / ms * 1e-9, instr / 256 / (ms * 1e3));
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 2);
    run<2>("v_dot2c_f32_bf16", 2);
    run<3>("v_exp_f32", 1);
    run<4>("v_rcp_f32", 1);
    run<5>("silu_f (5 ops)", 1);
    run<6>("v_add_f32", 1);
    return 0;
}
