// LDS-fill ceiling of one MI355X: persistent 512-thread workgroups stream a buffer into LDS with `buffer_load_dwordx4 ... lds`
// (16 B per lane, 1 KiB per wave-instruction, the staging instruction of csrc/conv_glds.hip), `depth` 32-KiB steps in flight per
// workgroup, no consumers, no barriers: what the global -> LDS path delivers per CU and chip-wide when the source is L2-resident
// (4 MiB), MALL-resident (96 MiB) or HBM (2 GiB), at one and two workgroups per CU.  The number conv_glds's fill traffic is priced against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/fillrate tools/micro/fillrate.hip && tools/micro/fillrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lptr;
#define WAITCNT_VM(n) (0x0F70 | ((n) & 15) | ((((n) >> 4) & 3) << 14))

// each workgroup: `steps` steps of 32 KiB (8 waves x 4 instructions x 1 KiB); step s of workgroup g reads 32 KiB at
// ((g * steps + s) * 32 KiB) mod bytes: disjoint streaming when bytes is large, a re-read working set when small
template <int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const char* src, unsigned bytes, int steps, unsigned* sink) {
    extern __shared__ char smem[];   // DEPTH * 32 KiB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes, 0x00020000);
    const unsigned mask = bytes - 1;   // power of two
    unsigned base = ((unsigned)blockIdx.x * (unsigned)steps) * 32768u;
    auto issue = [&](int s) {
        const unsigned off = (base + (unsigned)s * 32768u + (unsigned)wave * 4096u) & mask;
        char* dst = smem + (s % DEPTH) * 32768 + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr)(dst + j * 1024), 16, (int)(lane * 16), (int)(off + j * 1024), 0, 0);
    };
#pragma unroll
    for (int p = 0; p < DEPTH - 1; ++p) issue(p);
    for (int s = 0; s < steps; ++s) {
        if (s + DEPTH - 1 < steps) issue(s + DEPTH - 1);
        // wait until step s has landed: DEPTH - 1 younger steps (4 instructions each) may still travel
        if (DEPTH == 1) __builtin_amdgcn_s_waitcnt(WAITCNT_VM(0));
        else if (DEPTH == 2) __builtin_amdgcn_s_waitcnt(WAITCNT_VM(4));
        else if (DEPTH == 3) __builtin_amdgcn_s_waitcnt(WAITCNT_VM(8));
        else __builtin_amdgcn_s_waitcnt(WAITCNT_VM(12));
    }
    __builtin_amdgcn_s_waitcnt(WAITCNT_VM(0));
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem);
}

// the same stream into registers (global_load_dwordx4), for comparison
__global__ __launch_bounds__(512) void reg_kernel(const uint4* src, unsigned bytes, int steps, unsigned* sink) {
    const unsigned mask = bytes / 16 - 1;
    unsigned idx = ((unsigned)blockIdx.x * (unsigned)steps) * 2048u + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = src[(idx + j * 512) & mask];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
        idx += 2048u;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x] = 1;
}

template <int DEPTH>
static float run(const char* src, unsigned bytes, int wgs, int steps, unsigned* sink) {
    hipFuncSetAttribute((const void*)fill_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(wgs), dim3(512), DEPTH * 32768, 0, src, bytes, steps, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(wgs), dim3(512), DEPTH * 32768, 0, src, bytes, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    char* buf;
    unsigned* sink;
    const size_t big = 1ull << 31;
    if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, big);
    // warm the clocks
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(fill_kernel<2>, dim3(512), dim3(512), 65536, 0, buf, 1u << 22, 64, sink);
    hipDeviceSynchronize();
    const unsigned sizes[3] = {1u << 22, 3u << 25, 1u << 31};   // 4 MiB, 96 MiB, 2 GiB
    const char* names[3] = {"4 MiB (L2)", "64 MiB (MALL)", "2 GiB (HBM)"};
    printf("%-16s %6s %6s %10s %12s %12s\n", "source", "WGs", "depth", "us", "TB/s chip", "GB/s per CU");
    for (int si = 0; si < 3; ++si) {
        const unsigned bytes = si == 1 ? (1u << 26) : sizes[si];
        for (int wgs : {256, 512}) {
            const int steps = 512;
            for (int depth = 1; depth <= 4; ++depth) {
                if (wgs == 512 && depth > 2) continue;   // two workgroups per CU: 2 x 64 KiB of LDS
                float ms = depth == 1 ? run<1>(buf, bytes, wgs, steps, sink) : depth == 2 ? run<2>(buf, bytes, wgs, steps, sink)
                         : depth == 3 ? run<3>(buf, bytes, wgs, steps, sink) : run<4>(buf, bytes, wgs, steps, sink);
                const double total = (double)wgs * steps * 32768.0;
                printf("%-16s %6d %6d %10.1f %12.2f %12.1f\n", names[si], wgs, depth, ms * 1e3, total / ms / 1e9, total / ms / 1e6 / 256);
            }
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int wgs : {256, 1024}) {
            hipLaunchKernelGGL(reg_kernel, dim3(wgs), dim3(512), 0, 0, (const uint4*)buf, bytes, 512 * 256 / wgs, sink);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(reg_kernel, dim3(wgs), dim3(512), 0, 0, (const uint4*)buf, bytes, 512 * 256 / wgs, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 10;
            const double total = 256.0 * 512 * 32768.0;
            printf("%-16s %6d %6s %10.1f %12.2f %12.1f   (global_load_dwordx4 to registers)\n", names[si], wgs, "-", ms * 1e3, total / ms / 1e9, total / ms / 1e6 / 256);
        }
    }
    return 0;
}
