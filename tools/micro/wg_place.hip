// Where does the dispatcher put consecutive workgroups?  Every workgroup records (XCC, SE, CU) and a timestamp at its start; the host prints the
// placement of the first workgroups and how many distinct CUs the ids i, i+1, i+2, i+3 / i, i+8, ... / i, i+256, ... share.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/wg_place.hip -o /tmp/wg_place && /tmp/wg_place [threads] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <set>

__global__ void place(unsigned* out, int spin) {
    extern __shared__ char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)(wall_clock64() & 0xffffffffu);
    }
    // stay resident for a while so that the first wave of workgroups fills the chip
    volatile char* s = smem;
    for (int i = 0; i < spin; ++i) s[threadIdx.x] = (char)i;
    if (threadIdx.x == 0) out[blockIdx.x * 4 + 3] = (unsigned)(__builtin_readcyclecounter() - t0);
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 256;
    const int lds = argc > 2 ? atoi(argv[2]) : 39456;
    const int n = 4096;
    unsigned* d;
    hipMalloc(&d, n * 16);
    hipFuncSetAttribute((const void*)place, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(place, dim3(n), dim3(threads), lds, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    auto cu_of = [&](int i) {
        const unsigned hw = h[i * 4], xcc = h[i * 4 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        return (int)(((xcc * 8 + se) * 2 + sh) * 16 + cu);
    };
    printf("first 24 workgroups: id -> xcc se sh cu simd\n");
    for (int i = 0; i < 24; ++i) {
        const unsigned hw = h[i * 4];
        printf("  %3d -> xcc %u se %u sh %u cu %2u simd %u\n", i, h[i * 4 + 1] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3);
    }
    std::map<int, std::vector<int>> by_cu;
    for (int i = 0; i < 1024; ++i) by_cu[cu_of(i)].push_back(i);
    printf("distinct CUs used by the first 1024 workgroups: %zu\n", by_cu.size());
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ >= 6) break;
        printf("  cu %4d:", kv.first);
        for (int i : kv.second) printf(" %d", i);
        printf("\n");
    }
    return 0;
}
