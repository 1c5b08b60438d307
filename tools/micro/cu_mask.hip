// Which CUs does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) use on MI355X?  For a few mask patterns: workgroups per XCC and the number
// of distinct CUs a 4096-workgroup launch on that stream touched.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/cu_mask.hip -o tools/micro/cu_mask.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>

__global__ void place(unsigned* out, int spin) {
    extern __shared__ char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    volatile char* s = smem;
    for (int i = 0; i < spin; ++i) s[threadIdx.x] = (char)i;
}

static void run(const char* what, const std::vector<unsigned>& mask) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", what); return; }
    const int n = 4096;
    unsigned* d;
    hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(place, dim3(n), dim3(256), 32768, st, d, 4000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(n * 2);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    int per_xcc[16] = {0};
    std::set<unsigned> cus;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[i * 2], xcc = h[i * 2 + 1] & 0xf;
        ++per_xcc[xcc];
        cus.insert((xcc << 16) | (hw & 0xff00u) | ((hw >> 13) & 7) << 20);
    }
    printf("%-46s distinct CUs %3zu  workgroups per XCC:", what, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("\n");
    hipFree(d);
    hipStreamDestroy(st);
}

int main() {
    std::vector<unsigned> all(8, 0xffffffffu);
    run("all 256 bits", all);
    std::vector<unsigned> lo(8, 0u), hi(8, 0u), xlo(8, 0u), xhi(8, 0u), x01(8, 0u);
    for (int i = 0; i < 256; ++i) {
        (i < 128 ? lo : hi)[i / 32] |= 1u << (i % 32);
        ((i % 8) < 4 ? xlo : xhi)[i / 32] |= 1u << (i % 32);
        if ((i % 8) < 2) x01[i / 32] |= 1u << (i % 32);
    }
    run("bits 0..127", lo);
    run("bits 128..255", hi);
    run("bits with (i % 8) < 4", xlo);
    run("bits with (i % 8) >= 4", xhi);
    run("bits with (i % 8) < 2", x01);
    return 0;
}
