// Self-test of the CPU lane emulator's primitives (tests/hostemu/hip/hip_runtime.h): each kernel writes what the HARDWARE
// semantics of the primitive prescribe; tests/test_hostemu_selftest.py checks the results.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[t] = value held by lane t ^ mask of the same wave; sums[block] = workgroup sum through LDS and two barriers
__global__ void shfl_barrier_kernel(int mask, float* out, float* sums) {
    __shared__ float part[4];
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const float mine = (float)(t * 3 + 1);
    out[t] = __shfl_xor(mine, mask);
    float v = mine;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    if (threadIdx.x == 255) part[0] = -1.f;   // a late write must not be seen by the read above
}
// D = A x B on one wave: A[i][k] = i + 0.25 k, B[k][j] = (k == j) -> D[i][j] = A[i][j] for j < 16 (k runs to 32)
__global__ void mfma_kernel(float* d) {
    const int l = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = (l / 16) * 8 + e;
        a[e] = (__bf16)((float)(l % 16) + 0.25f * (float)k);
        b[e] = (__bf16)(k == (l % 16) ? 1.0f : 0.0f);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[((l / 16) * 4 + r) * 16 + (l % 16)] = c[r];
}
// LDS-DMA: lane l's 16 bytes land at base + 16 l
__global__ void dma_kernel(const uint32_t* src, uint32_t* dst) {
    extern __shared__ uint32_t lds[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    __builtin_amdgcn_global_load_lds(src + ((w * 64 + (63 - l)) * 4), lds + w * 256, 16, 0, 0);   // lane l fetches chunk 63 - l
    __syncthreads();
    for (int i = 0; i < 4; ++i) dst[threadIdx.x * 4 + i] = lds[threadIdx.x * 4 + i];
}

int main() {
    int bad = 0;
    {
        float out[512], sums[2];
        hipLaunchKernelGGL(shfl_barrier_kernel, dim3(2), dim3(256), 0, 0, 5, out, sums);
        for (int t = 0; t < 512; ++t) bad += out[t] != (float)(((t & ~63) | ((t ^ 5) & 63)) * 3 + 1);
        for (int b = 0; b < 2; ++b) {
            float s = 0.f;
            for (int t = b * 256; t < (b + 1) * 256; ++t) s += (float)(t * 3 + 1);
            bad += sums[b] != s;
        }
    }
    {
        float d[256];
        hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, d);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) bad += d[i * 16 + j] != (float)i + 0.25f * (float)j;
    }
    {
        uint32_t src[512 * 4], dst[512 * 4];
        for (int i = 0; i < 512 * 4; ++i) src[i] = 0x1000u + i;
        hipLaunchKernelGGL(dma_kernel, dim3(1), dim3(512), 8192, 0, src, dst);
        for (int t = 0; t < 512; ++t)
            for (int i = 0; i < 4; ++i) bad += dst[t * 4 + i] != 0x1000u + (((t & ~63) + (63 - (t & 63))) * 4 + i);
    }
    printf(bad ? "SELFTEST FAILED: %d mismatches\n" : "SELFTEST OK\n", bad);
    return bad ? 1 : 0;
}
