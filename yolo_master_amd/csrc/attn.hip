// Area attention core on MFMA: out = softmax(q^T k / sqrt(d)) v per (image, area, head).
// Reference: AAttn.forward (ultralytics/nn/modules/block.py:1696-1726): the two matmuls
// `attn = (q*scale)^T @ k`, `softmax(-1)`, `x = v @ attn^T` over N/area tokens, head_dim 32.
//
// Structure: workgroup = 64 queries (4 waves x 16) of one (image, area, head); keys/values
// are streamed in chunks of 256 tokens through LDS (K row-major, V transposed) with an
// online softmax across chunks.  S^T = K Q^T is computed with keys as MFMA rows, so the
// softmax reduction over keys is lane-local plus two cross-lane steps, and P feeds the
// second MFMA (O^T = V^T P^T) straight from registers (the k-slot permutation induced by the
// accumulator layout is applied identically to the V^T operand).
#include "igemm.h"

#define AT_KC 256
#define AT_NT 256   // threads per workgroup: 4 waves x 16 queries (512 measured ~5% slower on the A2C2f layers)

// WPE: waves per SIMD the register allocation is held to (left alone the compiler takes ~280 registers and one
// wave per SIMD; the kernel needs neighbours on the SIMD to overlap staging, softmax VALU and MFMA)
template <typename T, int WPE>
__global__ __launch_bounds__(AT_NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void area_attn_kernel(const T* __restrict__ qkv, int ldq, T* __restrict__ out,
                                                       int ldo, int N, int Na, int heads, int area, float scale) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NF = sizeof(T) == 2 ? 1 : 2;  // 16-byte fragments per 32-wide head row per lane
    constexpr int VPAD = AT_KC + VEC;
    constexpr bool PRECISE = sizeof(T) == 4;
    __shared__ __attribute__((aligned(16))) T sK[AT_KC * 32];
    __shared__ __attribute__((aligned(16))) T sVt[32 * VPAD];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fi = lane & 15, g = lane >> 4;
    const int h = blockIdx.y;
    const int b = blockIdx.z / area, ar = blockIdx.z % area;
    const int tok0 = ar * Na;  // first token of this area inside the image
    const int Cq = heads * 32;
    const T* base = qkv + (size_t)b * N * ldq;
    const int q0 = blockIdx.x * (AT_NT / 4) + wave * 16;
    const bool wave_on = q0 < Na;

    // query fragment(s): B operand, lane (query fi, k-group g)
    u32x4 qf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        qf[f] = u32x4{0u, 0u, 0u, 0u};
        if (wave_on && q0 + fi < Na)
            qf[f] = *reinterpret_cast<const u32x4*>(base + (size_t)(tok0 + q0 + fi) * ldq + h * 32 + f * 16 + g * VEC);
    }

    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float mrun = -INFINITY, lrun = 0.f;

    for (int c0 = 0; c0 < Na; c0 += AT_KC) {
        const int kc = min(AT_KC, Na - c0);
        const int kc32 = (kc + 31) & ~31;  // processed keys (zero padded)
        __syncthreads();                   // previous chunk fully consumed
        // stage K (row-major [key][32]) and V (transposed [d][key])
        constexpr int CPR = 32 / VEC;              // 16-byte chunks per row
        {
            constexpr int NL = AT_KC * CPR / AT_NT;   // staged chunks per thread (K and V each)
            u32x4 kreg[NL], vreg[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) {            // all loads first (independent, in flight together)
                const int i = t + l * AT_NT;
                const int key = i / CPR, ch = i % CPR;
                u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
                if (key < kc) {
                    const T* p = base + (size_t)(tok0 + c0 + key) * ldq + h * 32 + ch * VEC;
                    kv = *reinterpret_cast<const u32x4*>(p + Cq);
                    vv = *reinterpret_cast<const u32x4*>(p + 2 * Cq);
                }
                kreg[l] = kv; vreg[l] = vv;
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int i = t + l * AT_NT;
                const int key = i / CPR, ch = i % CPR;
                if (key < kc32) {
                    *reinterpret_cast<u32x4*>(&sK[key * 32 + ch * VEC]) = kreg[l];
                    const T* ve = reinterpret_cast<const T*>(&vreg[l]);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) sVt[(ch * VEC + q) * VPAD + key] = ve[q];
                }
            }
        }
        __syncthreads();
        if (!wave_on) continue;

        const int ntile = kc32 / 16;
        f32x4 sacc[AT_KC / 16];
        float cmax = -INFINITY;
        // bf16 path: scores stay unscaled; scale * log2(e) is folded into the one FMA that feeds v_exp_f32
        // (p = 2^((s - m) * c)), and the key-validity mask is applied only to tiles that reach past kc
        const float c2 = scale * 1.4426950408889634f;
#pragma unroll
        for (int tk = 0; tk < AT_KC / 16; ++tk) {
            if (tk < ntile) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(&sK[(tk * 16 + fi) * 32 + f * 16 + g * VEC]);
                    mma16<T>(acc, kf, qf[f]);
                }
                if (PRECISE) acc *= scale;
                if (tk * 16 + 16 > kc) {  // wave-uniform: only the zero-padded tail tiles
                    const int key0 = tk * 16 + g * 4;
                    acc.x = key0 + 0 < kc ? acc.x : -INFINITY;
                    acc.y = key0 + 1 < kc ? acc.y : -INFINITY;
                    acc.z = key0 + 2 < kc ? acc.z : -INFINITY;
                    acc.w = key0 + 3 < kc ? acc.w : -INFINITY;
                }
                cmax = fmaxf(cmax, fmaxf(fmaxf(acc.x, acc.y), fmaxf(acc.z, acc.w)));
                sacc[tk] = acc;
            }
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float mnew = fmaxf(mrun, cmax);  // finite: every chunk has >= 1 valid key
        const float resc = PRECISE ? expf(mrun - mnew) : __builtin_amdgcn_exp2f((mrun - mnew) * c2);
        const float nmc = -mnew * c2;
        mrun = mnew;
        lrun *= resc;
        o[0] *= resc;
        o[1] *= resc;
        float lsum = 0.f;
#pragma unroll
        for (int tk = 0; tk < AT_KC / 16; ++tk) {
            if (tk < ntile) {
                f32x4 p = sacc[tk];
                if (PRECISE) {
                    p.x = expf(p.x - mnew); p.y = expf(p.y - mnew); p.z = expf(p.z - mnew); p.w = expf(p.w - mnew);
                } else {
                    p.x = __builtin_amdgcn_exp2f(fmaf(p.x, c2, nmc)); p.y = __builtin_amdgcn_exp2f(fmaf(p.y, c2, nmc));
                    p.z = __builtin_amdgcn_exp2f(fmaf(p.z, c2, nmc)); p.w = __builtin_amdgcn_exp2f(fmaf(p.w, c2, nmc));
                }
                lsum += (p.x + p.y) + (p.z + p.w);
                sacc[tk] = p;
            }
        }
        lrun += lsum;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int u = 0; u < AT_KC / 32; ++u) {
                if (2 * u < ntile) {
                    u32x4 pb;
                    pb.x = pack_bf16x2(sacc[2 * u].x, sacc[2 * u].y);
                    pb.y = pack_bf16x2(sacc[2 * u].z, sacc[2 * u].w);
                    pb.z = pack_bf16x2(sacc[2 * u + 1].x, sacc[2 * u + 1].y);
                    pb.w = pack_bf16x2(sacc[2 * u + 1].z, sacc[2 * u + 1].w);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const T* vr = &sVt[(dt * 16 + fi) * VPAD + g * 4];
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(vr + (2 * u) * 16);
                        const u32x2 hi = *reinterpret_cast<const u32x2*>(vr + (2 * u + 1) * 16);
                        const u32x4 va = {lo.x, lo.y, hi.x, hi.y};
                        mma16<T>(o[dt], va, pb);
                    }
                }
            }
        } else {
#pragma unroll
            for (int tk = 0; tk < AT_KC / 16; ++tk) {
                if (tk < ntile) {
                    u32x4 pb;
                    pb.x = __float_as_uint(sacc[tk].x); pb.y = __float_as_uint(sacc[tk].y);
                    pb.z = __float_as_uint(sacc[tk].z); pb.w = __float_as_uint(sacc[tk].w);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const u32x4 va = *reinterpret_cast<const u32x4*>(&sVt[(dt * 16 + fi) * VPAD + tk * 16 + g * 4]);
                        mma16<T>(o[dt], va, pb);
                    }
                }
            }
        }
    }
    if (!wave_on) return;
    lrun += __shfl_xor(lrun, 16);
    lrun += __shfl_xor(lrun, 32);
    if (q0 + fi < Na) {
        const float inv = 1.0f / lrun;
        T* op = out + (size_t)(b * (size_t)N + tok0 + q0 + fi) * ldo + h * 32 + g * 4;
        store4(op, o[0].x * inv, o[0].y * inv, o[0].z * inv, o[0].w * inv);
        store4(op + 16, o[1].x * inv, o[1].y * inv, o[1].z * inv, o[1].w * inv);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Resident variant: ONE workgroup per (image, area, head).  All Na keys and values of the head (Na = 400 in the detector:
// 51 KB in bf16) are staged into LDS once, then the workgroup's four waves walk the area's 16-query tiles.  The streaming
// kernel above gives every 64-query workgroup its own copy of K / V (7 stagings per head at Na = 400, measured 4.7x the
// algorithmic HBM/L2 traffic and as many transposing LDS writes, profiles/r01_*); here qkv is read exactly once.  The
// arithmetic (score tiles, exp2-domain online softmax in 256-key chunks, P fed to the second MFMA from registers) is the
// streaming kernel's, so the two agree to rounding.  Used whenever K + V^T fit the LDS (Na <= 1024 bf16 / 512 fp32).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int WPE>
__global__ __launch_bounds__(AT_NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void area_attn_resident_kernel(
    const T* __restrict__ qkv, int ldq, T* __restrict__ out, int ldo, int N, int Na, int heads, int area, float scale) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NF = sizeof(T) == 2 ? 1 : 2;
    constexpr bool PRECISE = sizeof(T) == 4;
    constexpr int CPR = 32 / VEC;              // 16-byte chunks per 32-wide head row
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int Nk = (Na + 31) & ~31;            // keys incl. zero padding (the P V product walks 32 keys at a time)
    const int Nr = (Na + 15) & ~15;            // K rows kept: score tiles past them read into sVt and are masked to -inf
    const int VP = Nk + VEC;                   // V^T row pitch (elements); columns Na..Nk-1 are zeros (p = 0 there)
    T* sK = reinterpret_cast<T*>(at_smem);     // [Nr][32]
    T* sVt = sK + (size_t)Nr * 32;             // [32][VP]        (Na = 400, bf16: 25,600 + 27,136 bytes -> 3 workgroups per CU)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fi = lane & 15, g = lane >> 4;
    const int h = blockIdx.x % heads;
    const int ba = blockIdx.x / heads;
    const int b = ba / area, ar = ba % area;
    const int tok0 = ar * Na;
    const int Cq = heads * 32;
    const T* base = qkv + (size_t)b * N * ldq;

    // ---- stage every key / value of the head once: K row-major, V transposed ------------------------------------------------
    for (int i0 = 0; i0 < Nk * CPR; i0 += AT_NT * 4) {
        u32x4 kreg[4], vreg[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {          // four independent loads per thread in flight
            const int i = i0 + t + l * AT_NT;
            const int key = i / CPR, ch = i % CPR;
            u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (key < Na) {
                const T* p = base + (size_t)(tok0 + key) * ldq + h * 32 + ch * VEC;
                kv = *reinterpret_cast<const u32x4*>(p + Cq);
                vv = *reinterpret_cast<const u32x4*>(p + 2 * Cq);
            }
            kreg[l] = kv; vreg[l] = vv;
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int i = i0 + t + l * AT_NT;
            const int key = i / CPR, ch = i % CPR;
            if (key < Nk) {
                if (key < Nr) *reinterpret_cast<u32x4*>(&sK[key * 32 + ch * VEC]) = kreg[l];
                const T* ve = reinterpret_cast<const T*>(&vreg[l]);
#pragma unroll
                for (int q = 0; q < VEC; ++q) sVt[(ch * VEC + q) * VP + key] = ve[q];
            }
        }
    }
    __syncthreads();

    const float c2 = scale * 1.4426950408889634f;
    for (int q0 = wave * 16; q0 < Na; q0 += (AT_NT / 64) * 16) {
        u32x4 qf[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            qf[f] = u32x4{0u, 0u, 0u, 0u};
            if (q0 + fi < Na)
                qf[f] = *reinterpret_cast<const u32x4*>(base + (size_t)(tok0 + q0 + fi) * ldq + h * 32 + f * 16 + g * VEC);
        }
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        float mrun = -INFINITY, lrun = 0.f;
        for (int c0 = 0; c0 < Na; c0 += AT_KC) {
            const int kc = min(AT_KC, Na - c0);
            const int ntile = ((kc + 31) & ~31) / 16;
            f32x4 sacc[AT_KC / 16];
            float cmax = -INFINITY;
#pragma unroll
            for (int tk = 0; tk < AT_KC / 16; ++tk) {
                if (tk < ntile) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const u32x4 kf = *reinterpret_cast<const u32x4*>(&sK[(c0 + tk * 16 + fi) * 32 + f * 16 + g * VEC]);
                        mma16<T>(acc, kf, qf[f]);
                    }
                    if (PRECISE) acc *= scale;
                    if (tk * 16 + 16 > kc) {  // wave-uniform: only the zero-padded tail tiles
                        const int key0 = tk * 16 + g * 4;
                        acc.x = key0 + 0 < kc ? acc.x : -INFINITY;
                        acc.y = key0 + 1 < kc ? acc.y : -INFINITY;
                        acc.z = key0 + 2 < kc ? acc.z : -INFINITY;
                        acc.w = key0 + 3 < kc ? acc.w : -INFINITY;
                    }
                    cmax = fmaxf(cmax, fmaxf(fmaxf(acc.x, acc.y), fmaxf(acc.z, acc.w)));
                    sacc[tk] = acc;
                }
            }
            cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            const float mnew = fmaxf(mrun, cmax);
            const float resc = PRECISE ? expf(mrun - mnew) : __builtin_amdgcn_exp2f((mrun - mnew) * c2);
            const float nmc = -mnew * c2;
            mrun = mnew;
            lrun *= resc;
            o[0] *= resc;
            o[1] *= resc;
            float lsum = 0.f;
#pragma unroll
            for (int tk = 0; tk < AT_KC / 16; ++tk) {
                if (tk < ntile) {
                    f32x4 p = sacc[tk];
                    if (PRECISE) {
                        p.x = expf(p.x - mnew); p.y = expf(p.y - mnew); p.z = expf(p.z - mnew); p.w = expf(p.w - mnew);
                    } else {
                        p.x = __builtin_amdgcn_exp2f(fmaf(p.x, c2, nmc)); p.y = __builtin_amdgcn_exp2f(fmaf(p.y, c2, nmc));
                        p.z = __builtin_amdgcn_exp2f(fmaf(p.z, c2, nmc)); p.w = __builtin_amdgcn_exp2f(fmaf(p.w, c2, nmc));
                    }
                    lsum += (p.x + p.y) + (p.z + p.w);
                    sacc[tk] = p;
                }
            }
            lrun += lsum;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int u = 0; u < AT_KC / 32; ++u) {
                    if (2 * u < ntile) {
                        u32x4 pb;
                        pb.x = pack_bf16x2(sacc[2 * u].x, sacc[2 * u].y);
                        pb.y = pack_bf16x2(sacc[2 * u].z, sacc[2 * u].w);
                        pb.z = pack_bf16x2(sacc[2 * u + 1].x, sacc[2 * u + 1].y);
                        pb.w = pack_bf16x2(sacc[2 * u + 1].z, sacc[2 * u + 1].w);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const T* vr = &sVt[(dt * 16 + fi) * VP + c0 + g * 4];
                            const u32x2 lo = *reinterpret_cast<const u32x2*>(vr + (2 * u) * 16);
                            const u32x2 hi = *reinterpret_cast<const u32x2*>(vr + (2 * u + 1) * 16);
                            const u32x4 va = {lo.x, lo.y, hi.x, hi.y};
                            mma16<T>(o[dt], va, pb);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int tk = 0; tk < AT_KC / 16; ++tk) {
                    if (tk < ntile) {
                        u32x4 pb;
                        pb.x = __float_as_uint(sacc[tk].x); pb.y = __float_as_uint(sacc[tk].y);
                        pb.z = __float_as_uint(sacc[tk].z); pb.w = __float_as_uint(sacc[tk].w);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const u32x4 va = *reinterpret_cast<const u32x4*>(&sVt[(dt * 16 + fi) * VP + c0 + tk * 16 + g * 4]);
                            mma16<T>(o[dt], va, pb);
                        }
                    }
                }
            }
        }
        lrun += __shfl_xor(lrun, 16);
        lrun += __shfl_xor(lrun, 32);
        if (q0 + fi < Na) {
            const float inv = 1.0f / lrun;
            T* op = out + (size_t)(b * (size_t)N + tok0 + q0 + fi) * ldo + h * 32 + g * 4;
            store4(op, o[0].x * inv, o[0].y * inv, o[0].z * inv, o[0].w * inv);
            store4(op + 16, o[1].x * inv, o[1].y * inv, o[1].z * inv, o[1].w * inv);
        }
    }
}

template <typename T, int WPE>
static int launch_attn_resident(const T* qkv, int ldq, T* out, int ldo, int B, int N, int Na, int heads, int area, float scale,
                                hipStream_t s) {
    const int Nk = (Na + 31) & ~31, Nr = (Na + 15) & ~15;
    const size_t shm = ((size_t)Nr * 32 + (size_t)32 * (Nk + 16 / sizeof(T))) * sizeof(T);
    static bool attr_set = false;
    if (shm > 64 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&area_attn_resident_kernel<T, WPE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((area_attn_resident_kernel<T, WPE>), dim3((unsigned)((size_t)B * area * heads)), dim3(AT_NT), shm, s, qkv, ldq,
                       out, ldo, N, Na, heads, area, scale);
    return ymk_launch_status();
}

#define YMK_OFF_ATTN_RESIDENT 256u   // YMK_DISABLE bit: resident K/V attention -> streaming kernel (A/B runs)

extern "C" int ymk_area_attn(int32_t dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo, int32_t B,
                             int32_t N, int32_t heads, int32_t area, void* stream) {
    if (!qkv || !out || heads < 1 || area < 1 || N % area) return YMK_E_BADARG;
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (ldq % vec || ldo % 4) return YMK_E_BADARG;
    if (B <= 0 || N <= 0) return YMK_OK;
    const int Na = N / area;
    if ((int64_t)B * area > 65535 || heads > 65535) return YMK_E_BADARG;
    dim3 grid((Na + AT_NT / 4 - 1) / (AT_NT / 4), heads, B * area), blk(AT_NT);
    const float scale = 0.17677669529663687f;  // 32^-0.5
    hipStream_t s = (hipStream_t)stream;
    if (!(ymk_disabled() & YMK_OFF_ATTN_RESIDENT) && (int64_t)B * area * heads < (1ll << 31)) {
        if (dtype == YMK_BF16 && Na <= 1024)
            return launch_attn_resident<bf16_t, 3>((const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, B, N, Na, heads, area, scale, s);
        if (dtype == YMK_F32 && Na <= 512)
            return launch_attn_resident<float, 1>((const float*)qkv, ldq, (float*)out, ldo, B, N, Na, heads, area, scale, s);
    }
    if (dtype == YMK_F32)
        hipLaunchKernelGGL((area_attn_kernel<float, 1>), grid, blk, 0, s, (const float*)qkv, ldq, (float*)out, ldo, N, Na,
                           heads, area, scale);
    else if (dtype == YMK_BF16)
        hipLaunchKernelGGL((area_attn_kernel<bf16_t, 3>), grid, blk, 0, s, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, Na, heads, area,
                           scale);   // 3 waves per SIMD measured best (4: small spill, 1: 2x slower)
    else
        return YMK_E_BADARG;
    return ymk_launch_status();
}
