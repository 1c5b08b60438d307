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

#define YMK_OFF_ATTN_RESIDENT 256u   // YMK_DISABLE bit: resident K/V attention -> streaming kernel (A/B runs)
#define YMK_OFF_ATTN_QKV 2097152u   // YMK_DISABLE bit: qkv projection inside the attention kernel -> 1x1 convolution + ymk_area_attn
#define YMK_OFF_ATTN_WIDE 524288u    // YMK_DISABLE bit: long areas (> 1024 keys) on the 256-query kernel of csrc/mixattn.hip -> 64-query streaming kernel
#define AT_KC 256   // streaming kernel: keys per staged chunk
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
                    pb.x = pack_h16x2(sacc[2 * u].x, sacc[2 * u].y);
                    pb.y = pack_h16x2(sacc[2 * u].z, sacc[2 * u].w);
                    pb.z = pack_h16x2(sacc[2 * u + 1].x, sacc[2 * u + 1].y);
                    pb.w = pack_h16x2(sacc[2 * u + 1].z, sacc[2 * u + 1].w);
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
// max over the lanes that share (lane & 15): the two cross-row steps as register swaps (v_permlane16/32_swap: no LDS round trip)
__device__ __forceinline__ float at_rowgroup_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float at_rowgroup_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// One chunk of NP pairs of 16-key tiles (32 NP keys from c0) against a wave's 16 queries: score tiles, online-softmax update,
// P V.  Every loop bound is a compile-time constant, so the whole chunk is ONE basic block: the per-tile `if (tk < ntile)`
// guards this replaces put a branch, an LDS round trip and an MFMA latency in series 25 times per query tile (113 branches in
// the kernel; a SIMD with three waves was busy a third of the time).  `valid` < 32 NP only in the last chunk: keys past it
// (zero padding / the rows after K) get -inf in the last pair, which is where a ragged end falls by construction.
#ifndef AT_QKV_ABLATE
#define AT_QKV_ABLATE 0   // tools/micro stage ablation of the projection phase (bits): 1 no x loads, 2 one weight load for all passes, 4 no q / v stores, 16 no attention phase
#endif
#ifndef AT_QKV_WLDS
#define AT_QKV_WLDS 1   // the K / V weight rows of area_attn_qkv_kernel staged through LDS at kernel start (0: fetched per pass by every wave)
#endif
#define AT_QKV_MAXT 7   // token tiles per wave in the projection phase of area_attn_qkv_kernel (all resident in registers): Na <= 16 * 4 * 7
#ifndef AT_NQ2_NP
#define AT_NQ2_NP 3   // key-tile pairs per chunk when two query tiles walk together (4: 61 spilled registers at three waves per SIMD, 3: 18)
#endif
#ifndef AT_SETPRIO
#define AT_SETPRIO 1   // the two MFMA clusters of a chunk at raised wave priority (cdna_hip_programming.md T5): 63.8 -> 62.5 / 37.7 -> 36.9 / 84.0 -> 82.7 us, three interleaved runs
#endif
#if AT_SETPRIO && !defined(YMK_HOST_EMU)
#define AT_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define AT_PRIO(p) ((void)0)
#endif
#ifndef AT_ABLATE
#define AT_ABLATE 0   // tools/micro stage ablation (bits): 1 K fragments not read, 2 V fragments not read, 4 no exp2, 8 no score MFMAs, 16 no P V MFMAs
#endif
// NQ query tiles of a wave walk the chunk TOGETHER (NQ = 2: the K / V^T fragments are read from LDS once for 32 queries — the fragment reads
// are 51 KB per 16-query tile, as many LDS cycles per CU as the softmax arithmetic costs VALU cycles).
template <typename T, int NP, int NQ>
__device__ __forceinline__ void at_chunk(const T* __restrict__ sK, const T* __restrict__ sVt, int VP, int c0, int valid,
                                         const u32x4 (&qf)[NQ][sizeof(T) == 2 ? 1 : 2], f32x4 (&o)[NQ][2], float (&mrun)[NQ], float (&lrun)[NQ],
                                         float scale, int fi, int g) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NF = sizeof(T) == 2 ? 1 : 2;
    constexpr bool PRECISE = sizeof(T) == 4;
    constexpr int NTILE = 2 * NP;
    const float c2 = scale * 1.4426950408889634f;
    f32x4 sacc[NQ][NTILE];
    // all K fragments of the chunk are requested before the first MFMA, and all V fragments before the softmax arithmetic: left to
    // itself the compiler waited for every LDS read right before its use (27 waits per chunk, ~100 cycles each, in series)
    u32x4 kf[NTILE][NF];
#pragma unroll
    for (int tk = 0; tk < NTILE; ++tk)
#pragma unroll
        for (int f = 0; f < NF; ++f) {   // 16-bit: the chunk's swizzled slot depends on (fi >> 2) only (c0 and tk * 16 are multiples of 16)
            if (AT_ABLATE & 1) kf[tk][f] = u32x4{(unsigned)tk, (unsigned)fi, 0u, 0u};
            else kf[tk][f] = *reinterpret_cast<const u32x4*>(&sK[(c0 + tk * 16 + fi) * 32 + f * 16 + (sizeof(T) == 2 ? (g ^ ((0 - (fi >> 2)) & 3)) : g) * VEC]);
        }
    AT_PRIO(1);
#pragma unroll
    for (int tk = 0; tk < NTILE; ++tk)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (AT_ABLATE & 8) acc = f32x4{__uint_as_float(kf[tk][0].x & 0x3fffffffu), __uint_as_float(qf[q][0].y & 0x3fffffffu), 0.f, 1.f};
            else {
#pragma unroll
                for (int f = 0; f < NF; ++f) mma16<T>(acc, kf[tk][f], qf[q][f]);
            }
            if (PRECISE) acc *= scale;
            sacc[q][tk] = acc;
        }
    AT_PRIO(0);
    u32x4 va[sizeof(T) == 2 ? NP : 1][2];
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (AT_ABLATE & 2) { va[u][dt] = u32x4{(unsigned)u, (unsigned)g, 0u, 0u}; continue; }
                const T* vr = &sVt[(dt * 16 + fi) * VP + c0 + g * 4];
                const u32x2 lo = *reinterpret_cast<const u32x2*>(vr + (2 * u) * 16);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(vr + (2 * u + 1) * 16);
                va[u][dt] = u32x4{lo.x, lo.y, hi.x, hi.y};
            }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (valid < 32 * NP) {   // wave-uniform
#pragma unroll
            for (int tk = NTILE - 2; tk < NTILE; ++tk) {
                const int key0 = tk * 16 + g * 4;
                sacc[q][tk].x = key0 + 0 < valid ? sacc[q][tk].x : -INFINITY;
                sacc[q][tk].y = key0 + 1 < valid ? sacc[q][tk].y : -INFINITY;
                sacc[q][tk].z = key0 + 2 < valid ? sacc[q][tk].z : -INFINITY;
                sacc[q][tk].w = key0 + 3 < valid ? sacc[q][tk].w : -INFINITY;
            }
        }
        float cmax = -INFINITY;
#pragma unroll
        for (int tk = 0; tk < NTILE; ++tk) cmax = fmaxf(cmax, fmaxf(fmaxf(sacc[q][tk].x, sacc[q][tk].y), fmaxf(sacc[q][tk].z, sacc[q][tk].w)));
        cmax = at_rowgroup_max(cmax);
        const float mnew = fmaxf(mrun[q], cmax);
        const float resc = PRECISE ? expf(mrun[q] - mnew) : __builtin_amdgcn_exp2f((mrun[q] - mnew) * c2);
        const float nmc = -mnew * c2;
        mrun[q] = mnew;
        o[q][0] *= resc;
        o[q][1] *= resc;
        float lsum = 0.f;
#pragma unroll
        for (int tk = 0; tk < NTILE; ++tk) {
            f32x4 p = sacc[q][tk];
            if (PRECISE) {
                p.x = expf(p.x - mnew); p.y = expf(p.y - mnew); p.z = expf(p.z - mnew); p.w = expf(p.w - mnew);
            } else if (AT_ABLATE & 4) {
                p.x = fmaf(p.x, c2, nmc); p.y = fmaf(p.y, c2, nmc); p.z = fmaf(p.z, c2, nmc); p.w = fmaf(p.w, c2, nmc);
            } else {
                p.x = __builtin_amdgcn_exp2f(fmaf(p.x, c2, nmc)); p.y = __builtin_amdgcn_exp2f(fmaf(p.y, c2, nmc));
                p.z = __builtin_amdgcn_exp2f(fmaf(p.z, c2, nmc)); p.w = __builtin_amdgcn_exp2f(fmaf(p.w, c2, nmc));
            }
            lsum += (p.x + p.y) + (p.z + p.w);
            sacc[q][tk] = p;
        }
        lrun[q] = lrun[q] * resc + lsum;
    }
    if constexpr (sizeof(T) == 2) {
        AT_PRIO(1);
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                u32x4 pb;
                pb.x = pack_h16x2(sacc[q][2 * u].x, sacc[q][2 * u].y);
                pb.y = pack_h16x2(sacc[q][2 * u].z, sacc[q][2 * u].w);
                pb.z = pack_h16x2(sacc[q][2 * u + 1].x, sacc[q][2 * u + 1].y);
                pb.w = pack_h16x2(sacc[q][2 * u + 1].z, sacc[q][2 * u + 1].w);
                if (AT_ABLATE & 16) { o[q][0].x += __uint_as_float(pb.x & 0x3fffffffu) + __uint_as_float(va[u][0].x & 0x3fffffffu); o[q][1].y += __uint_as_float(pb.w & 0x3fffffffu) + __uint_as_float(va[u][1].z & 0x3fffffffu); continue; }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) mma16<T>(o[q][dt], va[u][dt], pb);
            }
        AT_PRIO(0);
    } else {
#pragma unroll
        for (int tk = 0; tk < NTILE; ++tk)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                u32x4 pb;
                pb.x = __float_as_uint(sacc[q][tk].x); pb.y = __float_as_uint(sacc[q][tk].y);
                pb.z = __float_as_uint(sacc[q][tk].z); pb.w = __float_as_uint(sacc[q][tk].w);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const u32x4 va = *reinterpret_cast<const u32x4*>(&sVt[(dt * 16 + fi) * VP + c0 + tk * 16 + g * 4]);
                    mma16<T>(o[q][dt], va, pb);
                }
            }
    }
}

// the NQ query tiles from q0 against every key of the head (chunks of at most MAXNP key-tile pairs, as equal as they come), normalised and stored
template <typename T, int NQ, int MAXNP>
__device__ __forceinline__ void at_tiles(const T* __restrict__ sK, const T* __restrict__ sVt, int VP, int Nk, int Na, int q0,
                                         const u32x4 (&qf)[NQ][sizeof(T) == 2 ? 1 : 2], T* __restrict__ orow, int ldo, float scale, int fi, int g) {
    const int npair = Nk / 32;
    const int nchunk = (npair + MAXNP - 1) / MAXNP;
    const int cbase = npair / nchunk, cextra = npair % nchunk;
    f32x4 o[NQ][2];
    float mrun[NQ], lrun[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        o[q][0] = o[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        mrun[q] = -INFINITY; lrun[q] = 0.f;
    }
    int c0 = 0;
    for (int c = 0; c < nchunk; ++c) {
        const int np = cbase + (c < cextra ? 1 : 0);
        const int valid = Na - c0;   // >= 32 np except in the last chunk
        switch (np) {
            case 1: at_chunk<T, 1, NQ>(sK, sVt, VP, c0, valid, qf, o, mrun, lrun, scale, fi, g); break;
            case 2: at_chunk<T, 2, NQ>(sK, sVt, VP, c0, valid, qf, o, mrun, lrun, scale, fi, g); break;
            case 3: at_chunk<T, 3, NQ>(sK, sVt, VP, c0, valid, qf, o, mrun, lrun, scale, fi, g); break;
            case 4: at_chunk<T, (MAXNP >= 4 ? 4 : 1), NQ>(sK, sVt, VP, c0, valid, qf, o, mrun, lrun, scale, fi, g); break;
            case 5: at_chunk<T, (MAXNP >= 5 ? 5 : 1), NQ>(sK, sVt, VP, c0, valid, qf, o, mrun, lrun, scale, fi, g); break;
            case 6: at_chunk<T, (MAXNP >= 6 ? 6 : 1), NQ>(sK, sVt, VP, c0, valid, qf, o, mrun, lrun, scale, fi, g); break;
            default: at_chunk<T, (MAXNP >= 7 ? 7 : 1), NQ>(sK, sVt, VP, c0, valid, qf, o, mrun, lrun, scale, fi, g); break;
        }
        c0 += 32 * np;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float l = at_rowgroup_sum(lrun[q]);
        if (q0 + q * 16 + fi < Na) {
            const float inv = 1.0f / l;
            T* op = orow + (size_t)(q * 16) * ldo;
            store4(op, o[q][0].x * inv, o[q][0].y * inv, o[q][0].z * inv, o[q][0].w * inv);
            store4(op + 16, o[q][1].x * inv, o[q][1].y * inv, o[q][1].z * inv, o[q][1].w * inv);
        }
    }
}

// NT = threads of the workgroup: the area's ceil(Na / 16) query tiles are dealt round-robin to NT / 64 waves — Na = 400 is 25 tiles: four
// waves take 7 / 6 / 6 / 6 (the workgroup lasts seven tile-times), five take 5 each (launch_attn_resident picks NT)
template <typename T, int WPE, int NT, bool NQ2 = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void area_attn_resident_kernel(
    const T* __restrict__ qkv, int ldq, T* __restrict__ out, int ldo, int N, int Na, int heads, int area, float scale) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NF = sizeof(T) == 2 ? 1 : 2;
    constexpr int CPR = 32 / VEC;              // 16-byte chunks per 32-wide head row
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int Nk = (Na + 31) & ~31;            // keys incl. zero padding (the P V product walks 32 keys at a time)
    const int Nr = (Na + 15) & ~15;            // K rows kept: score tiles past them read into sVt and are masked to -inf
    // V^T row pitch (elements); columns Na..Nk-1 are zeros (p = 0 there).  16-bit: Nk + 4 = a pitch of 2 x odd dwords modulo 32, so the
    // sixteen rows a ds_read2_b64 lane group reads (8 bytes each) cover the 32 banks once (Nk + 8 = 212 dwords at 400 keys put rows fi
    // and fi + 8 on the same banks: half of the kernel's LDS cycles were conflict cycles, profiles/r03_sq_summary.txt)
    const int VP = Nk + (sizeof(T) == 2 ? 4 : VEC);
    T* sK = reinterpret_cast<T*>(at_smem);     // [Nr][32]
    T* sVt = sK + (size_t)Nr * 32;             // [32][VP]        (Na = 400, bf16: 25,600 + 27,136 bytes -> 3 workgroups per CU)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fi = lane & 15, g = lane >> 4;
    // Bijective XCD remap (workgroup i runs on XCD i % 8): logically consecutive problems — the heads of one (image, area), whose 64-byte
    // q / k / v slices share 128-byte lines pairwise — are given to ONE XCD back to back.  With blockIdx taken as is, head h and h + 1
    // ran on different XCDs, each L2 fetched the shared line for itself, and the fabric counter showed qkv read twice (162 MB for 79 MB
    // at 40^2, profiles/r04_step_dispatch_pmc.txt).
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int h = bid % heads;
    const int ba = bid / heads;
    const int b = ba / area, ar = ba % area;
    const int tok0 = ar * Na;
    const int Cq = heads * 32;
    const T* base = qkv + (size_t)b * N * ldq;

    // ---- stage every key / value of the head once: K row-major, V transposed ------------------------------------------------
    auto kload = [&](int i0, u32x4 (&kreg)[4]) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int i = i0 + t + l * NT;
            const int key = i / CPR, ch = i % CPR;
            u32x4 kv = {0u, 0u, 0u, 0u};
            if (key < Na) kv = *reinterpret_cast<const u32x4*>(base + (size_t)(tok0 + key) * ldq + h * 32 + ch * VEC + Cq);
            kreg[l] = kv;
        }
    };
    auto kstore = [&](int i0, const u32x4 (&kreg)[4]) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int i = i0 + t + l * NT;
            // 16-bit rows are 64 bytes: chunk c of row r goes to slot c ^ ((-(r >> 2)) & 3), which spreads the sixteen rows of a
            // ds_read_b128 lane group over the 64 banks (unswizzled, rows r and r + 4 shared their banks)
            const int slot = sizeof(T) == 2 ? (i & ~3) + ((i & 3) ^ ((0 - (i >> 4)) & 3)) : i;
            if (i < Nr * CPR) *reinterpret_cast<u32x4*>(&sK[(size_t)slot * VEC]) = kreg[l];
        }
    };
    // V^T (bf16) as (key, key + 1) words: a thread takes the same 8 channels of two consecutive keys and stores 8 dwords; the 32 lanes of
    // a store group hold 32 consecutive key pairs of ONE channel chunk, i.e. 32 consecutive dwords of each V^T row (the 2-byte stores
    // this replaces were 8-way conflicted and twice as many)
    auto vload = [&](int i0, u32x4 (&v0)[4], u32x4 (&v1)[4]) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int i = i0 + t + l * NT;
            const int kp = (i & 31) + ((i >> 7) << 5), ch = (i >> 5) & 3, key = 2 * kp;
            u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
            const T* p = base + (size_t)(tok0 + key) * ldq + h * 32 + ch * VEC + 2 * Cq;
            if (key < Na) a = *reinterpret_cast<const u32x4*>(p);
            if (key + 1 < Na) c = *reinterpret_cast<const u32x4*>(p + ldq);
            v0[l] = a; v1[l] = c;
        }
    };
    auto vstore = [&](int i0, const u32x4 (&v0)[4], const u32x4 (&v1)[4]) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int i = i0 + t + l * NT;
            const int kp = (i & 31) + ((i >> 7) << 5), ch = (i >> 5) & 3;
            if (kp < Nk / 2) {
                uint32_t* d = reinterpret_cast<uint32_t*>(sVt) + kp;
                const int vp2 = VP / 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d[(ch * 8 + 2 * q) * vp2] = (v0[l][q] & 0xffffu) | (v1[l][q] << 16);
                    d[(ch * 8 + 2 * q + 1) * vp2] = (v0[l][q] >> 16) | (v1[l][q] & 0xffff0000u);
                }
            }
        }
    };
    if (sizeof(T) == 2 && Na <= 512) {
        // the whole head in ONE round trip: 8 K + 8 V loads per thread in flight (three dependent round trips in the loops below)
        u32x4 ka[4], kb[4], v0[4], v1[4];
        kload(0, ka);
        kload(NT * 4, kb);
        vload(0, v0, v1);
        kstore(0, ka);
        kstore(NT * 4, kb);
        vstore(0, v0, v1);
    } else {
        for (int i0 = 0; i0 < Nr * CPR; i0 += NT * 4) {
            u32x4 kreg[4];
            kload(i0, kreg);
            kstore(i0, kreg);
        }
    }
    if (sizeof(T) == 2 && Na <= 512) {
    } else if constexpr (sizeof(T) == 2) {
        for (int i0 = 0; i0 < (Nk / 2) * CPR; i0 += NT * 4) {
            u32x4 v0[4], v1[4];
            vload(i0, v0, v1);
            vstore(i0, v0, v1);
        }
    } else {
        for (int i0 = 0; i0 < Nk * CPR; i0 += NT * 4) {
            u32x4 vreg[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int i = i0 + t + l * NT;
                const int key = i / CPR, ch = i % CPR;
                u32x4 vv = {0u, 0u, 0u, 0u};
                if (key < Na) vv = *reinterpret_cast<const u32x4*>(base + (size_t)(tok0 + key) * ldq + h * 32 + ch * VEC + 2 * Cq);
                vreg[l] = vv;
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int i = i0 + t + l * NT;
                const int key = i / CPR, ch = i % CPR;
                if (key < Nk) {
                    const T* ve = reinterpret_cast<const T*>(&vreg[l]);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) sVt[(ch * VEC + q) * VP + key] = ve[q];
                }
            }
        }
    }
    __syncthreads();

    auto qload = [&](int q0, u32x4 (&q)[NF]) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            q[f] = u32x4{0u, 0u, 0u, 0u};
            if (q0 + fi < Na)
                q[f] = *reinterpret_cast<const u32x4*>(base + (size_t)(tok0 + q0 + fi) * ldq + h * 32 + f * 16 + g * VEC);
        }
    };
    constexpr int NW = NT / 64;
    if constexpr (sizeof(T) == 2 && NQ2) {
        // query tiles in PAIRS (32 queries share every K / V^T fragment read), chunks of at most AT_NQ2_NP key-tile pairs; an odd last tile goes,
        // alone, to the wave with the fewest pairs (Na = 400: 12 pairs + 1 tile over four waves = 3 + 3 + 3 + 3 pairs, wave 0 the tile)
        const int ntile = (Na + 15) >> 4, ndbl = ntile >> 1;
        u32x4 qf[2][NF], qn[2][NF];
        qload(wave * 32, qn[0]);
        qload(wave * 32 + 16, qn[1]);
        for (int d = wave; d < ndbl; d += NW) {
            const int q0 = d * 32;
#pragma unroll
            for (int f = 0; f < NF; ++f) { qf[0][f] = qn[0][f]; qf[1][f] = qn[1][f]; }
            if (d + NW < ndbl) {   // the next pair's queries arrive during this pair (a global round trip per pair otherwise)
                qload(q0 + NW * 32, qn[0]);
                qload(q0 + NW * 32 + 16, qn[1]);
            }
            at_tiles<T, 2, AT_NQ2_NP>(sK, sVt, VP, Nk, Na, q0, qf, out + (size_t)(b * (size_t)N + tok0 + q0 + fi) * ldo + h * 32 + g * 4, ldo, scale, fi, g);
        }
        if ((ntile & 1) && wave == ndbl % NW) {
            const int q0 = (ntile - 1) * 16;
            u32x4 q1[1][NF];
            qload(q0, q1[0]);
            at_tiles<T, 1, 7>(sK, sVt, VP, Nk, Na, q0, q1, out + (size_t)(b * (size_t)N + tok0 + q0 + fi) * ldo + h * 32 + g * 4, ldo, scale, fi, g);
        }
    } else {
        // one query tile at a time, chunks of at most 7 key-tile pairs (8 spills at three waves per SIMD; Na = 400: 13 pairs -> 7 + 6)
        u32x4 qf[1][NF], qn[NF];
        qload(wave * 16, qn);
        for (int q0 = wave * 16; q0 < Na; q0 += NW * 16) {
#pragma unroll
            for (int f = 0; f < NF; ++f) qf[0][f] = qn[f];
            qload(q0 + NW * 16, qn);   // the next tile's queries arrive during this tile (a global round trip per tile otherwise)
            at_tiles<T, 1, 7>(sK, sVt, VP, Nk, Na, q0, qf, out + (size_t)(b * (size_t)N + tok0 + q0 + fi) * ldo + h * 32 + g * 4, ldo, scale, fi, g);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Area attention with the qkv projection INSIDE (16-bit, C = 32 KS channels, KS <= 4): one workgroup per (image, area, head) as above,
// but K and V^T of the head are PRODUCED in LDS — phase 0 streams the area's x tile (Na tokens x C) through the head's 96 rows of the
// folded qkv weights (AAttn.qkv: 1x1 convolution + BatchNorm, nn/modules/block.py:1687,1708) held as register-resident MFMA fragments —
// instead of being loaded from a qkv tensor another kernel wrote.  Per step of the S detector this removes the four 128 -> 384 launches of
// the 40^2 A2C2f row, their 79 MB of output and its 82 MB re-read by this kernel (profiles/r05_step_dispatch_pmc.txt rows 17-36); what is
// still stored is v (the positional 7x7 depthwise branch `pe(v)` reads it: 26 MB) and the attention output.
//
// A wave owns the 16-token tiles wave, wave + 4, ...: the same tiles whose queries it walks in phase 1.  Passes Q / K (rows of W_q, W_k): for
// each tile the x fragment (lane (token fi, channel chunk g): 16 bytes per 32-channel k-step — as B operand) meets W fragments whose lane
// fi holds output row 8 (fi >> 2) + 4 mt + (fi & 3) of the head for row block mt: the accumulators of a lane are then channels 8 g + 4 mt +
// j of ITS token, i.e. after bias + rounding exactly the 16-byte word (token fi, channels 8 g .. 8 g + 7) that is the attention's Q operand
// (parked in the head's slice of `out`, re-read per tile in phase 1 and overwritten by the result) and a row chunk of K (written to
// its swizzled LDS slot).  Pass V (rows of W_v): the same product once more for the v tensor in memory, and with the x fragment as
// the A operand (same registers) the transposed product D[token 4 g + j][channel], whose accumulators are four consecutive tokens of one
// channel: one 8-byte store into V^T.  x is read ONCE per head (the wave's tiles stay in registers through the passes), W once per wave (8 KB per pass).
template <int KS, int WPE>
__global__ __launch_bounds__(AT_NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void area_attn_qkv_kernel(
    const h16_t* __restrict__ x, int ldx, const h16_t* __restrict__ wqkv, int Kpad, const float* __restrict__ bias, h16_t* __restrict__ out,
    int ldo, h16_t* __restrict__ vout, int ldv, int N, int Na, int heads, int area, float scale) {
    typedef h16_t T;
    constexpr int NW = AT_NT / 64;
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    const int Nk = (Na + 31) & ~31, Nr = (Na + 15) & ~15;
    const int VP = Nk + 4;
    T* sK = reinterpret_cast<T*>(at_smem);     // [Nr][32], chunk c of row r in slot c ^ ((-(r >> 2)) & 3)
    T* sVt = sK + (size_t)Nr * 32;             // [32][VP] as (key, key + 1) words
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fi = lane & 15, g = lane >> 4;
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int h = bid % heads;
    const int ba = bid / heads;
    const int b = ba / area, ar = ba % area;
    const int tok0 = ar * Na;
    const int C = heads * 32;
    const T* xb = x + ((size_t)b * N + tok0) * ldx;
    T* ob = out + ((size_t)b * N + tok0) * ldo + h * 32;
    T* vb = vout + ((size_t)b * N + tok0) * ldv + h * 32;
    const int ntile = Nr >> 4;
    const int wrow = 8 * (fi >> 2) + (fi & 3);   // + 4 mt: the output row of the head this lane's W fragment holds

    auto wload = [&](int part, u32x4 (&wf)[2][KS]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[mt][ks] = *reinterpret_cast<const u32x4*>(wqkv + (size_t)(part * C + h * 32 + wrow + 4 * mt) * Kpad + ks * 32 + g * 8);
    };
    // channels 8 g .. 8 g + 7 of this lane's token, bias added, rounded: row-block accumulators a0 (channels 8 g + j), a1 (8 g + 4 + j)
    auto pack8 = [&](const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1) {
        return u32x4{pack_h16x2(a0.x + b0.x, a0.y + b0.y), pack_h16x2(a0.z + b0.z, a0.w + b0.w),
                     pack_h16x2(a1.x + b1.x, a1.y + b1.y), pack_h16x2(a1.z + b1.z, a1.w + b1.w)};
    };
    // EVERY x tile of the wave is requested at once and stays in registers through the three passes (AT_QKV_MAXT x KS x 4 registers: the
    // attention phase's registers are not live yet): one exposed round trip for x instead of one per tile and pass.  Measured against the forms
    // that streamed the tiles two / three deep through two passes: 85.9 us per launch against 85.5 / 91.6 — the phase costs ~22 us either way
    // (44 us run alone: profiles/r06_negative_results.txt item 1); this form reads x once
    // (pass order V, Q, K — the storing passes first, their stores retiring under the later ones — measured 90.5 us against 85.9: nine more spilled registers)
    u32x4 w[2][KS];
    wload(0, w);
    u32x4 xf[AT_QKV_MAXT][KS];
#pragma unroll
    for (int i = 0; i < AT_QKV_MAXT; ++i) {
        // tokens past the area (the ragged last tile; tiles past the last: never used) read the area's last token instead of branching: their K
        // rows are masked, their V^T columns meet probabilities that are exactly zero, their q / v rows are not stored
        const int tok = min((wave + i * NW) * 16 + fi, Na - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (AT_QKV_ABLATE & 1) xf[i][ks] = u32x4{(unsigned)tok, (unsigned)ks, 0x3c003c00u, 0u};
            else xf[i][ks] = *reinterpret_cast<const u32x4*>(xb + (size_t)tok * ldx + ks * 32 + g * 8);
        }
    }
#if AT_QKV_WLDS
    // The K and V weight rows of the head travel through LDS (the V^T region: not written before pass V) instead of being fetched by every wave at
    // the top of its pass: requested here, together with x and W_q, they cost no round trip of their own.  Image: [part][row block mt][lane fi] rows of
    // 64 KS + 16 bytes (consecutive fragment rows 68 dwords apart: the sixteen rows of a ds_read_b128 lane group cover the 64 banks once).
    constexpr int WPITCH = 64 * KS + 16;
    char* sWst = reinterpret_cast<char*>(sVt);
    {
        constexpr int CPRW = 4 * KS;                   // 16-byte chunks per weight row
        for (int i = t; i < 2 * 32 * CPRW; i += AT_NT) {
            const int part = i / (32 * CPRW), r = (i / CPRW) % 32, c = i % CPRW;     // r = mt * 16 + fi
            const int row = 8 * ((r & 15) >> 2) + ((r & 15) & 3) + 4 * (r >> 4);
            const u32x4 v = *reinterpret_cast<const u32x4*>(wqkv + (size_t)((1 + part) * C + h * 32 + row) * Kpad + c * 8);
            *reinterpret_cast<u32x4*>(sWst + (size_t)(part * 32 + r) * WPITCH + c * 16) = v;
        }
    }
    auto wlds = [&](int part, u32x4 (&wf)[2][KS]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[mt][ks] = *reinterpret_cast<const u32x4*>(sWst + (size_t)(part * 32 + mt * 16 + fi) * WPITCH + ks * 64 + g * 16);
    };
#endif
    {   // ---- pass Q: parked in `out` --------------------------------------------------------------------------------------------
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + h * 32 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(bias + h * 32 + 8 * g + 4);
#pragma unroll
        for (int i = 0; i < AT_QKV_MAXT; ++i) {
            const int tile = wave + i * NW;
            if (tile < ntile) {   // wave-uniform
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) mma16<T>(acc[mt], w[mt][ks], xf[i][ks]);
                const int key = tile * 16 + fi;
                if (key < Na && (!(AT_QKV_ABLATE & 4) || acc[0].x == 12345.6f)) *reinterpret_cast<u32x4*>(ob + (size_t)key * ldo + 8 * g) = pack8(acc[0], acc[1], b0, b1);
            }
        }
    }
    {   // ---- pass K: row chunks into their swizzled LDS slots ---------------------------------------------------------------------
#ifndef YMK_HOST_EMU
        __builtin_amdgcn_sched_barrier(0);   // the next pass's weight loads stay BEHIND this pass (hoisted, three weight sets beside the resident x tiles spill)
#endif
#if AT_QKV_WLDS
        __syncthreads();                       // the staged weight rows are complete (their loads had pass Q to arrive)
        wlds(0, w);
#else
        if (!(AT_QKV_ABLATE & 2)) wload(1, w);
#endif
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + C + h * 32 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(bias + C + h * 32 + 8 * g + 4);
#pragma unroll
        for (int i = 0; i < AT_QKV_MAXT; ++i) {
            const int tile = wave + i * NW;
            if (tile < ntile) {
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) mma16<T>(acc[mt], w[mt][ks], xf[i][ks]);
                const int key = tile * 16 + fi;
                *reinterpret_cast<u32x4*>(&sK[(size_t)(key * 4 + (g ^ ((0 - (key >> 2)) & 3))) * 8]) = pack8(acc[0], acc[1], b0, b1);
            }
        }
    }
    {   // ---- pass V: v (memory) and V^T (LDS) ----------------------------------------------------------------------------------
#ifndef YMK_HOST_EMU
        __builtin_amdgcn_sched_barrier(0);   // the next pass's weight loads stay BEHIND this pass (hoisted, three weight sets beside the resident x tiles spill)
#endif
#if AT_QKV_WLDS
        wlds(1, w);
        __syncthreads();                       // every wave holds its W_v fragments and is done with the staged rows: V^T may be written over them
#else
        if (!(AT_QKV_ABLATE & 2)) wload(2, w);
#endif

        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 2 * C + h * 32 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(bias + 2 * C + h * 32 + 8 * g + 4);
        const float bt0 = bias[2 * C + h * 32 + wrow], bt1 = bias[2 * C + h * 32 + wrow + 4];   // the channel of this lane in the transposed product
        uint32_t* vt = reinterpret_cast<uint32_t*>(sVt);
        const int vp2 = VP / 2;
#pragma unroll
        for (int i = 0; i < AT_QKV_MAXT; ++i) {
            const int tile = wave + i * NW;
            if (tile < ntile) {
                const int key = tile * 16 + fi;
                {
                    f32x4 av[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) mma16<T>(av[mt], w[mt][ks], xf[i][ks]);   // [channel][token]: lane = token fi, channels 8 g + 4 mt + j
                    if (key < Na && (!(AT_QKV_ABLATE & 4) || av[0].x == 12345.6f)) *reinterpret_cast<u32x4*>(vb + (size_t)key * ldv + 8 * g) = pack8(av[0], av[1], b0, b1);
                }
                f32x4 at[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) mma16<T>(at[mt], xf[i][ks], w[mt][ks]);       // [token][channel]: lane = channel wrow + 4 mt, tokens 4 g + j
                // (tokens past Na inside the last tile: a copy of the last token's v — finite, and their probabilities are exactly zero)
                u32x2 w0 = {pack_h16x2(at[0].x + bt0, at[0].y + bt0), pack_h16x2(at[0].z + bt0, at[0].w + bt0)};
                u32x2 w1 = {pack_h16x2(at[1].x + bt1, at[1].y + bt1), pack_h16x2(at[1].z + bt1, at[1].w + bt1)};
                *reinterpret_cast<u32x2*>(vt + (size_t)wrow * vp2 + tile * 8 + 2 * g) = w0;
                *reinterpret_cast<u32x2*>(vt + (size_t)(wrow + 4) * vp2 + tile * 8 + 2 * g) = w1;
            }
        }
        // V^T columns Nr .. Nk - 1 (the key-tile pair walk reads them; p = 0 there, but 0 x an uninitialised NaN pattern would not be 0)
        for (int i = t; i < 32 * ((Nk - Nr) / 2); i += AT_NT) vt[(size_t)(i / ((Nk - Nr) / 2)) * vp2 + Nr / 2 + i % ((Nk - Nr) / 2)] = 0u;
    }
    __syncthreads();
    if (AT_QKV_ABLATE & 16) return;

    // ---- phase 1: the resident kernel's walk; q comes back from `out` (written by this very lane) ---------------------------------
    auto qload = [&](int q0, u32x4 (&q)[1]) {
        q[0] = u32x4{0u, 0u, 0u, 0u};
        if (q0 + fi < Na) q[0] = *reinterpret_cast<const u32x4*>(ob + (size_t)(q0 + fi) * ldo + g * 8);
    };
    u32x4 qf[1][1], qn[1];
    qload(wave * 16, qn);
    for (int q0 = wave * 16; q0 < Na; q0 += NW * 16) {
        qf[0][0] = qn[0];
        qload(q0 + NW * 16, qn);
        at_tiles<T, 1, 7>(sK, sVt, VP, Nk, Na, q0, qf, ob + (size_t)(q0 + fi) * ldo + g * 4, ldo, scale, fi, g);
    }
}

extern "C" int ymk_area_attn_qkv_supported(int32_t dtype, int32_t C, int32_t heads, int32_t N, int32_t area) {
    if (dtype != YMK_H16 || heads < 1 || area < 1 || C != heads * 32 || N % area) return 0;
    const int Na = N / area;
    return (C == 64 || C == 128) && Na >= 16 && Na <= 16 * (AT_NT / 64) * AT_QKV_MAXT && !(ymk_disabled() & (YMK_OFF_ATTN_RESIDENT | YMK_OFF_ATTN_QKV));
}

// AAttn.forward up to (not including) pe / proj (nn/modules/block.py:1696-1726) from the block's INPUT: qkv = W x + b (the folded
// 1x1 convolution, rows ordered [Q of all heads | K | V], head-major, 32 per head: AAttn.qkv.cout_perm of nn/modules.py), attention per
// (image, area, head); writes the attention output [B][N][C] and v [B][N][C].  Supported shapes: ymk_area_attn_qkv_supported.
extern "C" int ymk_area_attn_qkv(int32_t dtype, const void* x, int32_t ldx, const void* w, int32_t Kpad, const float* bias, void* out,
                                 int32_t ldo, void* v, int32_t ldv, int32_t B, int32_t N, int32_t C, int32_t heads, int32_t area, void* stream) {
    if (!x || !w || !bias || !out || !v) return YMK_E_BADARG;
    if (!ymk_area_attn_qkv_supported(dtype, C, heads, N, area) || Kpad < C || Kpad % 8 || ldx % 8 || ldo % 8 || ldv % 8) return YMK_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(v)) & 15) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    if ((int64_t)B * area * heads >= (1ll << 31)) return YMK_E_BADARG;
    const int Na = N / area, Nk = (Na + 31) & ~31, Nr = (Na + 15) & ~15;
    size_t shm = ((size_t)Nr * 32 + (size_t)32 * (Nk + 4)) * sizeof(h16_t);
#if AT_QKV_WLDS
    {   // short areas: the staged K / V weight rows (behind K) outgrow the V^T region they borrow
        const size_t wst = (size_t)Nr * 32 * sizeof(h16_t) + (size_t)2 * 32 * (64 * (C / 32) + 16);
        shm = shm > wst ? shm : wst;
    }
#endif
    const float scale = 0.17677669529663687f;  // 32^-0.5
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((size_t)B * area * heads));
#define AT_QKV_LAUNCH(KS)                                                                                                              \
    {                                                                                                                                  \
        static YmkOncePerDevice once;                                                                                                  \
        if (shm > 64 * 1024 && once.need()) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&area_attn_qkv_kernel<KS, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            once.done();                                                                                                               \
        }                                                                                                                              \
        hipLaunchKernelGGL((area_attn_qkv_kernel<KS, 3>), grid, dim3(AT_NT), shm, s, (const h16_t*)x, ldx, (const h16_t*)w, Kpad, bias, (h16_t*)out, ldo, \
                           (h16_t*)v, ldv, N, Na, heads, area, scale);                                                                 \
    }
    if (C == 64) AT_QKV_LAUNCH(2) else AT_QKV_LAUNCH(4)
#undef AT_QKV_LAUNCH
    return ymk_launch_status();
}

template <typename T, int WPE, int NT = AT_NT, bool NQ2 = false>
static int launch_attn_resident(const T* qkv, int ldq, T* out, int ldo, int B, int N, int Na, int heads, int area, float scale,
                                hipStream_t s) {
    const int Nk = (Na + 31) & ~31, Nr = (Na + 15) & ~15;
    const size_t shm = ((size_t)Nr * 32 + (size_t)32 * (Nk + (sizeof(T) == 2 ? 4 : 16 / sizeof(T)))) * sizeof(T);
    static YmkOncePerDevice attr_once;
    if (shm > 64 * 1024 && attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&area_attn_resident_kernel<T, WPE, NT, NQ2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_once.done();
    }
    hipLaunchKernelGGL((area_attn_resident_kernel<T, WPE, NT, NQ2>), dim3((unsigned)((size_t)B * area * heads)), dim3(NT), shm, s, qkv, ldq,
                       out, ldo, N, Na, heads, area, scale);
    return ymk_launch_status();
}

extern "C" int ymk_attention(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                             int32_t ldo, int32_t B, int32_t Nq, int32_t Nk, int32_t heads, int32_t hd, float scale, void* stream);

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
        if (dtype == YMK_BF16 && Na <= 1024) {
            // waves per workgroup: FOUR.  Five or six balance the 25 query tiles of a 400-token area better (5 tile-times per workgroup
            // instead of 7) but leave two workgroups per CU instead of three at 168 registers: measured 0.54 / 0.51 ms per step against 0.41
            // (profiles/r05_negative_results.txt).  YMK_ATTN_WAVES=5|6 forces them for A/B runs.
            static const int forced = [] { const char* e = getenv("YMK_ATTN_WAVES"); return e ? atoi(e) : 0; }();
            const int nw = (forced >= 4 && forced <= 6) ? forced : 4;
            if (nw == 5) return launch_attn_resident<h16_t, 3, 320>((const h16_t*)qkv, ldq, (h16_t*)out, ldo, B, N, Na, heads, area, scale, s);
            if (nw == 6) return launch_attn_resident<h16_t, 3, 384>((const h16_t*)qkv, ldq, (h16_t*)out, ldo, B, N, Na, heads, area, scale, s);
            static const int nq = [] { const char* e = getenv("YMK_ATTN_NQ"); return e ? atoi(e) : 1; }();
            if (nq == 2) return launch_attn_resident<h16_t, 3, AT_NT, true>((const h16_t*)qkv, ldq, (h16_t*)out, ldo, B, N, Na, heads, area, scale, s);
            return launch_attn_resident<h16_t, 3>((const h16_t*)qkv, ldq, (h16_t*)out, ldo, B, N, Na, heads, area, scale, s);
        }
        if (dtype == YMK_F32 && Na <= 512)
            return launch_attn_resident<float, 1>((const float*)qkv, ldq, (float*)out, ldo, B, N, Na, heads, area, scale, s);
    }
    // Areas too long for the resident kernel (the L-scale A2C2f blocks: 1600 keys; the whole-map local attention of the MoT blocks of
    // BASELINE config 5: 6400): the streaming kernel below re-stages every 256-key chunk for each 64 queries (64 FLOP per staged byte);
    // attention_mfma_kernel (csrc/mixattn.hip) holds 256 queries per workgroup against each 64-key block (256 FLOP per staged byte).
    // An area is a contiguous run of tokens, so (image, area) is simply a batch index there.
    if (dtype == YMK_H16 && !(ymk_disabled() & YMK_OFF_ATTN_WIDE) && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
        const h16_t* qp = static_cast<const h16_t*>(qkv);
        return ymk_attention(dtype, qp, ldq, qp + heads * 32, ldq, qp + 2 * heads * 32, ldq, out, ldo, B * area, Na, Na, heads, 32, scale, stream);
    }
    if (dtype == YMK_F32)
        hipLaunchKernelGGL((area_attn_kernel<float, 1>), grid, blk, 0, s, (const float*)qkv, ldq, (float*)out, ldo, N, Na,
                           heads, area, scale);
    else if (dtype == YMK_BF16)
        hipLaunchKernelGGL((area_attn_kernel<h16_t, 3>), grid, blk, 0, s, (const h16_t*)qkv, ldq, (h16_t*)out, ldo, N, Na, heads, area,
                           scale);   // 3 waves per SIMD measured best (4: small spill, 1: 2x slower)
    else
        return YMK_E_BADARG;
    return ymk_launch_status();
}
