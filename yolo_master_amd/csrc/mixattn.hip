// Config-5 rows: the four attention forms of the MoA heads and MoT experts (include/ymk_mixture.h).
// FIRST IMPLEMENTATION, correctness-first: one query per thread with the head vector in registers, keys / values staged
// in LDS as fp32 and read as broadcasts, online softmax on the VALU.  No MFMA yet: the shapes are small (head_dim 8-64,
// 49-key windows, <= 4096 pooled keys) and the tile engine of attn.hip takes over once parity is established on hardware.
#include <type_traits>

#include "ymk_common.h"
#include "../../include/ymk_mixture.h"

#define YMK_OFF_WINATTN_MFMA 262144u   // YMK_DISABLE bit: matrix-core window / general attention -> one query per thread on the VALU (A/B runs)

namespace {

__device__ __forceinline__ float ldv(const void* p, int dt, int64_t i) {
    return dt == YMK_BF16 ? h16_to_f32(static_cast<const h16_t*>(p)[i]) : static_cast<const float*>(p)[i];
}
__device__ __forceinline__ void stv(void* p, int dt, int64_t i, float v) {
    if (dt == YMK_BF16) static_cast<h16_t*>(p)[i] = f32_to_h16(v);
    else static_cast<float*>(p)[i] = v;
}
inline bool bad_dt(int dt) { return dt != YMK_F32 && dt != YMK_BF16; }

// online-softmax update of one query with one key / value row held in LDS
template <int HD>
__device__ __forceinline__ void attend(const float (&q)[HD], const float* kr, const float* vr, float scale, float& m, float& l,
                                       float (&acc)[HD]) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += q[d] * kr[d];
    s *= scale;
    if (s > m) {   // new running maximum: rescale what has been accumulated (exp(-inf) = 0 on the first key)
        const float c = __expf(m - s);
        l *= c;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] *= c;
        m = s;
    }
    const float p = __expf(s - m);
    l += p;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] += p * vr[d];
}

// ------------------------------------------------------------------------------------------------ general attention
// grid (ceil(Nq / 128), heads, B), 128 threads; key tiles of 64 rows
template <int HD>
__global__ __launch_bounds__(128) void attention_kernel(int dt, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv_,
                                                         void* out, int ldo, int Nq, int Nk, float scale) {
    __shared__ float sk[64][HD], sv[64][HD];
    const int h = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 128 + threadIdx.x;
    const bool live = qi < Nq;
    float qv[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        qv[d] = live ? ldv(q, dt, ((int64_t)b * Nq + qi) * ldq + h * HD + d) : 0.f;
        acc[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int n0 = 0; n0 < Nk; n0 += 64) {
        const int nt = min(64, Nk - n0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * HD; i += 128) {
            const int r = i / HD, d = i % HD;
            sk[r][d] = ldv(k, dt, ((int64_t)b * Nk + n0 + r) * ldk + h * HD + d);
            sv[r][d] = ldv(v, dt, ((int64_t)b * Nk + n0 + r) * ldv_ + h * HD + d);
        }
        __syncthreads();
        if (live)
            for (int r = 0; r < nt; ++r) attend<HD>(qv, sk[r], sv[r], scale, m, l, acc);
    }
    if (live) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) stv(out, dt, ((int64_t)b * Nq + qi) * ldo + h * HD + d, acc[d] * inv);
    }
}

// ------------------------------------------------------------------------------------------------ window attention
// grid (windows, heads, B), 256 threads >= win*win; the window's keys / values (real tokens or the pad vectors) in LDS
template <int HD>
__global__ __launch_bounds__(256) void window_attention_kernel(int dt, const void* q, int ldq, const void* k, int ldk, const void* v,
                                                                int ldv_, void* out, int ldo, int H, int W, float scale, int win,
                                                                int shift, const float* pad_q, const float* pad_k,
                                                                const float* pad_v) {
    extern __shared__ float smem[];   // [2][win*win][HD]
    const int T = win * win;
    float* sk = smem;
    float* sv = smem + T * HD;
    const int Hp = (H + win - 1) / win * win, Wp = (W + win - 1) / win * win;
    const int nwx = Wp / win;
    const int wy = blockIdx.x / nwx, wx = blockIdx.x % nwx;
    const int h = blockIdx.y, b = blockIdx.z;
    const int t = threadIdx.x;
    // token t of the window sits at (gy, gx) of the rolled grid = ((gy + shift) % Hp, (gx + shift) % Wp) of the padded map
    int oy = 0, ox = 0;
    bool real = false;
    if (t < T) {
        oy = (wy * win + t / win + shift) % Hp;
        ox = (wx * win + t % win + shift) % Wp;
        real = oy < H && ox < W;
    }
    const int64_t pix = ((int64_t)b * H + oy) * W + ox;
    float qv[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        acc[d] = 0.f;
        qv[d] = 0.f;
        if (t < T) {
            qv[d] = real ? ldv(q, dt, pix * ldq + h * HD + d) : (pad_q ? pad_q[h * HD + d] : 0.f);
            sk[t * HD + d] = real ? ldv(k, dt, pix * ldk + h * HD + d) : (pad_k ? pad_k[h * HD + d] : 0.f);
            sv[t * HD + d] = real ? ldv(v, dt, pix * ldv_ + h * HD + d) : (pad_v ? pad_v[h * HD + d] : 0.f);
        }
    }
    __syncthreads();
    if (!real) return;
    float m = -INFINITY, l = 0.f;
    for (int r = 0; r < T; ++r) attend<HD>(qv, sk + r * HD, sv + r * HD, scale, m, l, acc);
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < HD; ++d) stv(out, dt, pix * ldo + h * HD + d, acc[d] * inv);
}

// ------------------------------------------------------------------------------------------------ window attention on the matrix cores
// 16-bit inputs, win * win <= 64 tokens, head_dim 16 / 32 / 64: ONE WAVE per (window, head), four windows per workgroup.
//   S^T = K Q^T   4 x 4 tiles of v_mfma_f32_16x16x32 (K / Q fragments are 16-byte global loads: a lane owns 8 channels of one token)
//   softmax over keys = over the ROWS of S^T: a lane's 16 accumulators of a query column + two xor-shuffles (lane groups 16 / 32)
//   O^T = V^T P^T: the contraction index of the second product is PERMUTED — MFMA slot (g, e) stands for key 16 (2 kb + (e >> 2)) +
//         4 g + (e & 3) — so that the B operand (P^T) is exactly what the lane already holds in its S^T accumulators (no cross-lane
//         movement), and the A operand (V^T) is two 8-byte reads of a V^T image staged once per wave in LDS.
// Out-of-image tokens of the padded / rolled grid carry the pad vectors and take part as keys, as in window_attention_kernel.
template <int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void window_attention_mfma_kernel(const h16_t* q, int ldq, const h16_t* k, int ldk, const h16_t* v, int ldv_,
                                                                    h16_t* out, int ldo, int H, int W, float scale, int win, int shift,
                                                                    const float* pad_q, const float* pad_k, const float* pad_v, int nwin) {
    constexpr int KB = (HD + 31) / 32;          // 32-channel blocks of the first contraction (channels past head_dim are zero)
    constexpr int DT = (HD + 15) / 16;          // 16-channel output tiles (rows past head_dim are zero and never stored)
    constexpr int VP = 64 + 4;                  // V^T row pitch in elements (68: the 8-byte reads of 16 rows x 4 lane groups spread over the banks)
    __shared__ __attribute__((aligned(16))) h16_t svt[4][DT * 16 * VP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int widx = blockIdx.x * 4 + wave;
    if (widx >= nwin) return;                   // wave-uniform; no workgroup barrier below
    const int T = win * win;
    const int Hp = (H + win - 1) / win * win, Wp = (W + win - 1) / win * win;
    const int nwx = Wp / win;
    const int wy = widx / nwx, wx = widx % nwx;
    const int h = blockIdx.y, b = blockIdx.z;
    h16_t* vt = svt[wave];
    auto token = [&](int t, int64_t& pix) {   // -> 0: beyond the window, 1: real pixel, 2: padding of the grid
        if (t >= T) return 0;
        const int oy = (wy * win + t / win + shift) % Hp, ox = (wx * win + t % win + shift) % Wp;
        pix = ((int64_t)b * H + oy) * W + ox;
        return (oy < H && ox < W) ? 1 : 2;
    };
    auto pad_frag = [&](const float* pv, int c0) {
        u32x4 f = {0u, 0u, 0u, 0u};
        if (pv) {
            f.x = pack_h16x2(pv[h * HD + c0 + 0], pv[h * HD + c0 + 1]); f.y = pack_h16x2(pv[h * HD + c0 + 2], pv[h * HD + c0 + 3]);
            f.z = pack_h16x2(pv[h * HD + c0 + 4], pv[h * HD + c0 + 5]); f.w = pack_h16x2(pv[h * HD + c0 + 6], pv[h * HD + c0 + 7]);
        }
        return f;
    };
    // ---- Q and K fragments: token 16 i + fr, channels kb * 32 + g * 8 .. + 8 ------------------------------------------------------
    u32x4 qf[4][KB], kf[4][KB];
    int64_t mypix[4];
    int kind[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kind[i] = token(16 * i + fr, mypix[i]);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int c0 = kb * 32 + g * 8;
            u32x4 fq = {0u, 0u, 0u, 0u}, fk = {0u, 0u, 0u, 0u};
            if (c0 < HD) {
                if (kind[i] == 1) {
                    fq = *reinterpret_cast<const u32x4*>(q + mypix[i] * ldq + h * HD + c0);
                    fk = *reinterpret_cast<const u32x4*>(k + mypix[i] * ldk + h * HD + c0);
                } else if (kind[i] == 2) {
                    fq = pad_frag(pad_q, c0);
                    fk = pad_frag(pad_k, c0);
                }
            }
            qf[i][kb] = fq; kf[i][kb] = fk;
        }
    }
    // ---- V^T image of this window in LDS: vt[d][key], keys past the window zero ---------------------------------------------------
    if (DT * 16 > HD)
        for (int c = lane; c < (DT * 16 - HD) * VP; c += 64) vt[HD * VP + c] = (h16_t)0;
    for (int c = lane; c < 64 * (HD / 8); c += 64) {
        const int t = c / (HD / 8), c0 = (c % (HD / 8)) * 8;
        int64_t pix;
        const int kd = token(t, pix);
        u32x4 f = {0u, 0u, 0u, 0u};
        if (kd == 1) f = *reinterpret_cast<const u32x4*>(v + pix * ldv_ + h * HD + c0);
        else if (kd == 2) f = pad_frag(pad_v, c0);
        const uint32_t w4[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) vt[(c0 + e) * VP + t] = (h16_t)((w4[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    }
    // ---- S^T tiles: st[j][i] = K_j Q_i^T; lane (g, fr): keys 16 j + 4 g + r (r = 0..3) of query 16 i + fr -------------------------
    f32x4 st[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) acc = mfma16x16x32_h16(kf[j][kb], qf[i][kb], acc);
            st[j][i] = acc;
        }
    // ---- softmax over the keys of each query column ---------------------------------------------------------------------------------
    float inv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* a4 = reinterpret_cast<float*>(&st[j][i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a4[r] = (16 * j + 4 * g + r < T) ? a4[r] * scale : -INFINITY;
                m = fmaxf(m, a4[r]);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* a4 = reinterpret_cast<float*>(&st[j][i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a4[r] = __expf(a4[r] - m);      // exp(-inf) = 0 for the masked keys
                l += a4[r];
            }
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        inv[i] = 1.0f / l;
    }
    __builtin_amdgcn_wave_barrier();            // this wave's V^T stores are complete before its reads (same wave: program order + fence)
    __threadfence_block();
    // ---- O^T tiles: (16 channels) x (16 queries), contraction over the permuted key index ----------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4 pf[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            pf[kb].x = pack_h16x2(st[2 * kb][i].x, st[2 * kb][i].y);         pf[kb].y = pack_h16x2(st[2 * kb][i].z, st[2 * kb][i].w);
            pf[kb].z = pack_h16x2(st[2 * kb + 1][i].x, st[2 * kb + 1][i].y); pf[kb].w = pack_h16x2(st[2 * kb + 1][i].z, st[2 * kb + 1][i].w);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const h16_t* row = vt + (dt * 16 + fr) * VP + 32 * kb + 4 * g;     // keys 32 kb + 4 g .. + 3 and 32 kb + 16 + 4 g .. + 3
                const u32x2 lo = *reinterpret_cast<const u32x2*>(row), hi = *reinterpret_cast<const u32x2*>(row + 16);
                const u32x4 af = {lo.x, lo.y, hi.x, hi.y};
                o = mfma16x16x32_h16(af, pf[kb], o);
            }
            // lane (g, fr): channels dt * 16 + 4 g + r of query 16 i + fr
            if (kind[i] == 1 && dt * 16 + 4 * g < HD) {
                u32x2 pk;
                pk.x = pack_h16x2(o.x * inv[i], o.y * inv[i]);
                pk.y = pack_h16x2(o.z * inv[i], o.w * inv[i]);
                *reinterpret_cast<u32x2*>(out + mypix[i] * ldo + h * HD + dt * 16 + 4 * g) = pk;
            }
        }
    }
}

// max / sum over the four lane groups that share (lane & 15): register swaps (v_permlane16_swap / v_permlane32_swap), no LDS round trip
__device__ __forceinline__ float mx_rowgroup_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float mx_rowgroup_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ------------------------------------------------------------------------------------------------ general attention on the matrix cores
// 16-bit inputs, head_dim 16 / 32 / 64: a workgroup of four waves takes 256 queries of one (image, head) and walks the keys in blocks
// of 64 (online softmax).  Per key block the workgroup stages K (row-major, 16-byte chunks XOR-swizzled by the row) and V^T in LDS
// once; every wave then forms S^T = K Q^T for its 64 queries (Q fragments stay in registers for the whole kernel), rescales its O^T
// accumulators by exp(m_old - m_new) — a per-lane scalar, because a lane's accumulators all belong to query column (lane & 15) —
// and adds V^T P^T with the permuted contraction index of window_attention_mfma_kernel (P^T is used where the MFMA left it).
// NI = 16-query groups per wave (4: 256 queries per workgroup, ~190 registers = two waves per SIMD; 2: 128 queries, <= 128 registers = four
// waves per SIMD — each wave's chain S -> max -> exp -> P -> O is latency-bound, more resident waves hide it)
template <int HD, int NI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD > 32 ? 1 : NI == 2 ? 4 : 2))) void attention_mfma_kernel(const h16_t* q, int ldq, const h16_t* k, int ldk, const h16_t* v, int ldv_, h16_t* out,
                                                             int ldo, int Nq, int Nk, float scale) {
    constexpr int KB = (HD + 31) / 32, DT = (HD + 15) / 16, CH = HD / 8;   // 16-byte chunks per K row
    constexpr int VP = 64 + 4;
    constexpr int KC = CH <= 4 ? 4 : 8;          // row pitch in chunks: a power of two (XOR swizzle); chunks past head_dim are zero
    __shared__ __attribute__((aligned(16))) u32x4 sk[64 * KC];                   // K block: row r, chunk c at r * KC + (c ^ (r & (KC - 1)))
    __shared__ __attribute__((aligned(16))) h16_t svt[DT * 16 * VP];             // V^T block (rows past head_dim zero)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (64 * NI) + wave * (16 * NI);
    const h16_t* qb = q + (int64_t)b * Nq * ldq + h * HD;
    const h16_t* kbp = k + (int64_t)b * Nk * ldk + h * HD;
    const h16_t* vb = v + (int64_t)b * Nk * ldv_ + h * HD;
    u32x4 qf[NI][KB];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int qi = q0 + 16 * i + fr, c0 = kb * 32 + g * 8;
            qf[i][kb] = (qi < Nq && c0 < HD) ? *reinterpret_cast<const u32x4*>(qb + (int64_t)qi * ldq + c0) : u32x4{0u, 0u, 0u, 0u};
        }
    if (DT * 16 > HD)
        for (int c = threadIdx.x; c < (DT * 16 - HD) * VP; c += 256) svt[HD * VP + c] = (h16_t)0;
    f32x4 o[NI][DT];
    float m[NI];
    // softmax denominators on the matrix cores: a block of ones next to V^T makes one more output tile whose every row is
    // sum_keys P^T — the sum of exactly the rounded weights the numerator uses, complete per lane (the MFMA contracts over all 64 keys:
    // no per-score add, no cross-lane reduction).  The softmax is VALU-bound here (~5 VALU slots per score beside 2 x 64 MFMA FLOPs);
    // the matrix pipe has the room.  Only element .x of a tile is rescaled and read.
    f32x4 den[NI];
    const uint32_t one2 = pack_h16x2(1.0f, 1.0f);
    const u32x4 ones = {one2, one2, one2, one2};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        m[i] = -INFINITY; den[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // K / V of the next 64-key block travel in registers while the current block is multiplied (the loads are issued right after the
    // barrier that publishes the current block and are consumed after the barrier that retires it): global latency off the critical path
    constexpr int NKL = (64 * KC + 255) / 256, NVL = (64 * CH + 255) / 256;
    u32x4 kreg[NKL], vreg[NVL];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < NKL; ++u) {
            const int c = threadIdx.x + u * 256, r = c / KC, cc = c % KC;
            kreg[u] = u32x4{0u, 0u, 0u, 0u};
            if (c < 64 * KC && n0 + r < Nk && cc < CH) kreg[u] = *reinterpret_cast<const u32x4*>(kbp + (int64_t)(n0 + r) * ldk + cc * 8);
        }
#pragma unroll
        for (int u = 0; u < NVL; ++u) {
            const int c = threadIdx.x + u * 256, t = c / CH, c0 = (c % CH) * 8;
            vreg[u] = u32x4{0u, 0u, 0u, 0u};
            if (c < 64 * CH && n0 + t < Nk) vreg[u] = *reinterpret_cast<const u32x4*>(vb + (int64_t)(n0 + t) * ldv_ + c0);
        }
    };
    fetch(0);
    const float c2 = scale * 1.4426950408889634f;   // softmax scale folded into the exp2 argument: p = 2^((s - m) * scale * log2 e)
    for (int n0 = 0; n0 < Nk; n0 += 64) {
        __syncthreads();                          // the previous block's fragment reads are finished
#pragma unroll
        for (int u = 0; u < NKL; ++u) {
            const int c = threadIdx.x + u * 256, r = c / KC, cc = c % KC;
            if (c < 64 * KC) sk[r * KC + (cc ^ (r & (KC - 1)))] = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < NVL; ++u) {
            const int c = threadIdx.x + u * 256, t = c / CH, c0 = (c % CH) * 8;
            if (c < 64 * CH) {
                const uint32_t w4[4] = {vreg[u].x, vreg[u].y, vreg[u].z, vreg[u].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) svt[(c0 + e) * VP + t] = (h16_t)((w4[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
            }
        }
        __syncthreads();
        if (n0 + 64 < Nk) fetch(n0 + 64);
        // fragments of this key block, read once: K rows (A operand of S^T = K Q^T) and V^T rows (A operand of O^T += V^T P^T)
        constexpr bool HOIST_V = DT <= 2 && NI == 4;   // head_dim > 32 or the 128-register form: the V^T fragments are re-read per query group
        constexpr bool HOIST_K = NI == 4;
        u32x4 kf[HOIST_K ? 4 : 1][KB], vf[HOIST_V ? DT : 1][2];
        auto k_frag = [&](int j, int kb) {
            const int r = 16 * j + fr, cc = kb * 4 + g;
            return cc < KC ? sk[r * KC + (cc ^ (r & (KC - 1)))] : u32x4{0u, 0u, 0u, 0u};
        };
        if constexpr (HOIST_K) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) kf[j][kb] = k_frag(j, kb);
        }
        auto vt_frag = [&](int dt, int kb) {
            const h16_t* row = svt + (dt * 16 + fr) * VP + 32 * kb + 4 * g;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(row), hi = *reinterpret_cast<const u32x2*>(row + 16);
            return u32x4{lo.x, lo.y, hi.x, hi.y};
        };
        if constexpr (HOIST_V) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) vf[dt][kb] = vt_frag(dt, kb);
        }
        // one 16-query group at a time: its four S^T tiles (16 registers) live only from their MFMAs to the packed P^T — the whole
        // 64 x 64 score block held at once (64 registers) spilled into the accumulation registers and came back through 160 copies
        auto softmax_pv = [&](auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 st[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc = mfma16x16x32_h16(HOIST_K ? kf[HOIST_K ? j : 0][kb] : k_frag(j, kb), qf[i][kb], acc);
                st[j] = acc;
            }
            if constexpr (TAIL) {                     // last block only (compiled twice: as a runtime condition the sixteen selects per group were issued in every block)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float* a4 = reinterpret_cast<float*>(&st[j]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + 16 * j + 4 * g + r >= Nk) a4[r] = -INFINITY;
                }
            }
            float mb = st[0].x;
#pragma unroll
            for (int j = 0; j < 4; ++j) mb = fmaxf(fmaxf(fmaxf(fmaxf(mb, st[j].x), st[j].y), st[j].z), st[j].w);   // v_max3 chains
            mb = mx_rowgroup_max(mb);
            const float mn = fmaxf(m[i], mb);     // finite: every block holds at least one real key (scores are UNSCALED here)
            const float c = __builtin_amdgcn_exp2f((m[i] - mn) * c2);    // 2^(-inf) = 0 on the first block
            const float nmc = -mn * c2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st[j].x = __builtin_amdgcn_exp2f(fmaf(st[j].x, c2, nmc));
                st[j].y = __builtin_amdgcn_exp2f(fmaf(st[j].y, c2, nmc));
                st[j].z = __builtin_amdgcn_exp2f(fmaf(st[j].z, c2, nmc));
                st[j].w = __builtin_amdgcn_exp2f(fmaf(st[j].w, c2, nmc));
            }
            m[i] = mn;
            u32x4 pf[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                pf[kb].x = pack_h16x2(st[2 * kb].x, st[2 * kb].y);         pf[kb].y = pack_h16x2(st[2 * kb].z, st[2 * kb].w);
                pf[kb].z = pack_h16x2(st[2 * kb + 1].x, st[2 * kb + 1].y); pf[kb].w = pack_h16x2(st[2 * kb + 1].z, st[2 * kb + 1].w);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x4 acc = o[i][dt];
                acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) acc = mfma16x16x32_h16(HOIST_V ? vf[HOIST_V ? dt : 0][kb] : vt_frag(dt, kb), pf[kb], acc);
                o[i][dt] = acc;
            }
            {
                f32x4 acc = den[i];
                acc.x *= c;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) acc = mfma16x16x32_h16(ones, pf[kb], acc);
                den[i] = acc;
            }
        }
        };
        if (n0 + 64 > Nk) softmax_pv(std::true_type{}); else softmax_pv(std::false_type{});   // only the last block can hold keys past the end (workgroup-uniform)
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int qi = q0 + 16 * i + fr;
        const float inv = 1.0f / den[i].x;
        if (qi >= Nq) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if (dt * 16 + 4 * g >= HD) continue;
            u32x2 pk;
            pk.x = pack_h16x2(o[i][dt].x * inv, o[i][dt].y * inv);
            pk.y = pack_h16x2(o[i][dt].z * inv, o[i][dt].w * inv);
            *reinterpret_cast<u32x2*>(out + ((int64_t)b * Nq + qi) * ldo + h * HD + dt * 16 + 4 * g) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------ random-feature attention
__device__ __forceinline__ float phi(float t, float sc) { return fminf(fmaxf(t * sc, 0.f) + 1e-6f, 1e4f); }

// pass 1, grid (heads, B, chunks), 256 threads: kv[f][d] = sum_n phi(k_n)[f] v_n[d], ksum[f] = sum_n phi(k_n)[f] over the chunk's
// LINATTN_CHUNK tokens (one workgroup per (image, head) looped over 25 600 tokens on 96 workgroups: 9 ms per call at config 5);
// pass 1b adds the chunk partials in chunk order (deterministic)
#define LINATTN_CHUNK 512
__global__ __launch_bounds__(256) void linattn_kv_kernel(int dt, const void* k, int ldk, const void* v, int ldv_, const float* rf, int nb,
                                                          int Ntot, int hd, float* ws) {
    // K is staged next to V (it was read element by element from global memory inside the feature dot product: 64 x nb x hd scalar
    // loads per 64 tokens); the random-feature matrix is kept transposed so that lanes of consecutive features read consecutive words
    __shared__ float sphi[64][64], svv[64][64], skk[64][64], srfT[64][64];
    const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x, chunk = blockIdx.z, nchunk = gridDim.z;
    const int per = nb * hd + nb;
    float* kv = ws + (((int64_t)b * heads + h) * (nchunk + 1) + 1 + chunk) * per;   // slot 0 of an (image, head): the reduced result
    float* ksum = kv + nb * hd;
    const int N = min(Ntot, (chunk + 1) * LINATTN_CHUNK);
    const float sc = 1.0f / sqrtf((float)nb);
    for (int i = threadIdx.x; i < nb * hd; i += 256) srfT[i % hd][i / hd] = rf[i];
    // each thread owns up to 16 (f, d) pairs and, for t < nb, one ksum entry
    float acc[16], ks = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int n0 = chunk * LINATTN_CHUNK; n0 < N; n0 += 64) {
        const int nt = min(64, N - n0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * hd; i += 256) {
            svv[i / hd][i % hd] = ldv(v, dt, ((int64_t)b * Ntot + n0 + i / hd) * ldv_ + h * hd + i % hd);
            skk[i / hd][i % hd] = ldv(k, dt, ((int64_t)b * Ntot + n0 + i / hd) * ldk + h * hd + i % hd);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nt * nb; i += 256) {
            const int r = i / nb, f = i % nb;
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += skk[r][d] * srfT[d][f];   // same order of the terms as before
            sphi[r][f] = phi(s, sc);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = threadIdx.x + j * 256;
            if (e < nb * hd) {
                const int f = e / hd, d = e % hd;
                float s = 0.f;
                for (int r = 0; r < nt; ++r) s += sphi[r][f] * svv[r][d];
                acc[j] += s;
            }
        }
        if (threadIdx.x < nb)
            for (int r = 0; r < nt; ++r) ks += sphi[r][threadIdx.x];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int e = threadIdx.x + j * 256;
        if (e < nb * hd) kv[e] = acc[j];
    }
    if (threadIdx.x < nb) ksum[threadIdx.x] = ks;
}
// pass 1b, grid (heads * B): slot 0 = sum of the chunk partials, in chunk order
__global__ __launch_bounds__(256) void linattn_reduce_kernel(int per, int nchunk, float* ws) {
    float* base = ws + (int64_t)blockIdx.x * (nchunk + 1) * per;
    for (int i = threadIdx.x; i < per; i += 256) {
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) s += base[(int64_t)(1 + c) * per + i];
        base[i] = s;
    }
}
// pass 2, grid (ceil(N / 128), heads, B): out = clamp(phi(q) kv, +-1e4) / max(phi(q) . ksum, 1e-6)
template <int HD>
__global__ __launch_bounds__(128) void linattn_out_kernel(int dt, const void* q, int ldq, const float* rf, int nb, void* out, int ldo,
                                                           int N, const float* ws, int nchunk) {
    __shared__ float skv[64 * HD + 64], srf[64 * HD];
    const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
    const float* kv = ws + ((int64_t)b * heads + h) * (nchunk + 1) * (nb * HD + nb);
    for (int i = threadIdx.x; i < nb * HD + nb; i += 128) skv[i] = kv[i];
    for (int i = threadIdx.x; i < nb * HD; i += 128) srf[i] = rf[i];
    __syncthreads();
    const int n = blockIdx.x * 128 + threadIdx.x;
    if (n >= N) return;
    const float sc = 1.0f / sqrtf((float)nb);
    float qv[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        qv[d] = ldv(q, dt, ((int64_t)b * N + n) * ldq + h * HD + d);
        acc[d] = 0.f;
    }
    float den = 0.f;
    for (int f = 0; f < nb; ++f) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += qv[d] * srf[f * HD + d];
        const float pf = phi(s, sc);
        den += pf * skv[nb * HD + f];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] += pf * skv[f * HD + d];
    }
    den = fmaxf(den, 1e-6f);
#pragma unroll
    for (int d = 0; d < HD; ++d) stv(out, dt, ((int64_t)b * N + n) * ldo + h * HD + d, fminf(fmaxf(acc[d], -1e4f), 1e4f) / den);
}

// ------------------------------------------------------------------------------------------------ deformable sampling
// one thread per (token, head, channel of the head)
__global__ __launch_bounds__(256) void deform_kernel(int dt, const void* v, int ldv_, const float* off, int ldoff, const float* aw, int ldaw,
                                                      void* out, int ldo, int B, int H, int W, int heads, int hd, int np, int align) {
    const int64_t total = (int64_t)B * H * W * heads * hd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % hd);
        int64_t p = i / hd;
        const int h = (int)(p % heads);
        p /= heads;   // b*H*W + token
        const int n = (int)(p % ((int64_t)H * W));
        const int b = (int)(p / ((int64_t)H * W));
        const float refx = (float)(n % W) / (float)max(W - 1, 1) * 2.f - 1.f;
        const float refy = (float)(n / W) / (float)max(H - 1, 1) * 2.f - 1.f;
        const float* ar = aw + p * ldaw + h * np;
        float m = -INFINITY;
        for (int j = 0; j < np; ++j) m = fmaxf(m, ar[j]);
        float den = 0.f;
        for (int j = 0; j < np; ++j) den += expf(ar[j] - m);
        float acc = 0.f;
        for (int j = 0; j < np; ++j) {
            const float* o = off + p * ldoff + (h * np + j) * 2;
            const float gx = fminf(fmaxf(refx + 0.25f * tanhf(o[0]), -1.f), 1.f);
            const float gy = fminf(fmaxf(refy + 0.25f * tanhf(o[1]), -1.f), 1.f);
            // F.grid_sample, bilinear, zeros padding
            const float ix = align ? (gx + 1.f) * 0.5f * (float)(W - 1) : ((gx + 1.f) * (float)W - 1.f) * 0.5f;
            const float iy = align ? (gy + 1.f) * 0.5f * (float)(H - 1) : ((gy + 1.f) * (float)H - 1.f) * 0.5f;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float tx = ix - fx, ty = iy - fy;
            float s = 0.f;
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {
                    const int xx = x0 + cx, yy = y0 + cy;
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H)
                        s += (cx ? tx : 1.f - tx) * (cy ? ty : 1.f - ty) * ldv(v, dt, (((int64_t)b * H + yy) * W + xx) * ldv_ + h * hd + d);
                }
            acc += expf(ar[j] - m) / den * s;
        }
        stv(out, dt, p * ldo + h * hd + d, acc);
    }
}

// 16-bit maps, head_dim % 8 == 0: one thread per (token, head, 8-channel chunk) — the sampling positions, bilinear weights and the
// softmax over the points are computed once per chunk instead of once per channel, and every corner is one 16-byte load
__global__ __launch_bounds__(256) void deform_vec_kernel(const h16_t* v, int ldv_, const float* off, int ldoff, const float* aw, int ldaw,
                                                          h16_t* out, int ldo, int B, int H, int W, int heads, int hd, int np, int align) {
    const int cpb = hd / 8;
    const int64_t total = (int64_t)B * H * W * heads * cpb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ck = (int)(i % cpb);
        int64_t p = i / cpb;
        const int h = (int)(p % heads);
        p /= heads;   // b*H*W + token
        const int n = (int)(p % ((int64_t)H * W));
        const int b = (int)(p / ((int64_t)H * W));
        const float refx = (float)(n % W) / (float)max(W - 1, 1) * 2.f - 1.f;
        const float refy = (float)(n / W) / (float)max(H - 1, 1) * 2.f - 1.f;
        const float* ar = aw + p * ldaw + h * np;
        float m = -INFINITY;
        for (int j = 0; j < np; ++j) m = fmaxf(m, ar[j]);
        float den = 0.f;
        for (int j = 0; j < np; ++j) den += expf(ar[j] - m);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < np; ++j) {
            const float* o = off + p * ldoff + (h * np + j) * 2;
            const float gx = fminf(fmaxf(refx + 0.25f * tanhf(o[0]), -1.f), 1.f);
            const float gy = fminf(fmaxf(refy + 0.25f * tanhf(o[1]), -1.f), 1.f);
            const float ix = align ? (gx + 1.f) * 0.5f * (float)(W - 1) : ((gx + 1.f) * (float)W - 1.f) * 0.5f;
            const float iy = align ? (gy + 1.f) * 0.5f * (float)(H - 1) : ((gy + 1.f) * (float)H - 1.f) * 0.5f;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float tx = ix - fx, ty = iy - fy;
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {
                    const int xx = x0 + cx, yy = y0 + cy;
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                        const float wgt = (cx ? tx : 1.f - tx) * (cy ? ty : 1.f - ty);
                        float vv[8];
                        load_vec_f32(v + (((int64_t)b * H + yy) * W + xx) * ldv_ + h * hd + ck * 8, vv);
#pragma unroll
                        for (int q = 0; q < 8; ++q) s[q] += wgt * vv[q];   // same corner order and products as the scalar kernel
                    }
                }
            const float wj = expf(ar[j] - m) / den;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += wj * s[q];
        }
        store_vec_f32(out + p * ldo + h * hd + ck * 8, acc);
    }
}

}  // namespace

#define HD_SWITCH(hd, CALL)          \
    switch (hd) {                    \
        case 8: CALL(8); break;      \
        case 16: CALL(16); break;    \
        case 24: CALL(24); break;    \
        case 32: CALL(32); break;    \
        case 40: CALL(40); break;    \
        case 48: CALL(48); break;    \
        case 56: CALL(56); break;    \
        case 64: CALL(64); break;    \
        default: return YMK_E_BADARG; \
    }

extern "C" int ymk_attention(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                             void* out, int32_t ldo, int32_t B, int32_t Nq, int32_t Nk, int32_t heads, int32_t hd, float scale,
                             void* stream) {
    if (!q || !k || !v || !out || bad_dt(dtype) || heads < 1 || heads > 65535 || B > 65535 || Nk < 1) return YMK_E_BADARG;
    const int C = heads * hd;
    if (ldq < C || ldk < C || ldv < C || ldo < C) return YMK_E_BADARG;
    if (B <= 0 || Nq <= 0) return YMK_OK;
    {   // 16-bit maps with 16-byte aligned head slices: 256 queries per workgroup on the matrix cores
        auto al = [](const void* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; };
        if (dtype == YMK_H16 && hd % 8 == 0 && hd >= 8 && hd <= 64 && al(q, ldq) && al(k, ldk) && al(v, ldv) &&
            (reinterpret_cast<uintptr_t>(out) & 7) == 0 && ldo % 4 == 0 && !(ymk_disabled() & YMK_OFF_WINATTN_MFMA)) {
            // head_dim <= 32: 2 x 16 queries per wave = 128 per workgroup at four waves per SIMD (config 5's 6400-key areas: 15.5 -> 13.5 ms per
            // step against the 256-query form; eight-wave workgroups of 256 queries at the same occupancy: 15.9); wider heads: 4 x 16 queries per wave
            const bool narrow = hd <= 32;
            const int qpw = narrow ? 128 : 256;
            const dim3 mgrid((Nq + qpw - 1) / qpw, heads, B);
#define MARGS (const h16_t*)q, ldq, (const h16_t*)k, ldk, (const h16_t*)v, ldv, (h16_t*)out, ldo, Nq, Nk, scale
#define MCALL(HDV)                                                                                                                               \
    if (narrow) hipLaunchKernelGGL((attention_mfma_kernel<(HDV <= 32 ? HDV : 32), 2>), mgrid, dim3(256), 0, (hipStream_t)stream, MARGS);   \
    else hipLaunchKernelGGL((attention_mfma_kernel<(HDV > 32 ? HDV : 64), 4>), mgrid, dim3(256), 0, (hipStream_t)stream, MARGS)
            HD_SWITCH(hd, MCALL)
#undef MCALL
#undef MARGS
            return ymk_launch_status();
        }
    }
    const dim3 grid((Nq + 127) / 128, heads, B);
#define CALL(HDV) \
    hipLaunchKernelGGL(attention_kernel<HDV>, grid, dim3(128), 0, (hipStream_t)stream, dtype, q, ldq, k, ldk, v, ldv, out, ldo, Nq, Nk, scale)
    HD_SWITCH(hd, CALL)
#undef CALL
    return ymk_launch_status();
}

extern "C" int ymk_window_attention(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                                    void* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t heads, int32_t hd, float scale,
                                    int32_t win, int32_t shift, const float* pad_q, const float* pad_k, const float* pad_v,
                                    void* stream) {
    if (!q || !k || !v || !out || bad_dt(dtype) || heads < 1 || heads > 65535 || B > 65535 || win < 1 || win > 16 || shift < 0 ||
        H < 1 || W < 1)
        return YMK_E_BADARG;
    const int C = heads * hd;
    if (ldq < C || ldk < C || ldv < C || ldo < C) return YMK_E_BADARG;
    const size_t lds = (size_t)2 * win * win * hd * sizeof(float);
    if (lds > 64 * 1024) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    const int nwy = (H + win - 1) / win, nwx = (W + win - 1) / win;
    // 16-bit maps with 16-byte aligned head slices: one wave per (window, head) on the matrix cores
    auto al = [](const void* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; };
    if (dtype == YMK_H16 && win * win <= 64 && hd % 8 == 0 && hd >= 8 && hd <= 64 && al(q, ldq) && al(k, ldk) && al(v, ldv) &&
        (reinterpret_cast<uintptr_t>(out) & 7) == 0 && ldo % 4 == 0 && !(ymk_disabled() & YMK_OFF_WINATTN_MFMA)) {
        const int nwin = nwy * nwx;
        const dim3 mgrid((nwin + 3) / 4, heads, B);
#define MCALL(HDV)                                                                                                                          \
    hipLaunchKernelGGL(window_attention_mfma_kernel<HDV>, mgrid, dim3(256), 0, (hipStream_t)stream, (const h16_t*)q, ldq, (const h16_t*)k, ldk, \
                       (const h16_t*)v, ldv, (h16_t*)out, ldo, H, W, scale, win, shift, pad_q, pad_k, pad_v, nwin)
        HD_SWITCH(hd, MCALL)
#undef MCALL
        return ymk_launch_status();
    }
    const dim3 grid(nwy * nwx, heads, B);
#define CALL(HDV)                                                                                                                   \
    hipLaunchKernelGGL(window_attention_kernel<HDV>, grid, dim3(256), lds, (hipStream_t)stream, dtype, q, ldq, k, ldk, v, ldv, out, ldo, \
                       H, W, scale, win, shift, pad_q, pad_k, pad_v)
    HD_SWITCH(hd, CALL)
#undef CALL
    return ymk_launch_status();
}

extern "C" int ymk_linear_attention(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                                    const float* rf, int32_t nb, void* out, int32_t ldo, int32_t B, int32_t N, int32_t heads,
                                    int32_t hd, float* ws, void* stream) {
    if (!q || !k || !v || !rf || !out || !ws || bad_dt(dtype) || heads < 1 || heads > 65535 || B > 65535 || nb < 1 || nb > 64 ||
        hd < 1 || hd > 64)
        return YMK_E_BADARG;
    const int C = heads * hd;
    if (ldq < C || ldk < C || ldv < C || ldo < C) return YMK_E_BADARG;
    if (B <= 0 || N <= 0) return YMK_OK;
    const int nchunk = (N + LINATTN_CHUNK - 1) / LINATTN_CHUNK;
    if (nchunk > 65535) return YMK_E_BADARG;
    hipLaunchKernelGGL(linattn_kv_kernel, dim3(heads, B, nchunk), dim3(256), 0, (hipStream_t)stream, dtype, k, ldk, v, ldv, rf, nb, N, hd, ws);
    hipLaunchKernelGGL(linattn_reduce_kernel, dim3(heads * B), dim3(256), 0, (hipStream_t)stream, nb * hd + nb, nchunk, ws);
    const dim3 grid((N + 127) / 128, heads, B);
#define CALL(HDV) \
    hipLaunchKernelGGL(linattn_out_kernel<HDV>, grid, dim3(128), 0, (hipStream_t)stream, dtype, q, ldq, rf, nb, out, ldo, N, (const float*)ws, nchunk)
    HD_SWITCH(hd, CALL)
#undef CALL
    return ymk_launch_status();
}

extern "C" int ymk_deform_attention(int32_t dtype, const void* v, int32_t ldv, const float* off, int32_t ldoff, const float* aw,
                                    int32_t ldaw, void* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t heads, int32_t hd,
                                    int32_t n_points, int32_t align_corners, void* stream) {
    if (!v || !off || !aw || !out || bad_dt(dtype) || heads < 1 || hd < 1 || n_points < 1 || n_points > 8 || H < 1 || W < 1)
        return YMK_E_BADARG;
    const int C = heads * hd;
    if (ldv < C || ldo < C || ldoff < heads * n_points * 2 || ldaw < heads * n_points) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
#ifndef YMK_MAX_BLOCKS
#define YMK_MAX_BLOCKS 16384
#endif
    if (dtype == YMK_H16 && hd % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int64_t tv = (int64_t)B * H * W * heads * (hd / 8), nbv = (tv + 255) / 256;
        hipLaunchKernelGGL(deform_vec_kernel, dim3((unsigned)(nbv > YMK_MAX_BLOCKS ? YMK_MAX_BLOCKS : nbv)), dim3(256), 0, (hipStream_t)stream,
                           (const h16_t*)v, ldv, off, ldoff, aw, ldaw, (h16_t*)out, ldo, B, H, W, heads, hd, n_points, align_corners);
        return ymk_launch_status();
    }
    const int64_t total = (int64_t)B * H * W * C;
    const int64_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(deform_kernel, dim3((unsigned)(nb > YMK_MAX_BLOCKS ? YMK_MAX_BLOCKS : nb)), dim3(256), 0, (hipStream_t)stream, dtype, v, ldv, off, ldoff,
                       aw, ldaw, out, ldo, B, H, W, heads, hd, n_points, align_corners);
    return ymk_launch_status();
}
