// Config-5 rows: the four attention forms of the MoA heads and MoT experts (include/ymk_mixture.h).
// FIRST IMPLEMENTATION, correctness-first: one query per thread with the head vector in registers, keys / values staged
// in LDS as fp32 and read as broadcasts, online softmax on the VALU.  No MFMA yet: the shapes are small (head_dim 8-64,
// 49-key windows, <= 4096 pooled keys) and the tile engine of attn.hip takes over once parity is established on hardware.
#include "ymk_common.h"
#include "../../include/ymk_mixture.h"

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

// ------------------------------------------------------------------------------------------------ random-feature attention
__device__ __forceinline__ float phi(float t, float sc) { return fminf(fmaxf(t * sc, 0.f) + 1e-6f, 1e4f); }

// pass 1, grid (heads, B), 256 threads: kv[f][d] = sum_n phi(k_n)[f] v_n[d], ksum[f] = sum_n phi(k_n)[f]
__global__ __launch_bounds__(256) void linattn_kv_kernel(int dt, const void* k, int ldk, const void* v, int ldv_, const float* rf, int nb,
                                                          int N, int hd, float* ws) {
    __shared__ float sphi[64][65], svv[64][65], srf[64][65];
    const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
    float* kv = ws + ((int64_t)b * heads + h) * (nb * hd + nb);
    float* ksum = kv + nb * hd;
    const float sc = 1.0f / sqrtf((float)nb);
    for (int i = threadIdx.x; i < nb * hd; i += 256) srf[i / hd][i % hd] = rf[i];
    // each thread owns up to 16 (f, d) pairs and, for t < nb, one ksum entry
    float acc[16], ks = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int nt = min(64, N - n0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * hd; i += 256) svv[i / hd][i % hd] = ldv(v, dt, ((int64_t)b * N + n0 + i / hd) * ldv_ + h * hd + i % hd);
        for (int i = threadIdx.x; i < nt * nb; i += 256) {
            const int r = i / nb, f = i % nb;
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += ldv(k, dt, ((int64_t)b * N + n0 + r) * ldk + h * hd + d) * srf[f][d];
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
// pass 2, grid (ceil(N / 128), heads, B): out = clamp(phi(q) kv, +-1e4) / max(phi(q) . ksum, 1e-6)
template <int HD>
__global__ __launch_bounds__(128) void linattn_out_kernel(int dt, const void* q, int ldq, const float* rf, int nb, void* out, int ldo,
                                                           int N, const float* ws) {
    __shared__ float skv[64 * HD + 64], srf[64 * HD];
    const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
    const float* kv = ws + ((int64_t)b * heads + h) * (nb * HD + nb);
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
    hipLaunchKernelGGL(linattn_kv_kernel, dim3(heads, B), dim3(256), 0, (hipStream_t)stream, dtype, k, ldk, v, ldv, rf, nb, N, hd, ws);
    const dim3 grid((N + 127) / 128, heads, B);
#define CALL(HDV) \
    hipLaunchKernelGGL(linattn_out_kernel<HDV>, grid, dim3(128), 0, (hipStream_t)stream, dtype, q, ldq, rf, nb, out, ldo, N, (const float*)ws)
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
    const int64_t total = (int64_t)B * H * W * C;
    const int64_t nb = (total + 255) / 256;
#ifndef YMK_MAX_BLOCKS
#define YMK_MAX_BLOCKS 16384
#endif
    hipLaunchKernelGGL(deform_kernel, dim3((unsigned)(nb > YMK_MAX_BLOCKS ? YMK_MAX_BLOCKS : nb)), dim3(256), 0, (hipStream_t)stream, dtype, v, ldv, off, ldoff,
                       aw, ldaw, out, ldo, B, H, W, heads, hd, n_points, align_corners);
    return ymk_launch_status();
}
