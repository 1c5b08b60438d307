// ES-MoE router + sparse dispatch (CSR) + pointwise grouped GEMM.
// Reference: ultralytics/nn/modules/moe/modules.py:535-704 (ES_MOE.forward,
// _sparse_forward), moe/routers.py:458-527 (DynamicRoutingLayer), moe/experts.py:280-296,
// nn/modules/_numeric.py:85-90 (stable_normalize).
#include "igemm.h"

#define RT_MAX_E 16
#define RT_MAX_HID 256

// pixels per partial-GAP workgroup: small maps are cut finer so that the launch still covers the chip
static inline int rt_chunk_pixels(int HW) { return HW >= 4096 ? 256 : 64; }

extern "C" size_t ymk_esmoe_route_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
    const int cp = rt_chunk_pixels(H * W);
    const size_t chunks = ((size_t)H * W + cp - 1) / cp;
    return (size_t)B * chunks * C * sizeof(float);
}

// stage 1: deterministic partial global-average-pool sums + finite check of x.
// A thread owns 16 bytes of channels and every rows-th pixel of the chunk; four independent loads are in
// flight per thread, the adds keep a fixed order (thread-sequential, then row 0 adds rows 1.. in order).
template <typename T>
__global__ __launch_bounds__(256) void gap_partial_kernel(const T* __restrict__ x, int HW, int C, int ldx, int cpix,
                                                         float* __restrict__ part, int* __restrict__ flags) {
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ float red[256 * VEC];
    const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
    const int ncvt = C / VEC;
    const int t = threadIdx.x;
    bool bad = false;
    for (int cv0 = 0; cv0 < ncvt; cv0 += 256) {
        const int ncv = min(256, ncvt - cv0);
        const int rows = 256 / ncv;
        const int cv = t % ncv, pr = t / ncv;
        float s[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) s[q] = 0.f;
        if (pr < rows) {
            const int p1 = min(HW, (chunk + 1) * cpix);
            const T* base = x + ((size_t)b * HW) * ldx + (cv0 + cv) * VEC;
            int p = chunk * cpix + pr;
            for (; p + 3 * rows < p1; p += 4 * rows) {
                float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                load_vec_f32(base + (size_t)p * ldx, v0);
                load_vec_f32(base + (size_t)(p + rows) * ldx, v1);
                load_vec_f32(base + (size_t)(p + 2 * rows) * ldx, v2);
                load_vec_f32(base + (size_t)(p + 3 * rows) * ldx, v3);
#pragma unroll
                for (int q = 0; q < VEC; ++q) { s[q] += v0[q]; s[q] += v1[q]; s[q] += v2[q]; s[q] += v3[q]; }
            }
            for (; p < p1; p += rows) {
                float v0[VEC];
                load_vec_f32(base + (size_t)p * ldx, v0);
#pragma unroll
                for (int q = 0; q < VEC; ++q) s[q] += v0[q];
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) { bad |= !isfinite(s[q]); red[t * VEC + q] = s[q]; }
        __syncthreads();
        if (pr == 0) {
            for (int r = 1; r < rows; ++r) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) s[q] += red[(r * ncv + cv) * VEC + q];
            }
            float* o = part + ((size_t)b * nchunk + chunk) * C + (cv0 + cv) * VEC;
#pragma unroll
            for (int q = 0; q < VEC; q += 4) store4(o + q, s[q], s[q + 1], s[q + 2], s[q + 3]);
        }
        __syncthreads();
    }
    // NaN/Inf anywhere in x makes its partial sum non-finite (finite inputs cannot
    // overflow an fp32 sum of <= 256 bf16/fp32 activations in practice): this folds
    // _validate_router_input's isnan/isinf scan (routers.py:51) into the GAP read.
    if (bad) atomicOr(flags, YMK_FLAG_NONFINITE_INPUT);
}

// stage 2: one workgroup per image: finish the mean, run the two 1x1 layers, softmax,
// hard top-k, stable_normalize, threshold pruning + renormalisation.
__global__ __launch_bounds__(256) void route_finalize_kernel(
    const float* __restrict__ part, int nchunk, int HW, int C, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, int hidden, int E,
    int top_k, float thr, float* __restrict__ route_w, float* __restrict__ gate_w, int* __restrict__ sel,
    int* __restrict__ flags) {
    extern __shared__ float sm[];  // pooled[C], h[hidden], logits[E], w2[E][hidden]
    float* pooled = sm;
    float* h = sm + C;
    float* logits = h + hidden;
    float* sw2 = logits + E;
    const int b = blockIdx.x, t = threadIdx.x;
    // second-layer weights: fetched by all threads up front (the per-expert dot product below is a sequential
    // fmaf chain in a fixed order; reading it from global memory costs one exposed L2 round trip per term)
    for (int i = t; i < E * hidden; i += 256) sw2[i] = w2[i];
    for (int c = t; c < C; c += 256) {
        // fixed-order sum of the per-chunk partials; sixteen independent loads are in flight at a time
        const float* pp = part + (size_t)b * nchunk * C + c;
        float s = 0.f;
        int k = 0;
        for (; k + 16 <= nchunk; k += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = pp[(size_t)(k + q) * C];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += v[q];
        }
        for (; k < nchunk; ++k) s += pp[(size_t)k * C];
        if (!isfinite(s)) atomicOr(flags, YMK_FLAG_NONFINITE_INPUT);   // (partials written by a producer kernel carry no flag of their own)
        pooled[c] = s / (float)HW;
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    // four hidden units per wave at a time: their weight rows are independent loads in flight together (one unit
    // at a time exposes a global-load latency per unit); per-unit arithmetic order is unchanged
    for (int j0 = wave * 4; j0 < hidden; j0 += 16) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        int c = lane;
        for (; c + 192 < C; c += 256) {   // four column steps x four units = 16 independent loads in flight
            float wv[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) wv[q][u] = j0 + u < hidden ? w1[(size_t)(j0 + u) * C + c + q * 64] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pc = pooled[c + q * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) s4[u] = fmaf(wv[q][u], pc, s4[u]);
            }
        }
        for (; c < C; c += 64) {
            const float pc = pooled[c];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < hidden) s4[u] = fmaf(w1[(size_t)(j0 + u) * C + c], pc, s4[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s = s4[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0 && j0 + u < hidden) h[j0 + u] = silu_exact(s + b1[j0 + u]);
        }
    }
    __syncthreads();
    if (t < E) {
        float s = 0.f;
        for (int j = 0; j < hidden; ++j) s = fmaf(sw2[t * hidden + j], h[j], s);
        logits[t] = s + b2[t];
    }
    __syncthreads();
    // the serial decision tail indexes its small arrays dynamically: kept in LDS (as private arrays they would live in
    // scratch memory and every access would be a dependent global round trip)
    __shared__ float p[RT_MAX_E], rw[RT_MAX_E];
    __shared__ int idx[RT_MAX_E], used[RT_MAX_E], keep[RT_MAX_E];
    if (t == 0) {
        bool fin = true;
        float mx = -INFINITY;
        for (int e = 0; e < E; ++e) {
            fin &= isfinite(logits[e]);
            p[e] = fminf(fmaxf(logits[e], -30.f), 30.f);
            mx = fmaxf(mx, p[e]);
        }
        if (!fin) atomicOr(flags, YMK_FLAG_NONFINITE_LOGITS);
        float den = 0.f;
        for (int e = 0; e < E; ++e) { p[e] = expf(p[e] - mx); den += p[e]; }
        for (int e = 0; e < E; ++e) p[e] = p[e] / den;
        // hard top-k (descending, ties -> lower index), renormalised over the selected set
        for (int e = 0; e < E; ++e) used[e] = 0;
        float vsum = 0.f;
        for (int k = 0; k < top_k; ++k) {
            int best = -1;
            for (int e = 0; e < E; ++e)
                if (!used[e] && (best < 0 || p[e] > p[best])) best = e;
            used[best] = 1;
            idx[k] = best;
            vsum += p[best];
        }
        vsum = fmaxf(vsum, 1e-6f);
        for (int e = 0; e < E; ++e) rw[e] = 0.f;
        for (int k = 0; k < top_k; ++k) rw[idx[k]] = p[idx[k]] / vsum;
        // sparse dispatch decision (modules.py:665-684): importance == rw (spatially constant).
        // thr < 0 selects the DENSE forward (modules.py:648-656; use_sparse_inference=False): every expert is summed with
        // its routing weight, nothing pruned or renormalised — experts outside the top-k set have weight exactly 0 and
        // are skipped (their term is 0 * finite).
        const bool dense = top_k >= E || thr < 0.f;
        for (int e = 0; e < E; ++e) keep[e] = 0;
        if (top_k >= E) {
            for (int e = 0; e < E; ++e) keep[e] = 1;  // every expert, unpruned
        } else {
            for (int k = 0; k < top_k; ++k)
                keep[idx[k]] = (k == 0) || !(thr > 0.f) || (rw[idx[k]] >= thr);
        }
        float nsum = 0.f;
        for (int e = 0; e < E; ++e) nsum += keep[e] ? rw[e] : 0.f;
        nsum = dense ? 1.0f : fmaxf(nsum, 1.1920929e-07f);
        int ns = 0;
        for (int e = 0; e < E; ++e) {
            route_w[b * E + e] = rw[e];
            const float g = keep[e] ? (dense ? rw[e] : rw[e] / nsum) : 0.f;
            gate_w[b * E + e] = g;
            if (keep[e] && ns < top_k) sel[b * top_k + ns++] = e;
        }
        for (; ns < top_k; ++ns) sel[b * top_k + ns] = -1;
    }
}

// stage 3: image->expert CSR permutation, one wavefront: per 64-image chunk and expert,
// a ballot gives the member mask, popcount-of-lower-lanes the position inside the chunk.
// The same wavefront produces the eval-time state ES_MOE keeps (modules.py:706-741, moe/loss.py:16-26):
// state[e] = expert_usage_counts[e] = mean_b route_w[b][e], state[E] = load_balancing_loss = E * sum_e (u_e / max(sum u, 1e-6))^2.
__global__ __launch_bounds__(64) void route_csr_kernel(const int* __restrict__ sel, int B, int E, int top_k,
                                                      int* __restrict__ csr_off, int* __restrict__ csr_pair,
                                                      const float* __restrict__ route_w, float* __restrict__ state) {
    const int lane = threadIdx.x;
    if (state) {
        float usum = 0.f, u2 = 0.f, u[RT_MAX_E];
        for (int e = 0; e < E; ++e) {
            float s = 0.f;
            for (int b = lane; b < B; b += 64) s += route_w[b * E + e];   // fixed order: lane-strided, then butterfly
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            u[e] = s / (float)B;
            usum += u[e];
        }
        const float den = fmaxf(usum, 1e-6f);
        for (int e = 0; e < E; ++e) {
            const float un = u[e] / den;
            u2 += un * un;
            if (lane == 0) state[e] = u[e];
        }
        if (lane == 0) state[E] = (float)E * u2;
    }
    int cnt[RT_MAX_E];
    for (int e = 0; e < E; ++e) cnt[e] = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        for (int e = 0; e < E; ++e) {
            bool f = false;
            if (b < B)
                for (int s = 0; s < top_k; ++s) f |= (sel[b * top_k + s] == e);
            cnt[e] += __popcll(__ballot(f));
        }
    }
    int off[RT_MAX_E + 1];
    off[0] = 0;
    for (int e = 0; e < E; ++e) off[e + 1] = off[e] + cnt[e];
    if (lane == 0)
        for (int e = 0; e <= E; ++e) csr_off[e] = off[e];
    for (int e = 0; e < E; ++e) cnt[e] = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        for (int e = 0; e < E; ++e) {
            int slot = -1;
            if (b < B)
                for (int s = 0; s < top_k; ++s)
                    if (sel[b * top_k + s] == e) slot = s;
            const unsigned long long m = __ballot(slot >= 0);
            const int pre = __popcll(m & ((1ull << lane) - 1ull));
            if (slot >= 0) csr_pair[off[e] + cnt[e] + pre] = b * top_k + slot;
            cnt[e] += __popcll(m);
        }
    }
}

extern "C" int ymk_esmoe_route(int32_t dtype, const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                               int32_t ldx, const float* w1, const float* b1, const float* w2,
                               const float* b2, int32_t hidden, int32_t E, int32_t top_k,
                               float dynamic_threshold, float* route_w, float* gate_w, int32_t* sel,
                               int32_t* csr_off, int32_t* csr_pair, float* state, int32_t* flags, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !route_w || !gate_w || !sel || !csr_off || !csr_pair || !flags)
        return YMK_E_BADARG;
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (C % vec || ldx % vec || E < 1 || E > RT_MAX_E || top_k < 1 || top_k > E || hidden < 1 ||
        hidden > RT_MAX_HID)
        return YMK_E_BADARG;
    const int HW = H * W;
    if (B <= 0 || HW <= 0) return YMK_OK;
    if (B > 65535) return YMK_E_BADARG;
    const int cpix = rt_chunk_pixels(HW);
    const int nchunk = (HW + cpix - 1) / cpix;
    if (!workspace || workspace_bytes < ymk_esmoe_route_workspace_bytes(B, C, H, W)) return YMK_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    dim3 g1(nchunk, B), blk(256);
    if (dtype == YMK_F32)
        hipLaunchKernelGGL(gap_partial_kernel<float>, g1, blk, 0, s, (const float*)x, HW, C, ldx, cpix, part, flags);
    else if (dtype == YMK_BF16)
        hipLaunchKernelGGL(gap_partial_kernel<h16_t>, g1, blk, 0, s, (const h16_t*)x, HW, C, ldx, cpix, part, flags);
    else
        return YMK_E_BADARG;
    const size_t shm = (size_t)(C + hidden + E + E * hidden) * sizeof(float);
    hipLaunchKernelGGL(route_finalize_kernel, dim3(B), blk, shm, s, part, nchunk, HW, C, w1, b1, w2, b2, hidden,
                       E, top_k, dynamic_threshold, route_w, gate_w, sel, flags);
    hipLaunchKernelGGL(route_csr_kernel, dim3(1), dim3(64), 0, s, sel, B, E, top_k, csr_off, csr_pair, route_w, state);
    return ymk_launch_status();
}

// The router on per-chunk channel sums the PRODUCER of x already wrote (ymk_c3k2_fused_pooled): stage 1 — a full read of x — is skipped.
// part fp32 [B][nchunk][C]: any partition of each image's pixels into nchunk chunks (sum over chunks = sum over H * W pixels).
extern "C" int ymk_esmoe_route_pooled(const float* part, int32_t nchunk, int32_t B, int32_t H, int32_t W, int32_t C, const float* w1,
                                      const float* b1, const float* w2, const float* b2, int32_t hidden, int32_t E, int32_t top_k,
                                      float dynamic_threshold, float* route_w, float* gate_w, int32_t* sel, int32_t* csr_off,
                                      int32_t* csr_pair, float* state, int32_t* flags, void* stream) {
    if (!part || !w1 || !b1 || !w2 || !b2 || !route_w || !gate_w || !sel || !csr_off || !csr_pair || !flags) return YMK_E_BADARG;
    if (nchunk < 1 || C % 4 || E < 1 || E > RT_MAX_E || top_k < 1 || top_k > E || hidden < 1 || hidden > RT_MAX_HID) return YMK_E_BADARG;
    const int HW = H * W;
    if (B <= 0 || HW <= 0) return YMK_OK;
    if (B > 65535) return YMK_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t shm = (size_t)(C + hidden + E + E * hidden) * sizeof(float);
    hipLaunchKernelGGL(route_finalize_kernel, dim3(B), dim3(256), shm, s, part, nchunk, HW, C, w1, b1, w2, b2, hidden, E, top_k,
                       dynamic_threshold, route_w, gate_w, sel, flags);
    hipLaunchKernelGGL(route_csr_kernel, dim3(1), dim3(64), 0, s, sel, B, E, top_k, csr_off, csr_pair, route_w, state);
    return ymk_launch_status();
}

// ---------------------------------------------------------------------------
// Pointwise stage: grouped GEMM over the retained experts of each image.
// One workgroup = (image, pixel tile, cout tile); it loops over the image's
// slots, so the output tile is produced once, in registers, without atomics
// or a zero-initialised accumulator in HBM (the reference's index_add_, :702).
// ---------------------------------------------------------------------------
struct MoePwArgs {
    const void* dw;
    const void* pw_w;
    const float* pw_b;
    const float* nscale;
    const float* nshift;
    const int* sel;
    const float* gate;
    void* y;
    int B, HW, C, Cout, Kpad, E, top_k, ldy, tiles;
};

template <typename T, int BCO, int BPX, int WCO, int WPX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void moe_pw_kernel(MoePwArgs a) {
    using G = IGemm<T, BCO, BPX, WCO, WPX, 1>;
    __shared__ u32x4 smem[G::SMEM_U4];
    const int t = threadIdx.x;
    const int ncot = (a.Cout + BCO - 1) / BCO;  // cout tile = fast block index (L2 reuse of the pixel tile)
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = (lid / ncot) / a.tiles, tile = (lid / ncot) % a.tiles;
    const int co0 = (int)(lid % ncot) * BCO;
    const int m0 = tile * BPX;

    typename G::Rows rows;
#pragma unroll
    for (int i = 0; i < G::NB; ++i) {
        const int m = m0 + (t >> 3) + i * G::RPP;
        rows.ok[i] = m < a.HW;
        rows.pix[i] = rows.ok[i] ? m : 0; rows.iy0[i] = 0; rows.ix0[i] = 0;  // KS==1: pix = input pixel index
    }
    const int lane = t & 63, wave = t >> 6;
    const int wco = wave / WPX;
    constexpr bool PRECISE = sizeof(T) == 4;

    f32x4 out[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j) out[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s = 0; s < a.top_k; ++s) {
        const int pair = b * a.top_k + s;
        const int e = a.sel[pair];
        if (e < 0) continue;  // workgroup-uniform
        const float g = a.gate[b * a.E + e];
        f32x4 acc[G::TM][G::TN];
#pragma unroll
        for (int i = 0; i < G::TM; ++i)
#pragma unroll
            for (int j = 0; j < G::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* xin = reinterpret_cast<const T*>(a.dw) + (size_t)pair * a.HW * a.C;
        const T* wt = reinterpret_cast<const T*>(a.pw_w) + ((size_t)e * a.Cout + co0) * a.Kpad;
        G::run(acc, xin, a.C, 1, a.HW, a.C, rows, wt, a.Kpad, a.Cout - co0, smem);
#pragma unroll
        for (int i = 0; i < G::TM; ++i) {
            const int co = co0 + (wco * G::TM + i) * 16 + (lane >> 4) * 4;
            if (co >= a.Cout) continue;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.pw_b + (size_t)e * a.Cout + co);
#pragma unroll
            for (int j = 0; j < G::TN; ++j) {
                float v0 = acc[i][j].x + bv.x, v1 = acc[i][j].y + bv.y;
                float v2 = acc[i][j].z + bv.z, v3 = acc[i][j].w + bv.w;
                if (PRECISE) {
                    v0 = silu_exact(v0); v1 = silu_exact(v1); v2 = silu_exact(v2); v3 = silu_exact(v3);
                } else {
                    v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3);
                }
                out[i][j].x += v0 * g; out[i][j].y += v1 * g;
                out[i][j].z += v2 * g; out[i][j].w += v3 * g;
            }
        }
    }
    // trailing ES_MOE.norm: BatchNorm(eval) + SiLU, then the LDS-staged coalesced store
    f32x4 sc[G::TM], sh[G::TM];
#pragma unroll
    for (int i = 0; i < G::TM; ++i) {
        const int co = co0 + (wco * G::TM + i) * 16 + (lane >> 4) * 4;
        const bool ok = co < a.Cout;
        sc[i] = ok ? *reinterpret_cast<const f32x4*>(a.nscale + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        sh[i] = ok ? *reinterpret_cast<const f32x4*>(a.nshift + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto val = [&](int i, int j, int r) {
        const float v = out[i][j][r] * sc[i][r] + sh[i][r];
        return PRECISE ? silu_exact(v) : silu_f(v);
    };
    auto emit = [&](int px, int co_l, const f32x4& v) {
        const int m = m0 + px, co = co0 + co_l;
        if (m >= a.HW || co >= a.Cout) return;
        store4(reinterpret_cast<T*>(a.y) + ((size_t)b * a.HW + m) * a.ldy + co, v.x, v.y, v.z, v.w);
    };
    G::epilogue(smem, val, emit);
}


// ---------------------------------------------------------------------------
// Streaming pointwise stage (Cout > 64, K <= 4 x 128 bytes): persistent workgroups, expert weights resident in LDS.
// Work is the image-major list of (image, pixel tile, retained expert) steps; every workgroup takes a contiguous
// range of whole (image, tile) items of equal step count, so an expert's [128 cout][K] weight tile is loaded once
// per image run (once per tile for images that kept two experts: the tile order alternates so that the expert
// left in LDS by one tile is the first one the next tile needs).  The next step's activations are prefetched into
// registers while the current step is multiplied; the gated SiLU of each expert accumulates in registers and the
// trailing BatchNorm + SiLU is applied before the only store.  Same arithmetic as moe_pw_kernel.
// ---------------------------------------------------------------------------
#define PWS_MAXB 1024
#define PWS_MAXSEL 2048

template <typename T, int KG>
__global__ __launch_bounds__(512) void moe_pw_stream_kernel(MoePwArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BK = 8 * VEC;  // elements per 128-byte K group
    constexpr int RS = KG * 8;   // u32x4 per staged row
    constexpr bool PRECISE = sizeof(T) == 4;
    __shared__ u32x4 sW[128 * RS];
    __shared__ u32x4 sA[128 * RS];
    __shared__ int s_sel[PWS_MAXSEL];
    __shared__ int s_pref[PWS_MAXB + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wco = wave >> 1, wpx = wave & 1;   // 8 waves: 4 (cout, 32 each) x 2 (pixels, 64 each)
    const int srow = t >> 3, cq = t & 7;         // 64 staged rows per pass
    const int fr = lane & 15, fc = lane >> 4;
    const int ncot = (a.Cout + 127) / 128;
    // cout tiles of one pixel range sit on the same XCD (shared L2 for the activation tile)
    int ct, j;
    if (gridDim.x % (8 * ncot) == 0) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        ct = q % ncot; j = (q / ncot) * 8 + xcd;
    } else {
        ct = blockIdx.x % ncot; j = blockIdx.x / ncot;
    }
    const int co0 = ct * 128;
    const int nblk = gridDim.x / ncot;
    const int tiles = a.tiles;
    const T* dw = reinterpret_cast<const T*>(a.dw);
    auto swz = [](int row, int c) { return (c & ~7) | ((c & 7) ^ (row & 7)); };

    for (int i = t; i < a.B * a.top_k; i += 512) s_sel[i] = a.sel[i];
    __syncthreads();
    if (t == 0) {  // steps per image (an image without a retained expert still owns one, empty, step)
        int run = 0;
        for (int b = 0; b < a.B; ++b) {
            s_pref[b] = run;
            int nv = 0;
            for (int q = 0; q < a.top_k; ++q) nv += s_sel[b * a.top_k + q] >= 0;
            run += nv > 0 ? nv : 1;
        }
        s_pref[a.B] = run;
    }
    __syncthreads();
    const int64_t nitems = (int64_t)a.B * tiles;
    const int64_t total = (int64_t)s_pref[a.B] * tiles;
    const int64_t per = (total + nblk - 1) / nblk;
    auto locate = [&](int64_t target) -> int64_t {  // first item whose first step index is >= target
        if (target >= total) return nitems;
        int lo = 0, hi = a.B - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((int64_t)s_pref[mid] * tiles <= target) lo = mid; else hi = mid - 1;
        }
        const int nv = s_pref[lo + 1] - s_pref[lo];
        const int64_t local = target - (int64_t)s_pref[lo] * tiles;
        return (int64_t)lo * tiles + (local + nv - 1) / nv;
    };
    int64_t it = locate((int64_t)j * per);
    const int64_t it_end = locate((int64_t)(j + 1) * per);
    if (it >= it_end) return;

    struct Step { int b, tile, pair, e, nv; };
    auto decode = [&](int64_t item, int k) {
        Step st;
        st.b = (int)(item / tiles); st.tile = (int)(item % tiles);
        st.nv = s_pref[st.b + 1] - s_pref[st.b];
        const int slot = (st.tile & 1) ? st.nv - 1 - k : k;
        st.pair = st.b * a.top_k + slot;
        st.e = s_sel[st.pair];
        return st;
    };
    u32x4 ra[2][KG];
    auto gload = [&](const Step& st) {
        const T* xin = dw + (size_t)st.pair * a.HW * a.C;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = st.tile * 128 + srow + i * 64;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (st.e >= 0 && m < a.HW && g * BK + cq * VEC < a.C)
                    v = *reinterpret_cast<const u32x4*>(xin + (size_t)m * a.C + g * BK + cq * VEC);
                ra[i][g] = v;
            }
        }
    };

    int k = 0, e_lds = -1;
    Step cur = decode(it, 0);
    gload(cur);
    f32x4 part[2][4];
    while (true) {
        int64_t nit = it;
        int nk = k + 1;
        if (nk >= cur.nv) { nk = 0; ++nit; }
        const bool more = nit < it_end;
        const bool wneed = cur.e >= 0 && cur.e != e_lds;  // workgroup-uniform
        u32x4 rw[2][KG];
        if (wneed) {
            const T* wt = reinterpret_cast<const T*>(a.pw_w) + ((size_t)cur.e * a.Cout + co0) * a.Kpad;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = srow + i * 64;
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (co0 + r < a.Cout) v = *reinterpret_cast<const u32x4*>(wt + (size_t)r * a.Kpad + g * BK + cq * VEC);
                    rw[i][g] = v;
                }
            }
        }
        __syncthreads();  // the previous step's fragment reads are finished
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow + i * 64;
#pragma unroll
            for (int g = 0; g < KG; ++g) sA[r * RS + swz(r, g * 8 + cq)] = ra[i][g];
        }
        if (wneed) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = srow + i * 64;
#pragma unroll
                for (int g = 0; g < KG; ++g) sW[r * RS + swz(r, g * 8 + cq)] = rw[i][g];
            }
            e_lds = cur.e;
        }
        __syncthreads();
        Step nxt = cur;
        if (more) { nxt = decode(nit, nk); gload(nxt); }  // in flight during the MFMA + epilogue below

        const bool first = k == 0, last = k == cur.nv - 1;
        if (cur.e >= 0) {
            f32x4 bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int co = co0 + (wco * 2 + i) * 16 + fc * 4;
                bv[i] = co < a.Cout ? *reinterpret_cast<const f32x4*>(a.pw_b + (size_t)cur.e * a.Cout + co)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float gw = a.gate[cur.b * a.E + cur.e];
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 af[2], bfr[4];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int r = (wco * 2 + i) * 16 + fr;
                        af[i] = sW[r * RS + swz(r, g * 8 + kk * 4 + fc)];
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int r = (wpx * 4 + jj) * 16 + fr;
                        bfr[jj] = sA[r * RS + swz(r, g * 8 + kk * 4 + fc)];
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) mma16<T>(acc[i][jj], af[i], bfr[jj]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u = acc[i][jj][r] + bv[i][r];
                        v[r] = (PRECISE ? silu_exact(u) : silu_f(u)) * gw;
                    }
                    if (first) part[i][jj] = v; else part[i][jj] += v;
                }
        } else if (first) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) part[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (last) {  // trailing ES_MOE.norm: BatchNorm(eval) + SiLU
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int co = co0 + (wco * 2 + i) * 16 + fc * 4;
                if (co >= a.Cout) continue;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(a.nscale + co);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(a.nshift + co);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int m = cur.tile * 128 + (wpx * 4 + jj) * 16 + fr;
                    if (m >= a.HW) continue;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u = part[i][jj][r] * sc[r] + sh[r];
                        v[r] = PRECISE ? silu_exact(u) : silu_f(u);
                    }
                    store4(reinterpret_cast<T*>(a.y) + ((size_t)cur.b * a.HW + m) * a.ldy + co, v[0], v[1], v[2], v[3]);
                }
            }
        }
        if (!more) break;
        it = nit; k = nk; cur = nxt;
    }
}

// ---------------------------------------------------------------------------
// Lean streaming pointwise stage (Cout % 128 == 0, K == Kpad <= 4 x 128 bytes, E <= 8).
// Same work list, tile shape and arithmetic as moe_pw_stream_kernel, but that kernel spent ~1100 instructions per wave and
// step (index decode with divisions and dependent LDS reads, 64-bit address arithmetic, tail masks) around 32 MFMAs: with two
// waves per SIMD it was issue-bound at ~6 us per step (tools ablation: 4 us per step with loads, stores, MFMA and SiLU all
// removed).  Here every workgroup decodes its steps ONCE, in parallel, into an LDS table of 16-byte descriptors
// {activation row, output row, expert | first | last, gate weight}; the loop body reads one descriptor, addresses with a
// uniform base + a per-lane constant, keeps up to four expert weight tiles resident in LDS (an image that kept two experts
// alternates between them tile after tile), prefetches the next weight tile together with the next activations, starts the
// accumulators at the expert bias, and stages rows at a 32-byte-padded pitch (conflict-free ds_read_b128 without a swizzle).
// Weight rows are staged in the order that leaves a lane with EIGHT consecutive output channels after its two MFMA row
// blocks (16-byte stores, 64 contiguous bytes per pixel and wave instead of 32).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pwl_store8(float* p, const float (&v)[8]) {
    store4(p, v[0], v[1], v[2], v[3]);
    store4(p + 4, v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void pwl_store8(h16_t* p, const float (&v)[8]) { store_vec_f32(p, v); }
#define PWL_MAXSTEP 512
#define PWL_MAXE 8
__host__ __device__ constexpr int pwl_slots(int kg) { return kg == 1 ? 4 : kg == 2 ? 2 : 1; }
__host__ __device__ constexpr int pwl_pitch(int kg) { return kg * 8 + 2; }   // u32x4 per staged row

template <typename T, int KG>
__global__ __launch_bounds__(512) void moe_pw_lean_kernel(MoePwArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int RP = pwl_pitch(KG);
    constexpr int NSLOT = pwl_slots(KG);
    constexpr int TILE = 128 * RP;
    constexpr bool PRECISE = sizeof(T) == 4;
    extern __shared__ u32x4 pwl_dyn[];
    u32x4* const sWall = pwl_dyn;
    u32x4* const sA = pwl_dyn + NSLOT * TILE;
    int* const s_sel = reinterpret_cast<int*>(sA);   // the two prologue tables live in the (not yet used) activation tile
    int* const s_pref = s_sel + PWS_MAXSEL;
    __shared__ int4 s_step[PWL_MAXSTEP];
    __shared__ float s_bias[PWL_MAXE * 128];
    __shared__ float s_norm[256];
    __shared__ int s_wsum[8];
    __shared__ int s_rng[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wco = wave >> 1, wpx = wave & 1;   // 8 waves: 4 (cout, 32 each) x 2 (pixels, 64 each)
    const int srow = t >> 3, cq = t & 7;         // 64 staged rows per pass
    const int fr = lane & 15, fc = lane >> 4;
    const int ncot = a.Cout >> 7;
    const int tail = a.HW & 127;                 // rows of the last tile of an image (0: whole)
    int ct, j;   // cout tiles of one pixel range sit on the same XCD (shared L2 for the activation tile)
    if (gridDim.x % (8 * ncot) == 0) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        ct = q % ncot; j = (q / ncot) * 8 + xcd;
    } else {
        ct = blockIdx.x % ncot; j = blockIdx.x / ncot;
    }
    const int co0 = ct * 128;
    const int nblk = gridDim.x / ncot;
    const int tiles = a.tiles;

    for (int i = t; i < a.B * a.top_k; i += 512) s_sel[i] = a.sel[i];
    for (int i = t; i < a.E * 128; i += 512) s_bias[i] = a.pw_b[(size_t)(i >> 7) * a.Cout + co0 + (i & 127)];
    if (t < 256) s_norm[t] = t < 128 ? a.nscale[co0 + t] : a.nshift[co0 + t - 128];
    __syncthreads();
    {   // exclusive prefix of the steps per image (an image without a retained expert still owns one, empty, step): two images per thread
        int c[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = 2 * t + q;
            int nv = 0;
            if (b < a.B) {
                for (int k = 0; k < a.top_k; ++k) nv += s_sel[b * a.top_k + k] >= 0;
                nv = nv > 0 ? nv : 1;
            }
            c[q] = nv;
        }
        const int own = c[0] + c[1];
        int inc = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(inc, d, 64);
            if (lane >= d) inc += v;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int ex = inc - own;
        for (int w = 0; w < wave; ++w) ex += s_wsum[w];
        if (2 * t < a.B) s_pref[2 * t] = ex;
        if (2 * t + 1 < a.B) s_pref[2 * t + 1] = ex + c[0];
        if (2 * t == a.B - 1 || 2 * t + 1 == a.B - 1) s_pref[a.B] = ex + own;
    }
    __syncthreads();
    {   // this workgroup's range of steps, cut at whole (image, tile) items
        const int total = s_pref[a.B] * tiles;
        const int per = (total + nblk - 1) / nblk;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int target = (j + q) * per;
            if (target >= total) {
                if (t == 0) s_rng[q] = total;
            } else {
                for (int b = t; b < a.B; b += 512) {
                    const int lo = s_pref[b] * tiles, hi = s_pref[b + 1] * tiles;
                    if (lo <= target && target < hi) {
                        const int c = s_pref[b + 1] - s_pref[b];
                        s_rng[q] = lo + (target - lo + c - 1) / c * c;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int g0 = s_rng[0];
    const int n = s_rng[1] - g0;
    if (n <= 0) return;
    for (int i = t; i < n; i += 512) {   // n <= PWL_MAXSTEP (launcher)
        const int g = g0 + i;
        int lo = 0, hi = a.B - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_pref[mid] * tiles <= g) lo = mid; else hi = mid - 1;
        }
        const int b = lo, c = s_pref[b + 1] - s_pref[b];
        const int local = g - s_pref[b] * tiles;
        const int tile = local / c, k = local - tile * c;
        const int pair = b * a.top_k + ((tile & 1) ? c - 1 - k : k);   // the expert left in LDS by one tile is the first the next tile needs
        const int e = s_sel[pair];
        int4 d;
        d.x = pair * a.HW + tile * 128;
        d.y = b * a.HW + tile * 128;
        d.z = (e & 0xffff) | (k == 0 ? 0x10000 : 0) | (k == c - 1 ? 0x20000 : 0) | (tail && tile == tiles - 1 ? 0x40000 : 0);
        d.w = e >= 0 ? __float_as_int(a.gate[b * a.E + e]) : 0;
        s_step[i] = d;
    }
    __syncthreads();   // s_sel / s_pref are dead from here on: the activation tile may be written

    const T* const dw = reinterpret_cast<const T*>(a.dw);
    const T* const pw = reinterpret_cast<const T*>(a.pw_w) + (size_t)co0 * a.Kpad;
    T* const yb = reinterpret_cast<T*>(a.y) + co0;
    const int w_off = srow * a.Kpad + cq * VEC;
    const int st_off = srow * RP + cq;                // LDS: row srow (+64), chunk g * 8 + cq
    // weight row c of a wave's 32 = f2 * 8 + q * 4 + f0 goes to MFMA row block q, row f2 * 4 + f0
    const int sw_off = ((srow & ~31) + ((srow >> 2) & 1) * 16 + ((srow >> 3) & 3) * 4 + (srow & 3)) * RP + cq;
    const int fa_off = ((wco * 2) * 16 + fr) * RP + fc;   // weight fragment rows (wco*2 + i) * 16 + fr
    const int fb_off = ((wpx * 4) * 16 + fr) * RP + fc;   // pixel fragment rows (wpx*4 + jj) * 16 + fr
    const int o_off = ((wpx * 4) * 16 + fr) * a.ldy + wco * 32 + fc * 8;
    auto fetch = [&](int i) {
        const int4 d = s_step[i];
        int4 r;
        r.x = __builtin_amdgcn_readfirstlane(d.x); r.y = __builtin_amdgcn_readfirstlane(d.y);
        r.z = __builtin_amdgcn_readfirstlane(d.z); r.w = __builtin_amdgcn_readfirstlane(d.w);
        return r;
    };
    auto expert_of = [](const int4& d) { return (int)(short)(d.z & 0xffff); };

    // Activation tiles are requested TWO steps ahead (PWL_AHEAD; register sets ra / rb alternate, the loop below is unrolled by two so
    // that both stay statically indexed): with one step of lead a tile had exactly one step's arithmetic (~1-1.5 us) to arrive while every
    // CU streams — the wait at the top of the next step was exposed (SQ: waves parked 32 %, profiles/r04_sq_summary.txt).
#ifndef PWL_AHEAD
#define PWL_AHEAD 1
#endif
    u32x4 ra[2][KG], rb[2][KG], rw[2][KG];
    auto gload_into = [&](const int4& d, u32x4 (&dst)[2][KG]) {
        const T* xin = dw + (size_t)d.x * a.C + cq * VEC;
        const int last_row = ((d.z & 0x40000) ? tail : 128) - 1;   // rows past the image repeat its last row (never stored)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = min(srow + i * 64, last_row);
#pragma unroll
            for (int g = 0; g < KG; ++g) dst[i][g] = *reinterpret_cast<const u32x4*>(xin + r * a.C + g * 8 * VEC);
        }
    };
    auto gload = [&](const int4& d) { gload_into(d, ra); };
    auto wload = [&](int e) {
        const T* wt = pw + (size_t)e * a.Cout * a.Kpad;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < KG; ++g)
                rw[i][g] = *reinterpret_cast<const u32x4*>(wt + w_off + i * 64 * a.Kpad + g * 8 * VEC);
    };
    int slot_e[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) slot_e[q] = -1;
    int mru = 0;
    // slot of expert e; a miss picks a victim (never the slot of the step before), loads the tile into rw and reports it pending
    auto claim = [&](int e, bool& pend) {
        int sl = -1;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) if (slot_e[q] == e) sl = q;
        pend = sl < 0;
        if (pend) {
            sl = NSLOT == 1 ? 0 : (mru + 1) % NSLOT;
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) if (q == sl) slot_e[q] = e;
            wload(e);
        }
        mru = sl;
        return sl;
    };

    int4 cur = fetch(0);
    bool pend = false;
    int cslot = 0;
    if (expert_of(cur) >= 0) { gload(cur); cslot = claim(expert_of(cur), pend); }
    if (PWL_AHEAD == 2 && n > 1) {
        const int4 d1 = fetch(1);
        if (expert_of(d1) >= 0) gload_into(d1, rb);
    }
    f32x4 part[2][4];
    // one step; `mine` holds this step's activation rows, and is refilled with the rows of step i + PWL_AHEAD once they are in LDS
    auto step = [&](int i, u32x4 (&mine)[2][KG]) {
        __syncthreads();   // the previous step's fragment reads are finished
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < KG; ++g) sA[st_off + q * 64 * RP + g * 8] = mine[q][g];
        if (pend) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int g = 0; g < KG; ++g) sWall[cslot * TILE + sw_off + q * 64 * RP + g * 8] = rw[q][g];
        }
        __syncthreads();
        int4 nx = cur;
        int nslot = cslot;
        pend = false;
        if (i + 1 < n) {
            nx = fetch(i + 1);
            if (expert_of(nx) >= 0) {
                if (PWL_AHEAD == 1) gload_into(nx, mine);
                nslot = claim(expert_of(nx), pend);      // (the weight tile of a miss keeps one step of lead: misses are rare)
            }
        }
        if (PWL_AHEAD == 2 && i + 2 < n) {
            const int4 d2 = fetch(i + 2);
            if (expert_of(d2) >= 0) gload_into(d2, mine);
        }
        const int e = expert_of(cur);
        const bool first = cur.z & 0x10000, last = cur.z & 0x20000;
        if (e >= 0) {
            const u32x4* sW = sWall + cslot * TILE;
            const float gw = __int_as_float(cur.w);
            f32x4 acc[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(s_bias + e * 128 + wco * 32 + fc * 8 + q * 4);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[q][jj] = bv;
            }
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 af[2], bfr[4];
#pragma unroll
                    for (int q = 0; q < 2; ++q) af[q] = sW[fa_off + q * 16 * RP + g * 8 + kk * 4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bfr[jj] = sA[fb_off + jj * 16 * RP + g * 8 + kk * 4];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) mma16<T>(acc[q][jj], af[q], bfr[jj]);
                }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // 16-bit builds: the weighted expert output is a ROUNDED product that is then added (modules.py:697-702:
                        // `expert_out * w`, index_add_) — no FMA contraction, so the sum of an image's two experts does not depend on
                        // the order they are visited in (it alternates tile by tile here; the one-kernel form of round 4, tools/micro/parked/esfused.hip.txt, visits them in slot order
                        // and must produce the same bits).  fp32 keeps the arithmetic its fixtures were recorded with.
                        if constexpr (PRECISE) v[r] = silu_exact(acc[q][jj][r]) * gw;
                        else v[r] = ymk_mul_rn(silu_f(acc[q][jj][r]), gw);
                    }
                    if (first) part[q][jj] = v;
                    else if constexpr (PRECISE) part[q][jj] += v;
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[q][jj][r] = ymk_add_rn(part[q][jj][r], v[r]);
                    }
                }
        } else if (first) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) part[q][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (last) {  // trailing ES_MOE.norm: BatchNorm(eval) + SiLU
            T* const yo = yb + (size_t)cur.y * a.ldy;
            const int rows = (cur.z & 0x40000) ? tail : 128;
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                sc[q] = *reinterpret_cast<const f32x4*>(s_norm + wco * 32 + fc * 8 + q * 4);
                sh[q] = *reinterpret_cast<const f32x4*>(s_norm + 128 + wco * 32 + fc * 8 + q * 4);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u = part[q][jj][r] * sc[q][r] + sh[q][r];
                        v[q * 4 + r] = PRECISE ? silu_exact(u) : silu_f(u);
                    }
                if ((wpx * 4 + jj) * 16 + fr < rows) pwl_store8(yo + o_off + jj * 16 * a.ldy, v);
            }
        }
        cur = nx; cslot = nslot;
    };
    for (int i = 0; i < n; i += 2) {
        step(i, ra);
        if (i + 1 < n) {
            if (PWL_AHEAD == 2) step(i + 1, rb); else step(i + 1, ra);
        }
    }
}

template <typename T>
static bool launch_pw_lean(MoePwArgs a, hipStream_t s) {
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    const int kg = a.Kpad / BK;
    if (ymk_disabled() & (YMK_OFF_MOE_STREAM | YMK_OFF_MOE_LEAN)) return false;
    if (a.Kpad % BK || kg < 1 || kg > 4 || a.C != a.Kpad || a.Cout % 128 || a.E > PWL_MAXE || a.B > PWS_MAXB ||
        (int64_t)a.B * a.top_k > PWS_MAXSEL || (int64_t)a.B * a.top_k * a.HW >= (1ll << 31) ||
        (a.ldy * sizeof(T)) % 16 || reinterpret_cast<uintptr_t>(a.y) % 16)   // 16-byte output stores
        return false;
    const int ncot = a.Cout / 128;
    a.tiles = (a.HW + 127) / 128;
    int64_t nblk = 256 / ncot;
    if (nblk < 1) nblk = 1;
    if (nblk > (int64_t)a.B * a.tiles) nblk = (int64_t)a.B * a.tiles;
    // the step table holds one workgroup's share: at most ceil(all steps / workgroups) rounded up to a whole (image, tile) item
    if (ceil_div64((int64_t)a.B * a.top_k * a.tiles, nblk) + a.top_k > PWL_MAXSTEP) return false;
    dim3 grid((unsigned)(nblk * ncot)), blk(512);
    auto lds = [](int g) { return (size_t)(pwl_slots(g) + 1) * 128 * pwl_pitch(g) * sizeof(u32x4); };
    static YmkOncePerDevice attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_pw_lean_kernel<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(1));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_pw_lean_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(2));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_pw_lean_kernel<T, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(3));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_pw_lean_kernel<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(4));
        attr_once.done();
    }
    switch (kg) {
        case 1: hipLaunchKernelGGL((moe_pw_lean_kernel<T, 1>), grid, blk, lds(1), s, a); break;
        case 2: hipLaunchKernelGGL((moe_pw_lean_kernel<T, 2>), grid, blk, lds(2), s, a); break;
        case 3: hipLaunchKernelGGL((moe_pw_lean_kernel<T, 3>), grid, blk, lds(3), s, a); break;
        default: hipLaunchKernelGGL((moe_pw_lean_kernel<T, 4>), grid, blk, lds(4), s, a); break;
    }
    return true;
}

template <typename T>
static bool launch_pw_stream(MoePwArgs a, hipStream_t s) {
    constexpr int BK = 8 * (16 / (int)sizeof(T));
    const int kg = a.Kpad / BK;
    if (ymk_disabled() & YMK_OFF_MOE_STREAM) return false;
    if (a.Kpad % BK || kg < 1 || kg > 4 || a.Cout <= 64 || a.B > PWS_MAXB || (int64_t)a.B * a.top_k > PWS_MAXSEL) return false;
    const int ncot = (a.Cout + 127) / 128;
    a.tiles = (a.HW + 127) / 128;
    // one 8-wave workgroup per CU (two waves per SIMD overlap one wave's MFMA with the other's SiLU epilogue);
    // LDS = kg * 32 KB of tiles + 12 KB of tables
    int64_t nblk = 256 / ncot;
    if (nblk < 1) nblk = 1;
    if (nblk > (int64_t)a.B * a.tiles) nblk = (int64_t)a.B * a.tiles;
    dim3 grid((unsigned)(nblk * ncot)), blk(512);
    switch (kg) {
        case 1: hipLaunchKernelGGL((moe_pw_stream_kernel<T, 1>), grid, blk, 0, s, a); break;
        case 2: hipLaunchKernelGGL((moe_pw_stream_kernel<T, 2>), grid, blk, 0, s, a); break;
        case 3: hipLaunchKernelGGL((moe_pw_stream_kernel<T, 3>), grid, blk, 0, s, a); break;
        default: hipLaunchKernelGGL((moe_pw_stream_kernel<T, 4>), grid, blk, 0, s, a); break;
    }
    return true;
}

template <typename T>
static int launch_pw(MoePwArgs a, hipStream_t s) {
    dim3 blk(256);
    if (launch_pw_lean<T>(a, s)) return ymk_launch_status();
    if (launch_pw_stream<T>(a, s)) return ymk_launch_status();
    if (a.Cout > 64) {
        a.tiles = (a.HW + 127) / 128;
        dim3 grid(a.B * a.tiles * ((a.Cout + 127) / 128));
        hipLaunchKernelGGL((moe_pw_kernel<T, 128, 128, 2, 2>), grid, blk, 0, s, a);
    } else if (a.Cout > 32) {
        a.tiles = (a.HW + 255) / 256;
        dim3 grid(a.B * a.tiles, 1);
        hipLaunchKernelGGL((moe_pw_kernel<T, 64, 256, 1, 4>), grid, blk, 0, s, a);
    } else {
        a.tiles = (a.HW + 255) / 256;
        dim3 grid(a.B * a.tiles, 1);
        hipLaunchKernelGGL((moe_pw_kernel<T, 32, 256, 1, 4>), grid, blk, 0, s, a);
    }
    return ymk_launch_status();
}

extern "C" int ymk_esmoe_pw(int32_t dtype, const void* dw_out, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t Cout, int32_t Kpad, const void* pw_w, const float* pw_b,
                            const float* norm_scale, const float* norm_shift, int32_t E, int32_t top_k,
                            const int32_t* sel, const float* gate_w, void* y, int32_t ldy, void* stream) {
    if (!dw_out || !pw_w || !pw_b || !norm_scale || !norm_shift || !sel || !gate_w || !y) return YMK_E_BADARG;
    const int vec = dtype == YMK_BF16 ? 8 : 4;
    if (dtype != YMK_F32 && dtype != YMK_BF16) return YMK_E_BADARG;
    if (C % vec || Cout % 4 || ldy % 4 || Kpad % 64 || Kpad < C || Cout > 32 * 1024) return YMK_E_BADARG;
    if (B <= 0 || H * W <= 0) return YMK_OK;
    MoePwArgs a{dw_out, pw_w, pw_b, norm_scale, norm_shift, sel, gate_w, y, B, H * W, C, Cout, Kpad, E, top_k, ldy, 0};
    return dtype == YMK_F32 ? launch_pw<float>(a, (hipStream_t)stream) : launch_pw<h16_t>(a, (hipStream_t)stream);
}
