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
        hipLaunchKernelGGL(gap_partial_kernel<bf16_t>, g1, blk, 0, s, (const bf16_t*)x, HW, C, ldx, cpix, part, flags);
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
    return dtype == YMK_F32 ? launch_pw<float>(a, (hipStream_t)stream) : launch_pw<bf16_t>(a, (hipStream_t)stream);
}
