// Config-5 rows (MoA / MoT / gated MoE): normalisation, gates, pooling, router tails, expert gather, shuffle.
// include/ymk_mixture.h has the contracts and the reference lines.  FIRST IMPLEMENTATION, correctness-first: scalar
// element accesses with run-time dtype codes, grid-stride loops, fp32 arithmetic; every kernel here is HBM-bound and
// small next to the convolutions, vector-width loads and fusion into producers / consumers come after hardware parity.
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
__device__ __forceinline__ float act_f(float v, int act) {
    switch (act) {
        case YMK_ACT_SILU: return v / (1.0f + expf(-v));
        case YMK_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case YMK_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        default: return v;
    }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over a 256-thread workgroup; sh: 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
inline bool bad_dt(int dt) { return dt != YMK_F32 && dt != YMK_BF16; }
#ifndef YMK_MAX_BLOCKS
#define YMK_MAX_BLOCKS (256 * 64)   // grid-stride kernels: enough workgroups to fill the chip many times over
#endif
inline int blocks_for(int64_t total, int per_block = 256, int cap = YMK_MAX_BLOCKS) {
    const int64_t b = (total + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
#define GRID_STRIDE(i, total) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------ activation
__global__ __launch_bounds__(256) void activation_kernel(void* x, int dt, int ldx, int64_t npix, int C, int act) {
    const int64_t total = npix * C;
    GRID_STRIDE(i, total) {
        const int64_t p = i / C;
        const int c = (int)(i % C);
        stv(x, dt, p * ldx + c, act_f(ldv(x, dt, p * ldx + c), act));
    }
}

// ------------------------------------------------------------------------------------------------ group norm
// Statistics of one (image, group) slab, split over `nchunk` workgroups (grid.y) so that 16 images x 8 groups do not leave half
// the chip idle on a 200 MB map: each workgroup takes a run of pixels, computes ITS mean and centred sum of squares (two passes
// over a slab that stays in L2), and the partials are combined with the exact pairwise update (Chan et al.) in chunk order —
// deterministic, and as accurate as the reference's single two-pass evaluation.  nchunk == 1: final (mean, rstd) written directly.
#define GN_MAX_CHUNKS 256
__device__ __forceinline__ void gn_emit(float mean, float m2, float n, int nchunk, float eps, float* stats, float* part, int slab, int chunk) {
    if (threadIdx.x != 0) return;
    if (nchunk == 1) {
        stats[2 * slab] = mean;
        stats[2 * slab + 1] = 1.0f / sqrtf(m2 / n + eps);
    } else {
        float* o = part + ((size_t)slab * GN_MAX_CHUNKS + chunk) * 3;
        o[0] = mean; o[1] = m2; o[2] = n;
    }
}
__global__ __launch_bounds__(64) void gn_finalize_kernel(int nchunk, float eps, float* stats, const float* part) {
    const int slab = blockIdx.x;
    if (threadIdx.x != 0) return;
    const float* p = part + (size_t)slab * GN_MAX_CHUNKS * 3;
    float mean = p[0], m2 = p[1], n = p[2];
    for (int c = 1; c < nchunk; ++c) {   // (mean, M2, n) + (mean_c, M2_c, n_c)
        const float mc = p[3 * c], m2c = p[3 * c + 1], nc = p[3 * c + 2];
        const float tot = n + nc, d = mc - mean;
        mean += d * (nc / tot);
        m2 += m2c + d * d * (n * nc / tot);
        n = tot;
    }
    stats[2 * slab] = mean;
    stats[2 * slab + 1] = 1.0f / sqrtf(m2 / n + eps);
}
__global__ __launch_bounds__(256) void gn_stats_kernel(const void* x, int dt, int ldx, int HW, int C, int groups, float eps,
                                                        float* stats, float* part, int nchunk) {
    __shared__ float sh[4];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cg = C / groups;
    const int p0 = (int)((int64_t)HW * blockIdx.y / nchunk), p1 = (int)((int64_t)HW * (blockIdx.y + 1) / nchunk);
    const int64_t n = (int64_t)(p1 - p0) * cg;
    const int64_t base = ((int64_t)b * HW + p0) * ldx + (int64_t)g * cg;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += ldv(x, dt, base + (i / cg) * ldx + (i % cg));
    const float mean = block_sum(s, sh) / (float)n;
    float q = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float d = ldv(x, dt, base + (i / cg) * ldx + (i % cg)) - mean;
        q += d * d;
    }
    gn_emit(mean, block_sum(q, sh), (float)n, nchunk, eps, stats, part, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* x, int dt, int ldx, void* y, int odt, int ldy, const void* res,
                                                        int ldr, int B, int HW, int C, int groups, const float* weight,
                                                        const float* bias, const int32_t* rows, int act, const float* stats) {
    const int cg = C / groups;
    const int64_t total = (int64_t)B * HW * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int64_t p = i / C;   // b*HW + pixel
        const int b = (int)(p / HW);
        const float* st = stats + 2 * ((int64_t)b * groups + c / cg);
        float v = (ldv(x, dt, p * ldx + c) - st[0]) * st[1];
        if (weight) {
            const int64_t r = rows ? (int64_t)rows[b] * C : 0;
            v = v * weight[r + c] + bias[r + c];
        }
        v = act_f(v, act);
        if (res) v += ldv(res, odt, p * ldr + c);
        stv(y, odt, p * ldy + c, v);
    }
}

// ------------------------------------------------------------------------------------------------ layer norm
// one wave per token
__global__ __launch_bounds__(256) void ln_kernel(const void* x, int dt, int ldx, void* y, int ldy, int64_t npix, int C,
                                                  const float* weight, const float* bias, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p = wave0; p < npix; p += nwaves) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += ldv(x, dt, p * ldx + c);
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = ldv(x, dt, p * ldx + c) - mean;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        for (int c = lane; c < C; c += 64) stv(y, dt, p * ldy + c, (ldv(x, dt, p * ldx + c) - mean) * rstd * weight[c] + bias[c]);
    }
}

// ------------------------------------------------------------------------------------------------ elementwise family
__global__ __launch_bounds__(256) void eltwise_kernel(int op, int dt, const void* a, int lda, const void* b, int ldb, void* y,
                                                       int ldy, int64_t npix, int C, float alpha) {
    const int64_t total = npix * C;
    GRID_STRIDE(i, total) {
        const int64_t p = i / C;
        const int c = (int)(i % C);
        const float av = ldv(a, dt, p * lda + c), bv = ldv(b, dt, p * ldb + c);
        float r;
        if (op == YMK_ELT_MUL) r = av * bv;
        else if (op == YMK_ELT_SIGMOID_MUL) r = bv / (1.0f + expf(-av));
        else if (op == YMK_ELT_CLAMP_ADD) r = fminf(fmaxf(av, -alpha), alpha) + bv;
        else r = (1.0f - alpha) * av + alpha * bv;
        stv(y, dt, p * ldy + c, r);
    }
}
__global__ __launch_bounds__(256) void fma_gate_kernel(int dt, const void* x, int ldx, const void* a, int lda, const void* b,
                                                        int bdt, int ldb, int b_per_image, float scale, void* y, int ldy, int B,
                                                        int HW, int C) {
    const int64_t total = (int64_t)B * HW * C;
    GRID_STRIDE(i, total) {
        const int64_t p = i / C;
        const int c = (int)(i % C);
        const float bv = b_per_image ? static_cast<const float*>(b)[(p / HW) * C + c] : ldv(b, bdt, p * ldb + c);
        stv(y, dt, p * ldy + c, ldv(x, dt, p * ldx + c) + scale * ldv(a, dt, p * lda + c) * bv);
    }
}
__global__ __launch_bounds__(256) void channel_gate_kernel(int dt, const void* x, int ldx, const float* gate, void* y, int ldy,
                                                            int B, int HW, int C) {
    const int64_t total = (int64_t)B * HW * C;
    GRID_STRIDE(i, total) {
        const int64_t p = i / C;
        const int c = (int)(i % C);
        stv(y, dt, p * ldy + c, ldv(x, dt, p * ldx + c) * gate[(p / HW) * C + c]);
    }
}
struct Parts4 {
    const void* p[4];
};
__global__ __launch_bounds__(256) void weighted_sum_kernel(int dt, const float* w, int ldw, int per_image, int E, Parts4 parts,
                                                            int ldp, void* y, int ldy, int B, int HW, int C) {
    const int64_t total = (int64_t)B * HW * C;
    GRID_STRIDE(i, total) {
        const int64_t p = i / C;
        const int c = (int)(i % C);
        const float* wr = w + (per_image ? p / HW : p) * ldw;
        float r = 0.f;
        for (int e = 0; e < E; ++e) r += wr[e] * ldv(parts.p[e], dt, p * ldp + c);
        stv(y, dt, p * ldy + c, r);
    }
}
struct Pyr4 {
    const void* p[4];
    int h[4], w[4], ld[4];
};
__global__ __launch_bounds__(256) void mean_upsampled_kernel(int dt, int n, Pyr4 a, void* y, int ldy, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * H * W * C;
    const float inv = 1.0f / (float)n;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int ox = (int)(p % W);
        p /= W;
        const int oy = (int)(p % H);
        const int b = (int)(p / H);
        float r = 0.f;
        for (int j = 0; j < n; ++j) {   // F.interpolate(mode="nearest"): src = floor(dst * h / H)
            const int sy = (int)(((int64_t)oy * a.h[j]) / H), sx = (int)(((int64_t)ox * a.w[j]) / W);
            r += ldv(a.p[j], dt, (((int64_t)b * a.h[j] + sy) * a.w[j] + sx) * a.ld[j] + c);
        }
        stv(y, dt, (((int64_t)b * H + oy) * W + ox) * ldy + c, r * inv);
    }
}

// ------------------------------------------------------------------------------------------------ pooling / statistics
__global__ __launch_bounds__(256) void pool_kernel(int dt, const void* x, int ldx, void* y, int odt, int ldy, int B, int H, int W,
                                                    int C, int Ho, int Wo, int k /* 0: adaptive bins, else k x k stride k */) {
    const int64_t total = (int64_t)B * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        int y0, y1, x0, x1;
        if (k) {
            y0 = oy * k; y1 = y0 + k; x0 = ox * k; x1 = x0 + k;
        } else {
            y0 = (int)(((int64_t)oy * H) / Ho); y1 = (int)((((int64_t)oy + 1) * H + Ho - 1) / Ho);
            x0 = (int)(((int64_t)ox * W) / Wo); x1 = (int)((((int64_t)ox + 1) * W + Wo - 1) / Wo);
        }
        float s = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) s += ldv(x, dt, (((int64_t)b * H + yy) * W + xx) * ldx + c);
        stv(y, odt, (((int64_t)b * Ho + oy) * Wo + ox) * ldy + c, s / (float)((y1 - y0) * (x1 - x0)));
    }
}
// workgroup = 4 pixel lanes x 64 channels of one image; grid (ceil(C/64), B)
__global__ __launch_bounds__(256) void channel_stats_kernel(int dt, const void* x, int ldx, float* out, int HW, int C, int want_std) {
    __shared__ float sh[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t base = (int64_t)b * HW * ldx;
    float s = 0.f;
    if (c < C)
        for (int p = r; p < HW; p += 4) s += ldv(x, dt, base + (int64_t)p * ldx + c);
    sh[r][cl] = s;
    __syncthreads();
    const float mean = (sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) / (float)HW;
    __syncthreads();
    const int oc = want_std ? 2 * C : C;
    if (r == 0 && c < C) out[(int64_t)b * oc + c] = mean;
    if (!want_std) return;
    float q = 0.f;
    if (c < C)
        for (int p = r; p < HW; p += 4) {
            const float d = ldv(x, dt, base + (int64_t)p * ldx + c) - mean;
            q += d * d;
        }
    sh[r][cl] = q;
    __syncthreads();
    if (r == 0 && c < C) out[(int64_t)b * oc + C + c] = HW > 1 ? sqrtf((sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) / (float)HW) : 0.f;
}

// ------------------------------------------------------------------------------------------------ routers
// bias: nullptr or fp32 [B][n] added to every token's logits of image b before the temperature (the scene-aware residual of
// mot/router.py:224-240, and the whole logit of the image-level router: logits == nullptr then)
__global__ __launch_bounds__(256) void token_softmax_kernel(const float* logits, int ldl, const float* bias, float* w, int ldw,
                                                             int32_t* active, int B, int HW, int n, float inv_temp, int top_k) {
    const int64_t total = (int64_t)B * HW;
    GRID_STRIDE(p, total) {
        float v[8];
        float m = -INFINITY;
        const int bb = (int)(p / HW);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float l = -INFINITY;
            if (e < n) {
                l = logits ? logits[p * ldl + e] : 0.f;
                if (bias) l = l + bias[bb * n + e];
                l = l * inv_temp;
            }
            v[e] = l;
            m = fmaxf(m, v[e]);
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = e < n ? expf(v[e] - m) : 0.f;
            s += v[e];
        }
        unsigned sel = (1u << n) - 1u;
        if (top_k > 0 && top_k < n) {
            sel = 0u;
            float ssel = 0.f;
            for (int j = 0; j < top_k; ++j) {   // largest not yet taken, lower index wins ties
                int best = -1;
                float bv = -1.f;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < n && !((sel >> e) & 1u) && v[e] > bv) { bv = v[e]; best = e; }
                sel |= 1u << best;
                ssel += bv / s;
            }
            s *= fmaxf(ssel, 1e-6f);   // renormalise over the selected set (sum clamped at 1e-6)
        }
        const int b = (int)(p / HW);
        // "expert e is active on image b": the 64 tokens of a wave almost always lie in one image, so the wave votes and ONE lane
        // raises the flag (every token doing it serialised ~10^4 atomics on three addresses: 0.6 ms per call at 16 x 80 x 80 tokens)
        const int b0 = __builtin_amdgcn_readfirstlane(b);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < n) {
                const bool on = (sel >> e) & 1u;
                w[p * ldw + e] = on ? v[e] / s : 0.f;
                const unsigned long long vote = __ballot(on && b == b0);   // the wave's tokens of image b0 that selected e
                if (b == b0) {
                    if (vote && (int)(threadIdx.x & 63) == __builtin_ffsll((long long)vote) - 1 && active[b * n + e] == 0) atomicOr(&active[b * n + e], 1);
                } else if (on && active[b * n + e] == 0) atomicOr(&active[b * n + e], 1);   // a wave straddling two images
            }
    }
}

// one workgroup: complexity (batch mean) first, then one image per thread
__global__ __launch_bounds__(256) void gated_decide_kernel(const float* g, int ldg, const float* loc, int ldloc, const float* cplx,
                                                            int ldc, int B, int E, float alpha, float inv_temp, int clamp_mode, int top_k,
                                                            float* w, int32_t* idx, int32_t* idx_sm, float* probs) {
    __shared__ float sh[4];
    float cs = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) cs += 1.0f / (1.0f + expf(-cplx[(int64_t)b * ldc]));
    float c = block_sum(cs, sh) / (float)B;
    c = (c == c && fabsf(c) <= 3.0e38f) ? fminf(fmaxf(c, 0.3f), 1.5f) : 1.0f;
    const int keep = (int)fminf(fmaxf(rintf(c * (float)top_k), 1.0f), (float)top_k);   // torch.round: half to even
    const float a = 1.0f / (1.0f + expf(-alpha));
    for (int b = threadIdx.x; b < B; b += 256) {
        float* pr = probs + (int64_t)b * E;
        float m = -INFINITY;
        for (int e = 0; e < E; ++e) {
            // clamp_mode 1: clamp(logits, +-30) / T (moe/gated.py:141-142); 2: clamp(logits / T, +-30) (gated.py:972); 0: no clamp — a
            // router's own plain softmax (gated.py:958 nn.Softmax, routers.py:207 `_process_logits`)
            float l = a * g[(int64_t)b * ldg + e] + (1.0f - a) * loc[(int64_t)b * ldloc + e];
            if (clamp_mode == 1) l = fminf(fmaxf(l, -30.0f), 30.0f);
            l *= inv_temp;
            if (clamp_mode == 2) l = fminf(fmaxf(l, -30.0f), 30.0f);
            pr[e] = l;
            m = fmaxf(m, l);
        }
        float s = 0.f;
        for (int e = 0; e < E; ++e) {
            pr[e] = expf(pr[e] - m);
            s += pr[e];
        }
        for (int e = 0; e < E; ++e) pr[e] /= s;
        unsigned long long taken = 0ull;
        float tw[8];
        float tsum = 0.f;
        for (int j = 0; j < top_k; ++j) {   // descending, lower index wins ties (torch.topk on CPU)
            int best = 0;
            float bv = -1.f;
            for (int e = 0; e < E; ++e)
                if (!((taken >> e) & 1ull) && pr[e] > bv) { bv = pr[e]; best = e; }
            taken |= 1ull << best;
            idx[(int64_t)b * top_k + j] = best;
            idx_sm[(int64_t)j * B + b] = best;
            tw[j] = bv;
            tsum += bv;
        }
        float ksum = 0.f;
        for (int j = 0; j < top_k; ++j) {
            tw[j] = tw[j] / (tsum + 1e-6f);
            if (top_k > 1 && j >= keep) tw[j] = 0.f;
            ksum += tw[j];
        }
        for (int j = 0; j < top_k; ++j) w[(int64_t)b * top_k + j] = top_k > 1 ? tw[j] / fmaxf(ksum, 1e-6f) : tw[j];
    }
}

// one workgroup: c = clamp(mean_b sigmoid(logit_b), lo, hi) (1 when not finite), then w *= c
__global__ __launch_bounds__(256) void batch_scale_kernel(float* w, int B, int K, const float* logit, int ldl, float lo, float hi) {
    __shared__ float sh[4];
    float cs = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) cs += 1.0f / (1.0f + expf(-logit[(int64_t)b * ldl]));
    float c = block_sum(cs, sh) / (float)B;
    c = (c == c && fabsf(c) <= 3.0e38f) ? fminf(fmaxf(c, lo), hi) : 1.0f;
    for (int i = threadIdx.x; i < B * K; i += 256) w[i] *= c;
}

// ------------------------------------------------------------------------------------------------ MoA sparse inference
// moa/block.py:194-234 (eval, `sparse_inference=True`): a head group is skipped when its gate is at or below the threshold for EVERY
// token of the batch; if none is above it, the group with the largest mean gate runs alone; the retained gates are renormalised per
// token (sum clamped at the fp32 epsilon).  Two launches: batch-wide statistics (maximum as the bit pattern of a non-negative float,
// sum in fp64: its only consumer is the argmax of the fall-back), then the decision + the blend weights, COMPACTED: column j of the
// output belongs to the j-th active group (what ymk_weighted_sum takes next to the list of head outputs that were computed).
__global__ __launch_bounds__(256) void moa_gate_stats_kernel(const float* __restrict__ w, int ldw, int64_t npix, int n,
                                                             unsigned* __restrict__ gmax, double* __restrict__ gsum) {
    __shared__ float sh[4];
    __shared__ unsigned shm[4];
    for (int g = 0; g < n; ++g) {
        float mx = 0.f, sm = 0.f;
        GRID_STRIDE(i, npix) {
            const float v = w[i * ldw + g];
            mx = fmaxf(mx, v);
            sm += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = __float_as_uint(mx);
        const float tot = block_sum(sm, sh);     // (barriers inside: shm is complete afterwards)
        if (threadIdx.x == 0) {
            unsigned m = shm[0];
            for (int q = 1; q < 4; ++q) m = shm[q] > m ? shm[q] : m;
            atomicMax(&gmax[g], m);
            atomicAdd(&gsum[g], (double)tot);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void moa_gate_apply_kernel(const float* __restrict__ w, int ldw, int64_t npix, int n, float thr,
                                                             const unsigned* __restrict__ gmax, const double* __restrict__ gsum,
                                                             float* __restrict__ blend, int ldb, int* __restrict__ active) {
    bool act[8];
    int nact = 0, best = 0;
    for (int g = 0; g < n; ++g) {
        act[g] = __uint_as_float(gmax[g]) > thr;
        nact += act[g];
        if (gsum[g] > gsum[best]) best = g;      // first maximum, as torch.argmax
    }
    if (nact == 0)
        for (int g = 0; g < n; ++g) act[g] = g == best;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int g = 0; g < n; ++g) active[g] = act[g];
    GRID_STRIDE(i, npix) {
        float s = 0.f;
        for (int g = 0; g < n; ++g) s += act[g] ? w[i * ldw + g] : 0.f;
        s = fmaxf(s, 1.1920929e-07f);
        int j = 0;
        for (int g = 0; g < n; ++g)
            if (act[g]) blend[i * ldb + j++] = w[i * ldw + g] / s;
        for (; j < n; ++j) blend[i * ldb + j] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ gather / shuffle
__global__ __launch_bounds__(256) void expert_gather_kernel(int dt, const void* f, int ldf, const int32_t* idx, int B, int HW,
                                                             int OC, int K, void* out) {
    const int64_t total = (int64_t)K * B * HW * OC;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % OC);
        int64_t p = i / OC;
        const int px = (int)(p % HW);
        p /= HW;
        const int b = (int)(p % B);
        const int j = (int)(p / B);
        stv(out, dt, i, ldv(f, dt, ((int64_t)b * HW + px) * ldf + (int64_t)idx[b * K + j] * OC + c));
    }
}
// out[(j*B + b)][y][x][c] = sum_taps w[e][tap][c] * x[b][y + (ky-1)*d][x + (kx-1)*d][c], e = idx[b][j], d = dil[e] (zero padding)
__global__ __launch_bounds__(256) void expert_dw3_kernel(int dt, const void* x, int ldx, const void* w, const int32_t* dil, const int32_t* idx,
                                                          int B, int H, int W, int C, int K, void* out) {
    const int64_t total = (int64_t)K * B * H * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int xx = (int)(p % W);
        p /= W;
        const int yy = (int)(p % H);
        p /= H;
        const int b = (int)(p % B);
        const int j = (int)(p / B);
        const int e = idx[b * K + j];
        const int d = dil[e];
        float acc = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = yy + (tap / 3 - 1) * d, ix = xx + (tap % 3 - 1) * d;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                acc = fmaf(ldv(x, dt, (((int64_t)b * H + iy) * W + ix) * ldx + c), ldv(w, dt, ((int64_t)e * 9 + tap) * C + c), acc);
        }
        stv(out, dt, i, acc);
    }
}
__global__ __launch_bounds__(256) void shuffle_cat_kernel(int dt, const void* a, int lda, int Ca, const void* b, int ldb, int Cb,
                                                           int groups, void* y, int ldy, int64_t npix) {
    const int C = Ca + Cb, cpg = C / groups;
    const int64_t total = npix * C;
    GRID_STRIDE(i, total) {
        const int o = (int)(i % C);
        const int64_t p = i / C;
        const int s = (o % groups) * cpg + o / groups;
        stv(y, dt, p * ldy + o, s < Ca ? ldv(a, dt, p * lda + s) : ldv(b, dt, p * ldb + (s - Ca)));
    }
}

// ------------------------------------------------------------------------------------------------ 16-byte vector paths
// Same arithmetic as the scalar kernels above, 8 bf16 / 4 fp32 channels per lane and memory instruction; taken when every
// operand has the compute dtype, C and all pixel strides are multiples of the vector width and the bases are 16-byte
// aligned (every buffer the host code builds for channel counts that are multiples of 8 qualifies).
template <typename T>
__global__ __launch_bounds__(256) void activation_vec_kernel(T* x, int ldx, int64_t npix, int C, int act) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = npix * ncv;
    GRID_STRIDE(i, total) {
        T* px = x + (i / ncv) * ldx + (i % ncv) * VEC;
        float v[VEC];
        load_vec_f32(px, v);
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = act_f(v[q], act);
        store_vec_f32(px, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_vec_kernel(const T* x, int ldx, int B, int HW, int C, int groups, float eps, float* stats,
                                                            float* part, int nchunk) {
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ float sh[4];
    // 1-D grid, XCD-aware: workgroup L runs on XCD L % 8.  The groups of one (image, pixel chunk) read neighbouring pieces of the SAME
    // rows (C / groups channels of each: 64 bytes at 256 channels in 8 groups), so they are given to one XCD back to back — its L2 then
    // serves the row's other groups — instead of to eight XCDs that each pull the whole cache line from HBM.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = slot % groups, bc = (slot / groups) * 8 + xcd;
    if (bc >= B * nchunk) return;
    const int b = bc / nchunk, chunk = bc % nchunk, slab = b * groups + g;
    const int cg = C / groups, ncv = cg / VEC;
    const int p0 = (int)((int64_t)HW * chunk / nchunk), p1 = (int)((int64_t)HW * (chunk + 1) / nchunk);
    const int64_t nv = (int64_t)(p1 - p0) * ncv;
    const T* base = x + ((int64_t)b * HW + p0) * ldx + (int64_t)g * cg;
    const float n = (float)(p1 - p0) * (float)cg;
    constexpr int RV = 8;                        // vectors a thread can keep in registers between the two passes
    if (nv <= (int64_t)RV * 256) {               // the usual case (chunks of ~16k elements): ONE read of the slab, exact two-pass arithmetic
        u32x4 keep[RV];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            const int64_t i = threadIdx.x + (int64_t)r * 256;
            keep[r] = u32x4{0u, 0u, 0u, 0u};
            if (i < nv) {
                keep[r] = *reinterpret_cast<const u32x4*>(base + (i / ncv) * ldx + (i % ncv) * VEC);
                float v[VEC];
                load_vec_f32(reinterpret_cast<const T*>(&keep[r]), v);
#pragma unroll
                for (int q = 0; q < VEC; ++q) s += v[q];
            }
        }
        const float mean = block_sum(s, sh) / n;
        float qq = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            if (threadIdx.x + (int64_t)r * 256 < nv) {
                float v[VEC];
                load_vec_f32(reinterpret_cast<const T*>(&keep[r]), v);
#pragma unroll
                for (int q = 0; q < VEC; ++q) qq += (v[q] - mean) * (v[q] - mean);
            }
        }
        gn_emit(mean, block_sum(qq, sh), n, nchunk, eps, stats, part, slab, chunk);
        return;
    }
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
        float v[VEC];
        load_vec_f32(base + (i / ncv) * ldx + (i % ncv) * VEC, v);
#pragma unroll
        for (int q = 0; q < VEC; ++q) s += v[q];
    }
    const float mean = block_sum(s, sh) / n;
    float qq = 0.f;
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
        float v[VEC];
        load_vec_f32(base + (i / ncv) * ldx + (i % ncv) * VEC, v);
#pragma unroll
        for (int q = 0; q < VEC; ++q) qq += (v[q] - mean) * (v[q] - mean);
    }
    gn_emit(mean, block_sum(qq, sh), n, nchunk, eps, stats, part, slab, chunk);
}
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_vec_kernel(const T* x, int ldx, T* y, int ldy, const T* res, int ldr, int B, int HW, int C,
                                                            int groups, const float* weight, const float* bias, const int32_t* rows, int act,
                                                            const float* stats) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int cg = C / groups, ncv = C / VEC;
    const int64_t total = (int64_t)B * HW * ncv;
    GRID_STRIDE(i, total) {
        const int c0 = (int)(i % ncv) * VEC;
        const int64_t p = i / ncv;
        const int b = (int)(p / HW);
        float v[VEC];
        load_vec_f32(x + p * ldx + c0, v);
        const int64_t r = rows ? (int64_t)rows[b] * C : 0;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const float* st = stats + 2 * ((int64_t)b * groups + (c0 + q) / cg);   // a vector may straddle two groups
            float t = (v[q] - st[0]) * st[1];
            if (weight) t = t * weight[r + c0 + q] + bias[r + c0 + q];
            v[q] = act_f(t, act);
        }
        if (res) {
            float rr[VEC];
            load_vec_f32(res + p * ldr + c0, rr);
#pragma unroll
            for (int q = 0; q < VEC; ++q) v[q] += rr[q];
        }
        store_vec_f32(y + p * ldy + c0, v);
    }
}
// The same with the per-channel constants hoisted: a thread owns ONE channel vector of one image and walks pixels, so scale = rstd * w
// and shift = b - mean * rstd * w are computed once (the grid-stride form above re-loads two statistics and 2 x VEC affine values per
// 16 bytes of data: four times the bytes it normalises).  grid (pixel chunks, B); 256 threads = (256 / ncv) pixel lanes x ncv vectors,
// ncv = C / VEC <= 256.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const T* x, int ldx, T* y, int ldy, const T* res, int ldr, int HW, int C, int groups,
                                                             const float* weight, const float* bias, const int32_t* rows, int act,
                                                             const float* stats, int ppb /* pixels per workgroup */) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC, cg = C / groups;
    const int b = blockIdx.y;
    const int cv = threadIdx.x % ncv, lane = threadIdx.x / ncv, lanes = 256 / ncv;
    if (lane >= lanes) return;
    const int c0 = cv * VEC;
    const int64_t r = rows ? (int64_t)rows[b] * C : 0;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        const float* st = stats + 2 * ((int64_t)b * groups + (c0 + q) / cg);
        const float w = weight ? weight[r + c0 + q] : 1.0f, bb = weight ? bias[r + c0 + q] : 0.0f;
        sc[q] = st[1] * w;
        sh[q] = bb - st[0] * st[1] * w;
    }
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    const T* xb = x + (int64_t)b * HW * ldx + c0;
    T* yb = y + (int64_t)b * HW * ldy + c0;
    const T* rb = res ? res + (int64_t)b * HW * ldr + c0 : nullptr;
    for (int p = p0 + lane; p < p1; p += lanes) {
        float v[VEC];
        load_vec_f32(xb + (int64_t)p * ldx, v);
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = act_f((v[q] - 0.0f) * sc[q] + sh[q], act);
        if (rb) {
            float rr[VEC];
            load_vec_f32(rb + (int64_t)p * ldr, rr);
#pragma unroll
            for (int q = 0; q < VEC; ++q) v[q] += rr[q];
        }
        store_vec_f32(yb + (int64_t)p * ldy, v);
    }
}
// one wave per token, one vector per lane and step
template <typename T>
__global__ __launch_bounds__(256) void ln_vec_kernel(const T* x, int ldx, T* y, int ldy, int64_t npix, int C, const float* weight,
                                                      const float* bias, float eps) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63, ncv = C / VEC;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p = wave0; p < npix; p += nwaves) {
        float s = 0.f;
        for (int cv = lane; cv < ncv; cv += 64) {
            float v[VEC];
            load_vec_f32(x + p * ldx + cv * VEC, v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) s += v[q];
        }
        const float mean = wave_sum(s) / (float)C;
        float qq = 0.f;
        for (int cv = lane; cv < ncv; cv += 64) {
            float v[VEC];
            load_vec_f32(x + p * ldx + cv * VEC, v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) qq += (v[q] - mean) * (v[q] - mean);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)C + eps);
        for (int cv = lane; cv < ncv; cv += 64) {
            float v[VEC];
            load_vec_f32(x + p * ldx + cv * VEC, v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) v[q] = (v[q] - mean) * rstd * weight[cv * VEC + q] + bias[cv * VEC + q];
            store_vec_f32(y + p * ldy + cv * VEC, v);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void eltwise_vec_kernel(int op, const T* a, int lda, const T* b, int ldb, T* y, int ldy, int64_t npix, int C,
                                                           float alpha) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = npix * ncv;
    GRID_STRIDE(i, total) {
        const int64_t p = i / ncv;
        const int c0 = (int)(i % ncv) * VEC;
        float av[VEC], bv[VEC];
        load_vec_f32(a + p * lda + c0, av);
        load_vec_f32(b + p * ldb + c0, bv);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            if (op == YMK_ELT_MUL) av[q] = av[q] * bv[q];
            else if (op == YMK_ELT_SIGMOID_MUL) av[q] = bv[q] / (1.0f + expf(-av[q]));
            else if (op == YMK_ELT_CLAMP_ADD) av[q] = fminf(fmaxf(av[q], -alpha), alpha) + bv[q];
            else av[q] = (1.0f - alpha) * av[q] + alpha * bv[q];
        }
        store_vec_f32(y + p * ldy + c0, av);
    }
}
// y = x + scale * a * b (b a map of T, or an fp32 per-image gate when gate != nullptr); a == nullptr: y = x * gate
template <typename T>
__global__ __launch_bounds__(256) void gate_vec_kernel(const T* x, int ldx, const T* a, int lda, const T* b, int ldb, const float* gate,
                                                        float scale, T* y, int ldy, int B, int HW, int C) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = (int64_t)B * HW * ncv;
    GRID_STRIDE(i, total) {
        const int64_t p = i / ncv;
        const int c0 = (int)(i % ncv) * VEC;
        float xv[VEC], av[VEC], bv[VEC];
        load_vec_f32(x + p * ldx + c0, xv);
        if (gate) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) bv[q] = gate[(p / HW) * C + c0 + q];
        } else {
            load_vec_f32(b + p * ldb + c0, bv);
        }
        if (a) {
            load_vec_f32(a + p * lda + c0, av);
#pragma unroll
            for (int q = 0; q < VEC; ++q) xv[q] = xv[q] + scale * av[q] * bv[q];
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) xv[q] = xv[q] * bv[q];
        }
        store_vec_f32(y + p * ldy + c0, xv);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void weighted_sum_vec_kernel(const float* w, int ldw, int per_image, int E, Parts4 parts, int ldp, T* y, int ldy,
                                                                int B, int HW, int C) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = (int64_t)B * HW * ncv;
    GRID_STRIDE(i, total) {
        const int64_t p = i / ncv;
        const int c0 = (int)(i % ncv) * VEC;
        const float* wr = w + (per_image ? p / HW : p) * ldw;
        float r[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) r[q] = 0.f;
        for (int e = 0; e < E; ++e) {
            float v[VEC];
            load_vec_f32(static_cast<const T*>(parts.p[e]) + p * ldp + c0, v);
            const float we = wr[e];
#pragma unroll
            for (int q = 0; q < VEC; ++q) r[q] += we * v[q];
        }
        store_vec_f32(y + p * ldy + c0, r);
    }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void pool_vec_kernel(const T* x, int ldx, TO* y, int ldy, int B, int H, int W, int C, int Ho, int Wo, int k) {
    constexpr int VEC = 16 / (int)sizeof(T);   // input vector; an fp32 output of a bf16 input is written as two 16-byte stores
    const int ncv = C / VEC;
    const int64_t total = (int64_t)B * Ho * Wo * ncv;
    GRID_STRIDE(i, total) {
        const int c0 = (int)(i % ncv) * VEC;
        int64_t p = i / ncv;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        int y0, y1, x0, x1;
        if (k) {
            y0 = oy * k; y1 = y0 + k; x0 = ox * k; x1 = x0 + k;
        } else {
            y0 = (int)(((int64_t)oy * H) / Ho); y1 = (int)((((int64_t)oy + 1) * H + Ho - 1) / Ho);
            x0 = (int)(((int64_t)ox * W) / Wo); x1 = (int)((((int64_t)ox + 1) * W + Wo - 1) / Wo);
        }
        float s[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) s[q] = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
                float v[VEC];
                load_vec_f32(x + (((int64_t)b * H + yy) * W + xx) * ldx + c0, v);
#pragma unroll
                for (int q = 0; q < VEC; ++q) s[q] += v[q];
            }
        const float inv = 1.0f / (float)((y1 - y0) * (x1 - x0));
        TO* py = y + (((int64_t)b * Ho + oy) * Wo + ox) * ldy + c0;
#pragma unroll
        for (int q = 0; q < VEC; ++q) from_f32(py[q], s[q] * inv);   // contiguous: the compiler merges these into wide stores
    }
}
// workgroup = 16 pixel lanes x 16 channel vectors of one image; grid (ceil(C / (16 * VEC)), B)
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_vec_kernel(const T* x, int ldx, float* out, int HWtot, int C, int want_std, float* part,
                                                                int nchunk) {
    // grid.z = pixel chunks (a [16, 160, 160, 128] map on (1, 16) workgroups was 16 workgroups for 105 MB): chunk partials (mean, M2) go to
    // `part` and are combined exactly, in chunk order, by channel_stats_finalize_kernel; nchunk == 1 writes the result directly
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ float sh[16][16 * VEC + 1];
    const int b = blockIdx.y, cl = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int c0 = (blockIdx.x * 16 + cl) * VEC;
    const int p0 = (int)((int64_t)HWtot * blockIdx.z / nchunk), HW = (int)((int64_t)HWtot * (blockIdx.z + 1) / nchunk) - p0;
    const T* base = x + ((int64_t)b * HWtot + p0) * ldx;
    const int oc = want_std ? 2 * C : C;
    float* pm = part ? part + (((int64_t)b * nchunk + blockIdx.z) * 2) * C : nullptr;   // [mean | M2] of this chunk
    if (nchunk > 1) out = nullptr;
    float s[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) s[q] = 0.f;
    if (c0 < C)
        for (int p = r; p < HW; p += 16) {
            float v[VEC];
            load_vec_f32(base + (int64_t)p * ldx + c0, v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) s[q] += v[q];
        }
#pragma unroll
    for (int q = 0; q < VEC; ++q) sh[r][cl * VEC + q] = s[q];
    __syncthreads();
    float mean[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        float t = 0.f;
        for (int rr = 0; rr < 16; ++rr) t += sh[rr][cl * VEC + q];
        mean[q] = t / (float)HW;
    }
    __syncthreads();
    if (r == 0 && c0 < C)
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            if (out) out[(int64_t)b * oc + c0 + q] = mean[q];
            else pm[c0 + q] = mean[q];
        }
    if (!want_std) return;
#pragma unroll
    for (int q = 0; q < VEC; ++q) s[q] = 0.f;
    if (c0 < C)
        for (int p = r; p < HW; p += 16) {
            float v[VEC];
            load_vec_f32(base + (int64_t)p * ldx + c0, v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) s[q] += (v[q] - mean[q]) * (v[q] - mean[q]);
        }
#pragma unroll
    for (int q = 0; q < VEC; ++q) sh[r][cl * VEC + q] = s[q];
    __syncthreads();
    if (r == 0 && c0 < C)
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float t = 0.f;
            for (int rr = 0; rr < 16; ++rr) t += sh[rr][cl * VEC + q];
            if (out) out[(int64_t)b * oc + C + c0 + q] = HW > 1 ? sqrtf(t / (float)HW) : 0.f;
            else pm[C + c0 + q] = t;
        }
}
// grid (ceil(C / 256), B): chunk partials (mean_c, M2_c over n_c pixels) -> mean and biased std, pairwise-exact update in chunk order
__global__ __launch_bounds__(256) void channel_stats_finalize_kernel(const float* part, float* out, int HW, int C, int want_std, int nchunk) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* p = part + (int64_t)b * nchunk * 2 * C;
    float mean = 0.f, m2 = 0.f, n = 0.f;
    for (int k = 0; k < nchunk; ++k) {
        const float nc = (float)((int)((int64_t)HW * (k + 1) / nchunk) - (int)((int64_t)HW * k / nchunk));
        const float mc = p[(int64_t)k * 2 * C + c], m2c = want_std ? p[(int64_t)k * 2 * C + C + c] : 0.f;
        const float tot = n + nc, d = mc - mean;
        mean += d * (nc / tot);
        m2 += m2c + d * d * (n * nc / tot);
        n = tot;
    }
    const int oc = want_std ? 2 * C : C;
    out[(int64_t)b * oc + c] = mean;
    if (want_std) out[(int64_t)b * oc + C + c] = HW > 1 ? sqrtf(m2 / (float)HW) : 0.f;
}
template <typename T>
__global__ __launch_bounds__(256) void mean_upsampled_vec_kernel(int n, Pyr4 a, T* y, int ldy, int B, int H, int W, int C) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int ncv = C / VEC;
    const int64_t total = (int64_t)B * H * W * ncv;
    const float inv = 1.0f / (float)n;
    GRID_STRIDE(i, total) {
        const int c0 = (int)(i % ncv) * VEC;
        int64_t p = i / ncv;
        const int ox = (int)(p % W);
        p /= W;
        const int oy = (int)(p % H);
        const int b = (int)(p / H);
        float r[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) r[q] = 0.f;
        for (int j = 0; j < n; ++j) {
            const int sy = (int)(((int64_t)oy * a.h[j]) / H), sx = (int)(((int64_t)ox * a.w[j]) / W);
            float v[VEC];
            load_vec_f32(static_cast<const T*>(a.p[j]) + (((int64_t)b * a.h[j] + sy) * a.w[j] + sx) * a.ld[j] + c0, v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) r[q] += v[q];
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) r[q] *= inv;
        store_vec_f32(y + (((int64_t)b * H + oy) * W + ox) * ldy + c0, r);
    }
}
// raw 16-byte copies: dtype only sets the element size
__global__ __launch_bounds__(256) void expert_gather_vec_kernel(int es, const char* f, int64_t ldf_b, const int32_t* idx, int B, int HW, int ocv,
                                                                 int K, int64_t oc_b, char* out) {
    const int64_t total = (int64_t)K * B * HW * ocv;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % ocv);
        int64_t p = i / ocv;
        const int px = (int)(p % HW);
        p /= HW;
        const int b = (int)(p % B);
        const int j = (int)(p / B);
        const u32x4 v = *reinterpret_cast<const u32x4*>(f + ((int64_t)b * HW + px) * ldf_b + (int64_t)idx[b * K + j] * oc_b + (int64_t)cv * 16);
        *reinterpret_cast<u32x4*>(out + i * 16) = v;
    }
}
// groups == 2, Ca == Cb: out = [a0 b0 a1 b1 ...]; one lane interleaves VEC/2 channels of each source into one vector
template <typename T>
__global__ __launch_bounds__(256) void shuffle2_vec_kernel(const T* a, int lda, const T* b, int ldb, int Ch, T* y, int ldy, int64_t npix) {
    constexpr int VEC = 16 / (int)sizeof(T), HV = VEC / 2;
    const int nh = Ch / HV;   // output vectors per pixel
    const int64_t total = npix * nh;
    GRID_STRIDE(i, total) {
        const int h0 = (int)(i % nh) * HV;
        const int64_t p = i / nh;
        T o[VEC];
#pragma unroll
        for (int q = 0; q < HV; ++q) {
            o[2 * q] = a[p * lda + h0 + q];
            o[2 * q + 1] = b[p * ldb + h0 + q];
        }
        *reinterpret_cast<u32x4*>(y + p * ldy + 2 * h0) = *reinterpret_cast<const u32x4*>(o);
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int vecw(int dt) { return dt == YMK_BF16 ? 8 : 4; }

}  // namespace

#define LAUNCH(kern, total, ...) \
    hipLaunchKernelGGL(kern, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int ymk_activation(int32_t dtype, void* x, int32_t ldx, int64_t npix, int32_t C, int32_t act, void* stream) {
    if (!x || bad_dt(dtype) || C < 1 || ldx < C || act < YMK_ACT_NONE || act > YMK_ACT_GELU) return YMK_E_BADARG;
    if (npix <= 0 || act == YMK_ACT_NONE) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && ldx % V == 0 && al16(x)) {
        if (dtype == YMK_BF16) LAUNCH(activation_vec_kernel<h16_t>, npix * (C / V), (h16_t*)x, ldx, npix, C, act);
        else LAUNCH(activation_vec_kernel<float>, npix * (C / V), (float*)x, ldx, npix, C, act);
        return ymk_launch_status();
    }
    LAUNCH(activation_kernel, npix * C, x, dtype, ldx, npix, C, act);
    return ymk_launch_status();
}

extern "C" int ymk_group_norm(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy,
                              const void* residual, int32_t ldr, int32_t B, int32_t HW, int32_t C, int32_t groups,
                              const float* weight, const float* bias, const int32_t* affine_rows, float eps, int32_t act,
                              float* stats_ws, void* stream) {
    if (!x || !y || !stats_ws || bad_dt(dtype) || bad_dt(out_dtype) || C < 1 || groups < 1 || C % groups || ldx < C || ldy < C)
        return YMK_E_BADARG;
    if ((weight == nullptr) != (bias == nullptr) || (affine_rows && !weight) || (residual && ldr < C)) return YMK_E_BADARG;
    if (act != YMK_ACT_NONE && act != YMK_ACT_SILU) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    const int V = vecw(dtype);
    const bool vin = C % V == 0 && ldx % V == 0 && al16(x);
    // chunks of ~16k elements per workgroup, enough of them to fill the chip, at most GN_MAX_CHUNKS and one per pixel
    const int64_t celems = dtype == YMK_F32 ? 8192 : 16384;   // what a workgroup keeps in registers between its two passes (gn_stats_vec_kernel)
    int64_t want = ((int64_t)HW * (C / groups) + celems - 1) / celems;
    const int nchunk = (int)(want < 1 ? 1 : want > GN_MAX_CHUNKS ? GN_MAX_CHUNKS : want > HW ? HW : want);
    float* part = stats_ws + (size_t)B * groups * 2;
    const dim3 sgrid(B * groups, nchunk);
    const dim3 vgrid((unsigned)(((int64_t)B * nchunk + 7) / 8 * 8 * groups));
    if (vin && (C / groups) % V == 0) {   // statistics: whole vectors inside a group
        if (dtype == YMK_BF16)
            hipLaunchKernelGGL(gn_stats_vec_kernel<h16_t>, vgrid, dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, ldx, B, HW, C, groups, eps, stats_ws, part, nchunk);
        else
            hipLaunchKernelGGL(gn_stats_vec_kernel<float>, vgrid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, B, HW, C, groups, eps, stats_ws, part, nchunk);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel, sgrid, dim3(256), 0, (hipStream_t)stream, x, dtype, ldx, HW, C, groups, eps, stats_ws, part, nchunk);
    }
    if (nchunk > 1) hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, (hipStream_t)stream, nchunk, eps, stats_ws, (const float*)part);
    if (vin && out_dtype == dtype && ldy % V == 0 && al16(y) && (!residual || (ldr % V == 0 && al16(residual))) && C / V <= 256 && B <= 65535) {
        // per-thread channel constants: ~2048 workgroups of one image each
        int ppb = (int)(((int64_t)B * HW + 2047) / 2048);
        ppb = ppb < 64 ? 64 : ppb;
        const dim3 agrid((HW + ppb - 1) / ppb, B);
        if (dtype == YMK_BF16)
            hipLaunchKernelGGL(gn_apply_rows_kernel<h16_t>, agrid, dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, ldx, (h16_t*)y, ldy, (const h16_t*)residual, ldr,
                               HW, C, groups, weight, bias, affine_rows, act, (const float*)stats_ws, ppb);
        else
            hipLaunchKernelGGL(gn_apply_rows_kernel<float>, agrid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, (const float*)residual, ldr,
                               HW, C, groups, weight, bias, affine_rows, act, (const float*)stats_ws, ppb);
        return ymk_launch_status();
    }
    if (vin && out_dtype == dtype && ldy % V == 0 && al16(y) && (!residual || (ldr % V == 0 && al16(residual)))) {
        const int64_t total = (int64_t)B * HW * (C / V);
        if (dtype == YMK_BF16)
            LAUNCH(gn_apply_vec_kernel<h16_t>, total, (const h16_t*)x, ldx, (h16_t*)y, ldy, (const h16_t*)residual, ldr, B, HW, C, groups, weight,
                   bias, affine_rows, act, (const float*)stats_ws);
        else
            LAUNCH(gn_apply_vec_kernel<float>, total, (const float*)x, ldx, (float*)y, ldy, (const float*)residual, ldr, B, HW, C, groups, weight,
                   bias, affine_rows, act, (const float*)stats_ws);
        return ymk_launch_status();
    }
    LAUNCH(gn_apply_kernel, (int64_t)B * HW * C, x, dtype, ldx, y, out_dtype, ldy, residual, ldr, B, HW, C, groups, weight, bias,
           affine_rows, act, (const float*)stats_ws);
    return ymk_launch_status();
}

extern "C" int ymk_layer_norm(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t npix, int32_t C,
                              const float* weight, const float* bias, float eps, void* stream) {
    if (!x || !y || !weight || !bias || bad_dt(dtype) || C < 1 || ldx < C || ldy < C) return YMK_E_BADARG;
    if (npix <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && ldx % V == 0 && ldy % V == 0 && al16(x) && al16(y)) {
        if (dtype == YMK_BF16) LAUNCH(ln_vec_kernel<h16_t>, npix * 64, (const h16_t*)x, ldx, (h16_t*)y, ldy, npix, C, weight, bias, eps);
        else LAUNCH(ln_vec_kernel<float>, npix * 64, (const float*)x, ldx, (float*)y, ldy, npix, C, weight, bias, eps);
        return ymk_launch_status();
    }
    LAUNCH(ln_kernel, npix * 64, x, dtype, ldx, y, ldy, npix, C, weight, bias, eps);
    return ymk_launch_status();
}

extern "C" int ymk_eltwise(int32_t op, int32_t dtype, const void* a, int32_t lda, const void* b, int32_t ldb, void* y, int32_t ldy,
                           int64_t npix, int32_t C, float alpha, void* stream) {
    if (!a || !b || !y || bad_dt(dtype) || op < YMK_ELT_MUL || op > YMK_ELT_CLAMP_ADD || C < 1 || lda < C || ldb < C || ldy < C)
        return YMK_E_BADARG;
    if (npix <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && lda % V == 0 && ldb % V == 0 && ldy % V == 0 && al16(a) && al16(b) && al16(y)) {
        if (dtype == YMK_BF16) LAUNCH(eltwise_vec_kernel<h16_t>, npix * (C / V), op, (const h16_t*)a, lda, (const h16_t*)b, ldb, (h16_t*)y, ldy, npix, C, alpha);
        else LAUNCH(eltwise_vec_kernel<float>, npix * (C / V), op, (const float*)a, lda, (const float*)b, ldb, (float*)y, ldy, npix, C, alpha);
        return ymk_launch_status();
    }
    LAUNCH(eltwise_kernel, npix * C, op, dtype, a, lda, b, ldb, y, ldy, npix, C, alpha);
    return ymk_launch_status();
}

extern "C" int ymk_fma_gate(int32_t dtype, const void* x, int32_t ldx, const void* a, int32_t lda, const void* b, int32_t b_dtype,
                            int32_t ldb, int32_t b_per_image, float scale, void* y, int32_t ldy, int32_t B, int32_t HW, int32_t C,
                            void* stream) {
    if (!x || !a || !b || !y || bad_dt(dtype) || C < 1 || ldx < C || lda < C || ldy < C) return YMK_E_BADARG;
    if (!b_per_image && (bad_dt(b_dtype) || ldb < C)) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && ldx % V == 0 && lda % V == 0 && ldy % V == 0 && al16(x) && al16(a) && al16(y) &&
        (b_per_image || (b_dtype == dtype && ldb % V == 0 && al16(b)))) {
        const int64_t total = (int64_t)B * HW * (C / V);
        if (dtype == YMK_BF16)
            LAUNCH(gate_vec_kernel<h16_t>, total, (const h16_t*)x, ldx, (const h16_t*)a, lda, b_per_image ? nullptr : (const h16_t*)b, ldb,
                   b_per_image ? (const float*)b : nullptr, scale, (h16_t*)y, ldy, B, HW, C);
        else
            LAUNCH(gate_vec_kernel<float>, total, (const float*)x, ldx, (const float*)a, lda, b_per_image ? nullptr : (const float*)b, ldb,
                   b_per_image ? (const float*)b : nullptr, scale, (float*)y, ldy, B, HW, C);
        return ymk_launch_status();
    }
    LAUNCH(fma_gate_kernel, (int64_t)B * HW * C, dtype, x, ldx, a, lda, b, b_dtype, ldb, b_per_image, scale, y, ldy, B, HW, C);
    return ymk_launch_status();
}

extern "C" int ymk_channel_gate(int32_t dtype, const void* x, int32_t ldx, const float* gate, void* y, int32_t ldy, int32_t B,
                                int32_t HW, int32_t C, void* stream) {
    if (!x || !gate || !y || bad_dt(dtype) || C < 1 || ldx < C || ldy < C) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && ldx % V == 0 && ldy % V == 0 && al16(x) && al16(y)) {
        const int64_t total = (int64_t)B * HW * (C / V);
        if (dtype == YMK_BF16)
            LAUNCH(gate_vec_kernel<h16_t>, total, (const h16_t*)x, ldx, (const h16_t*)nullptr, 0, (const h16_t*)nullptr, 0, gate, 0.f, (h16_t*)y, ldy, B, HW, C);
        else
            LAUNCH(gate_vec_kernel<float>, total, (const float*)x, ldx, (const float*)nullptr, 0, (const float*)nullptr, 0, gate, 0.f, (float*)y, ldy, B, HW, C);
        return ymk_launch_status();
    }
    LAUNCH(channel_gate_kernel, (int64_t)B * HW * C, dtype, x, ldx, gate, y, ldy, B, HW, C);
    return ymk_launch_status();
}

extern "C" int ymk_weighted_sum(int32_t dtype, const float* w, int32_t ldw, int32_t w_per_image, int32_t E, const void* p0,
                                const void* p1, const void* p2, const void* p3, int32_t ldp, void* y, int32_t ldy, int32_t B,
                                int32_t HW, int32_t C, void* stream) {
    if (!w || !y || bad_dt(dtype) || E < 1 || E > 4 || ldw < E || C < 1 || ldp < C || ldy < C) return YMK_E_BADARG;
    Parts4 parts{{p0, p1, p2, p3}};
    for (int e = 0; e < E; ++e)
        if (!parts.p[e]) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    const int V = vecw(dtype);
    bool vok = C % V == 0 && ldp % V == 0 && ldy % V == 0 && al16(y);
    for (int e = 0; e < E; ++e) vok = vok && al16(parts.p[e]);
    if (vok) {
        const int64_t total = (int64_t)B * HW * (C / V);
        if (dtype == YMK_BF16) LAUNCH(weighted_sum_vec_kernel<h16_t>, total, w, ldw, w_per_image, E, parts, ldp, (h16_t*)y, ldy, B, HW, C);
        else LAUNCH(weighted_sum_vec_kernel<float>, total, w, ldw, w_per_image, E, parts, ldp, (float*)y, ldy, B, HW, C);
        return ymk_launch_status();
    }
    LAUNCH(weighted_sum_kernel, (int64_t)B * HW * C, dtype, w, ldw, w_per_image, E, parts, ldp, y, ldy, B, HW, C);
    return ymk_launch_status();
}

extern "C" int ymk_mean_upsampled(int32_t dtype, int32_t n, const void* p0, const void* p1, const void* p2, const void* p3,
                                  const int32_t* hs, const int32_t* ws, const int32_t* lds, void* y, int32_t ldy, int32_t B,
                                  int32_t H, int32_t W, int32_t C, void* stream) {
    if (!y || !hs || !ws || !lds || bad_dt(dtype) || n < 1 || n > 4 || C < 1 || ldy < C) return YMK_E_BADARG;
    Pyr4 a{{p0, p1, p2, p3}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int j = 0; j < n; ++j) {
        if (!a.p[j] || hs[j] < 1 || ws[j] < 1 || lds[j] < C) return YMK_E_BADARG;
        a.h[j] = hs[j]; a.w[j] = ws[j]; a.ld[j] = lds[j];
    }
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    const int V = vecw(dtype);
    bool vok = C % V == 0 && ldy % V == 0 && al16(y);
    for (int j = 0; j < n; ++j) vok = vok && a.ld[j] % V == 0 && al16(a.p[j]);
    if (vok) {
        const int64_t total = (int64_t)B * H * W * (C / V);
        if (dtype == YMK_BF16) LAUNCH(mean_upsampled_vec_kernel<h16_t>, total, n, a, (h16_t*)y, ldy, B, H, W, C);
        else LAUNCH(mean_upsampled_vec_kernel<float>, total, n, a, (float*)y, ldy, B, H, W, C);
        return ymk_launch_status();
    }
    LAUNCH(mean_upsampled_kernel, (int64_t)B * H * W * C, dtype, n, a, y, ldy, B, H, W, C);
    return ymk_launch_status();
}

extern "C" int ymk_adaptive_avg_pool(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy, int32_t B,
                                     int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void* stream) {
    if (!x || !y || bad_dt(dtype) || bad_dt(out_dtype) || C < 1 || ldx < C || ldy < C || Ho < 1 || Wo < 1 || H < 1 || W < 1)
        return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && ldx % V == 0 && ldy % V == 0 && al16(x) && al16(y) && (out_dtype == dtype || out_dtype == YMK_F32)) {
        const int64_t total = (int64_t)B * Ho * Wo * (C / V);
        if (dtype == YMK_F32) LAUNCH((pool_vec_kernel<float, float>), total, (const float*)x, ldx, (float*)y, ldy, B, H, W, C, Ho, Wo, 0);
        else if (out_dtype == YMK_BF16) LAUNCH((pool_vec_kernel<h16_t, h16_t>), total, (const h16_t*)x, ldx, (h16_t*)y, ldy, B, H, W, C, Ho, Wo, 0);
        else LAUNCH((pool_vec_kernel<h16_t, float>), total, (const h16_t*)x, ldx, (float*)y, ldy, B, H, W, C, Ho, Wo, 0);
        return ymk_launch_status();
    }
    LAUNCH(pool_kernel, (int64_t)B * Ho * Wo * C, dtype, x, ldx, y, out_dtype, ldy, B, H, W, C, Ho, Wo, 0);
    return ymk_launch_status();
}

extern "C" int ymk_avg_pool(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy, int32_t B, int32_t H,
                            int32_t W, int32_t C, int32_t k, void* stream) {
    if (!x || !y || bad_dt(dtype) || bad_dt(out_dtype) || C < 1 || ldx < C || ldy < C || k < 1 || H < k || W < k) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && ldx % V == 0 && ldy % V == 0 && al16(x) && al16(y) && (out_dtype == dtype || out_dtype == YMK_F32)) {
        const int64_t total = (int64_t)B * (H / k) * (W / k) * (C / V);
        if (dtype == YMK_F32) LAUNCH((pool_vec_kernel<float, float>), total, (const float*)x, ldx, (float*)y, ldy, B, H, W, C, (H / k), (W / k), k);
        else if (out_dtype == YMK_BF16) LAUNCH((pool_vec_kernel<h16_t, h16_t>), total, (const h16_t*)x, ldx, (h16_t*)y, ldy, B, H, W, C, (H / k), (W / k), k);
        else LAUNCH((pool_vec_kernel<h16_t, float>), total, (const h16_t*)x, ldx, (float*)y, ldy, B, H, W, C, (H / k), (W / k), k);
        return ymk_launch_status();
    }
    LAUNCH(pool_kernel, (int64_t)B * (H / k) * (W / k) * C, dtype, x, ldx, y, out_dtype, ldy, B, H, W, C, H / k, W / k, k);
    return ymk_launch_status();
}

extern "C" int ymk_channel_stats(int32_t dtype, const void* x, int32_t ldx, float* out, int32_t B, int32_t HW, int32_t C,
                                 int32_t want_std, float* ws, void* stream) {
    if (!x || !out || bad_dt(dtype) || C < 1 || ldx < C || HW < 1 || B > 65535) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (C % V == 0 && ldx % V == 0 && al16(x)) {
        // pixel chunks of ~1024 per workgroup, at most 64 (ws: [B][nchunk][2][C] floats); without a workspace one workgroup per image
        int nchunk = ws ? (HW + 1023) / 1024 : 1;
        nchunk = nchunk < 1 ? 1 : nchunk > 64 ? 64 : nchunk;
        const dim3 grid((C / V + 15) / 16, B, nchunk);
        if (dtype == YMK_BF16) hipLaunchKernelGGL(channel_stats_vec_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, ldx, out, HW, C, want_std, ws, nchunk);
        else hipLaunchKernelGGL(channel_stats_vec_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, out, HW, C, want_std, ws, nchunk);
        if (nchunk > 1)
            hipLaunchKernelGGL(channel_stats_finalize_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, (const float*)ws, out, HW, C, want_std, nchunk);
        return ymk_launch_status();
    }
    hipLaunchKernelGGL(channel_stats_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, dtype, x, ldx, out, HW, C,
                       want_std);
    return ymk_launch_status();
}

/* MoA sparse inference (moa/block.py:194-234).  w fp32 [npix][ldw] per-token gates of n <= 8 groups; stats: zero-initialised scratch of
 * n * 4 + n * 8 bytes (8-byte aligned); outputs: active int32 [n], blend fp32 [npix][ldb] with the retained groups' renormalised gates in
 * columns 0 .. (number of active groups) - 1, zeros behind. */
extern "C" int ymk_moa_sparse_gate(const float* w, int32_t ldw, int64_t npix, int32_t n, float threshold, void* stats, float* blend,
                                   int32_t ldb, int32_t* active, void* stream) {
    if (!w || !stats || !blend || !active || n < 1 || n > 8 || ldw < n || ldb < n || ((uintptr_t)stats & 7)) return YMK_E_BADARG;
    if (npix <= 0) return YMK_OK;
    double* gsum = static_cast<double*>(stats);
    unsigned* gmax = reinterpret_cast<unsigned*>(gsum + n);
    LAUNCH(moa_gate_stats_kernel, npix, w, ldw, npix, n, gmax, gsum);
    LAUNCH(moa_gate_apply_kernel, npix, w, ldw, npix, n, threshold, gmax, gsum, blend, ldb, active);
    return ymk_launch_status();
}

extern "C" int ymk_token_softmax(const float* logits, int32_t ldl, const float* bias, float* w, int32_t ldw, int32_t* active, int32_t B,
                                 int32_t HW, int32_t n, float inv_temp, int32_t top_k, void* stream) {
    if ((!logits && !bias) || !w || !active || n < 1 || n > 8 || (logits && ldl < n) || ldw < n || top_k < 0) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    LAUNCH(token_softmax_kernel, (int64_t)B * HW, logits, ldl, bias, w, ldw, active, B, HW, n, inv_temp, top_k);
    return ymk_launch_status();
}

// ---- UltraEfficientRouter decision tail (moe/routers.py:117-147, eval): per PIXEL of the (pooled) router map softmax(clamp(logits, +-30) *
// inv_temp) over the E experts, the MEAN of those weights over the pixels, top-k of the pooled weights (lower index first among equals),
// values / max(sum, 1e-6); routes whose weight is not above `threshold` get weight 0 (the inference-only cut of
// BatchedExpertComputation, moe/utils.py:166-169: such an expert contributes nothing).  One wave per image: lanes over the pixels,
// per-expert partial sums reduced across the wave in a fixed order.  E <= 32, top_k <= 4.
__global__ __launch_bounds__(64) void pooled_softmax_route_kernel(const float* __restrict__ logits, int ldl, int HW, int E, float inv_temp,
                                                                   int top_k, float threshold, int B, float* __restrict__ w,
                                                                   int32_t* __restrict__ idx, int32_t* __restrict__ rows,
                                                                   float* __restrict__ pooled) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    for (int p = lane; p < HW; p += 64) {
        const float* l = logits + ((size_t)b * HW + p) * ldl;
        float v[32];
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            v[e] = e < E ? fminf(fmaxf(l[e], -30.0f), 30.0f) * inv_temp : -INFINITY;
            m = fmaxf(m, v[e]);
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            v[e] = e < E ? expf(v[e] - m) : 0.f;
            s += v[e];
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] += v[e] / s;
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[e] += __shfl_xor(acc[e], o);
        acc[e] = acc[e] / (float)HW;
    }
    if (lane != 0) return;
    unsigned taken = 0u;
    float vals[4] = {0.f, 0.f, 0.f, 0.f};
    int sel[4] = {0, 0, 0, 0};
    float sum = 0.f;
    for (int j = 0; j < top_k; ++j) {
        int best = -1;
        float bv = -1.f;
#pragma unroll
        for (int e = 0; e < 32; ++e)
            if (e < E && !((taken >> e) & 1u) && acc[e] > bv) { bv = acc[e]; best = e; }
        taken |= 1u << best;
        vals[j] = bv; sel[j] = best;
        sum += bv;
    }
    sum = fmaxf(sum, 1e-6f);
    for (int j = 0; j < top_k; ++j) {
        const float wj = vals[j] / sum;
        w[b * top_k + j] = wj > threshold ? wj : 0.f;
        idx[b * top_k + j] = sel[j];
        rows[j * B + b] = sel[j];
    }
#pragma unroll
    for (int e = 0; e < 32; ++e)
        if (e < E) pooled[b * E + e] = acc[e];
}

extern "C" int ymk_pooled_softmax_route(const float* logits, int32_t ldl, int32_t B, int32_t HW, int32_t E, float inv_temp, int32_t top_k,
                                        float threshold, float* w, int32_t* idx, int32_t* rows, float* pooled, void* stream) {
    if (!logits || !w || !idx || !rows || !pooled || E < 1 || E > 32 || ldl < E || top_k < 1 || top_k > 4 || top_k > E) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    hipLaunchKernelGGL(pooled_softmax_route_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ldl, HW, E, inv_temp, top_k, threshold, B, w,
                       idx, rows, pooled);
    return ymk_launch_status();
}

// ---- scene statistics of the MoT router (mot/router.py:166-192 compute_scene_stats, :224-240 scene_projector) ----------------------------------
// Stage 1, workgroup = (image, chunk of rows), a wave per pixel with its lanes over the channels: sum |x(.., w+1) - x(.., w)|, sum |x(.., h+1, ..) -
// x(.., h, ..)|, and of the per-pixel energy e = mean_c x^2: sum e, sum e^2.  part fp32 [B][nchunk][4].
__global__ __launch_bounds__(256) void scene_partials_kernel(int dt, const void* x, int ldx, int H, int W, int C, int rpc, int nchunk,
                                                              float* __restrict__ part) {
    __shared__ float sh[4][4];
    const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y0 = ch * rpc, y1 = min(H, y0 + rpc);
    const int64_t base = (int64_t)b * H * W * ldx;
    float adx = 0.f, ady = 0.f, e1 = 0.f, e2 = 0.f;
    for (int p = y0 * W + wave; p < y1 * W; p += 4) {
        const int yy = p / W, xx = p - yy * W;
        const int64_t o = base + (int64_t)p * ldx;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = ldv(x, dt, o + c);
            s2 += v * v;
            if (xx + 1 < W) adx += fabsf(ldv(x, dt, o + ldx + c) - v);
            if (yy + 1 < H) ady += fabsf(ldv(x, dt, o + (int64_t)W * ldx + c) - v);
        }
#pragma unroll
        for (int q = 32; q > 0; q >>= 1) s2 += __shfl_xor(s2, q);
        const float e = s2 / (float)C;
        e1 += e;
        e2 += e * e;
    }
#pragma unroll
    for (int q = 32; q > 0; q >>= 1) { adx += __shfl_xor(adx, q); ady += __shfl_xor(ady, q); }
    if (lane == 0) { sh[wave][0] = adx; sh[wave][1] = ady; sh[wave][2] = e1; sh[wave][3] = e2; }
    __syncthreads();
    if (threadIdx.x < 4)
        part[((int64_t)b * nchunk + ch) * 4 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__device__ __forceinline__ double scene_block_sum(double v, double* sh) {   // 256 threads; every thread returns the total
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// Stage 2, one workgroup per image, fp64 combines of fp32 inputs: cs = per-channel [mean | biased std] (ymk_channel_stats), p4 / p2 = the
// adaptive average pools to (min(4,H), min(4,W)) / (min(2,H), min(2,W)) as fp32 [B][n4|n2][C]; stats = (high_frequency, heterogeneity,
// multi_scale); bias = W2 silu(W1 stats + b1) + b2 (+ base[b]).
__global__ __launch_bounds__(256) void scene_bias_kernel(const float* cs, const float* p4, int n4, const float* p2, int n2, const float* part,
                                                          int nchunk, int H, int W, int C, const float* w1, const float* b1,
                                                          const float* w2, const float* b2, int hidden, int E, const float* base,
                                                          float* stats, float* bias) {
    __shared__ double shd[256];
    __shared__ float sst[3];
    const int b = blockIdx.x, t = threadIdx.x;
    const double eps = 1.1920928955078125e-07;   // torch.finfo(torch.float32).eps
    double sm = 0.0, sq = 0.0;
    for (int c = t; c < C; c += 256) {
        const double m = cs[(int64_t)b * 2 * C + c], sd = cs[(int64_t)b * 2 * C + C + c];
        sm += m;
        sq += sd * sd + m * m;
    }
    const double ex = scene_block_sum(sm, shd) / C, ex2 = scene_block_sum(sq, shd) / C;
    const double var = ex2 - ex * ex;
    auto pool_var = [&](const float* pp, int n) {
        const int64_t tot = (int64_t)n * C;
        const float* q = pp + (int64_t)b * tot;
        double s = 0.0;
        for (int64_t i = t; i < tot; i += 256) s += q[i];
        const double mean = scene_block_sum(s, shd) / (double)tot;
        double d2 = 0.0;
        for (int64_t i = t; i < tot; i += 256) { const double d = q[i] - mean; d2 += d * d; }
        return scene_block_sum(d2, shd) / (double)tot;
    };
    const double v4 = pool_var(p4, n4), v2 = pool_var(p2, n2);
    if (t == 0) {
        double pdx = 0.0, pdy = 0.0, pe1 = 0.0, pe2 = 0.0;
        for (int k = 0; k < nchunk; ++k) {
            const float* q = part + ((int64_t)b * nchunk + k) * 4;
            pdx += q[0]; pdy += q[1]; pe1 += q[2]; pe2 += q[3];
        }
        const double rms = fmax(sqrt(fmax(ex2, 0.0)), eps);
        const double dx = W > 1 ? pdx / ((double)C * H * (W - 1)) : 0.0, dy = H > 1 ? pdy / ((double)C * (H - 1) * W) : 0.0;
        const double hw = (double)H * W, m1 = pe1 / hw, sde = sqrt(fmax(pe2 / hw - m1 * m1, 0.0));
        sst[0] = (float)(0.5 * (dx + dy) / rms);
        sst[1] = (float)(sde / fmax(m1, eps));
        sst[2] = (float)(fabs(v4 - v2) / fmax(var, eps));
        stats[b * 3 + 0] = sst[0]; stats[b * 3 + 1] = sst[1]; stats[b * 3 + 2] = sst[2];
    }
    __syncthreads();
    if (t < E) {
        float o = b2[t];
        for (int j = 0; j < hidden; ++j) {
            float h = b1[j] + w1[j * 3 + 0] * sst[0] + w1[j * 3 + 1] * sst[1] + w1[j * 3 + 2] * sst[2];
            h = h / (1.0f + expf(-h));
            o += w2[t * hidden + j] * h;
        }
        bias[b * E + t] = base ? o + base[b * E + t] : o;
    }
}

extern "C" size_t ymk_scene_workspace_bytes(int32_t B, int32_t H) { return (size_t)(B > 0 ? B : 0) * (H < 64 ? (H > 0 ? H : 1) : 64) * 4 * sizeof(float); }

extern "C" int ymk_scene_bias(int32_t dtype, const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, int32_t C, const float* chan_stats,
                              const float* pool4, const float* pool2, const float* w1, const float* b1, const float* w2, const float* b2,
                              int32_t hidden, int32_t E, const float* base, float* stats, float* bias, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!x || !chan_stats || !pool4 || !pool2 || !w1 || !b1 || !w2 || !b2 || !stats || !bias || !workspace || bad_dt(dtype) || C < 1 ||
        ldx < C || hidden < 1 || E < 1 || E > 8)
        return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    if (workspace_bytes < ymk_scene_workspace_bytes(B, H)) return YMK_E_WORKSPACE;
    const int nchunk = H < 64 ? H : 64, rpc = (H + nchunk - 1) / nchunk;
    const int nch = (H + rpc - 1) / rpc;   // chunks that own at least one row
    float* part = static_cast<float*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(scene_partials_kernel, dim3(B * nch), dim3(256), 0, s, dtype, x, ldx, H, W, C, rpc, nch, part);
    const int n4 = (H < 4 ? H : 4) * (W < 4 ? W : 4), n2 = (H < 2 ? H : 2) * (W < 2 ? W : 2);
    hipLaunchKernelGGL(scene_bias_kernel, dim3(B), dim3(256), 0, s, chan_stats, pool4, n4, pool2, n2, part, nch, H, W, C, w1, b1, w2, b2,
                       hidden, E, base, stats, bias);
    return ymk_launch_status();
}

extern "C" int ymk_gated_route_decide(const float* g, int32_t ldg, const float* loc, int32_t ldloc, const float* cplx, int32_t ldc,
                                      int32_t B, int32_t E, float alpha, float inv_temp, int32_t clamp_mode, int32_t top_k, float* w,
                                      int32_t* idx, int32_t* idx_slot_major, float* probs, void* stream) {
    if (!g || !loc || !cplx || !w || !idx || !idx_slot_major || !probs || E < 1 || E > 64 || top_k < 1 || top_k > 8 || top_k > E || ldg < E ||
        ldloc < E || ldc < 1 || clamp_mode < 0 || clamp_mode > 2)
        return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    hipLaunchKernelGGL(gated_decide_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, g, ldg, loc, ldloc, cplx, ldc, B, E, alpha,
                       inv_temp, clamp_mode, top_k, w, idx, idx_slot_major, probs);
    return ymk_launch_status();
}

extern "C" int ymk_batch_scale(float* w, int32_t B, int32_t K, const float* logit, int32_t ldl, float lo, float hi, void* stream) {
    if (!w || !logit || K < 1 || ldl < 1 || !(lo <= hi)) return YMK_E_BADARG;
    if (B <= 0) return YMK_OK;
    hipLaunchKernelGGL(batch_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w, B, K, logit, ldl, lo, hi);
    return ymk_launch_status();
}

extern "C" int ymk_expert_gather(int32_t dtype, const void* f_all, int32_t ldf, const int32_t* idx, int32_t B, int32_t HW, int32_t OC,
                                 int32_t K, int32_t E, void* out, void* stream) {
    if (!f_all || !idx || !out || bad_dt(dtype) || OC < 1 || K < 1 || E < 1 || ldf < E * OC) return YMK_E_BADARG;
    if (B <= 0 || HW <= 0) return YMK_OK;
    const int V = vecw(dtype), es = dtype == YMK_BF16 ? 2 : 4;
    if (OC % V == 0 && ldf % V == 0 && al16(f_all) && al16(out)) {
        LAUNCH(expert_gather_vec_kernel, (int64_t)K * B * HW * (OC / V), es, (const char*)f_all, (int64_t)ldf * es, idx, B, HW, OC / V, K,
               (int64_t)OC * es, (char*)out);
        return ymk_launch_status();
    }
    LAUNCH(expert_gather_kernel, (int64_t)K * B * HW * OC, dtype, f_all, ldf, idx, B, HW, OC, K, out);
    return ymk_launch_status();
}

extern "C" int ymk_expert_dw3(int32_t dtype, const void* x, int32_t ldx, const void* w, const int32_t* dil, const int32_t* idx, int32_t B,
                              int32_t H, int32_t W, int32_t C, int32_t K, int32_t E, void* out, void* stream) {
    if (!x || !w || !dil || !idx || !out || bad_dt(dtype) || C < 1 || K < 1 || E < 1 || ldx < C) return YMK_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0) return YMK_OK;
    LAUNCH(expert_dw3_kernel, (int64_t)K * B * H * W * C, dtype, x, ldx, w, dil, idx, B, H, W, C, K, out);
    return ymk_launch_status();
}

extern "C" int ymk_channel_shuffle_cat(int32_t dtype, const void* a, int32_t lda, int32_t Ca, const void* b, int32_t ldb, int32_t Cb,
                                       int32_t groups, void* y, int32_t ldy, int64_t npix, void* stream) {
    if (!a || !b || !y || bad_dt(dtype) || Ca < 1 || Cb < 1 || groups < 1 || (Ca + Cb) % groups || lda < Ca || ldb < Cb ||
        ldy < Ca + Cb)
        return YMK_E_BADARG;
    if (npix <= 0) return YMK_OK;
    const int V = vecw(dtype);
    if (groups == 2 && Ca == Cb && Ca % (V / 2) == 0 && ldy % V == 0 && al16(y)) {   // the gated block's case: even / odd interleave
        const int64_t total = npix * (Ca / (V / 2));
        if (dtype == YMK_BF16) LAUNCH(shuffle2_vec_kernel<h16_t>, total, (const h16_t*)a, lda, (const h16_t*)b, ldb, Ca, (h16_t*)y, ldy, npix);
        else LAUNCH(shuffle2_vec_kernel<float>, total, (const float*)a, lda, (const float*)b, ldb, Ca, (float*)y, ldy, npix);
        return ymk_launch_status();
    }
    LAUNCH(shuffle_cat_kernel, npix * (Ca + Cb), dtype, a, lda, Ca, b, ldb, Cb, groups, y, ldy, npix);
    return ymk_launch_status();
}
