// Batched NMS on the GPU, bit-compatible with the reference's CPU result.
// Reference: non_max_suppression (ultralytics/utils/nms.py:13-171) + TorchNMS.nms (:245-302).
//   candidates: conf > thres (best class :124-129, or every class when multi_label :119-123)
//   order: score descending (ties: lower candidate index first), cap max_nms (:142-146)
//   boxes + cls*max_wh (:148-154); greedy: drop j when !(IoU(i,j) <= thr) (:300); [:max_det] (:162)
// Pipeline (all per-image work runs batched over B):
//   count -> scan -> [top-max_nms selection: 3-level radix select on the score bits, only for images with more candidates
//   than max_nms] -> emit (ordered compaction) -> rank (counting sort, stable)
//   -> chunked greedy suppression against the kept list (one workgroup per image, no n^2 mask).
// Compile with -ffp-contract=off: IoU arithmetic must round exactly like the reference.
#include "ymk_common.h"

struct NmsWs {
    int* cnt;        // [B][A]
    float* bconf;    // [B][A]
    int* bcls;       // [B][A]
    int* blocksum;   // [B][NBLK]
    int* blockoff;   // [B][NBLK]
    int* ncand;      // [B] candidates (clamped to capc)
    int* nsort;      // [B] min(ncand, ns)
    float* cbox;     // [B][capc][4] xyxy
    float* cscore;   // [B][capc]
    int* ccls;       // [B][capc]
    int* canchor;    // [B][capc]
    float* sbox;     // [B][ns][4]
    float* sscore;   // [B][ns]
    int* scls;       // [B][ns]
    int* sanchor;    // [B][ns]
    int* keep_pos;   // [B][max_det_cap]
    // top-max_nms selection (images with more than max_nms candidates; utils/nms.py:142-146 keeps the max_nms best by score)
    int* sel;            // [B] 1: this image selects by threshold key
    unsigned* thr_key;   // [B] score bits T: every candidate with bits > T is taken, and the first eq_take (anchor-major order) with bits == T
    int* n_gt;           // [B] candidates with bits > T
    int* eq_take;        // [B]
    int* need;           // [B] running: how many still to take inside the current prefix
    unsigned* hist;      // [B][2048]
    int* blocksum_e;     // [B][NBLK] per-block counts of bits == T
    int* blockoff_e;     // [B][NBLK]
    int nblk, capc, ns, nw;
    int rows;            // rows of y per image: 4 + nc + extra (mask coefficients of a Segment head ride behind the class rows)
    size_t total;
};

#define NMS_MAXDET_CAP 1024

static NmsWs nms_layout(void* base, int B, int nc, int A, int multi, int max_nms) {
    NmsWs w;
    w.rows = 4 + nc;
    w.nblk = (A + 255) / 256;
    const int64_t full = multi ? (int64_t)A * nc : (int64_t)A;
    w.capc = (int)(full < (int64_t)max_nms ? full : (int64_t)max_nms);   // never more than max_nms rows reach the ordering stage
    w.ns = w.capc;
    w.nw = (w.ns + 63) / 64;
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) {
        char* r = p ? p + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return r;
    };
    w.cnt = (int*)take((size_t)B * A * 4);
    w.bconf = (float*)take((size_t)B * A * 4);
    w.bcls = (int*)take((size_t)B * A * 4);
    w.blocksum = (int*)take((size_t)B * w.nblk * 4);
    w.blockoff = (int*)take((size_t)B * w.nblk * 4);
    w.ncand = (int*)take((size_t)B * 4);
    w.nsort = (int*)take((size_t)B * 4);
    w.cbox = (float*)take((size_t)B * w.capc * 16);
    w.cscore = (float*)take((size_t)B * w.capc * 4);
    w.ccls = (int*)take((size_t)B * w.capc * 4);
    w.canchor = (int*)take((size_t)B * w.capc * 4);
    w.sbox = (float*)take((size_t)B * w.ns * 16);
    w.sscore = (float*)take((size_t)B * w.ns * 4);
    w.scls = (int*)take((size_t)B * w.ns * 4);
    w.sanchor = (int*)take((size_t)B * w.ns * 4);
    w.keep_pos = (int*)take((size_t)B * NMS_MAXDET_CAP * 4);
    w.sel = (int*)take((size_t)B * 4);
    w.thr_key = (unsigned*)take((size_t)B * 4);
    w.n_gt = (int*)take((size_t)B * 4);
    w.eq_take = (int*)take((size_t)B * 4);
    w.need = (int*)take((size_t)B * 4);
    w.hist = (unsigned*)take((size_t)B * 2048 * 4);
    w.blocksum_e = (int*)take((size_t)B * w.nblk * 4);
    w.blockoff_e = (int*)take((size_t)B * w.nblk * 4);
    w.total = off;
    return w;
}

extern "C" size_t ymk_nms_workspace_bytes(int32_t B, int32_t nc, int32_t A, int32_t multi_label, int32_t max_nms) {
    if (B <= 0 || A <= 0 || nc <= 0 || max_nms <= 0) return 0;
    return nms_layout(nullptr, B, nc, A, multi_label && nc > 1, max_nms).total;
}

// ---- 1. per-anchor candidate count (+ best class for single-label) ----------------
// class_keep (nullable, [nc] bytes): the `classes=` filter (utils/nms.py:63,132) — applied to the candidate's class AFTER the
// best-class choice of the single-label path, exactly where the reference filters its candidate rows.
__global__ __launch_bounds__(256) void nms_count_kernel(const float* __restrict__ y, int nc, int A, float conf, int multi,
                                                       const unsigned char* __restrict__ class_keep, NmsWs w) {
    __shared__ int wsum[4];
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (a < A) {
        const float* p = y + ((size_t)b * w.rows + 4) * A + a;
        // class scores are read eight at a time (independent loads in flight); the compares keep class order, so
        // the first maximum wins exactly as in the sequential scan (utils/nms.py:124-129 amax/argmax semantics)
        if (multi) {
            int k = 0;
            for (; k + 8 <= nc; k += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(k + q) * A];
#pragma unroll
                for (int q = 0; q < 8; ++q) c += (v[q] > conf) && (!class_keep || class_keep[k + q]);
            }
            for (; k < nc; ++k) c += (p[(size_t)k * A] > conf) && (!class_keep || class_keep[k]);
        } else {
            float best = p[0];
            int bi = 0;
            int k = 1;
            for (; k + 8 <= nc; k += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(k + q) * A];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (v[q] > best) { best = v[q]; bi = k + q; }
            }
            for (; k < nc; ++k) {
                const float v = p[(size_t)k * A];
                if (v > best) { best = v; bi = k; }
            }
            c = (best > conf) && (!class_keep || class_keep[bi]);
            w.bconf[(size_t)b * A + a] = best;
            w.bcls[(size_t)b * A + a] = bi;
        }
        w.cnt[(size_t)b * A + a] = c;
    }
    int s = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) w.blocksum[b * w.nblk + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---- 1b. the same from the per-anchor best class the producer of y already wrote (ymk_detect_decode): single-label only ---------
// w.bconf / w.bcls point at the producer's arrays; the nc class rows of y are not read.
__global__ __launch_bounds__(256) void nms_count_best_kernel(int A, int nc, float conf, const unsigned char* __restrict__ class_keep, NmsWs w) {
    __shared__ int wsum[4];
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (a < A) {
        // the class id is the PRODUCER's (Detect's decode epilogue): an id outside [0, nc) — a caller that handed over its own arrays — is
        // not a candidate rather than an index into class_keep
        const unsigned cls = (unsigned)w.bcls[(size_t)b * A + a];
        c = (w.bconf[(size_t)b * A + a] > conf) && cls < (unsigned)nc && (!class_keep || class_keep[cls]);
        w.cnt[(size_t)b * A + a] = c;
    }
    int s = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) w.blocksum[b * w.nblk + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---- 2. exclusive scan of block sums (one wavefront per image) ---------------------
__global__ __launch_bounds__(64) void nms_scan_kernel(NmsWs w, int* __restrict__ status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int run = 0;
    for (int i0 = 0; i0 < w.nblk; i0 += 64) {
        const int i = i0 + lane;
        const int v = i < w.nblk ? w.blocksum[b * w.nblk + i] : 0;
        int inc = v;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o);
            if (lane >= o) inc += n;
        }
        if (i < w.nblk) w.blockoff[b * w.nblk + i] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) {
        // more candidates than the ordering stage takes (max_nms, utils/nms.py:142-146): the image goes through the radix select
        const int over = run > w.capc;
        const int n = over ? w.capc : run;
        w.ncand[b] = n;
        w.nsort[b] = n;
        w.sel[b] = over;
        w.thr_key[b] = 0u;       // bits > 0: every candidate (scores that pass conf >= 0 are positive)
        w.n_gt[b] = 0;
        w.eq_take[b] = 0;
        w.need[b] = w.capc;
    }
    for (int i = lane; i < 2048; i += 64) w.hist[(size_t)b * 2048 + i] = 0u;
}

// ---- 2b. top-max_nms selection: radix select of the threshold key ------------------------------------------------------------
// The reference sorts ALL candidates by score and keeps the first max_nms (utils/nms.py:142-146; with conf 0.001 and multi_label a
// dense 640^2 image has up to 672 000 (anchor, class) pairs).  Instead of ordering them all: find the score bit pattern T with
// #{bits > T} < max_nms <= #{bits >= T} by three histogram levels over the 32 key bits (11 + 11 + 10), then emit every candidate
// above T and the first (max_nms - #{bits > T}) candidates equal to T in anchor-major order — the order the stable sort would give them.
// Level l histograms the candidates whose higher bits equal the prefix found so far.  Images with <= max_nms candidates skip all of it.
__device__ __forceinline__ int nms_level_shift(int level) { return level == 0 ? 21 : (level == 1 ? 10 : 0); }

__global__ __launch_bounds__(256) void nms_hist_kernel(const float* __restrict__ y, int nc, int A, float conf, int multi,
                                                      const unsigned char* __restrict__ class_keep, NmsWs w, int level) {
    __shared__ unsigned h[2048];
    const int b = blockIdx.y;
    if (!w.sel[b]) return;   // workgroup-uniform
    const int a = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0u;
    __syncthreads();
    const int shift = nms_level_shift(level);
    const unsigned prefix = w.thr_key[b];
    const int hs = level == 0 ? 32 : (level == 1 ? 21 : 10);   // bits above this position must match the prefix
    const unsigned mask = level == 2 ? 1023u : 2047u;
    auto add = [&](float v) {
        const unsigned k = __float_as_uint(v);
        if (hs < 32 && (k >> hs) != (prefix >> hs)) return;
        atomicAdd(&h[(k >> shift) & mask], 1u);
    };
    if (a < A) {
        if (multi) {
            const float* p = y + ((size_t)b * w.rows + 4) * A + a;
            for (int k = 0; k < nc; ++k) {
                const float v = p[(size_t)k * A];
                if (v > conf && (!class_keep || class_keep[k])) add(v);
            }
        } else if (w.cnt[(size_t)b * A + a]) add(w.bconf[(size_t)b * A + a]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (h[i]) atomicAdd(&w.hist[(size_t)b * 2048 + i], h[i]);
}

__global__ __launch_bounds__(256) void nms_pick_kernel(NmsWs w, int level) {
    __shared__ unsigned h[2048];
    __shared__ unsigned wtot[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!w.sel[b]) return;
    // read BEFORE the barriers: the one thread that finds the threshold bin rewrites need[b] below, and a wave that is still on its
    // way to this read would then test its bins against the NEXT level's count (a second "pick", OR-ed into thr_key)
    const unsigned need = (unsigned)w.need[b];
    unsigned* gh = w.hist + (size_t)b * 2048;
    for (int i = tid; i < 2048; i += 256) { h[i] = gh[i]; gh[i] = 0u; }   // cleared for the next level
    __syncthreads();
    // thread t owns bins [2047 - 8t - 7, 2047 - 8t]: walking threads upward walks the bins from the top down
    unsigned own = 0u;
#pragma unroll
    for (int q = 0; q < 8; ++q) own += h[2047 - (tid * 8 + q)];
    unsigned inc = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    unsigned above = inc - own;   // candidates in the bins above this thread's range
    for (int k = 0; k < wave; ++k) above += wtot[k];
    if (above < need && need <= above + own) {   // exactly one thread: the threshold bin is in its range
        unsigned acc = above;
        int t = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int bin = 2047 - (tid * 8 + q);
            const unsigned c = h[bin];
            if (acc < need && need <= acc + c) { t = bin; break; }
            acc += c;
        }
        w.thr_key[b] |= (unsigned)t << nms_level_shift(level);
        w.n_gt[b] += (int)acc;
        w.need[b] = (int)(need - acc);
        if (level == 2) w.eq_take[b] = (int)(need - acc);
    }
}

// per-anchor counts against the threshold: low 16 bits #{bits > T}, high 16 bits #{bits == T} (nc < 65536)
__global__ __launch_bounds__(256) void nms_count2_kernel(const float* __restrict__ y, int nc, int A, float conf, int multi,
                                                        const unsigned char* __restrict__ class_keep, NmsWs w) {
    __shared__ int wsum[4], wsum_e[4];
    const int b = blockIdx.y;
    if (!w.sel[b]) return;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const unsigned T = w.thr_key[b];
    int c = 0;
    if (a < A) {
        if (multi) {
            const float* p = y + ((size_t)b * w.rows + 4) * A + a;
            for (int k = 0; k < nc; ++k) {
                const float v = p[(size_t)k * A];
                if (v > conf && (!class_keep || class_keep[k])) {
                    const unsigned key = __float_as_uint(v);
                    c += key > T ? 1 : (key == T ? 65536 : 0);
                }
            }
        } else if (w.cnt[(size_t)b * A + a]) {
            const unsigned key = __float_as_uint(w.bconf[(size_t)b * A + a]);
            c = key > T ? 1 : (key == T ? 65536 : 0);
        }
        w.cnt[(size_t)b * A + a] = c;
    }
    int s = c & 0xffff, e = (int)((unsigned)c >> 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); e += __shfl_xor(e, o); }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = s; wsum_e[threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        w.blocksum[b * w.nblk + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        w.blocksum_e[b * w.nblk + blockIdx.x] = wsum_e[0] + wsum_e[1] + wsum_e[2] + wsum_e[3];
    }
}

__global__ __launch_bounds__(64) void nms_scan2_kernel(NmsWs w) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!w.sel[b]) return;
    int run = 0, run_e = 0;
    for (int i0 = 0; i0 < w.nblk; i0 += 64) {
        const int i = i0 + lane;
        const int v = i < w.nblk ? w.blocksum[b * w.nblk + i] : 0;
        const int e = i < w.nblk ? w.blocksum_e[b * w.nblk + i] : 0;
        int inc = v, ince = e;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o), ne = __shfl_up(ince, o);
            if (lane >= o) { inc += n; ince += ne; }
        }
        if (i < w.nblk) { w.blockoff[b * w.nblk + i] = run + inc - v; w.blockoff_e[b * w.nblk + i] = run_e + ince - e; }
        run += __shfl(inc, 63);
        run_e += __shfl(ince, 63);
    }
}

// ---- 3. ordered compaction of candidates -------------------------------------------
__global__ __launch_bounds__(256) void nms_emit_kernel(const float* __restrict__ y, int nc, int A, float conf, int multi,
                                                      const unsigned char* __restrict__ class_keep, NmsWs w) {
    __shared__ int wsum[4], wsum_e[4];
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sel = w.sel[b];
    // per-anchor counts.  An image that selects (more than max_nms candidates) holds them packed: low 16 bits = candidates above
    // its threshold key, high 16 bits = candidates equal to it; any other image holds its plain candidate count (T = 0: all above).
    const int c = a < A ? w.cnt[(size_t)b * A + a] : 0;
    const int cg = sel ? (c & 0xffff) : c, ce = sel ? (int)((unsigned)c >> 16) : 0;
    int inc = cg, ince = ce;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(inc, o), ne = __shfl_up(ince, o);
        if (lane >= o) { inc += n; ince += ne; }
    }
    if (lane == 63) { wsum[wv] = inc; wsum_e[wv] = ince; }
    __syncthreads();
    int pos = w.blockoff[b * w.nblk + blockIdx.x] + inc - cg;
    int pos_e = sel ? w.blockoff_e[b * w.nblk + blockIdx.x] + ince - ce : 0;
    for (int k = 0; k < wv; ++k) { pos += wsum[k]; pos_e += wsum_e[k]; }
    if (c == 0) return;
    const unsigned T = w.thr_key[b];
    const int n_gt = w.n_gt[b], eq_take = w.eq_take[b];
    const float* yb = y + (size_t)b * w.rows * A + a;
    const float cx = yb[0], cy = yb[(size_t)A], bw = yb[2 * (size_t)A], bh = yb[3 * (size_t)A];
    const float hw = bw / 2.0f, hh = bh / 2.0f;  // xywh2xyxy (utils/ops.py:248-264)
    const float x1 = cx - hw, y1 = cy - hh, x2 = cx + hw, y2 = cy + hh;
    auto put = [&](int at, float score, int cls) {
        if (at >= w.capc) return;
        const size_t o = (size_t)b * w.capc + at;
        w.cbox[o * 4 + 0] = x1; w.cbox[o * 4 + 1] = y1; w.cbox[o * 4 + 2] = x2; w.cbox[o * 4 + 3] = y2;
        w.cscore[o] = score; w.ccls[o] = cls; w.canchor[o] = a;
    };
    // Rows above the threshold fill [0, n_gt) in anchor-major order, the taken rows equal to it [n_gt, n_gt + eq_take): equal
    // scores never straddle the two ranges, so "lower row first" in the ordering stage is still "lower candidate index first".
    auto route = [&](float v, int cls) {
        const unsigned key = __float_as_uint(v);
        if (key > T) put(pos++, v, cls);
        else if (key == T) { if (pos_e < eq_take) put(n_gt + pos_e, v, cls); ++pos_e; }
    };
    if (multi) {
        for (int k = 0; k < nc; ++k) {
            const float v = yb[(size_t)(4 + k) * A];
            if (v > conf && (!class_keep || class_keep[k])) route(v, k);
        }
    } else {
        route(w.bconf[(size_t)b * A + a], w.bcls[(size_t)b * A + a]);
    }
}

// ---- 4. stable descending counting-rank sort ---------------------------------------
// rank(i) = #{j : key_j > key_i} with the 64-bit key (score bits << 32 | ~index): for the positive scores
// that pass the confidence filter the IEEE bit pattern is order preserving, and the inverted index in the
// low word makes the lower candidate index win ties (stable order) with ONE integer compare per pair.
// Opt-in (YMK_ENABLE bit 128): images with at most NMS_RANK_MAX candidates are ordered by THIS kernel on (n / 256) workgroups each instead
// of one 1024-thread workgroup per image walking a sorting network (64 workgroups on 256 CUs); larger images (O(n^2) would hurt) go to
// nms_sort_kernel.  Both are launched then; each returns at once for the images of the other.  rank_max < 0: every image (the fall-back
// when the sorting kernel is switched off).  Round 4: slower than the sort on the bench step (see ymk_nms_batched).
#ifndef NMS_RANK_MAX
#define NMS_RANK_MAX 4096   // (tests/hostemu builds with 1200: its small fixtures then reach both ordering kernels)
#endif
__global__ __launch_bounds__(256) void nms_rank_kernel(NmsWs w, int rank_max) {
    __shared__ unsigned long long tk[256];
    const int b = blockIdx.y;
    const int n = w.ncand[b];
    const int i0 = blockIdx.x * 256;
    if (i0 >= n || (rank_max >= 0 && n > rank_max)) return;
    const int i = i0 + threadIdx.x;
    const float* sc = w.cscore + (size_t)b * w.capc;
    const float si = i < n ? sc[i] : 0.f;
    const unsigned long long ki = ((unsigned long long)__float_as_uint(si) << 32) | (unsigned)(~i);
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
        tk[threadIdx.x] = j < n ? (((unsigned long long)__float_as_uint(sc[j]) << 32) | (unsigned)(~j)) : 0ull;
        __syncthreads();
#pragma unroll 8
        for (int jj = 0; jj < 256; ++jj) rank += tk[jj] > ki;   // padding keys are 0: never greater
    }
    if (i < n && rank < w.ns) {
        const size_t s = (size_t)b * w.capc + i, d = (size_t)b * w.ns + rank;
        const f32x4 bx = *reinterpret_cast<const f32x4*>(w.cbox + s * 4);
        *reinterpret_cast<f32x4*>(w.sbox + d * 4) = bx;
        w.sscore[d] = si; w.scls[d] = w.ccls[s]; w.sanchor[d] = w.canchor[s];
    }
}


// ---- 4b. same order by an in-LDS bitonic sort (one workgroup per image, up to 16384 candidates) ----------
// The 64-bit keys are unique (the low word carries the candidate index), so any correct sort yields exactly the
// permutation of the counting rank above; this one is O(n log^2 n) instead of O(n^2).
#define NMS_SORT_CAP 16384

// ---- 4c. same order again by a stable LSD radix sort of candidate INDICES (4-bit digits of the inverted score bits) -----------
// The bitonic network needs log^2 passes over the keys (105 stages at 16384 elements, 155 us for the batch-64 step with one
// workgroup per image); here a pass is: every thread takes 16 consecutive positions, counts its digits in registers (16 8-bit
// counters in two 64-bit registers — thread-private, so the order inside the thread is kept without atomics), the 16 digit
// channels are scanned across the workgroup as eight registers of two 16-bit fields (values <= 16384 never carry), and every
// index is scattered to  digit base + earlier threads' count + own earlier count.  Digits on which all keys agree are skipped
// (scores in (conf, 1) share their top bits).  LDS: 64 KB of keys (fixed) + two 32 KB index buffers, inside the bitonic
// kernel's 128 KB; the consumed source slots of a thread hold its 16 scanned bases (dynamic indexing without a register tree).
#define NMS_RADIX_MIN 1024   // below: the bitonic network is as fast
#define YMK_OFF_NMS_RADIX 65536u
#define YMK_ON_NMS_RANK_SMALL 128u   // YMK_ENABLE: images with <= NMS_RANK_MAX candidates ordered by nms_rank_kernel on many workgroups
__device__ __forceinline__ void nms_radix_sort(const NmsWs& w, int b, int n, unsigned* lds) {
    unsigned* key = lds;                                                       // [NMS_SORT_CAP]
    unsigned short* buf0 = reinterpret_cast<unsigned short*>(lds + NMS_SORT_CAP);   // [2][NMS_SORT_CAP]
    __shared__ unsigned wtot[16][8];
    __shared__ unsigned wbase[16][8];
    __shared__ unsigned red[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sc = w.cscore + (size_t)b * w.capc;
    unsigned orv = 0u, andv = ~0u;
    for (int i = tid; i < n; i += 1024) {
        const unsigned k = ~__float_as_uint(sc[i]);   // ascending on the inverted bits = descending score
        key[i] = k; orv |= k; andv &= k;
    }
    for (int i = tid; i < NMS_SORT_CAP / 2; i += 1024) reinterpret_cast<unsigned*>(buf0)[i] = (unsigned)(2 * i) | ((unsigned)(2 * i + 1) << 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { orv |= __shfl_xor(orv, o); andv &= __shfl_xor(andv, o); }
    if (lane == 0) { red[0][wave] = orv; red[1][wave] = andv; }
    __syncthreads();
    unsigned diff;
    {
        unsigned o2 = 0u, a2 = ~0u;
#pragma unroll
        for (int q = 0; q < 16; ++q) { o2 |= red[0][q]; a2 &= red[1][q]; }
        diff = o2 ^ a2;
    }
    int cur = 0;
    const int p0 = tid * 16;
    for (int shift = 0; shift < 32; shift += 4) {
        if (((diff >> shift) & 15u) == 0u) continue;   // every key has the same digit here: the pass would be the identity
        unsigned short* src = buf0 + cur * NMS_SORT_CAP;
        unsigned short* dst = buf0 + (cur ^ 1) * NMS_SORT_CAP;
        const u32x4 r0 = *reinterpret_cast<const u32x4*>(src + p0), r1 = *reinterpret_cast<const u32x4*>(src + p0 + 8);
        const unsigned raw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        unsigned long long clo = 0ull, chi = 0ull;
        unsigned meta[16];   // index | digit << 16 | own earlier count << 20
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned idx = (raw[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
            unsigned m = idx;
            if (p0 + k < n) {
                const unsigned d = (key[idx] >> shift) & 15u;
                const unsigned sh = (d & 7u) * 8u;
                const unsigned long long sel = (d & 8u) ? chi : clo;
                m |= (d << 16) | ((unsigned)((sel >> sh) & 0xffull) << 20);
                const unsigned long long one = 1ull << sh;
                if (d & 8u) chi += one; else clo += one;
            }
            meta[k] = m;
        }
        unsigned c2[8], in2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long sel = j < 4 ? clo : chi;
            const unsigned pairbits = (unsigned)(sel >> ((j & 3) * 16)) & 0xffffu;   // digits 2j (low byte), 2j + 1 (high byte)
            c2[j] = (pairbits & 0xffu) | ((pairbits >> 8) << 16);
            in2[j] = c2[j];
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned v = __shfl_up(in2[j], o);
                if (lane >= o) in2[j] += v;
            }
        if (lane == 63) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wtot[wave][j] = in2[j];
        }
        __syncthreads();
        if (wave == 0) {   // 16 wave totals -> exclusive base of every (wave, digit): digit base + earlier waves
            unsigned tt[8], ti[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { tt[j] = lane < 16 ? wtot[lane][j] : 0u; ti[j] = tt[j]; }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned v = __shfl_up(ti[j], o);
                    if (lane >= o) ti[j] += v;
                }
            unsigned acc = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned tot = __shfl(ti[j], 15);
                const unsigned lo = acc; acc += tot & 0xffffu;
                const unsigned hi = acc; acc += tot >> 16;
                if (lane < 16) wbase[lane][j] = (lo | (hi << 16)) + ti[j] - tt[j];
            }
        }
        __syncthreads();
        {   // this thread's 16 bases into its own (consumed) source slots
            u32x4 b0, b1;
            b0.x = wbase[wave][0] + in2[0] - c2[0]; b0.y = wbase[wave][1] + in2[1] - c2[1];
            b0.z = wbase[wave][2] + in2[2] - c2[2]; b0.w = wbase[wave][3] + in2[3] - c2[3];
            b1.x = wbase[wave][4] + in2[4] - c2[4]; b1.y = wbase[wave][5] + in2[5] - c2[5];
            b1.z = wbase[wave][6] + in2[6] - c2[6]; b1.w = wbase[wave][7] + in2[7] - c2[7];
            *reinterpret_cast<u32x4*>(src + p0) = b0;
            *reinterpret_cast<u32x4*>(src + p0 + 8) = b1;
        }
        __threadfence_block();   // the 2-byte reads below alias the 16-byte stores above (same thread, same wave: in order once emitted in order)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (p0 + k < n) {
                const unsigned m = meta[k];
                const unsigned r = (unsigned)src[p0 + ((m >> 16) & 15u)] + (m >> 20);
                dst[r] = (unsigned short)(m & 0xffffu);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    const unsigned short* fin = buf0 + cur * NMS_SORT_CAP;
    const int mo = n < w.ns ? n : w.ns;
    for (int r = tid; r < mo; r += 1024) {
        const int i = fin[r];
        const size_t s = (size_t)b * w.capc + i, d = (size_t)b * w.ns + r;
        *reinterpret_cast<f32x4*>(w.sbox + d * 4) = *reinterpret_cast<const f32x4*>(w.cbox + s * 4);
        w.sscore[d] = sc[i]; w.scls[d] = w.ccls[s]; w.sanchor[d] = w.canchor[s];
    }
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(NmsWs w, int radix, int rank_max) {
    __shared__ unsigned long long key[NMS_SORT_CAP];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = w.ncand[b];
    if (n <= rank_max) return;          // ordered by nms_rank_kernel (n <= 0 included: rank_max >= 0)
    if (n > NMS_RADIX_MIN && radix) {   // workgroup-uniform
        nms_radix_sort(w, b, n, reinterpret_cast<unsigned*>(key));
        return;
    }
    int np = 64;
    while (np < n) np <<= 1;
    const float* sc = w.cscore + (size_t)b * w.capc;
    for (int i = tid; i < np; i += 1024)
        key[i] = i < n ? (((unsigned long long)__float_as_uint(sc[i]) << 32) | (unsigned)(~i)) : 0ull;
    __syncthreads();
    // Compare-exchange t of a stage touches elements i and i|j with i = 2*(t - t%j) + t%j.  For j <= 32 the 64
    // exchanges of a wavefront stay inside its own 128 consecutive elements, stage after stage, so those stages
    // need no workgroup barrier — only the wave's own LDS traffic in order (fence) — which removes 3/4 of them.
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np >> 1); t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const unsigned long long ka = key[i], kb = key[l];
                const bool desc = (i & k) == 0;
                if ((ka < kb) == desc) { key[i] = kb; key[l] = ka; }
            }
            if (j > 32 || j == 1) __syncthreads();   // j == 1 closes the wave-local run before the next k
            else { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }
        }
    }
    const int m = n < w.ns ? n : w.ns;
    for (int r = tid; r < m; r += 1024) {
        const unsigned long long kr = key[r];
        const int i = (int)(~(unsigned)kr);
        const size_t s = (size_t)b * w.capc + i, d = (size_t)b * w.ns + r;
        *reinterpret_cast<f32x4*>(w.sbox + d * 4) = *reinterpret_cast<const f32x4*>(w.cbox + s * 4);
        w.sscore[d] = __uint_as_float((unsigned)(kr >> 32)); w.scls[d] = w.ccls[s]; w.sanchor[d] = w.canchor[s];
    }
}

// ---- 5. greedy suppression, one workgroup (4 wavefronts) per image --------------------
// Sorted candidates are consumed 64 at a time.  Every wavefront holds the same 64 candidates (one per
// lane).  Phase A: each wavefront tests them against its quarter of the boxes kept so far (LDS
// broadcast reads) and ballots a "suppressed by an earlier keep" word.  Phase B: each wavefront
// evaluates 16 of the 64 intra-chunk columns (v_readlane broadcast) into per-lane mask words.
// Then the 64-step in-order resolve runs (redundantly per wavefront, register only), survivors
// are appended to the kept list and written out.  Work is O(n * kept) <= n * max_det IoUs instead of
// the O(n^2) bitmask, nothing but the final detections touches HBM, and the loop stops at max_det.
// IoU arithmetic and decisions are identical to TorchNMS.nms (drop j when !(IoU <= thr)).
__device__ __forceinline__ float rdlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// NW waves per image (round 5: 16, was 4).  The pass is a chain of chunk steps — 64 candidates against the kept list (phase A, split
// over the waves), against each other (phase B), an in-order resolve — and with the parity-pinned weights an image hands over a few
// thousand candidates of which 300 survive: phase A is 64 x kept IoUs per chunk, a dependent VALU chain that ONE wave per SIMD issued at
// a fraction of the SIMD's rate (270 us of the 320 us NMS step, profiles/r05a_step_dispatch_pmc.txt).  Sixteen waves give every SIMD four
// chains to interleave and a quarter of the kept list each; the resolve visits only the candidates the kept list left alive.
template <int NW>
__global__ __launch_bounds__(NW * 64) void nms_greedy_kernel(NmsWs w, float thr, float cls_off, int max_det,
                                                             float* __restrict__ out_dets, int* __restrict__ out_counts,
                                                             int* __restrict__ out_idx) {
    static_assert(64 % NW == 0, "the 64 intra-chunk pivots are dealt evenly to the waves");
    constexpr int NT = NW * 64, PIV = 64 / NW;
    __shared__ float kx1[NMS_MAXDET_CAP], ky1[NMS_MAXDET_CAP], kx2[NMS_MAXDET_CAP], ky2[NMS_MAXDET_CAP],
        kar[NMS_MAXDET_CAP];
    __shared__ unsigned long long supw[NW];
    __shared__ unsigned long long part[NW][64];
    __shared__ unsigned long long kmask;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    const int n = w.nsort[b];
    const float* sb = w.sbox + (size_t)b * w.ns * 4;
    const int* sc = w.scls + (size_t)b * w.ns;
    int kept = 0;
    // the NEXT chunk's candidates are requested while this chunk is resolved: every chunk used to start with its own exposed global round trip
    // (a few thousand candidates = ~50 chunks per image: a third of the pass was that wait)
    f32x4 nbx = {0.f, 0.f, 0.f, 0.f};
    int ncl = 0, nan_ = 0;      // wave 0 (the one that emits the kept rows) also carries the candidates' scores and anchor ids: no load inside the resolve
    float nsc = 0.f;
    if (lane < n) {
        nbx = *reinterpret_cast<const f32x4*>(sb + (size_t)lane * 4); ncl = sc[lane];
        if (wave == 0) { nsc = w.sscore[(size_t)b * w.ns + lane]; nan_ = w.sanchor[(size_t)b * w.ns + lane]; }
    }
    for (int i0 = 0; i0 < n && kept < max_det; i0 += 64) {
        const int i = i0 + lane;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
        const f32x4 rbx = nbx;
        const int rcl = ncl, ran = nan_;
        const float rsc = nsc;
        if (i < n) {
            const float c = (float)rcl * cls_off;
            x1 = rbx.x + c; y1 = rbx.y + c; x2 = rbx.z + c; y2 = rbx.w + c;
        }
        if (i + 64 < n) {
            nbx = *reinterpret_cast<const f32x4*>(sb + (size_t)(i + 64) * 4); ncl = sc[i + 64];
            if (wave == 0) { nsc = w.sscore[(size_t)b * w.ns + i + 64]; nan_ = w.sanchor[(size_t)b * w.ns + i + 64]; }
        }
        const float ai = (x2 - x1) * (y2 - y1);
        // phase A: against this wavefront's share of the kept list
        const int kq = (kept + NW - 1) / NW;
        const int k0 = wave * kq, k1 = min(kept, k0 + kq);
        bool sup = false;
        for (int k = k0; k < k1; ++k) {
            const float xx1 = fmaxf(kx1[k], x1), yy1 = fmaxf(ky1[k], y1);
            const float xx2 = fminf(kx2[k], x2), yy2 = fminf(ky2[k], y2);
            const float ww = fmaxf(xx2 - xx1, 0.f), hh = fmaxf(yy2 - yy1, 0.f);
            const float inter = ww * hh;
            const float iou = inter / ((kar[k] + ai) - inter);   // (deciding from inter against thr * union with the division only in a 2^-21 band: 209 vs 184 us — the branches cost more than the division)
            sup |= !(iou <= thr);
        }
        // (leaving the loop early once every candidate of the chunk is known to be suppressed — the waves publishing their ballots every
        // eighth box — changed nothing at sixteen waves and cost the four-wave form 50 %: profiles/r05_negative_results.txt)
        const unsigned long long sw = __ballot(sup);
        // phase B: PIV intra-chunk columns per wavefront; row = this lane's candidate, earlier box = pivot jj
        unsigned long long bits = 0ull;  // bit jj set: candidate `lane` is suppressed by chunk member jj (jj < lane)
        const int nv = min(64, n - i0);
        for (int q = 0; q < PIV; ++q) {
            const int jj = wave * PIV + q;
            if (jj >= nv) break;
            const float px1 = rdlane(x1, jj), py1 = rdlane(y1, jj), px2 = rdlane(x2, jj), py2 = rdlane(y2, jj);
            const float pa = rdlane(ai, jj);
            const float xx1 = fmaxf(px1, x1), yy1 = fmaxf(py1, y1);
            const float xx2 = fminf(px2, x2), yy2 = fminf(py2, y2);
            const float ww = fmaxf(xx2 - xx1, 0.f), hh = fmaxf(yy2 - yy1, 0.f);
            const float inter = ww * hh;
            const float iou = inter / ((pa + ai) - inter);
            if (lane > jj && !(iou <= thr)) bits |= 1ull << jj;
        }
        if (lane == 0) supw[wave] = sw;
        part[wave][lane] = bits;
        __syncthreads();
        unsigned long long km = 0ull;
        if (wave == 0) {
            unsigned long long cur = 0ull, mine = 0ull;
#pragma unroll
            for (int q = 0; q < NW; ++q) { cur |= supw[q]; mine |= part[q][lane]; }
            if (nv < 64) cur |= ~0ull << nv;  // lanes past the end count as suppressed
            // in-order resolve: candidate t survives iff not suppressed by the kept list nor by an earlier survivor.  Only the candidates the
            // kept list left alive are visited (ascending: the order of the serial loop this replaces — the others never entered km)
            unsigned long long rem = ~cur;
            while (rem) {
                const int t = __builtin_ctzll(rem);
                rem &= rem - 1ull;
                // who (earlier in the chunk) suppresses t: t is wave-uniform, so this is two v_readlane_b32 — a __shfl here was a ds_bpermute
                // round trip through the LDS crossbar per visited candidate, in the one serial section of the pass
                const unsigned long long mt = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), t) << 32) |
                                              (unsigned)__builtin_amdgcn_readlane((int)(mine & 0xffffffffull), t);
                if (!(mt & km)) km |= 1ull << t;
            }
            int cntk = __popcll(km);
            if (kept + cntk > max_det) {  // keep only the first (max_det - kept) survivors
                int allow = max_det - kept;
                unsigned long long m2 = 0ull, tmp = km;
                while (allow-- > 0) { const unsigned long long low = tmp & (~tmp + 1ull); m2 |= low; tmp ^= low; }
                km = m2;
            }
            if ((km >> lane) & 1ull) {
                const int pos = kept + __popcll(km & ((1ull << lane) - 1ull));
                kx1[pos] = x1; ky1[pos] = y1; kx2[pos] = x2; ky2[pos] = y2; kar[pos] = ai;
                float* o = out_dets + ((size_t)b * max_det + pos) * 6;
                o[0] = rbx.x; o[1] = rbx.y; o[2] = rbx.z; o[3] = rbx.w;
                o[4] = rsc; o[5] = (float)rcl;
                out_idx[(size_t)b * max_det + pos] = ran;
                w.keep_pos[(size_t)b * NMS_MAXDET_CAP + pos] = i;
            }
            if (lane == 0) kmask = km;
        }
        __syncthreads();
        kept += __popcll(kmask);
    }
    const int nk = kept < max_det ? kept : max_det;
    if (threadIdx.x == 0) out_counts[b] = nk;
    // rows past the count are zero: the caller hands over uninitialised buffers (no fill kernels in the step)
    for (int i = nk * 6 + threadIdx.x; i < max_det * 6; i += NT) out_dets[(size_t)b * max_det * 6 + i] = 0.f;
    for (int i = nk + threadIdx.x; i < max_det; i += NT) out_idx[(size_t)b * max_det + i] = 0;
}

extern "C" int ymk_nms_batched(const float* y, int32_t B, int32_t nc, int32_t extra, int32_t A, float conf_thres, float iou_thres,
                               int32_t multi_label, int32_t agnostic, int32_t max_det, int32_t max_nms, float max_wh,
                               const uint8_t* class_keep, const float* best_conf, const int32_t* best_cls, float* out_dets,
                               int32_t* out_counts, int32_t* out_idx, int32_t* status, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (!y || !out_dets || !out_counts || !out_idx || !status || !workspace) return YMK_E_BADARG;
    if (B <= 0 || A <= 0 || nc <= 0 || extra < 0 || max_det <= 0 || max_nms <= 0 || B > 65535 || max_det > NMS_MAXDET_CAP)
        return YMK_E_BADARG;
    const int multi = (multi_label && nc > 1) ? 1 : 0;
    NmsWs w = nms_layout(workspace, B, nc, A, multi, max_nms);
    if (workspace_bytes < w.total) return YMK_E_WORKSPACE;
    w.rows = 4 + nc + extra;   // utils/nms.py:76-81: candidates come from rows [4, 4 + nc); the `extra` rows behind them are carried, not scored
    hipStream_t s = (hipStream_t)stream;
    if (best_conf && best_cls && !multi) {   // the producer's per-anchor best class (ymk_detect_decode): no pass over the class rows
        w.bconf = const_cast<float*>(best_conf);
        w.bcls = const_cast<int*>(best_cls);
        hipLaunchKernelGGL(nms_count_best_kernel, dim3(w.nblk, B), dim3(256), 0, s, A, nc, conf_thres, class_keep, w);
    } else {
        hipLaunchKernelGGL(nms_count_kernel, dim3(w.nblk, B), dim3(256), 0, s, y, nc, A, conf_thres, multi, class_keep, w);
    }
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, s, w, status);
    const int64_t full = multi ? (int64_t)A * nc : (int64_t)A;
    if (full > (int64_t)max_nms) {   // an image CAN hold more candidates than max_nms: selection kernels (no-ops for images that do not)
        for (int level = 0; level < 3; ++level) {
            hipLaunchKernelGGL(nms_hist_kernel, dim3(w.nblk, B), dim3(256), 0, s, y, nc, A, conf_thres, multi, class_keep, w, level);
            hipLaunchKernelGGL(nms_pick_kernel, dim3(B), dim3(256), 0, s, w, level);
        }
        hipLaunchKernelGGL(nms_count2_kernel, dim3(w.nblk, B), dim3(256), 0, s, y, nc, A, conf_thres, multi, class_keep, w);
        hipLaunchKernelGGL(nms_scan2_kernel, dim3(B), dim3(64), 0, s, w);
    }
    hipLaunchKernelGGL(nms_emit_kernel, dim3(w.nblk, B), dim3(256), 0, s, y, nc, A, conf_thres, multi, class_keep, w);
    if (w.capc <= NMS_SORT_CAP && !(ymk_disabled() & YMK_OFF_NMS_SORT)) {
        // YMK_ENABLE bit 128: small images by counting ranks on many workgroups, large ones by the in-LDS sort.  Measured SLOWER on the
        // bench step (64 images, 2-8 k candidates each: nms 0.248 ms against 0.191 ms with one sorting workgroup per image — the
        // 64-bit compare loop of the rank kernel costs more than the sorting network's barriers): off by default.
        const int rank_max = (ymk_enabled() & YMK_ON_NMS_RANK_SMALL) ? NMS_RANK_MAX : 0;
        const int rcap = w.capc < rank_max ? w.capc : rank_max;
        if (rcap > 0) hipLaunchKernelGGL(nms_rank_kernel, dim3((rcap + 255) / 256, B), dim3(256), 0, s, w, rank_max);
        if (w.capc > rank_max)
            hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(1024), 0, s, w, (ymk_disabled() & YMK_OFF_NMS_RADIX) ? 0 : 1, rank_max);
    } else {
        hipLaunchKernelGGL(nms_rank_kernel, dim3((w.capc + 255) / 256, B), dim3(256), 0, s, w, -1);
    }
    // YMK_NMS_GREEDY_WAVES=4: the four-wave form of rounds 1-4 (A/B runs)
    static const int greedy_waves = [] { const char* e = getenv("YMK_NMS_GREEDY_WAVES"); return e ? atoi(e) : 16; }();
    if (greedy_waves == 4)
        hipLaunchKernelGGL(nms_greedy_kernel<4>, dim3(B), dim3(256), 0, s, w, iou_thres, agnostic ? 0.0f : max_wh, max_det, out_dets, out_counts, out_idx);
    else if (greedy_waves == 8)
        hipLaunchKernelGGL(nms_greedy_kernel<8>, dim3(B), dim3(512), 0, s, w, iou_thres, agnostic ? 0.0f : max_wh, max_det, out_dets, out_counts, out_idx);
    else
        hipLaunchKernelGGL(nms_greedy_kernel<16>, dim3(B), dim3(1024), 0, s, w, iou_thres, agnostic ? 0.0f : max_wh, max_det, out_dets, out_counts, out_idx);
    return ymk_launch_status();
}

// ---- rows carried behind the class rows (Segment: mask coefficients; utils/nms.py:76-81,117: `mask` columns of the detections) ----
// out[b][j][k] = y[b][row0 + k][idx[b][j]] for j < counts[b]; zero rows past the count (the caller hands over uninitialised memory).
__global__ __launch_bounds__(256) void nms_gather_rows_kernel(const float* __restrict__ y, int rows, int A, int row0, int extra,
                                                             const int* __restrict__ idx, const int* __restrict__ counts, int max_det,
                                                             float* __restrict__ out) {
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= max_det * extra) return;
    const int j = t / extra, k = t - j * extra;
    float v = 0.f;
    if (j < counts[b]) v = y[((size_t)b * rows + row0 + k) * A + idx[(size_t)b * max_det + j]];
    out[(size_t)b * max_det * extra + t] = v;
}

extern "C" int ymk_nms_gather_rows(const float* y, int32_t B, int32_t rows, int32_t A, int32_t row0, int32_t extra, const int32_t* idx,
                                   const int32_t* counts, int32_t max_det, float* out, void* stream) {
    if (!y || !idx || !counts || !out || B <= 0 || B > 65535 || A <= 0 || extra <= 0 || row0 < 0 || row0 + extra > rows || max_det <= 0)
        return YMK_E_BADARG;
    hipLaunchKernelGGL(nms_gather_rows_kernel, dim3((max_det * extra + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, y, rows, A, row0,
                       extra, idx, counts, max_det, out);
    return ymk_launch_status();
}

// ---- CW-NMS refinement (spec: examples/.../cpp/src/common.cpp:150-185) ----------------
// One wavefront per kept detection: fp64 accumulation over the score-sorted pool.
// agnostic: the cluster of a survivor is every pool member it overlaps, whatever its class (the suppression that produced
// the survivors ignored classes too); the C++ spec only defines the per-class case.
__global__ __launch_bounds__(64) void cw_refine_kernel(NmsWs w, int max_det, float thr, double sigma, int pool_cap, int agnostic,
                                                      float* __restrict__ dets, const int* __restrict__ counts) {
    const int b = blockIdx.y, k = blockIdx.x, lane = threadIdx.x;
    if (k >= counts[b]) return;
    const int n = min(w.nsort[b], pool_cap);
    const int row = w.keep_pos[(size_t)b * NMS_MAXDET_CAP + k];
    const size_t base = (size_t)b * w.ns;
    const int ck = w.scls[base + row];
    const double kx1 = w.sbox[(base + row) * 4 + 0], ky1 = w.sbox[(base + row) * 4 + 1];
    const double kx2 = w.sbox[(base + row) * 4 + 2], ky2 = w.sbox[(base + row) * 4 + 3];
    const double ak = (kx2 - kx1) * (ky2 - ky1);
    double sw = 0, ax1 = 0, ay1 = 0, ax2 = 0, ay2 = 0;
    for (int m = lane; m < n; m += 64) {
        if (!agnostic && w.scls[base + m] != ck) continue;
        const double x1 = w.sbox[(base + m) * 4 + 0], y1 = w.sbox[(base + m) * 4 + 1];
        const double x2 = w.sbox[(base + m) * 4 + 2], y2 = w.sbox[(base + m) * 4 + 3];
        const double iw = fmin(kx2, x2) - fmax(kx1, x1), ih = fmin(ky2, y2) - fmax(ky1, y1);
        if (iw <= 0 || ih <= 0) continue;
        const double inter = iw * ih;
        const double uni = ak + (x2 - x1) * (y2 - y1) - inter;
        const double ov = uni > 0 ? inter / uni : 0.0;
        if (ov <= (double)thr) continue;
        const double wt = (double)w.sscore[base + m] * exp(-((1.0 - ov) * (1.0 - ov)) / sigma);
        sw += wt; ax1 += wt * x1; ay1 += wt * y1; ax2 += wt * x2; ay2 += wt * y2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sw += __shfl_xor(sw, o); ax1 += __shfl_xor(ax1, o); ay1 += __shfl_xor(ay1, o);
        ax2 += __shfl_xor(ax2, o); ay2 += __shfl_xor(ay2, o);
    }
    if (lane == 0 && sw > 1e-6) {
        float* o = dets + ((size_t)b * max_det + k) * 6;
        const double x0 = ax1 / sw, y0 = ay1 / sw;
        const double ww = fmax(0.0, ax2 / sw - x0), hh = fmax(0.0, ay2 / sw - y0);
        o[0] = (float)x0; o[1] = (float)y0; o[2] = (float)(x0 + ww); o[3] = (float)(y0 + hh);
    }
}

extern "C" int ymk_cw_refine(int32_t B, int32_t nc, int32_t A, int32_t multi_label, int32_t agnostic, int32_t max_nms, int32_t max_det,
                                float iou_thres, float sigma, int32_t pool_cap, float* out_dets,
                                const int32_t* out_counts, void* workspace, size_t workspace_bytes, void* stream) {
    if (!out_dets || !out_counts || !workspace || B <= 0 || max_det <= 0 || max_det > NMS_MAXDET_CAP || !(sigma > 0.f))
        return YMK_E_BADARG;
    const int multi = (multi_label && nc > 1) ? 1 : 0;
    NmsWs w = nms_layout(workspace, B, nc, A, multi, max_nms);
    if (workspace_bytes < w.total) return YMK_E_WORKSPACE;
    hipLaunchKernelGGL(cw_refine_kernel, dim3(max_det, B), dim3(64), 0, (hipStream_t)stream, w, max_det, iou_thres,
                       (double)sigma, pool_cap, agnostic ? 1 : 0, out_dets, out_counts);
    return ymk_launch_status();
}
