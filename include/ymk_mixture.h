/* libymk — C-ABI of the config-5 rows (MoA / MoT / gated MoE), SURVEY.md §8 rows a11 / a12 and §8(f) rank 1.
 *
 * STATUS: validated on MI355X since round 2 (tests/test_gpu_mixture.py: every entry point in fp32 / bf16 / fp16 against the
 * reference's module fixtures and the whole config-5 detector at N and at L scale, 2 x 1280 x 1280) and on the CPU lane emulator
 * (tests/test_hostemu_mixture.py).  Element-wise and normalisation forms: fp32 arithmetic on 16-byte vector memory paths, chunked
 * exact statistics; the softmax attention forms (ymk_attention, ymk_window_attention) run on the matrix cores for 16-bit inputs
 * (csrc/mixattn.hip); random-feature and deformable attention are VALU kernels.
 *
 * Conventions as in ymk.h: NHWC views with pixel strides (elements), activations YMK_F32 / YMK_BF16, statistics /
 * gates / router tensors fp32, `stream` a hipStream_t, return 0 or a negative YMK_E_* code, no allocation, no sync.
 * Each function cites the reference lines it replaces.
 */
#ifndef YMK_MIXTURE_H_
#define YMK_MIXTURE_H_
#include "ymk.h"
#ifdef __cplusplus
extern "C" {
#endif

/* YMK_ACT_SIGMOID (2) and YMK_ACT_GELU (3, exact erf form) are defined in ymk.h */

/* In-place activation on a channel-dense view (used after ymk_conv2d for the sigmoid / GELU epilogues of the gates and
 * the token FFNs: moe/gated.py:1171-1218, mot/experts.py:318-325). */
int ymk_activation(int32_t dtype, void* x, int32_t ldx, int64_t npix, int32_t C, int32_t act, void* stream);

/* torch.nn.GroupNorm over (HW, C/groups) per image and group, biased variance, fp32 statistics.
 * weight/bias: fp32 [C], or NULL (no affine), or [R][C] with affine_rows int32 [B] (row per image; FusedExpertGroup,
 * moe/gated.py:1058-1090).  act: YMK_ACT_NONE | YMK_ACT_SILU.  residual (dtype = out_dtype of y, may be NULL) is added
 * after the activation (MoTBlock out_norm(.) + x, mot/block.py:413-417).  y may alias x.
 * stats_ws: fp32 [B*groups*(2 + 3*256)] scratch: (mean, rstd) per slab, then up to 256 chunk partials (mean, M2, n) per slab —
 * a slab's statistics are computed by several workgroups and combined exactly, in chunk order (deterministic). */
int ymk_group_norm(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy,
                   const void* residual, int32_t ldr, int32_t B, int32_t HW, int32_t C, int32_t groups,
                   const float* weight, const float* bias, const int32_t* affine_rows, float eps, int32_t act,
                   float* stats_ws, void* stream);

/* torch.nn.LayerNorm(C) per token (mot/experts.py:285, 318, 470, 486). */
int ymk_layer_norm(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t npix, int32_t C,
                   const float* weight, const float* bias, float eps, void* stream);

/* y = a * b (op 0), sigmoid(a) * b (op 1: GLU of the MoT local expert, mot/experts.py:160-166),
 * (1 - alpha) * a + alpha * b (op 2: exact / linear attention blend, moa/heads.py:366-374). */
#define YMK_ELT_MUL 0
#define YMK_ELT_SIGMOID_MUL 1
#define YMK_ELT_LERP 2
#define YMK_ELT_CLAMP_ADD 3   /* y = clamp(a, -alpha, alpha) + b: UltraOptimizedMoE's `shared + expert_output.clamp_(-1e4, 1e4)` (moe/utils.py:203, moe/modules.py:224) */
int ymk_eltwise(int32_t op, int32_t dtype, const void* a, int32_t lda, const void* b, int32_t ldb, void* y, int32_t ldy,
                int64_t npix, int32_t C, float alpha, void* stream);

/* y = x + scale * a * b; b is a map (b_per_image = 0, dtype b_dtype, stride ldb) or an fp32 per-image channel gate
 * [B][C] (b_per_image = 1).  Detail gate, context mixer, refinement: moe/gated.py:1171-1218, hooks.py:60-68. */
int ymk_fma_gate(int32_t dtype, const void* x, int32_t ldx, const void* a, int32_t lda, const void* b, int32_t b_dtype,
                 int32_t ldb, int32_t b_per_image, float scale, void* y, int32_t ldy, int32_t B, int32_t HW, int32_t C,
                 void* stream);

/* w[b][j] *= c, c = clamp(mean_b sigmoid(logit[b]), lo, hi) with a non-finite mean replaced by 1 (UltimateOptimizedMoE's batch-level
 * complexity scale on the routing weights, moe/modules.py:1662-1672).  w fp32 [B][K] dense, logit fp32 [B] with stride ldl. */
int ymk_batch_scale(float* w, int32_t B, int32_t K, const float* logit, int32_t ldl, float lo, float hi, void* stream);

/* y = x * gate[b][c], gate fp32 [B][C] (squeeze-excite gate, moe/gated.py:333-341). */
int ymk_channel_gate(int32_t dtype, const void* x, int32_t ldx, const float* gate, void* y, int32_t ldy, int32_t B,
                     int32_t HW, int32_t C, void* stream);

/* y = sum_e w[..][e] * part_e, E <= 4; w fp32 per token (stride ldw) or per image (w_per_image = 1, [B][ldw]).
 * MoA head mix (moa/block.py:230-262), MoT expert blend (mot/block.py:360-417), gated expert mix. */
int ymk_weighted_sum(int32_t dtype, const float* w, int32_t ldw, int32_t w_per_image, int32_t E, const void* p0,
                     const void* p1, const void* p2, const void* p3, int32_t ldp, void* y, int32_t ldy, int32_t B,
                     int32_t HW, int32_t C, void* stream);

/* y = mean_i nearest_resize(part_i -> H x W), n <= 4 parts of sizes (h_i, w_i), F.interpolate(mode="nearest") index
 * rule src = floor(dst * h / H).  PyramidContextMixer.forward, moe/gated.py:1209-1216. */
int ymk_mean_upsampled(int32_t dtype, int32_t n, const void* p0, const void* p1, const void* p2, const void* p3,
                       const int32_t* hs, const int32_t* ws, const int32_t* lds /* host arrays [n] */, void* y,
                       int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* F.adaptive_avg_pool2d bins [floor(i*H/Ho), ceil((i+1)*H/Ho)) (moa/heads.py:225-233, moe/gated.py:1211-1213) and
 * F.avg_pool2d(k, stride k) (moe/gated.py:141-143). */
int ymk_adaptive_avg_pool(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy, int32_t B,
                          int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void* stream);
int ymk_avg_pool(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t out_dtype, int32_t ldy, int32_t B, int32_t H,
                 int32_t W, int32_t C, int32_t k, void* stream);

/* Per image and channel mean (and biased std) over HW: out fp32 [B][C] or [B][2C] = [mean | std]
 * (moe/gated.py:133-139, AdaptiveAvgPool2d(1) of the gates).  ws: fp32 [B * min(64, ceil(HW / 1024)) * 2 * C] scratch (an image's map
 * is reduced by several workgroups whose partials are combined exactly, in chunk order) or NULL (one workgroup per image). */
int ymk_channel_stats(int32_t dtype, const void* x, int32_t ldx, float* out, int32_t B, int32_t HW, int32_t C,
                      int32_t want_std, float* ws, void* stream);

/* Per-token softmax over n <= 8 fp32 logits scaled by inv_temp; 0 < top_k < n keeps the top_k largest (ties: lower index
 * first), renormalised with the sum clamped at 1e-6 (mot/router.py:243-295, moa/router.py:50-62).
 * w fp32 [npix][ldw]; active int32 [B][n] must be zero on entry (atomicOr 1 where a token selects expert e). */
/* UltraEfficientRouter's decision tail (moe/routers.py:117-147, eval) + the inference threshold of BatchedExpertComputation (moe/utils.py:166-169):
 * logits fp32 [B][HW][ldl >= E] of the (pooled) router map -> per pixel softmax(clamp(l, +-30) * inv_temp), mean over the pixels (pooled fp32
 * [B][E]), top-k (lower index first among equals), w = value / max(sum, 1e-6) with w <= threshold replaced by 0; idx int32 [B][top_k], rows int32
 * [top_k * B] = idx transposed (the expert of image j * B + b in a slot-major expert batch).  E <= 32, top_k <= 4. */
int ymk_pooled_softmax_route(const float* logits, int32_t ldl, int32_t B, int32_t HW, int32_t E, float inv_temp, int32_t top_k, float threshold,
                             float* w, int32_t* idx, int32_t* rows, float* pooled, void* stream);

int ymk_token_softmax(const float* logits, int32_t ldl, const float* bias /*[B][n] or NULL*/, float* w, int32_t ldw, int32_t* active,
                      int32_t B, int32_t HW, int32_t n, float inv_temp, int32_t top_k, void* stream);
/* bias: added to every token's logits of its image before the temperature — the scene-aware residual of the MoT router
 * (mot/router.py:224-240) and, with logits == NULL, the whole logit of the image-level router (use_spatial=False, :118-125).
 *
 * Scene statistics + projector of the MoT router (mot/router.py:166-192 compute_scene_stats: high-frequency, heterogeneity and
 * multi-scale statistics of the routed map in fp32; :145-160 scene_projector = Linear(3, hidden) -> SiLU -> Linear(hidden, E)).
 * x: the routed map [B][H][W][ldx]; chan_stats fp32 [B][2C] = ymk_channel_stats(want_std = 1); pool4 / pool2 fp32
 * [B][min(4,H) * min(4,W)][C] / [B][min(2,H) * min(2,W)][C] = ymk_adaptive_avg_pool; w1 [hidden][3], b1 [hidden], w2 [E][hidden], b2 [E];
 * base: NULL or fp32 [B][E] added to the result (the image-level router's own logits).  Outputs stats fp32 [B][3], bias fp32 [B][E]. */
size_t ymk_scene_workspace_bytes(int32_t B, int32_t H);
int ymk_scene_bias(int32_t dtype, const void* x, int32_t ldx, int32_t B, int32_t H, int32_t W, int32_t C, const float* chan_stats,
                   const float* pool4, const float* pool2, const float* w1, const float* b1, const float* w2, const float* b2,
                   int32_t hidden, int32_t E, const float* base, float* stats, float* bias, void* workspace, size_t workspace_bytes,
                   void* stream);

/* MoA sparse inference (moa/block.py:194-234, eval with `sparse_inference=True`): a head group whose gate is at or below `threshold`
 * for every token of the batch is skipped (none above it: the group with the largest mean gate runs alone); the retained gates are
 * renormalised per token.  w fp32 [npix][ldw], n <= 8 groups; stats: ZEROED scratch of 12 * n bytes, 8-byte aligned; active int32 [n];
 * blend fp32 [npix][ldb]: the retained groups' gates in columns 0 .. #active - 1 (the order of the groups), zeros behind. */
int ymk_moa_sparse_gate(const float* w, int32_t ldw, int64_t npix, int32_t n, float threshold, void* stats, float* blend, int32_t ldb,
                        int32_t* active, void* stream);

/* Decision tail of the gated MoE (moe/gated.py:124-166, 455-492), one workgroup.  g / loc fp32 [B][ld*] logits of the
 * two router streams, cplx fp32 [B][ldc] complexity logit; outputs w fp32 [B][top_k], idx int32 [B][top_k] (and its
 * transpose), probs fp32 [B][E].  E <= 64, top_k <= 8.
 * clamp_mode: 1 = softmax(clamp(l, +-30) * inv_temp) (gated.py:141-142); 2 = softmax(clamp(l * inv_temp, +-30)) (gated.py:972);
 * 0 = softmax(l * inv_temp), a router's own unclamped softmax (gated.py:958, routers.py:207). */
int ymk_gated_route_decide(const float* g, int32_t ldg, const float* loc, int32_t ldloc, const float* cplx, int32_t ldc,
                           int32_t B, int32_t E, float alpha, float inv_temp, int32_t clamp_mode, int32_t top_k, float* w, int32_t* idx,
                           int32_t* idx_slot_major /* [top_k][B]: the expert of image j*B + b of ymk_expert_gather's output */,
                           float* probs, void* stream);

/* out[(j*B + b)][p][:] = f_all[b][p][idx[b][j]*OC : +OC] — the routed experts' slices of the all-expert convolution
 * (FusedExpertGroup moe/gated.py:1058-1076, SharedInvertedExpertGroup moe/experts.py:235-269), slot-major. */
int ymk_expert_gather(int32_t dtype, const void* f_all, int32_t ldf, const int32_t* idx, int32_t B, int32_t HW,
                      int32_t OC, int32_t K, int32_t E, void* out, void* stream);

/* Per-image expert depthwise 3x3 with a per-expert dilation (DiversifiedExpertGroup.dw_layers, moe/gated.py:2265-2278): out[(j*B + b)]
 * = dw3x3(x[b]; w[idx[b][j]], dilation dil[idx[b][j]], zero padding = dilation), slot-major like ymk_expert_gather; w [E][9][C] in
 * the compute dtype, dil int32 [E], idx int32 [B][K]; no bias, no activation (a GroupNorm follows). */
int ymk_expert_dw3(int32_t dtype, const void* x, int32_t ldx, const void* w, const int32_t* dil, const int32_t* idx, int32_t B,
                   int32_t H, int32_t W, int32_t C, int32_t K, int32_t E, void* out, void* stream);

/* out[..][j*groups + i] = cat(a, b)[..][i*(C/groups) + j], C = Ca + Cb (_channel_shuffle, moe/gated.py:1333-1338). */
int ymk_channel_shuffle_cat(int32_t dtype, const void* a, int32_t lda, int32_t Ca, const void* b, int32_t ldb, int32_t Cb,
                            int32_t groups, void* y, int32_t ldy, int64_t npix, void* stream);

/* softmax(q k^T * scale) v per (image, head); q [B][Nq][heads*hd], k / v [B][Nk][heads*hd] (strides ldq/ldk/ldv),
 * hd <= 64.  moa/heads.py:208-253, 354-365; mot/experts.py:150-156. */
int ymk_attention(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                  void* out, int32_t ldo, int32_t B, int32_t Nq, int32_t Nk, int32_t heads, int32_t hd, float scale,
                  void* stream);

/* Attention inside win x win windows (win <= 16) of the map padded bottom / right to a multiple of win; out-of-image
 * tokens carry pad_q / pad_k / pad_v (fp32 [heads*hd], NULL = zeros) and take part as keys; shift > 0 rolls the padded
 * grid by -shift before the partition (no mask).  moa/heads.py:83-117, mot/experts.py:237-325. */
int ymk_window_attention(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v,
                         int32_t ldv, void* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t heads, int32_t hd,
                         float scale, int32_t win, int32_t shift, const float* pad_q, const float* pad_k,
                         const float* pad_v, void* stream);

/* ReLU random-feature attention (moa/heads.py:318-352), fp32: phi(t) = min(relu(t rf^T / sqrt(nb)) + 1e-6, 1e4);
 * out = clamp(phi(q) (phi(k)^T v), +-1e4) / max(phi(q) . sum_n phi(k_n), 1e-6).  rf fp32 [nb][hd], nb, hd <= 64.
 * ws: fp32 [B*heads*(ceil(N/512) + 1)*(nb*hd + nb)] scratch (per (image, head): the reduced phi(k)^T v | sum phi(k), then one partial
 * per 512-token chunk, added in chunk order). */
int ymk_linear_attention(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v,
                         int32_t ldv, const float* rf, int32_t nb, void* out, int32_t ldo, int32_t B, int32_t N,
                         int32_t heads, int32_t hd, float* ws, void* stream);

/* Deformable sampling attention (mot/experts.py:381-459): per token and head, n_points locations
 * clamp(ref + 0.25 * tanh(off), -1, 1) around the token's normalised position, softmax over the points of aw, bilinear
 * samples (zeros padding) of v's head slice, weighted sum.  off fp32 [npix][heads*n_points*2] (x, y), aw fp32
 * [npix][heads*n_points]; hd <= 64, n_points <= 8. */
int ymk_deform_attention(int32_t dtype, const void* v, int32_t ldv, const float* off, int32_t ldoff, const float* aw,
                         int32_t ldaw, void* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t heads, int32_t hd,
                         int32_t n_points, int32_t align_corners, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YMK_MIXTURE_H_ */
