// Shared device/host helpers for libymk (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/ymk.h"

#define YMK_WAVE 64

// The library's 16-bit element type.  Every kernel that handles 16-bit activations / weights is written against `h16_t` (raw bits)
// and the helpers below; the FORMAT is a property of the build: libymk.so = bfloat16, libymk_f16.so (the same sources compiled with
// -DYMK_H16_F16) = IEEE binary16, the reference's reduced-precision mode (`half=True`, engine/predictor.py:174,415).  Accumulation is
// fp32 in both; routers, statistics, DFL decode and NMS are fp32 in both.  `h16_t` remains as an alias in the bf16-era comments only.
typedef uint16_t h16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));

#ifdef YMK_H16_F16
#define YMK_H16_NAME "f16"
typedef _Float16 hw_h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hw_h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float h16_to_f32(h16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float h16lo(uint32_t w) { return (float)__builtin_bit_cast(hw_h16x2, w).x; }
__device__ __forceinline__ float h16hi(uint32_t w) { return (float)__builtin_bit_cast(hw_h16x2, w).y; }
// two packed words (e0,e1),(e2,e3) -> a = (e0, e2), b = (e1, e3) as fp32 pairs
__device__ __forceinline__ void h16x4_widen(const u32x2 t, f32x2& a, f32x2& b) {
    const uint32_t tx = t.x, ty = t.y;   // scalars first: __builtin_bit_cast of an ext-vector ELEMENT reads element 0 (clang 22)
    const hw_h16x2 p = __builtin_bit_cast(hw_h16x2, tx), q = __builtin_bit_cast(hw_h16x2, ty);
    a = f32x2{(float)p.x, (float)q.x};
    b = f32x2{(float)p.y, (float)q.y};
}
// fp32 -> binary16, round to nearest even (v_cvt_f16_f32 under the default rounding mode: torch's float -> half cast)
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
    const hw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_h16x2));
}
__device__ __forceinline__ f32x4 mfma16x16x32_h16(const u32x4& a, const u32x4& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_h16x8, a), __builtin_bit_cast(hw_h16x8, b), acc, 0, 0, 0);
}
#else
#define YMK_H16_NAME "bf16"
typedef __bf16 hw_h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hw_h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float h16_to_f32(h16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float h16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// two packed words (e0,e1),(e2,e3) -> a = (e0, e2), b = (e1, e3): VECTOR shift / mask, the results land in register pairs directly
__device__ __forceinline__ void h16x4_widen(const u32x2 t, f32x2& a, f32x2& b) {
    a = __builtin_bit_cast(f32x2, t << 16);
    b = __builtin_bit_cast(f32x2, t & 0xffff0000u);
}
// fp32 -> bf16: gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN quieted: the same rule as torch's
// float->bfloat16 cast), one instruction per two values.
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
    const hw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_h16x2));
}
__device__ __forceinline__ f32x4 mfma16x16x32_h16(const u32x4& a, const u32x4& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hw_h16x8, a), __builtin_bit_cast(hw_h16x8, b), acc, 0, 0, 0);
}
#endif
__device__ __forceinline__ h16_t f32_to_h16(float f) { return (h16_t)(pack_h16x2(f, 0.f) & 0xffffu); }
// c + a.lo * b.lo + a.hi * b.hi on two packed 16-bit pairs, fp32 accumulate: v_dot2_f32_bf16 / v_dot2_f32_f16 (one VALU instruction for
// two multiply-adds straight from the 16-bit words: no widening).  The hardware's internal rounding of the two-term sum is its own (not
// two separately rounded fmaf's); the host emulator restates it as two fmaf's, so kernels built on it are held to a tolerance there.
__device__ __forceinline__ float dot2_h16(uint32_t a, uint32_t b, float c) {
#ifdef YMK_HOST_EMU
    return __builtin_fmaf(h16hi(a), h16hi(b), __builtin_fmaf(h16lo(a), h16lo(b), c));
#elif defined(YMK_H16_F16)
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(hw_h16x2, a), __builtin_bit_cast(hw_h16x2, b), c, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_h16x2, a), __builtin_bit_cast(hw_h16x2, b), c, false);
#endif
}

// SiLU exactly as x * sigmoid(x) with sigmoid = 1/(1+exp(-x)) (torch CPU formula)
// fast form for bf16 outputs: v_exp_f32 + v_rcp_f32 (~1 ulp), 5 instructions instead of an IEEE division
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }   // torch.nn.GELU default (erf form)
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// a * b and a + b as separately rounded operations the compiler may not contract into an FMA (hipcc's default is -ffp-contract=fast)
__device__ __forceinline__ float ymk_mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float ymk_add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
    static constexpr int VEC = 4;  // elements per 16 bytes
    static constexpr int DT = YMK_F32;
};
template <>
struct ElemTraits<h16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = YMK_BF16;
};

// load VEC elements (16 B) as fp32 values
__device__ __forceinline__ void load_vec_f32(const float* p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load_vec_f32(const h16_t* p, float (&v)[8]) {
    u32x4 t = *reinterpret_cast<const u32x4*>(p);
    v[0] = h16lo(t.x); v[1] = h16hi(t.x); v[2] = h16lo(t.y); v[3] = h16hi(t.y);
    v[4] = h16lo(t.z); v[5] = h16hi(t.z); v[6] = h16lo(t.w); v[7] = h16hi(t.w);
}
__device__ __forceinline__ void store_vec_f32(float* p, const float (&v)[4]) {
    f32x4 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
    *reinterpret_cast<f32x4*>(p) = t;
}
__device__ __forceinline__ void store_vec_f32(h16_t* p, const float (&v)[8]) {
    u32x4 t;
    t.x = pack_h16x2(v[0], v[1]); t.y = pack_h16x2(v[2], v[3]);
    t.z = pack_h16x2(v[4], v[5]); t.w = pack_h16x2(v[6], v[7]);
    *reinterpret_cast<u32x4*>(p) = t;
}
// store 4 consecutive channels
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    f32x4 t; t.x = a; t.y = b; t.z = c; t.w = d;
    *reinterpret_cast<f32x4*>(p) = t;
}
__device__ __forceinline__ void store4(h16_t* p, float a, float b, float c, float d) {
    u32x2 t; t.x = pack_h16x2(a, b); t.y = pack_h16x2(c, d);
    *reinterpret_cast<u32x2*>(p) = t;
}
__device__ __forceinline__ void load4(const float* p, float& a, float& b, float& c, float& d) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    a = t.x; b = t.y; c = t.z; d = t.w;
}
__device__ __forceinline__ void load4(const h16_t* p, float& a, float& b, float& c, float& d) {
    u32x2 t = *reinterpret_cast<const u32x2*>(p);
    a = h16lo(t.x); b = h16hi(t.x); c = h16lo(t.y); d = h16hi(t.y);
}
// 4 consecutive channels as raw bits (register prefetch of residual operands) and their later widening
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef f32x4 type; };
template <> struct Raw4<h16_t> { typedef u32x2 type; };
__device__ __forceinline__ f32x4 load_raw4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x2 load_raw4(const h16_t* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void unpack_raw4(const f32x4& t, float& a, float& b, float& c, float& d) {
    a = t.x; b = t.y; c = t.z; d = t.w;
}
__device__ __forceinline__ void unpack_raw4(const u32x2& t, float& a, float& b, float& c, float& d) {
    a = h16lo(t.x); b = h16hi(t.x); c = h16lo(t.y); d = h16hi(t.y);
}
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(h16_t v) { return h16_to_f32(v); }
__device__ __forceinline__ void from_f32(float& d, float v) { d = v; }
__device__ __forceinline__ void from_f32(h16_t& d, float v) { d = f32_to_h16(v); }

// One-shot per (call site, DEVICE), safe for concurrent host threads: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the
// current device only, and a library that promises per-thread predictors may be entered by two threads at once.  Two
// threads racing through need() both set the (idempotent) attribute; nobody launches before it is set on his device.
struct YmkOncePerDevice {
    std::atomic<unsigned long long> mask{0ull};
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool need() const { return !(mask.load(std::memory_order_acquire) & bit()); }
    void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

static inline int ymk_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? YMK_OK : YMK_E_LAUNCH;
}
// Tuning switch for A/B runs (not part of the ABI): YMK_DISABLE=<bitmask> turns specialised kernels off so the
// same process image can be timed with and without them.  Every path computes the same result.
#define YMK_OFF_CONV_STREAM 1u   // streaming 1x1 + spatial-tile 3x3 convolutions -> tiled implicit GEMM
#define YMK_OFF_MOE_STREAM 2u    // streaming ES-MoE pointwise stage -> tiled grouped GEMM
#define YMK_OFF_NMS_SORT 4u      // LDS bitonic candidate sort -> rank-by-counting
#define YMK_OFF_STEM_FAST 8u     // fp32-MFMA stem -> one pixel per thread on the VALU
#define YMK_OFF_RES_PREFETCH 16u  // register prefetch of residual operands in the spatial-tile 3x3 kernel
#define YMK_OFF_STEM_ROWS 32u     // LDS-staged stem rows -> direct gathers from the image
#define YMK_OFF_MOE_LEAN 32768u    // table-driven ES-MoE pointwise stage -> moe_pw_stream_kernel
#define YMK_OFF_CONV_GLDS1 131072u  // tiled bf16 1x1 convolutions (K >= 256, Cout % 128 == 0) on the LDS-DMA core -> conv_igemm_kernel
#define YMK_OFF_CONV_GLDS3 128u   // bf16 3x3 convolutions with Cin >= 64 on the LDS-DMA tiled core -> conv_igemm_kernel
static inline unsigned ymk_disabled() {
    static const unsigned m = [] {
        const char* s = getenv("YMK_DISABLE");
        return s ? (unsigned)strtoul(s, nullptr, 0) : 0u;
    }();
    return m;
}
// Opt-in switch for code that is not validated / measured on hardware yet (not part of the ABI): YMK_ENABLE=<bitmask>.
#define YMK_ON_CONV_GLDS 1u        // tiled convolutions -> conv_glds_kernel (csrc/conv_glds.hip)
#define YMK_ON_GLDS_TWO_STAGE 2u   // ... with the 2-stage plain-barrier k-loop instead of the 3-stage counted-vmcnt one
static inline unsigned ymk_enabled() {
    static const unsigned m = [] {
        const char* s = getenv("YMK_ENABLE");
        return s ? (unsigned)strtoul(s, nullptr, 0) : 0u;
    }();
    return m;
}
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
