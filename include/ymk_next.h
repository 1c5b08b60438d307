/* libymk — opt-in entry points that are NOT part of the validated surface of ymk.h yet.
 *
 * ymk_conv2d_glds: the next tiled implicit-GEMM core (csrc/conv_glds.hip): 256-pixel x 64/128-cout
 * tiles on 8 waves, both operands staged into LDS by global_load_lds with a source-side swizzle, 2- or 3-stage k-loop,
 * XCD-aware tile order.  Same arguments and result as ymk_conv2d (ymk.h) plus `two_stage` (0 = three LDS stages with a
 * counted vmcnt, 1 = two stages with a plain barrier); bf16 only, Cin % 64 == 0, Cout % 64 == 0, otherwise YMK_E_BADARG.
 * ymk_conv2d itself dispatches to it only when the environment variable YMK_ENABLE has bit 0 set (bit 1 = two_stage):
 * its logic is verified on the CPU lane emulator (tests/test_hostemu_conv.py), its speed has not been measured.
 */
#ifndef YMK_NEXT_H_
#define YMK_NEXT_H_
#include "ymk.h"
#ifdef __cplusplus
extern "C" {
#endif
int ymk_conv2d_glds(const ymk_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y,
                    int32_t two_stage, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* YMK_NEXT_H_ */
