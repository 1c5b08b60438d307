// ABI/version entry points of libymk.
#include "ymk_common.h"

extern "C" int ymk_abi_version(void) { return YMK_ABI_VERSION; }

extern "C" const char* ymk_build_info(void) {
#define YMK_STR2(x) #x
#define YMK_STR(x) YMK_STR2(x)
    return "libymk gfx950 (CDNA4) hipcc " __VERSION__ " abi " YMK_STR(YMK_ABI_VERSION) " h16=" YMK_H16_NAME;
}

extern "C" int ymk_h16_format(void) {
#ifdef YMK_H16_F16
    return YMK_H16_FORMAT_F16;
#else
    return YMK_H16_FORMAT_BF16;
#endif
}
