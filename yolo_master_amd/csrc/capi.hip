// ABI/version entry points of libymk.
#include "ymk_common.h"

extern "C" int ymk_abi_version(void) { return YMK_ABI_VERSION; }

extern "C" const char* ymk_build_info(void) {
#define YMK_STR2(x) #x
#define YMK_STR(x) YMK_STR2(x)
    return "libymk gfx950 (CDNA4) hipcc " __VERSION__ " abi " YMK_STR(YMK_ABI_VERSION);
}
