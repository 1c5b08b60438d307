// ABI/version entry points of libymk.
#include "ymk_common.h"

extern "C" int ymk_abi_version(void) { return YMK_ABI_VERSION; }

extern "C" const char* ymk_build_info(void) {
    return "libymk gfx950 (CDNA4) hipcc " __VERSION__ " abi " "1";
}
