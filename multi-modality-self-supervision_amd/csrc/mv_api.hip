// Library-level entry points (version / implementation selector).  The two selectors (g_mv_impl here, the GEMM tile override in
// mv_gemm.hip) are process-global TEST HOOKS -- the only mutable state of the library; nothing on the product path sets them.
#include "mv_common.h"

int g_mv_impl = 0;

extern "C" int mv_abi_version(void) { return MV_ABI_VERSION; }
extern "C" void mv_set_impl(int impl) { g_mv_impl = impl ? 1 : 0; }
extern "C" int mv_get_impl(void) { return g_mv_impl; }
extern "C" const char* mv_build_info(void) { return "medvill-hip gfx950 (" __DATE__ " " __TIME__ ")"; }
