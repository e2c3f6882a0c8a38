// Library-level entry points (version / implementation selector).  The two selectors (g_mv_impl here, the GEMM tile override in
// mv_gemm.hip) are process-global TEST HOOKS -- the only mutable state of the library; nothing on the product path sets them.
#include "mv_common.h"

int g_mv_impl = 0;

extern "C" int mv_abi_version(void) { return MV_ABI_VERSION; }
extern "C" void mv_set_impl(int impl) { g_mv_impl = impl ? 1 : 0; }
extern "C" int mv_get_impl(void) { return g_mv_impl; }
extern "C" const char* mv_build_info(void) { return "medvill-hip gfx950 (" __DATE__ " " __TIME__ ")"; }

// ---- compute-unit partitioning between the streams of a step ---------------------------------------------------------------------
// The backward keeps two queues busy: the main chain (LayerNorm backward, dX GEMMs, attention backward: partly HBM- / VALU-bound) and
// the weight-gradient GEMMs (persistent, one 256x256 block per CU with the whole register file: a CU that runs one cannot take a block of
// anything else until the persistent kernel ends).  Two knobs let the host PARTITION the chip instead of letting the queues time-slice it:
//   mv_set_persistent_cus(n)     the persistent GEMM kernels launch at most n blocks (0 = one per CU of the device)
//   mv_stream_create_cumask(...) a HIP stream whose kernels only run on the CUs of a mask (hipExtStreamCreateWithCUMask).  On a
//                                multi-XCD device bit i of the mask is CU (i / n_xcd) of XCD (i % n_xcd): the first 8 k bits are k CUs of
//                                EVERY XCD of an MI355X, so a masked stream still spreads over all eight L2s.
int g_mv_persistent_cus = 0;
extern "C" void mv_set_persistent_cus(int n) { g_mv_persistent_cus = n > 0 ? n : 0; }
extern "C" int mv_get_persistent_cus(void) { return g_mv_persistent_cus; }
extern "C" int mv_stream_create_cumask(const uint32_t* mask_words, int n_words, void** stream_out) {
  if (!mask_words || n_words <= 0 || !stream_out) return MV_E_ARG;
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask_words);
  if (e != hipSuccess) return (int)e;
  *stream_out = (void*)st;
  return MV_OK;
}
extern "C" int mv_stream_destroy(void* stream) {
  if (!stream) return MV_E_ARG;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? MV_OK : (int)e;
}
