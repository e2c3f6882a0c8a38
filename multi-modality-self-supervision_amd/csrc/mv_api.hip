// Library-level entry points.  This is the ONE translation unit that differs between the product library (libmedvill_hip.so: the
// kernel-selection knobs are compile-time constants, no mutable state, no setter exported) and the debug library
// (libmedvill_hip_dbg.so, built with -DMV_DEBUG_KNOBS: include/medvill_debug.h) that tests and experiments use to force a kernel.
#include "mv_common.h"
#ifdef MV_DEBUG_KNOBS
#include "../../include/medvill_debug.h"
#endif

#define MV_KNOB_DEFAULTS {0, 0, 0, 0, 16, 0, 0, 0, 0, 1}
#ifdef MV_DEBUG_KNOBS
static int g_knobs[MV_KNOB_COUNT] = MV_KNOB_DEFAULTS;
int mv_knob(int id) { return g_knobs[id]; }
extern "C" int mv_debug_set_knob(int id, int value) {
  if (id < 0 || id >= MV_KNOB_COUNT) return MV_E_ARG;
  if (id == MV_KNOB_ATTN_PLANES && value != 8 && value != 12 && value != 16) return MV_E_ARG;
  g_knobs[id] = value;
  return MV_OK;
}
extern "C" int mv_debug_get_knob(int id) { return (id < 0 || id >= MV_KNOB_COUNT) ? MV_E_ARG : g_knobs[id]; }
#else
int mv_knob(int id) {
  static const int k[MV_KNOB_COUNT] = MV_KNOB_DEFAULTS;
  return k[id];
}
#endif

extern "C" int mv_abi_version(void) { return MV_ABI_VERSION; }
#ifdef MV_DEBUG_KNOBS
extern "C" const char* mv_build_info(void) { return "medvill-hip gfx950 debug-knobs (" __DATE__ " " __TIME__ ")"; }
#else
extern "C" const char* mv_build_info(void) { return "medvill-hip gfx950 (" __DATE__ " " __TIME__ ")"; }
#endif

// ---- compute-unit partitioning between the streams of a step ---------------------------------------------------------------------
// The backward keeps two queues busy: the main chain (LayerNorm backward, dX GEMMs, attention backward: partly HBM- / VALU-bound) and
// the weight-gradient GEMMs (persistent, one 256x256 block per CU with the whole register file: a CU that runs one cannot take a block of
// anything else until the persistent kernel ends).  Two knobs let the host PARTITION the chip instead of letting the queues time-slice it:
//   MV_KNOB_PERSISTENT_CUS       (debug library) the persistent GEMM kernels launch at most n blocks (0 = one per CU of the device)
//   mv_stream_create_cumask(...) a HIP stream whose kernels only run on the CUs of a mask (hipExtStreamCreateWithCUMask).  On a
//                                multi-XCD device bit i of the mask is CU (i / n_xcd) of XCD (i % n_xcd): the first 8 k bits are k CUs of
//                                EVERY XCD of an MI355X, so a masked stream still spreads over all eight L2s.
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
