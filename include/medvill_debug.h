/*
 * medvill_debug.h -- the two extra entry points of libmedvill_hip_dbg.so, the TEST / EXPERIMENT build of the library.
 *
 * libmedvill_hip.so (include/medvill.h) has no mutable state: every kernel choice follows from a call's arguments.  Parity tests that
 * cross-check one kernel against another (MFMA against the plain VALU kernels, one GEMM tile shape against another) and the timing
 * experiments under profiles/tools need to FORCE a kernel; they load this second library, which is built from the same objects
 * except for one translation unit (csrc/mv_api.hip with -DMV_DEBUG_KNOBS) holding a process-global knob table.  It exports everything
 * medvill.h declares plus the functions below.  Nothing on the product path loads it (medvill_amd._lib switches to it only while
 * a knob differs from its default).
 */
#ifndef MEDVILL_DEBUG_H_
#define MEDVILL_DEBUG_H_

#ifdef __cplusplus
extern "C" {
#endif

/* knob ids (csrc/mv_common.h): defaults in parentheses */
enum {
  MV_KNOB_ID_IMPL = 0,           /* (0) 0 auto: MFMA kernels for 16-bit data, VALU for f32 | 1 plain VALU kernels for every dtype */
  MV_KNOB_ID_GEMM_FORCE = 1,     /* (0) 0 auto | 1 the 128x128x64 kernel | 2 the 256-row LDS-DMA kernel */
  MV_KNOB_ID_GEMM_NJ = 2,        /* (0) 0 auto | ring-kernel variant: 14, 24, 2, 32 (csrc/mv_gemm.hip) */
  MV_KNOB_ID_GEMM_DBG = 3,       /* (0) timing-experiment bits of the ring kernels; results are wrong by construction */
  MV_KNOB_ID_ATTN_PLANES = 4,    /* (16) bits per uniform of mv_attn_dropmask: 16 | 12 | 8; P(drop) = round(p 2^n) / 2^n */
  MV_KNOB_ID_PERSISTENT_CUS = 5, /* (0) the persistent weight-gradient GEMM kernels launch at most n blocks; 0 = one per CU */
  MV_KNOB_ID_ROWOPS_VARIANT = 6, /* (0) mv_layernorm_bwd: low byte 0 prefetching kernel, 8 waves per block | 1 one row at a time, 4 waves |
                                    2, 3 prefetching, 4 / 16 waves; bits 8.. = grid cap (0 = default) */
  MV_KNOB_ID_ATTN_FWD = 8,       /* (0) reserved for profiles/r05_two_subtile_attention_experiment.patch (two 32-query sub-tiles per wave): no effect otherwise */
  MV_KNOB_ID_GEMM_ROUNDS = 9,    /* (1) mv_gemm chooses the ring tile height (256 / 320 rows) that minimises whole rounds of CUs; 0 = the choice of rounds 1-4 */
  MV_KNOB_ID_ATTN_ORDER = 7      /* (0) attention block order: 0 row block slowest | 1 a (sample, head)'s row blocks adjacent on one XCD
                                    (measured slower: profiles/r05_notes.txt) */
};
int mv_debug_set_knob(int id, int value);   /* 0, or MV_E_ARG for an unknown id / value */
int mv_debug_get_knob(int id);

#ifdef __cplusplus
}
#endif
#endif
