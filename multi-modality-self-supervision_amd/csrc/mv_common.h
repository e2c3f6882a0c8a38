// Shared device helpers for the gfx950 kernels (wave64, MFMA, LDS).  gfx950 only: no
// portability layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/medvill.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
// f16 encoding of the FORWARD operands of the 16-bit path (MV_F16): 11-bit significand instead of bf16's 8; the MFMA rate
// is the same (v_mfma_f32_16x16x32_f16 / 32x32x16_f16), gradients keep the bf16 encoding for its exponent range
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

#define MV_LDS __attribute__((address_space(3)))

// Kernel-selection knobs (test / experiment hooks).  The PRODUCT library has no mutable state: mv_knob() returns compile-time
// defaults there and nothing can change them.  Only the debug build of mv_api.hip (-DMV_DEBUG_KNOBS -> libmedvill_hip_dbg.so,
// include/medvill_debug.h) holds a table that mv_debug_set_knob() writes; every other translation unit is shared by both libraries.
enum {
  MV_KNOB_IMPL = 0,            // 0 auto (MFMA for 16-bit, VALU for f32) | 1 plain VALU kernels for every dtype
  MV_KNOB_GEMM_FORCE = 1,      // 0 auto | 1 the 128x128x64 kernel | 2 the 256-row LDS-DMA kernel
  MV_KNOB_GEMM_NJ = 2,         // 0 auto | kernel variant (mv_gemm.hip)
  MV_KNOB_GEMM_DBG = 3,        // timing-experiment bits of the ring kernels (results wrong by construction)
  MV_KNOB_ATTN_PLANES = 4,     // bits per uniform of the attention-dropout generator: 16 | 12 | 8
  MV_KNOB_PERSISTENT_CUS = 5,  // persistent GEMM kernels launch at most n blocks (0 = one per CU)
  MV_KNOB_ROWOPS_VARIANT = 6,  // mv_layernorm_bwd kernel form
  MV_KNOB_ATTN_ORDER = 7,      // attention block -> (row block, head, sample) order: 0 row block slowest (default) | 1 a pair's row blocks adjacent on one XCD
  MV_KNOB_ATTN_FWD = 8,        // reserved: the two-sub-tile forward experiment (profiles/r05_two_subtile_attention_experiment.patch) selects its kernel with it
  MV_KNOB_GEMM_ROUNDS = 9,     // 1 (default): mv_gemm picks the ring tile height (256 / 320 rows) that minimises whole rounds of CUs | 0: rounds 1-4's choice
  MV_KNOB_COUNT = 10
};
__attribute__((visibility("hidden"))) int mv_knob(int id);
#define g_mv_impl (mv_knob(MV_KNOB_IMPL))

#define MV_CHECK_LAUNCH()                           \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return (float)*p; }
template <> __device__ __forceinline__ float ldf<f16_t>(const f16_t* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = (bf16_t)v; }
template <> __device__ __forceinline__ void stf<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }

// 4-wide vector access (pointer must be 4-element aligned)
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
  bf16x4 v = *(const bf16x4*)p;
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
template <> __device__ __forceinline__ f32x4 ld4<f16_t>(const f16_t* p) {
  f16x4 v = *(const f16x4*)p;
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) {
  bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *(bf16x4*)p = r;
}

template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, f32x4 v) {
  f16x4 r = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
  *(f16x4*)p = r;
}

// runtime-typed access (dtype: MV_F32 / MV_BF16 / MV_F16)
__device__ __forceinline__ float ld_any(const void* p, size_t i, int dtype) {
  return dtype == MV_F32 ? ((const float*)p)[i] : dtype == MV_BF16 ? (float)((const bf16_t*)p)[i] : (float)((const f16_t*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, size_t i, int dtype, float v) {
  if (dtype == MV_F32) ((float*)p)[i] = v; else if (dtype == MV_BF16) ((bf16_t*)p)[i] = (bf16_t)v; else ((f16_t*)p)[i] = (f16_t)v;
}
// 4 consecutive elements at element offset i (i % 4 == 0, base aligned for the vector width)
__device__ __forceinline__ f32x4 ld4_any(const void* p, size_t i, int dtype) {
  return dtype == MV_F32 ? ld4<float>((const float*)p + i) : dtype == MV_BF16 ? ld4<bf16_t>((const bf16_t*)p + i) : ld4<f16_t>((const f16_t*)p + i);
}
__device__ __forceinline__ void st4_any(void* p, size_t i, int dtype, f32x4 v) {
  if (dtype == MV_F32) st4<float>((float*)p + i, v); else if (dtype == MV_BF16) st4<bf16_t>((bf16_t*)p + i, v); else st4<f16_t>((f16_t*)p + i, v);
}
__host__ __device__ __forceinline__ bool mv_is16(int dtype) { return dtype == MV_BF16 || dtype == MV_F16; }
__host__ __device__ __forceinline__ bool mv_dtype_ok(int dtype) { return dtype == MV_F32 || dtype == MV_BF16 || dtype == MV_F16; }

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level): 1 rcp + 1 exp + 6 fma instead
// of libdevice erff's ~40-instruction branchy polynomial.  The GELU epilogues run it 65536 times per 256x256 tile, where
// the arithmetic is what the epilogue costs (profiles/r02_gemm_variants.txt), so: the reciprocal is the bare v_rcp_f32
// (1 ulp; `__frcp_rn` / `1.0f / x` compile to the 12-instruction IEEE division sequence), 1/sqrt2 and the 0.5 of the cdf are
// folded into the constants, and the sign is handled by one select on 0.5 * erfc(|z| / sqrt2).
#define MV_AS_P 0.3275911f
#define MV_AS_A1 0.254829592f
#define MV_AS_A2 -0.284496736f
#define MV_AS_A3 1.421413741f
#define MV_AS_A4 -1.453152027f
#define MV_AS_A5 1.061405429f
// h = 0.5 * erfc(|z| / sqrt2) = upper tail of the standard normal; e = exp(-z^2 / 2)
__device__ __forceinline__ float mv_norm_tail(float z, float& e) {
  const float t = __builtin_amdgcn_rcpf(fmaf(MV_AS_P * 0.70710678118654752440f, fabsf(z), 1.0f));
  float p = fmaf(0.5f * MV_AS_A5, t, 0.5f * MV_AS_A4);
  p = fmaf(p, t, 0.5f * MV_AS_A3);
  p = fmaf(p, t, 0.5f * MV_AS_A2);
  p = fmaf(p, t, 0.5f * MV_AS_A1);
  e = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170368f);      // exp(-z^2/2) = 2^(-z^2 * log2(e) / 2)
  return p * t * e;
}
__device__ __forceinline__ float fast_erf(float x) {
  float e;
  const float h = mv_norm_tail(x * 1.41421356237309504880f, e);      // erf(x) = 1 - 2 * tail(x * sqrt2)
  return copysignf(fmaf(-2.0f, h, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float z) {
  float e;
  const float h = mv_norm_tail(z, e);
  return z * (z >= 0.f ? 1.0f - h : h);
}
// d/dz [ z * Phi(z) ] = Phi(z) + z * phi(z).  The tail's exp(-z^2/2) IS the Gaussian pdf's: one exponential and one
// reciprocal serve both terms.
__device__ __forceinline__ float dgelu_erf(float z) {
  float e;
  const float h = mv_norm_tail(z, e);
  const float cdf = z >= 0.f ? 1.0f - h : h;
  return fmaf(z * 0.39894228040143267794f, e, cdf);
}

// gelu(z) and gelu'(z) from one exponential and one reciprocal (forward epilogue MV_EPI_BIAS_GELU_D)
__device__ __forceinline__ void gelu_erf_and_grad(float z, float& g, float& d) {
  float e;
  const float h = mv_norm_tail(z, e);
  const float cdf = z >= 0.f ? 1.0f - h : h;
  g = z * cdf;
  d = fmaf(z * 0.39894228040143267794f, e, cdf);
}

// ---- dropout of the hidden-state sites: counter-based mask, regenerated (never stored) in the backward kernels -----------
// One 32-bit hash per PAIR of consecutive elements (linear index >> 1), 16 bits per element; an element is dropped when its
// half-word < thr, i.e. with probability thr/65536 (p = 0.1 -> thr = 6554 -> 0.100006; survivors are scaled by 1/(1 - thr/65536),
// so the expectation is preserved exactly).  Keys are derived on the host per (step, site, layer).  (The attention-probability
// site uses a stored keep-bit tensor instead: mv_attn_dropmask.)
struct DropCfg {
  unsigned k0, k1, thr;   // thr == 0: dropout off
  float inv_keep;
};
__device__ __forceinline__ unsigned mv_hash32(unsigned x, unsigned k0, unsigned k1) {
  x = (x ^ k0) * 0x9E3779B1u;      // two full 32-bit multiplies + three xor-shifts per 4 mask bytes
  x ^= x >> 15;
  x = (x ^ k1) * 0x85EBCA6Bu;
  x ^= x >> 13;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool mv_keep(unsigned h, int e, unsigned thr) { return ((h >> (16 * e)) & 0xffffu) >= thr; }    // e = 0, 1
// 4 consecutive elements starting at linear index idx (idx % 4 == 0)
__device__ __forceinline__ f32x4 mv_drop4(f32x4 v, size_t idx, const DropCfg& d) {
  const unsigned c = (unsigned)(idx >> 1);
  const unsigned h0 = mv_hash32(c, d.k0, d.k1), h1 = mv_hash32(c + 1u, d.k0, d.k1);
  v[0] = mv_keep(h0, 0, d.thr) ? v[0] * d.inv_keep : 0.f;
  v[1] = mv_keep(h0, 1, d.thr) ? v[1] * d.inv_keep : 0.f;
  v[2] = mv_keep(h1, 0, d.thr) ? v[2] * d.inv_keep : 0.f;
  v[3] = mv_keep(h1, 1, d.thr) ? v[3] * d.inv_keep : 0.f;
  return v;
}
__device__ __forceinline__ float mv_drop1(float v, size_t idx, const DropCfg& d) {
  const unsigned h = mv_hash32((unsigned)(idx >> 1), d.k0, d.k1);
  return mv_keep(h, (int)(idx & 1), d.thr) ? v * d.inv_keep : 0.f;
}
static inline DropCfg mv_make_drop(float p, unsigned long long key) {
  DropCfg d;
  int thr = (int)(p * 65536.0f + 0.5f);
  if (p <= 0.f) thr = 0;
  if (thr > 65535) thr = 65535;
  d.thr = (unsigned)thr;
  d.k0 = (unsigned)(key & 0xffffffffULL);
  d.k1 = (unsigned)(key >> 32);
  d.inv_keep = 65536.0f / (65536.0f - (float)thr);
  return d;
}

// ---- LDS-DMA (global -> LDS without registers) issued from inline assembly.  Through the builtin the compiler tracks the transfer as
// a pending LDS write and, having no alias information for `ds_read_b64_tr_b16`, puts `s_waitcnt vmcnt(0)` in front of the first
// transposed fragment read that follows -- which drains the transfer just issued for the NEXT stage before the current stage's MFMAs
// start.  Kernels that use these order their rings themselves (counted `s_waitcnt vmcnt(N)` + `s_barrier` before a stage is read).
typedef int dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const void* p, unsigned bytes) {      // raw buffer, stride 0, `bytes` records
  const unsigned long long v = (unsigned long long)(uintptr_t)p;
  return (dma_rsrc_t){(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// (M0 is compiler-reserved: it is written inside the SAME statement that reads it, as the CDNA guide prescribes; an "m0" clobber would
// only draw "clobber list contains reserved registers" from hipcc -- ROCm 7.2 -- on every instantiation)
__device__ __forceinline__ void lds_dma16(dma_rsrc_t rs, MV_LDS void* dst, unsigned voff) {   // dst: wave-uniform, lane i lands at +16 i
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"((unsigned)(uintptr_t)dst), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void lds_dma4(dma_rsrc_t rs, MV_LDS void* dst, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"((unsigned)(uintptr_t)dst), "v"(voff), "s"(rs) : "memory");
}

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }   // bare v_exp_f32

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int mv_dtype_size(int dt) { return dt == MV_F32 ? 4 : 2; }   // MV_BF16 and MV_F16 are both 2 bytes
