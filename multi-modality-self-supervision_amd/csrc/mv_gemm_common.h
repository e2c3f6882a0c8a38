// Shared device code of the MFMA GEMM kernels (mv_gemm*.hip): argument block, fused epilogues, LDS image layouts, LDS-DMA
// staging and fragment reads of the 256-row kernels.  Split out so that each kernel family is its own translation unit
// (they compile in parallel).
#pragma once
#include "mv_common.h"

struct GemmArgs {
  const void* A; const void* B; void* C; void* C2; const float* bias; const void* R;
  void* C3;       // optional copy of C in a second 16-bit encoding (c3_dtype): the forward writes the f16 operand of the next
                  // forward GEMM and the bf16 operand of the backward's weight-gradient GEMM from one accumulator tile
  int M, N, K, lda, ldb, ldc, ldc2, ldr, ldc3;
  int c_dtype, r_dtype, c3_dtype, epi, accumulate, vec_ok;
  int vec8_ok;    // 16-bit C (and C2 / C3): 16-byte aligned bases and leading dimensions that are multiples of 8 -> 16-byte stores
  int r8_ok;      // the elementwise operand R likewise: 16-bit, 16-byte aligned, ldr a multiple of 8 -> 16-byte loads
  int kchunk, splitk;
  float* ws;
  unsigned bytesA, bytesB;
  unsigned bytesC, bytesC2, bytesC3;   // 16-bit outputs of the 16-byte-store epilogue, as buffer sizes (that path needs them < 2 GiB)
  const float* alpha;   // nullable, MV_EPI_NONE with an f32 C only: C = *alpha * (A.B) (+ C when accumulating) -- 1 / loss scale of the
                        // f16-gradient path, applied where a weight gradient is written (mv_gemm's alpha_dev)
  float* csum;    // nullable (256x256 ring kernel, 16-bit C, N % 256 == 0): f32 [2 * ceil(M/256)][N] -- row 2*tm + half receives the sums over
                  // the 128 rows of that tile half of every output column (taken from the f32 values before they are rounded): the
                  // column sums of C without a second pass over it (mv_colsum_partials folds the rows)
  DropCfg drop;   // MV_EPI_BIAS_RES only: C = dropout(A.B + bias) + R
  int dbg;   // ablation bits (timing experiments only): 1 skip C stores, 2 skip operand loads, 4 skip LDS reads + MFMA
  // implicit convolution (mv_conv2d): A is an NHWC activation [B, cvH, cvW, cvC]; its logical row m = (b, oy, ox) and column
  // k = (ky*cvKw + kx)*cvC + c are gathered from pixel (oy*stride - pad + ky, ox*stride - pad + kx), zero outside
  int cvH, cvW, cvC, cvCshift, cvKw, cvStride, cvPad, cvHo, cvWo;
};

// ------------------------------------------------------------------------------------------
// fused epilogue on 4 consecutive columns (n .. n+3) of row m.
// Slow path (scalar, run-time epilogue selector): ragged right edge / unaligned leading dimensions.
__device__ __forceinline__ void epilogue4_slow(const GemmArgs& p, int m, int n, f32x4 v) {
  const int nv = p.N - n;
  if (m >= p.M || nv <= 0) return;
  const size_t co = (size_t)m * p.ldc + n;
  const int lim = nv < 4 ? nv : 4;
  const float al = p.alpha ? *p.alpha : 1.0f;
  for (int i = 0; i < lim; ++i) {
    float x = v[i] * al;
    const int e = p.epi;
    float b = 0.f, r = 0.f;
    if (e == MV_EPI_BIAS || e == MV_EPI_BIAS_GELU || e == MV_EPI_BIAS_RES || e == MV_EPI_BIAS_TANH || e == MV_EPI_BIAS_GELU_D || e == MV_EPI_BIAS_RELU || e == MV_EPI_BIAS_RES_RELU) b = p.bias[n + i];
    if (e == MV_EPI_BIAS_RES || e == MV_EPI_DGELU || e == MV_EPI_RES || e == MV_EPI_MUL || e == MV_EPI_BIAS_RES_RELU) r = ld_any(p.R, (size_t)m * p.ldr + n + i, p.r_dtype);
    switch (e) {
      case MV_EPI_BIAS: x += b; break;
      case MV_EPI_BIAS_GELU: x += b; break;
      case MV_EPI_BIAS_GELU_D: x += b; break;
      case MV_EPI_MUL: x *= r; break;
      case MV_EPI_BIAS_RELU: x = fmaxf(x + b, 0.f); break;
      case MV_EPI_BIAS_RES_RELU: x = fmaxf(x + b + r, 0.f); break;
      case MV_EPI_BIAS_RES:
        x += b;
        if (p.drop.thr) x = mv_drop1(x, (size_t)m * p.N + n + i, p.drop);
        x += r;
        break;
      case MV_EPI_DGELU: x *= dgelu_erf(r); break;
      case MV_EPI_RES: x += r; break;
      case MV_EPI_BIAS_TANH: x = tanhf(x + b); break;
      default: break;
    }
    if (e == MV_EPI_BIAS_GELU) {
      st_any(p.C2, (size_t)m * p.ldc2 + n + i, p.c_dtype, x);
      x = gelu_erf(x);
    }
    if (e == MV_EPI_BIAS_GELU_D) {
      float g_, d_;
      gelu_erf_and_grad(x, g_, d_);
      st_any(p.C2, (size_t)m * p.ldc2 + n + i, p.c_dtype, d_);
      x = g_;
    }
    if (p.c_dtype == MV_F32 && p.accumulate) x += ((const float*)p.C)[co + i];
    st_any(p.C, co + i, p.c_dtype, x);
    if (p.C3) st_any(p.C3, (size_t)m * p.ldc3 + n + i, p.c3_dtype, x);
  }
}

// Fast path: compile-time epilogue, 16-byte bias / residual loads, 8- or 16-byte stores, with its global loads taken out
// of the store stream: bias (b4, one load per tile: a lane keeps its 4 columns for all
// rows) and the residual operand (r4) are fetched by the caller AHEAD of the stores of the previous rows.  gfx9 retires
// loads and stores through one in-order counter (vmcnt), so a load issued after a store cannot be waited for without
// waiting for that store's round trip to L2 as well; with the loads one row-group ahead, the stores stream out
// back-to-back.
template <int E>
__device__ __forceinline__ f32x4 epi_load_res4(const GemmArgs& p, int m, int n) {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (m >= p.M) return r;
  if (E == MV_EPI_BIAS_RES || E == MV_EPI_DGELU || E == MV_EPI_RES || E == MV_EPI_MUL || E == MV_EPI_BIAS_RES_RELU) {
    const size_t ro = (size_t)m * p.ldr + n;
    r = ld4_any(p.R, ro, p.r_dtype);
  } else if (E == MV_EPI_NONE) {
    if (p.c_dtype == MV_F32 && p.accumulate) r = *(const f32x4*)((const float*)p.C + (size_t)m * p.ldc + n);
  }
  return r;
}
template <int E>
__device__ __forceinline__ void epilogue4v(const GemmArgs& p, int m, int n, f32x4 v, f32x4 b4, f32x4 r) {
  if (m >= p.M) return;
  const size_t co = (size_t)m * p.ldc + n;
  f32x4 o = v;
  if (E == MV_EPI_NONE && p.alpha) o *= *p.alpha;
  if (E == MV_EPI_BIAS || E == MV_EPI_BIAS_GELU || E == MV_EPI_BIAS_RES || E == MV_EPI_BIAS_TANH || E == MV_EPI_BIAS_GELU_D || E == MV_EPI_BIAS_RELU || E == MV_EPI_BIAS_RES_RELU) o += b4;
  if (E == MV_EPI_BIAS_RES && p.drop.thr) o = mv_drop4(o, (size_t)m * p.N + n, p.drop);
  if (E == MV_EPI_DGELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] *= dgelu_erf(r[i]);
  } else if (E == MV_EPI_MUL) {
    o *= r;
  } else if (E == MV_EPI_BIAS_RES || E == MV_EPI_RES || E == MV_EPI_NONE || E == MV_EPI_BIAS_RES_RELU) {
    o += r;
  }
  if (E == MV_EPI_BIAS_RELU || E == MV_EPI_BIAS_RES_RELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.f);
  }
  if (E == MV_EPI_BIAS_TANH) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = tanhf(o[i]);
  }
  if (E == MV_EPI_BIAS_GELU) {
    const size_t c2 = (size_t)m * p.ldc2 + n;
    st4_any(p.C2, c2, p.c_dtype, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = gelu_erf(o[i]);
  }
  if (E == MV_EPI_BIAS_GELU_D) {
    f32x4 d;
#pragma unroll
    for (int i = 0; i < 4; ++i) { float g_, d_; gelu_erf_and_grad(o[i], g_, d_); o[i] = g_; d[i] = d_; }
    const size_t c2 = (size_t)m * p.ldc2 + n;
    st4_any(p.C2, c2, p.c_dtype, d);
  }
  st4_any(p.C, co, p.c_dtype, o);
  if (p.C3) st4_any(p.C3, (size_t)m * p.ldc3 + n, p.c3_dtype, o);
}

// 8 consecutive columns of one row of a 16-bit output: ONE 16-byte store per output tensor instead of two 8-byte ones (the
// epilogue of a 16-bit tile is bound by the number of store instructions it issues, not by their bytes).  Epilogues without
// a residual operand only: bias, bias + GELU (+ derivative), plain.
// 8 consecutive 16-bit outputs as ONE 16-byte buffer store.  (Write-through `sc1` stores -- outputs that do not stay in the XCD's L2,
// where they evict the operand panels the XCD's other tiles re-read: the FFN-up kernel fetches its x panels ~4 times,
// profiles/r03_ffn1_kernel_stats.txt -- were measured slower on every shape: profiles/r03_notes.txt.)
__device__ __forceinline__ void st8_16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, int dtype, const float (&o)[8]) {
  u32x4 bits;
  if (dtype == MV_BF16) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)o[e];
    bits = __builtin_bit_cast(u32x4, r);
  } else {
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (f16_t)o[e];
    bits = __builtin_bit_cast(u32x4, r);
  }
  __builtin_amdgcn_raw_buffer_store_b128(bits, rs, byte_off, 0, 0);
}
// 8 consecutive 16-bit elements of the elementwise operand, as raw bits (zeros past the last row); decoded at the point of use
__device__ __forceinline__ u32x4 ld8_raw(const GemmArgs& p, int m, int n) {
  if (m >= p.M) return (u32x4){0u, 0u, 0u, 0u};
  return *(const u32x4*)((const char*)p.R + ((size_t)m * p.ldr + n) * 2);
}
__device__ __forceinline__ void dec8_16(u32x4 raw, int dtype, float (&r)[8]) {
  if (dtype == MV_BF16) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (float)v[e];
  } else {
    const f16x8 v = __builtin_bit_cast(f16x8, raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (float)v[e];
  }
}
struct OutRsrc { __amdgpu_buffer_rsrc_t c, c2, c3; };
template <int E>
__device__ __forceinline__ void epilogue8_16(const GemmArgs& p, const OutRsrc& rs, int m, int n, f32x4 v0, f32x4 v1, f32x4 b0, f32x4 b1, float (&cs)[8],
                                             bool do_cs, u32x4 rraw = (u32x4){0u, 0u, 0u, 0u}) {
  if (m >= p.M) return;
  if (p.dbg & 16) m &= 255;       // experiment: every tile stores into the same 256 rows (they stay in L2): what the epilogue costs without HBM writes
  float o[8] = {v0[0] + b0[0], v0[1] + b0[1], v0[2] + b0[2], v0[3] + b0[3], v1[0] + b1[0], v1[1] + b1[1], v1[2] + b1[2], v1[3] + b1[3]};
  if (E == MV_EPI_BIAS_RES && p.drop.thr) {      // hidden-state dropout of the projection output, before the residual is added
    const size_t li = (size_t)m * p.N + n;
    const f32x4 d0 = mv_drop4((f32x4){o[0], o[1], o[2], o[3]}, li, p.drop), d1 = mv_drop4((f32x4){o[4], o[5], o[6], o[7]}, li + 4, p.drop);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = d0[e]; o[4 + e] = d1[e]; }
  }
  if (E == MV_EPI_MUL || E == MV_EPI_RES || E == MV_EPI_BIAS_RES) {
    float r[8];
    dec8_16(rraw, p.r_dtype, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (E == MV_EPI_MUL) ? o[e] * r[e] : o[e] + r[e];
  }
  if (E == MV_EPI_BIAS_GELU_D) {
    float dd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { float g_, d_; gelu_erf_and_grad(o[e], g_, d_); o[e] = g_; dd[e] = d_; }
    st8_16(rs.c2, ((unsigned)m * (unsigned)p.ldc2 + (unsigned)n) * 2u, p.c_dtype, dd);
  }
  st8_16(rs.c, ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 2u, p.c_dtype, o);
  if (p.C3) st8_16(rs.c3, ((unsigned)m * (unsigned)p.ldc3 + (unsigned)n) * 2u, p.c3_dtype, o);
  if (do_cs) {
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] += o[e];
  }
}

// run BODY(E) with the run-time epilogue selector turned into a compile-time constant
#define MV_EPI_SWITCH(epi_, BODY)                          \
  switch (epi_) {                                          \
    case MV_EPI_BIAS: BODY(MV_EPI_BIAS); break;            \
    case MV_EPI_BIAS_GELU: BODY(MV_EPI_BIAS_GELU); break;  \
    case MV_EPI_BIAS_RES: BODY(MV_EPI_BIAS_RES); break;    \
    case MV_EPI_DGELU: BODY(MV_EPI_DGELU); break;          \
    case MV_EPI_RES: BODY(MV_EPI_RES); break;              \
    case MV_EPI_BIAS_TANH: BODY(MV_EPI_BIAS_TANH); break;  \
    case MV_EPI_BIAS_GELU_D: BODY(MV_EPI_BIAS_GELU_D); break; \
    case MV_EPI_MUL: BODY(MV_EPI_MUL); break;              \
    case MV_EPI_BIAS_RELU: BODY(MV_EPI_BIAS_RELU); break;  \
    case MV_EPI_BIAS_RES_RELU: BODY(MV_EPI_BIAS_RES_RELU); break; \
    default: BODY(MV_EPI_NONE); break;                     \
  }

// raw partial tile store for split-K (ws is [splitk][M][N] f32)
__device__ __forceinline__ void store_partial4(const GemmArgs& p, int split, int m, int n, f32x4 v) {
  const int nv = p.N - n;
  if (m >= p.M || nv <= 0) return;
  float* w = p.ws + ((size_t)split * p.M + m) * p.N + n;
  if ((p.N & 3) == 0 && nv >= 4) { *(f32x4*)w = v; return; }
  for (int i = 0; i < (nv < 4 ? nv : 4); ++i) w[i] = v[i];
}

// one 16x16x32 MFMA on 16-bit operands held as bf16x8 bit patterns: F16 selects the f16 encoding (forward operands of
// the MV_F16 path), otherwise bf16.  Same rate, same fragment layout.
template <bool F16>
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// k-contiguous ("row") tile image: [128 rows][64 k] bf16, 128-B rows, chunk ^= (row>>1)&7
__device__ __forceinline__ int row_img_off(int r, int ch) { return r * 128 + ((ch ^ ((r >> 1) & 7)) << 4); }
// contraction-major ("tr") tile image: [64 k][128 cols] bf16, 256-B rows, chunk ^= swz(k)
__device__ __forceinline__ int tr_swz(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }
__device__ __forceinline__ int tr_img_off(int kr, int ch) { return kr * 256 + ((ch ^ tr_swz(kr)) << 4); }

// Large-tile MFMA kernel ("ring" kernel): 256 x BN x 32 per stage, waves 2 x WN, each wave 128 x (16*NJ).
//   <NJ=4, WN=4, 4 stages>  256x256 tile, 8 waves, 128 KiB LDS, 1 block / CU
//   <NJ=3, WN=4, 4 stages>  256x192 tile (N = 768 / 2304 without a ragged last column of tiles)
//   <NJ=4, WN=2, 3 stages>  256x128 tile, 4 waves,  72 KiB LDS, 2 blocks / CU: the second block's MFMAs cover the
//                           first one's prologue / epilogue
// Operands go HBM -> LDS directly (buffer_load ... lds, 16 B per lane, out-of-range lanes write 0) into a ring of
// NSTAGE stages: while stage s feeds the MFMAs, the following NSTAGE-1 stages are in flight, tracked with a counted
// s_waitcnt vmcnt(N) and ONE raw s_barrier per stage -- the loads are never drained inside the loop.  The LDS image
// is lane-linear per 1-KiB piece (that is what an LDS-DMA writes); the bank-conflict swizzle is applied to the
// per-lane SOURCE address and again on the fragment read.  0.375 LDS fragment reads per MFMA.
#define G2_BM 256
#define G2_BK 32

// k-contiguous image: [rows][32 k] bf16 = 64-B rows, 4 chunks per row
__device__ __forceinline__ int r2_f(int r) { return (0 - (r >> 2)) & 3; }
__device__ __forceinline__ int r2_off(int r, int c) { return r * 64 + ((c ^ r2_f(r)) << 4); }
// contraction-major image, 512-B rows ([32 k][256 cols]): 32 chunks per row, XOR at 32-B granularity
__device__ __forceinline__ int t2_g(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }
__device__ __forceinline__ int t2_off(int kr, int ch) { return kr * 512 + ((ch ^ (t2_g(kr) << 1)) << 4); }
// contraction-major image, 256-B rows ([32 k][128 cols]): tr_img_off() of the 128x128 kernel

// PITCH512: contraction-major image with 512-B rows (tile width 192/256) or 256-B rows (tile width 128)
template <bool TR, bool PITCH512, int NPIECE, int NW, int KS>
__device__ __forceinline__ void g2_issue(dma_rsrc_t rs, unsigned bytes, int ld, int row0, int rows_total,
                                         int tile_rows, int k0, int kend, char* region, int wid, int lane, int dbg = 0) {
#pragma unroll
  for (int q = 0; q < NPIECE / NW; ++q) {
    const int pc = wid + NW * q;                // 1-KiB piece of the operand image
    unsigned off;
    bool ok;
    if (!TR && KS == 2) {       // 8 rows x 128 B per piece: whole cache lines
      const int r = 8 * pc + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      const int gr = row0 + r, gk = k0 + c * 8;
      ok = (r < tile_rows) && (gr < rows_total) && (gk < kend);
      off = ((unsigned)gr * (unsigned)ld + (unsigned)gk) * 2u;
    } else if (!TR) {
      const int r = 16 * pc + (lane >> 2), c = (lane & 3) ^ r2_f(16 * pc + (lane >> 2));
      const int gr = row0 + r, gk = k0 + c * 8;
      ok = (r < tile_rows) && (gr < rows_total) && (gk < kend);
      off = ((unsigned)gr * (unsigned)ld + (unsigned)gk) * 2u;
      if (dbg & 8) off = ((unsigned)(row0 + 8 * pc + (lane >> 3)) * (unsigned)ld + (unsigned)((k0 & ~63) + (lane & 7) * 8)) * 2u;
    } else if (PITCH512) {
      const int kr = 2 * pc + (lane >> 5), c = (lane & 31) ^ (t2_g(2 * pc + (lane >> 5)) << 1);
      const int gk = k0 + kr, gc = row0 + c * 8;
      ok = (c * 8 < tile_rows) && (gk < kend) && (gc < rows_total);
      off = ((unsigned)gk * (unsigned)ld + (unsigned)gc) * 2u;
    } else {
      const int kr = 4 * pc + (lane >> 4), c = (lane & 15) ^ tr_swz(4 * pc + (lane >> 4));
      const int gk = k0 + kr, gc = row0 + c * 8;
      ok = (c * 8 < tile_rows) && (gk < kend) && (gc < rows_total);
      off = ((unsigned)gk * (unsigned)ld + (unsigned)gc) * 2u;
    }
    lds_dma16(rs, (MV_LDS void*)(region + pc * 1024), ok ? off : bytes);     // asm: see mv_common.h (no compiler-inserted ring drain)
  }
}

template <bool TR, bool PITCH512, int KS>
__device__ __forceinline__ bf16x8 g2_frag(const char* tile, int base, int l15, int lq, int ks = 0) {
  if (!TR) {
    if (KS == 2) return *(const bf16x8*)(tile + row_img_off(base + l15, ks * 4 + lq));
    return *(const bf16x8*)(tile + r2_off(base + l15, lq));
  } else {
    const int kr = ks * 32 + 8 * lq + (l15 >> 2);
    const int ch = (base >> 3) + ((l15 & 3) >> 1);
    const int sub = (l15 & 1) * 8;
    const int o0 = PITCH512 ? t2_off(kr, ch) : tr_img_off(kr, ch);
    const int o1 = PITCH512 ? t2_off(kr + 4, ch) : tr_img_off(kr + 4, ch);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(tile + o0 + sub));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(tile + o1 + sub));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// Epilogue of the 256-row kernels, through LDS: an accumulator tile holds 4 columns x 16 rows per lane, which would
// store as sixteen 32-byte fragments per instruction (measured: ~1 TB/s).  Each wave transposes 16 rows at a time
// through its own 4.25-KiB scratch (272-B row pitch: conflict-free both ways) so that 16 lanes cover one full output
// row: whole 128/256-byte lines per store.  Loads run one row-group ahead of the stores (see epilogue4v).
// Expects in scope: p, acc, scr, split, m0, n0, wm, wn, l15, lq, rrow, c4, col_on, NJ, G2_NI (16-row groups per wave).
// 16-bit outputs, a wave's 64 whole columns: the accumulators of one 16-row group go through the wave's LDS scratch (272-B row pitch) so that
// a lane owns 8 consecutive columns of a row (8 lanes per row, 8 rows per pass): 16-byte stores, whole 128-byte lines per row.  NI_ = number
// of 16-row groups per wave (8 in the 256-row kernels, 4 in the 128x128 kernel).  Expects in scope: p, acc[NI_][4], scr, m0, n0, wm, wn, l15,
// lq, lane, WIDE_E / WIDE_R (constexpr), EE.
#define G2_WIDE_COND(NJ_) ((WIDE_E || (WIDE_R && p.r8_ok)) && (NJ_) == 4 && p.vec8_ok && p.c_dtype != MV_F32 && !p.accumulate && n0 + wn + 64 <= p.N)
#define G2_EPI_WIDE(E_, NI_)                                                                                   \
    {                                                                                                          \
      /* 16-bit outputs, 64 whole columns: a lane owns 8 consecutive columns of a row (8 lanes per row, 8 rows per pass) */ \
      const int r8 = lane >> 3, c8 = lane & 7;                                                                 \
      const int ncol8 = n0 + wn + c8 * 8;                                                                      \
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;                                                                \
      if ((WIDE_E && (E_) != MV_EPI_NONE) || (E_) == MV_EPI_BIAS_RES) { b0 = *(const f32x4*)(p.bias + ncol8); b1 = *(const f32x4*)(p.bias + ncol8 + 4); } \
      /* the elementwise operand's rows are requested two 16-row groups ahead of their use (4 x 16 bytes per lane in flight) */ \
      u32x4 rq[2][2];                                                                                          \
      float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                                  \
      OutRsrc ors;                                                                                             \
      ors.c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.bytesC, 0x00020000);                                 \
      ors.c2 = __builtin_amdgcn_make_buffer_rsrc(p.C2, 0, p.bytesC2, 0x00020000);                              \
      ors.c3 = __builtin_amdgcn_make_buffer_rsrc(p.C3, 0, p.bytesC3, 0x00020000);                              \
      const bool do_cs = p.csum != nullptr;                                                                    \
      if (WIDE_R) {                                                                                            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                          \
          _Pragma("unroll") for (int rr = 0; rr < 2; ++rr) rq[t][rr] = ld8_raw(p, m0 + wm + t * 16 + rr * 8 + r8, ncol8); \
      }                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < (NI_); ++i) {                                                          \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) *(f32x4*)(scr + l15 * 272 + j * 64 + lq * 16) = acc[i][j]; \
        _Pragma("unroll") for (int rr = 0; rr < 2; ++rr) {                                                     \
          const int row = rr * 8 + r8;                                                                         \
          const f32x4 v0 = *(const f32x4*)(scr + row * 272 + c8 * 32);                                         \
          const f32x4 v1 = *(const f32x4*)(scr + row * 272 + c8 * 32 + 16);                                    \
          if (WIDE_R) {                                                                                        \
            const u32x4 rcur = rq[i & 1][rr];                                                                  \
            if (i + 2 < (NI_)) rq[i & 1][rr] = ld8_raw(p, m0 + wm + (i + 2) * 16 + row, ncol8);                    \
            epilogue8_16<EE>(p, ors, m0 + wm + i * 16 + row, ncol8, v0, v1, b0, b1, cs, do_cs, rcur);          \
          } else {                                                                                             \
            epilogue8_16<EE>(p, ors, m0 + wm + i * 16 + row, ncol8, v0, v1, b0, b1, cs, do_cs);                \
          }                                                                                                    \
        }                                                                                                      \
      }                                                                                                        \
      if (do_cs) {          /* fold the 8 row-lanes that share this lane's 8 columns; one lane per column group stores */ \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                        \
          float v = cs[e];                                                                                     \
          v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);                   \
          cs[e] = v;                                                                                           \
        }                                                                                                      \
        if (r8 == 0) {                                                                                         \
          float* cp = p.csum + ((size_t)(2 * (m0 / G2_BM) + wm / 128)) * p.N + ncol8;                          \
          *(f32x4*)cp = (f32x4){cs[0], cs[1], cs[2], cs[3]};                                                   \
          *(f32x4*)(cp + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};                                             \
        }                                                                                                      \
      }                                                                                                        \
    }
#define G2_RG 4
#define G2_EPI_BODY(E_)                                                                                        \
  {                                                                                                            \
    constexpr int EE = (E_) < 0 ? 0 : (E_);                                                                    \
    constexpr bool HAS_R = (E_) == MV_EPI_BIAS_RES || (E_) == MV_EPI_RES || (E_) == MV_EPI_MUL || (E_) == MV_EPI_DGELU || (E_) == MV_EPI_BIAS_RES_RELU; \
    constexpr bool WIDE_E = (E_) == MV_EPI_NONE || (E_) == MV_EPI_BIAS || (E_) == MV_EPI_BIAS_GELU_D;          \
    constexpr bool WIDE_R = (E_) == MV_EPI_MUL || (E_) == MV_EPI_RES || (E_) == MV_EPI_BIAS_RES;      /* 16-bit elementwise operand, 16-byte loads */ \
    if (G2_WIDE_COND(NJ)) G2_EPI_WIDE(E_, G2_NI)                                                                \
    else {                                                                                                     \
    const int ncol = n0 + wn + c4 * 4;                                                                         \
    const bool lane_fast = ((E_) >= 0) && col_on && p.vec_ok && (p.N - ncol >= 4);                             \
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};                                                                           \
    if (lane_fast && (EE == MV_EPI_BIAS || EE == MV_EPI_BIAS_GELU || EE == MV_EPI_BIAS_RES || EE == MV_EPI_BIAS_TANH || \
                      EE == MV_EPI_BIAS_GELU_D || EE == MV_EPI_BIAS_RELU || EE == MV_EPI_BIAS_RES_RELU))                                                               \
      b4 = *(const f32x4*)(p.bias + ncol);                                                                     \
    if (HAS_R && __all(lane_fast || !col_on)) {                                                                \
      /* residual operand: a tile's worth comes from HBM, so G2_RG 16-row groups of row loads are kept in flight \
         per wave; each slot is re-requested as soon as it has been consumed */                                \
      f32x4 rb[4 * G2_RG];                                                                                     \
      _Pragma("unroll") for (int t = 0; t < 4 * G2_RG; ++t) {                                                  \
        rb[t] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                   \
        if (col_on) rb[t] = epi_load_res4<EE>(p, m0 + wm + (t >> 2) * 16 + (t & 3) * 4 + rrow, ncol);          \
      }                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < G2_NI; ++i) {                                                      \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) *(f32x4*)(scr + l15 * 272 + j * 64 + lq * 16) = acc[i][j]; \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                     \
          const int row = rr * 4 + rrow, mcur = m0 + wm + i * 16 + row;                                        \
          const f32x4 v = *(const f32x4*)(scr + row * 272 + c4 * 16);                                          \
          const f32x4 rc = rb[(i % G2_RG) * 4 + rr];                                                           \
          if (i + G2_RG < G2_NI && col_on) rb[(i % G2_RG) * 4 + rr] = epi_load_res4<EE>(p, mcur + 16 * G2_RG, ncol); \
          if (col_on) epilogue4v<EE>(p, mcur, ncol, v, b4, rc);                                                \
        }                                                                                                      \
      }                                                                                                        \
    } else {                                                                                                   \
      f32x4 rcur = {0.f, 0.f, 0.f, 0.f};                                                                       \
      if (lane_fast) rcur = epi_load_res4<EE>(p, m0 + wm + rrow, ncol);                                        \
      _Pragma("unroll") for (int i = 0; i < G2_NI; ++i) {                                                      \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) *(f32x4*)(scr + l15 * 272 + j * 64 + lq * 16) = acc[i][j]; \
        _Pragma("unroll 1") for (int rr = 0; rr < 4; ++rr) {                                                   \
          const int row = rr * 4 + rrow, mcur = m0 + wm + i * 16 + row;                                        \
          const int fn = i * 4 + rr + 1;                                                                       \
          f32x4 rnext = {0.f, 0.f, 0.f, 0.f};                                                                  \
          if (lane_fast && fn < 4 * G2_NI) rnext = epi_load_res4<EE>(p, m0 + wm + (fn >> 2) * 16 + (fn & 3) * 4 + rrow, ncol); \
          const f32x4 v = *(const f32x4*)(scr + row * 272 + c4 * 16);                                          \
          if (col_on) {                                                                                        \
            if ((E_) < 0) store_partial4(p, split, mcur, ncol, v);                                             \
            else if (lane_fast) epilogue4v<EE>(p, mcur, ncol, v, b4, rcur);                                    \
            else epilogue4_slow(p, mcur, ncol, v);                                                             \
          }                                                                                                    \
          rcur = rnext;                                                                                        \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
    }                                                                                                          \
  }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// launchers of the 256-row kernels: one translation unit per operand layout (mv_gemm_ring_{nt,nn,tn,tnn}.hip; they compile
// in parallel).  variant: 14 = ring 256x256, 64-deep stages x2;  24 = persistent form;  f16: f16-encoded operands
int mv_launch_ring_nt(const GemmArgs& p, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream);    // y = x.W^T
int mv_launch_ring_nn(const GemmArgs& p, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream);    // dx = dy.W
int mv_launch_ring_tn(const GemmArgs& p, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream);    // dW = dy^T.x
int mv_launch_ring_tnn(const GemmArgs& p, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream);   // A^T.B^T (bf16 only)
int mv_launch_ring_tn4(const GemmArgs& p, bool f16, int tiles, int splitk, hipStream_t stream);                           // dW, 32-deep stages x4
static inline int mv_launch_ring(const GemmArgs& p, int ta, int tb, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream) {
  if (variant == 4 && ta && tb) return mv_launch_ring_tn4(p, f16, tiles, splitk, stream);
  if (!ta && !tb) return mv_launch_ring_nt(p, f16, variant, tiles, splitk, n_cu, stream);
  if (!ta && tb) return mv_launch_ring_nn(p, f16, variant, tiles, splitk, n_cu, stream);
  if (ta && tb) return mv_launch_ring_tn(p, f16, variant, tiles, splitk, n_cu, stream);
  return mv_launch_ring_tnn(p, f16, variant, tiles, splitk, n_cu, stream);
}
