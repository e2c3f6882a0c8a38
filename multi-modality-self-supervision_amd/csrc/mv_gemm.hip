// Dense projections for the CXRBERT hot path on gfx950.
//
//  * gemm_mfma_kernel<TA,TB>: bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//    128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.  Operands are
//    staged HBM -> registers (16-byte buffer loads, out-of-range lanes return 0) -> LDS,
//    double-buffered, one barrier per K-tile, next tile's loads in flight under the MFMAs.
//    A k-contiguous operand is kept as [rows][64 k] with a 16-byte-chunk XOR swizzle
//    (conflict-free ds_read_b128); an operand whose contraction index is the slow memory
//    dimension (the weight in dX = dY.W, both operands in dW = dY^T.X) is kept as
//    [64 k][128 cols] and consumed through ds_read_b64_tr_b16 (hardware transpose), so the
//    three GEMM forms of a Linear layer need no transposed copies in HBM.
//    The MFMA is issued with the operands swapped (D = W_frag x X_frag) so that each lane
//    ends up with 4 CONSECUTIVE output columns of one row: 8/16-byte stores, vector bias /
//    residual loads in the fused epilogue.
//  * gemm_simple_kernel<T>: plain VALU 64x64 tile kernel, any dtype; it is the exact-fp32
//    path (MV_F32) and the on-GPU cross-check of the MFMA kernel.
//
// Reference work replaced: every nn.Linear on the path (see include/medvill.h, mv_gemm).
#include "mv_common.h"

struct GemmArgs {
  const void* A; const void* B; void* C; void* C2; const float* bias; const void* R;
  void* C3;       // optional copy of C in a second 16-bit encoding (c3_dtype): the forward writes the f16 operand of the next
                  // forward GEMM and the bf16 operand of the backward's weight-gradient GEMM from one accumulator tile
  int M, N, K, lda, ldb, ldc, ldc2, ldr, ldc3;
  int c_dtype, r_dtype, c3_dtype, epi, accumulate, vec_ok;
  int kchunk, splitk;
  float* ws;
  unsigned bytesA, bytesB;
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
  for (int i = 0; i < lim; ++i) {
    float x = v[i];
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

// ------------------------------------------------------------------------------------------
// MFMA kernel
#define GT_BM 128
#define GT_BN 128
#define GT_BK 64
#define GT_STAGE_BYTES 32768   // A tile 16 KiB + B tile 16 KiB

// k-contiguous ("row") tile image: [128 rows][64 k] bf16, 128-B rows, chunk ^= (row>>1)&7
__device__ __forceinline__ int row_img_off(int r, int ch) { return r * 128 + ((ch ^ ((r >> 1) & 7)) << 4); }
// contraction-major ("tr") tile image: [64 k][128 cols] bf16, 256-B rows, chunk ^= swz(k)
__device__ __forceinline__ int tr_swz(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }
__device__ __forceinline__ int tr_img_off(int kr, int ch) { return kr * 256 + ((ch ^ tr_swz(kr)) << 4); }

template <bool TR>
__device__ __forceinline__ void stage_load(u32x4 (&reg)[4], __amdgpu_buffer_rsrc_t rs, unsigned bytes, int ld,
                                           int row0, int rows_total, int k0, int kend, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    unsigned off;
    bool ok;
    if (!TR) {
      const int r = idx >> 3, ch = idx & 7;
      const int gr = row0 + r, gk = k0 + ch * 8;
      ok = (gr < rows_total) && (gk < kend);
      off = ((unsigned)gr * (unsigned)ld + (unsigned)gk) * 2u;
    } else {
      const int kr = idx >> 4, ch = idx & 15;
      const int gk = k0 + kr, gc = row0 + ch * 8;
      ok = (gk < kend) && (gc < rows_total);
      off = ((unsigned)gk * (unsigned)ld + (unsigned)gc) * 2u;
    }
    reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : bytes, 0, 0);
  }
}

// per-thread constants of the 4 activation rows a thread stages in the implicit-convolution form
struct ConvRows { int pix[4]; int iy0[4], ix0[4]; bool on[4]; };
__device__ __forceinline__ ConvRows conv_rows(const GemmArgs& p, int row0, int tid) {
  ConvRows c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gr = row0 + ((tid + 256 * i) >> 3);
    c.on[i] = gr < p.M;
    const int hw = p.cvHo * p.cvWo;
    const int b = gr / hw, rem = gr - b * hw;
    const int oy = rem / p.cvWo, ox = rem - oy * p.cvWo;
    c.iy0[i] = oy * p.cvStride - p.cvPad;
    c.ix0[i] = ox * p.cvStride - p.cvPad;
    c.pix[i] = b * p.cvH * p.cvW;              // first pixel of the image
  }
  return c;
}
__device__ __forceinline__ void stage_load_conv(u32x4 (&reg)[4], __amdgpu_buffer_rsrc_t rs, unsigned bytes, const GemmArgs& p,
                                                const ConvRows& c, int k0, int kend, int tid) {
  const int ch = tid & 7;
  const int gk = k0 + ch * 8;                   // 8 consecutive channels of one filter tap (cvC % 8 == 0)
  const int tap = gk >> p.cvCshift, cc = gk & (p.cvC - 1);
  const int ky = tap / p.cvKw, kx = tap - ky * p.cvKw;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int iy = c.iy0[i] + ky, ix = c.ix0[i] + kx;
    const bool ok = c.on[i] && gk < kend && iy >= 0 && iy < p.cvH && ix >= 0 && ix < p.cvW;
    const unsigned off = ((unsigned)(c.pix[i] + iy * p.cvW + ix) * (unsigned)p.cvC + (unsigned)cc) * 2u;
    reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : bytes, 0, 0);
  }
}

template <bool TR>
__device__ __forceinline__ void stage_store(const u32x4 (&reg)[4], char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    int off;
    if (!TR) off = row_img_off(idx >> 3, idx & 7);
    else off = tr_img_off(idx >> 4, idx & 15);
    *(u32x4*)(tile + off) = reg[i];
  }
}

// fragment X[idx = base + (lane&15)][k = ks*32 + 8*(lane>>4) + j], j = 0..7
template <bool TR>
__device__ __forceinline__ bf16x8 load_frag(const char* tile, int base, int ks, int l15, int lq) {
  if (!TR) {
    const int r = base + l15;
    return *(const bf16x8*)(tile + row_img_off(r, ks * 4 + lq));
  } else {
    const int kr = ks * 32 + 8 * lq + (l15 >> 2);
    const int ch = (base >> 3) + ((l15 & 3) >> 1);
    const int sub = (l15 & 1) * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(tile + tr_img_off(kr, ch) + sub));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(tile + tr_img_off(kr + 4, ch) + sub));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// SB: ONE 32-KiB LDS stage instead of two (the next K-tile waits in registers, two barriers per K-tile) so that three blocks
// share a CU: 768 tile slots instead of 512, which turns 2.34 rounds of tiles (N = 768 at ~25k rows) into 1.56.
template <bool TA, bool TB, bool CONV = false, bool SB = false, bool F16 = false>
__global__ __launch_bounds__(256, SB ? 3 : 2) void gemm_mfma_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;

  // XCD-aware tile order: the 8 XCDs each take a contiguous run of tiles (bijective remap)
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, in = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
  }
  // grouped rasterisation inside each XCD's run: the ~64 blocks resident on an XCD form an 8 x 8 window of
  // tiles, so every A / B panel fetched into the XCD's 4 MiB L2 is reused by 8 blocks instead of 2-3
  const int tiles_n = (p.N + GT_BN - 1) / GT_BN;
  const int tiles_m = (p.M + GT_BM - 1) / GT_BM;
  int tm, tn;
  {
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = bid / per_group, rem = bid - group * per_group;
    const int gm = min(GM, tiles_m - group * GM);
    tm = group * GM + rem % gm;
    tn = rem / gm;
  }
  const int m0 = tm * GT_BM, n0 = tn * GT_BN;
  const int split = blockIdx.y;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + GT_BK - 1) / GT_BK;

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.bytesB, 0x00020000);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  ConvRows crow;
  if constexpr (CONV) crow = conv_rows(p, m0, tid);
  if constexpr (CONV) stage_load_conv(ra, rsA, p.bytesA, p, crow, kbeg, kend, tid);
  else stage_load<TA>(ra, rsA, p.bytesA, p.lda, m0, p.M, kbeg, kend, tid);
  stage_load<TB>(rb, rsB, p.bytesB, p.ldb, n0, p.N, kbeg, kend, tid);
  stage_store<TA>(ra, smem, tid);
  stage_store<TB>(rb, smem + 16384, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const char* tA = smem + (SB ? 0 : (kt & 1)) * GT_STAGE_BYTES;
    const char* tB = tA + 16384;
    const bool more = (kt + 1 < nk);
    if (more) {
      const int k0 = kbeg + (kt + 1) * GT_BK;
      if constexpr (CONV) stage_load_conv(ra, rsA, p.bytesA, p, crow, k0, kend, tid);
      else stage_load<TA>(ra, rsA, p.bytesA, p.lda, m0, p.M, k0, kend, tid);
      stage_load<TB>(rb, rsB, p.bytesB, p.ldb, n0, p.N, k0, kend, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = load_frag<TA>(tA, wm + i * 16, ks, l15, lq);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = load_frag<TB>(tB, wn + j * 16, ks, l15, lq);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mma16<F16>(fb[j], fa[i], acc[i][j]);
    }
    if (SB) __syncthreads();                     // every wave is done reading the single stage before it is overwritten
    if (more) {
      char* nA = smem + (SB ? 0 : ((kt + 1) & 1)) * GT_STAGE_BYTES;
      stage_store<TA>(ra, nA, tid);
      stage_store<TB>(rb, nA + 16384, tid);
    }
    __syncthreads();
  }

  // D[r = n][c = m]: lane holds C[m = .. + l15][n = .. + 4*lq + 0..3]
  if (p.splitk > 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) store_partial4(p, split, m0 + wm + i * 16 + l15, n0 + wn + j * 16 + 4 * lq, acc[i][j]);
    return;
  }
  // bias once per tile, residual rows one 16-row group ahead of the stores (see epilogue4v)
#define EPI_BODY(E_)                                                                                          \
  {                                                                                                           \
    f32x4 b4[4], rc[4], rn[4];                                                                                \
    bool fast[4];                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
      const int n = n0 + wn + j * 16 + 4 * lq;                                                                \
      fast[j] = p.vec_ok && (p.N - n >= 4);                                                                   \
      b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; rc[j] = b4[j]; rn[j] = b4[j];                                      \
      if (fast[j]) {                                                                                          \
        if (E_ == MV_EPI_BIAS || E_ == MV_EPI_BIAS_GELU || E_ == MV_EPI_BIAS_RES || E_ == MV_EPI_BIAS_TANH || E_ == MV_EPI_BIAS_GELU_D || E_ == MV_EPI_BIAS_RELU || E_ == MV_EPI_BIAS_RES_RELU)    \
          b4[j] = *(const f32x4*)(p.bias + n);                                                                \
        rc[j] = epi_load_res4<E_>(p, m0 + wm + l15, n);                                                       \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
      const int m = m0 + wm + i * 16 + l15;                                                                   \
      if (i + 1 < 4) {                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
          if (fast[j]) rn[j] = epi_load_res4<E_>(p, m + 16, n0 + wn + j * 16 + 4 * lq);                       \
      }                                                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
        const int n = n0 + wn + j * 16 + 4 * lq;                                                              \
        if (fast[j]) epilogue4v<E_>(p, m, n, acc[i][j], b4[j], rc[j]);                                        \
        else epilogue4_slow(p, m, n, acc[i][j]);                                                              \
      }                                                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) rc[j] = rn[j];                                            \
    }                                                                                                         \
  }
  MV_EPI_SWITCH(p.epi, EPI_BODY)
#undef EPI_BODY
}


// ------------------------------------------------------------------------------------------
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
__device__ __forceinline__ void g2_issue(__amdgpu_buffer_rsrc_t rs, unsigned bytes, int ld, int row0, int rows_total,
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
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (MV_LDS void*)(region + pc * 1024), 16, ok ? off : bytes, 0, 0, 0);
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
// Expects in scope: p, acc, scr, split, m0, n0, wm, wn, l15, lq, rrow, c4, col_on, NJ.
#define G2_RG 4
#define G2_EPI_BODY(E_)                                                                                        \
  {                                                                                                            \
    constexpr int EE = (E_) < 0 ? 0 : (E_);                                                                    \
    constexpr bool HAS_R = (E_) == MV_EPI_BIAS_RES || (E_) == MV_EPI_RES || (E_) == MV_EPI_MUL || (E_) == MV_EPI_DGELU || (E_) == MV_EPI_BIAS_RES_RELU; \
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
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) *(f32x4*)(scr + l15 * 272 + j * 64 + lq * 16) = acc[i][j]; \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                     \
          const int row = rr * 4 + rrow, mcur = m0 + wm + i * 16 + row;                                        \
          const f32x4 v = *(const f32x4*)(scr + row * 272 + c4 * 16);                                          \
          const f32x4 rc = rb[(i % G2_RG) * 4 + rr];                                                           \
          if (i + G2_RG < 8 && col_on) rb[(i % G2_RG) * 4 + rr] = epi_load_res4<EE>(p, mcur + 16 * G2_RG, ncol); \
          if (col_on) epilogue4v<EE>(p, mcur, ncol, v, b4, rc);                                                \
        }                                                                                                      \
      }                                                                                                        \
    } else {                                                                                                   \
      f32x4 rcur = {0.f, 0.f, 0.f, 0.f};                                                                       \
      if (lane_fast) rcur = epi_load_res4<EE>(p, m0 + wm + rrow, ncol);                                        \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) *(f32x4*)(scr + l15 * 272 + j * 64 + lq * 16) = acc[i][j]; \
        _Pragma("unroll 1") for (int rr = 0; rr < 4; ++rr) {                                                   \
          const int row = rr * 4 + rrow, mcur = m0 + wm + i * 16 + row;                                        \
          const int fn = i * 4 + rr + 1;                                                                       \
          f32x4 rnext = {0.f, 0.f, 0.f, 0.f};                                                                  \
          if (lane_fast && fn < 32) rnext = epi_load_res4<EE>(p, m0 + wm + (fn >> 2) * 16 + (fn & 3) * 4 + rrow, ncol); \
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
  }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool TA, bool TB, int NJ, int WN, int NSTAGE, int KS, bool F16 = false>
__global__ __launch_bounds__(128 * WN, 2) void gemm_ring_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 2 * WN;                      // waves per block
  constexpr int BN = WN * 16 * NJ;
  constexpr bool BP512 = BN > 128;                // pitch of a contraction-major B image
  constexpr int BKS = G2_BK * KS;                 // contraction depth of one stage (32 or 64)
  constexpr int A_BYTES = 16384 * KS;
  constexpr int B_BYTES = (BP512 ? 16384 : 8192) * KS;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int LPS = (A_BYTES + B_BYTES) / 1024 / NW;   // LDS-DMA instructions per wave per stage
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int wm = (wid / WN) * 128, wn = (wid % WN) * (16 * NJ);

  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, in = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + G2_BM - 1) / G2_BM;
  int tm, tn;
  {
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = bid / per_group, rem = bid - group * per_group;
    const int gm = min(GM, tiles_m - group * GM);
    tm = group * GM + rem % gm;
    tn = rem / gm;
  }
  const int m0 = tm * G2_BM, n0 = tn * BN;
  const int split = blockIdx.y;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nst = (kend - kbeg + BKS - 1) / BKS;

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.bytesB, 0x00020000);

  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define G2_ISSUE(S_)                                                                                        \
  do {                                                                                                      \
    char* st__ = smem + ((S_) % NSTAGE) * STAGE;                                                            \
    const int k0__ = kbeg + (S_) * BKS;                                                                     \
    g2_issue<TA, true, A_BYTES / 1024, NW, KS>(rsA, p.bytesA, p.lda, m0, p.M, G2_BM, k0__, kend, st__, wid, lane, p.dbg); \
    g2_issue<TB, BP512, B_BYTES / 1024, NW, KS>(rsB, p.bytesB, p.ldb, n0, p.N, BN, k0__, kend, st__ + A_BYTES, wid, lane, p.dbg); \
  } while (0)

  const bool do_load = !(p.dbg & 2), do_mma = !(p.dbg & 4);
  // Software pipeline: all NSTAGE buffers are filled up front; while the MFMAs of stage s run, the fragments of
  // stage s+1 are already being read into the second register set and stages s+2.. are in flight.  Per stage: one
  // counted vmcnt wait + one barrier (stage s+1 visible to every wave, buffer of stage s free), then the refill of
  // that buffer with stage s+NSTAGE.
  // (The second fragment set does not fit in 256 registers next to the transposed-read addresses, so the kernels
  // with a contraction-major operand keep the simpler schedule: read the fragments after the barrier, then MFMA.)
  if (do_load) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nst) G2_ISSUE(s);
  }
#define G2_WAIT(YOUNGER_)                                           \
  do {                                                              \
    const int y__ = (YOUNGER_);                                     \
    if (y__ <= 0) wait_vmcnt<0>();                                  \
    else if (y__ == 1) wait_vmcnt<LPS>();                           \
    else if (y__ == 2) wait_vmcnt<2 * LPS>();                       \
    else wait_vmcnt<3 * LPS>();                                     \
  } while (0)
#define G2_FRAGS_K(FA_, FB_, S_, KS_)                                                            \
  do {                                                                                           \
    const char* tA__ = smem + ((S_) % NSTAGE) * STAGE;                                           \
    const char* tB__ = tA__ + A_BYTES;                                                           \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) FB_[j] = g2_frag<TB, BP512, KS>(tB__, wn + j * 16, l15, lq, KS_); \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) FA_[i] = g2_frag<TA, true, KS>(tA__, wm + i * 16, l15, lq, KS_);   \
  } while (0)
#define G2_FRAGS(FA_, FB_, S_) G2_FRAGS_K(FA_, FB_, S_, 0)
#define G2_MMA(FA_, FB_)                                                                         \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                               \
        acc[i][j] = mma16<F16>(FB_[j], FA_[i], acc[i][j]);                                       \
  } while (0)
  {
    for (int s = 0; s < nst; ++s) {
      G2_WAIT(min(nst - 1 - s, NSTAGE - 2));            // stage s landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (do_load && s + NSTAGE - 1 < nst) G2_ISSUE(s + NSTAGE - 1);   // refills the buffer everyone finished reading
      if (!do_mma) continue;
      if constexpr (KS == 2 && !TA && !TB) {
        // all 24 fragment reads of the 64-deep stage are issued before its first MFMA: the MFMAs then wait on a
        // counted lgkmcnt that only the first reads hold up, instead of a read-wait-MFMA ping-pong per 2 fragments
        bf16x8 fa0[8], fb0[NJ], fa1[8], fb1[NJ];
        G2_FRAGS_K(fa0, fb0, s, 0);
        G2_FRAGS_K(fa1, fb1, s, 1);
        __builtin_amdgcn_sched_barrier(0);
        G2_MMA(fa0, fb0);
        G2_MMA(fa1, fb1);
      } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 fa[8], fb[NJ];
          G2_FRAGS_K(fa, fb, s, ks);
          G2_MMA(fa, fb);
        }
      }
    }
  }
#undef G2_MMA
#undef G2_FRAGS
#undef G2_FRAGS_K
#undef G2_WAIT
#undef G2_ISSUE

  if (p.dbg & 1) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) ((float*)p.C)[0] = t;     // keeps the accumulators live without storing the tile
    return;
  }
  // Epilogue through LDS: an accumulator tile holds 4 columns x 16 rows per lane, which would store as sixteen
  // 32-byte fragments per instruction (measured: ~1 TB/s).  Each wave transposes 16 rows at a time through its own
  // 4.25-KiB scratch (272-B row pitch: conflict-free both ways) so that 16 lanes cover one full output row:
  // whole 128/256-byte lines per store, and coalesced bias / residual loads in the fused epilogue.
  __builtin_amdgcn_s_barrier();             // every wave is done with the operand ring before it becomes scratch
  char* scr = smem + wid * 4608;
  const int rrow = lane >> 4, c4 = lane & 15;
  const bool col_on = (c4 * 4) < 16 * NJ;
  if (p.splitk > 1) { G2_EPI_BODY(-1) return; }
  MV_EPI_SWITCH(p.epi, G2_EPI_BODY)
}

// ------------------------------------------------------------------------------------------
// Persistent form of the ring kernel with 64-deep stages: one block per CU walks its share of the (tile, K-slice)
// units, and the operand ring never drains between them -- the first stage(s) of the next unit are issued during the
// last K-tile of the current one and land while the epilogue runs.  The epilogue's stores are not waited for either:
// the first wait of the next unit is a COUNTED vmcnt that only requires the ring stage (older than the stores) to be
// complete (gfx9 vmcnt retires loads and stores in issue order), so a tile's 128 KiB of output drains to HBM under
// the next tile's MFMAs instead of in a chip-wide burst at the end of every round of tiles.
#undef G2_RG
#define G2_RG 2   // the persistent kernel keeps its issue cursor live across the epilogue: fewer registers to spare
template <bool TA, bool TB, int NJ, int WN, int NSTAGE>
__global__ __launch_bounds__(128 * WN, 1) void gemm_pring_kernel(GemmArgs p, int units, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 2;
  constexpr int NW = 2 * WN;
  constexpr int BN = WN * 16 * NJ;
  constexpr bool BP512 = BN > 128;
  constexpr int BKS = G2_BK * KS;
  constexpr int A_BYTES = 16384 * KS;
  constexpr int B_BYTES = (BP512 ? 16384 : 8192) * KS;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int LPS = STAGE / 1024 / NW;
  constexpr int EPI_OPS = 28;                     // lower bound of the VMEM ops a wave issues in a full-tile epilogue (32 stores)
  static_assert(NW * 4608 <= STAGE, "epilogue scratch must fit in one ring stage");
  static_assert((NSTAGE - 2) * LPS + EPI_OPS < 64, "vmcnt is a 6-bit counter");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int wm = (wid / WN) * 128, wn = (wid % WN) * (16 * NJ);
  const int G = gridDim.x;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + G2_BM - 1) / G2_BM;

  // unit -> (tile origin, K-slice).  Units that run at the same time on one XCD (blocks b, b+8, ... share an L2) are
  // neighbours in the grouped raster: 8 row-panels x consecutive column-panels.
  auto decode = [&](int u, int& m0, int& n0, int& kbeg, int& kend, int& split) {
    const int q = units >> 3, r = units & 7, xcd = u & 7, in = u >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
    split = v / tiles;
    const int bid = v - split * tiles;
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = bid / per_group, rem = bid - group * per_group;
    const int gm = min(GM, tiles_m - group * GM);
    m0 = (group * GM + rem % gm) * G2_BM;
    n0 = (rem / gm) * BN;
    kbeg = split * p.kchunk;
    kend = min(p.K, kbeg + p.kchunk);
  };

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.bytesB, 0x00020000);

  // issue cursor: runs NSTAGE-1 stages ahead of the compute cursor, across unit boundaries
  int iu = blockIdx.x, is = 0, im0 = 0, in0 = 0, ikbeg = 0, ikend = 0, isplit = 0, inst = 0;
  unsigned ifs = 0, cfs = 0;                      // flat stage counters (ring slot = counter % NSTAGE)
  if (iu < units) { decode(iu, im0, in0, ikbeg, ikend, isplit); inst = (ikend - ikbeg + BKS - 1) / BKS; }
  auto issue_one = [&]() {
    if (iu >= units) return;
    char* st = smem + (ifs % NSTAGE) * STAGE;
    const int k0 = ikbeg + is * BKS;
    g2_issue<TA, true, A_BYTES / 1024, NW, KS>(rsA, p.bytesA, p.lda, im0, p.M, G2_BM, k0, ikend, st, wid, lane);
    g2_issue<TB, BP512, B_BYTES / 1024, NW, KS>(rsB, p.bytesB, p.ldb, in0, p.N, BN, k0, ikend, st + A_BYTES, wid, lane);
    ++ifs;
    if (++is == inst) {
      iu += G; is = 0;
      if (iu < units) { decode(iu, im0, in0, ikbeg, ikend, isplit); inst = (ikend - ikbeg + BKS - 1) / BKS; }
    }
  };
#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i) issue_one();

  int epi_ops = 0;                                // VMEM ops this wave is known to have issued after its last ring load
  for (int cu = blockIdx.x; cu < units; cu += G) {
    int m0, n0, kbeg, kend, split;
    decode(cu, m0, n0, kbeg, kend, split);
    const int nst = (kend - kbeg + BKS - 1) / BKS;
    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int s = 0; s < nst; ++s) {
      // stage cfs must have landed: everything older than the (ifs - cfs - 1) younger stages and, right after an
      // epilogue, older than its stores
      const int younger = (int)(ifs - cfs) - 1;
      const bool after_epi = (s == 0) && epi_ops > 0;
      if (after_epi) {
        if (NSTAGE > 2 && younger >= 1) wait_vmcnt<(NSTAGE > 2 ? LPS : 0) + EPI_OPS>();
        else wait_vmcnt<EPI_OPS>();
      } else {
        if (NSTAGE > 2 && younger >= 1) wait_vmcnt<(NSTAGE > 2 ? LPS : 0)>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue_one();                                // refills the slot everyone finished reading (or used as scratch)
      const char* tA = smem + (cfs % NSTAGE) * STAGE;
      const char* tB = tA + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 fa[8], fb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = g2_frag<TB, BP512, KS>(tB, wn + j * 16, l15, lq, ks);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = g2_frag<TA, true, KS>(tA, wm + i * 16, l15, lq, ks);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
      ++cfs;
    }

    if (p.dbg & 1) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (t == 123.456f) ((float*)p.C)[0] = t;
      epi_ops = 0;
      continue;
    }
    // epilogue through the ring slot of the stage just consumed (see gemm_ring_kernel): 16 rows at a time per wave
    __builtin_amdgcn_s_barrier();
    char* scr = smem + ((cfs + NSTAGE - 1) % NSTAGE) * STAGE + wid * 4608;
    const int rrow = lane >> 4, c4 = lane & 15;
    const bool col_on = (c4 * 4) < 16 * NJ;
    if (p.splitk > 1) { G2_EPI_BODY(-1) }
    else { MV_EPI_SWITCH(p.epi, G2_EPI_BODY) }
    // whole tile inside the matrix and vector stores: every one of the 32 row-group stores above was issued
    const bool full = (m0 + G2_BM <= p.M) && (n0 + BN <= p.N) && ((p.N & 3) == 0) && (p.splitk > 1 || p.vec_ok);
    epi_ops = full ? EPI_OPS : 0;
  }
}

// ------------------------------------------------------------------------------------------
// plain VALU kernel (exact fp32 path + cross-check)
template <typename T>
__global__ __launch_bounds__(256) void gemm_simple_kernel(GemmArgs p, int ta, int tb) {
  __shared__ float As[16][65];
  __shared__ float Bs[16][65];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int split = blockIdx.z;
  const int kbeg = split * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const T* A = (const T*)p.A;
  const T* B = (const T*)p.B;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      int mm, kk;
      if (ta) { mm = idx & 63; kk = idx >> 6; } else { kk = idx & 15; mm = idx >> 4; }
      const int gm = m0 + mm, gk = k0 + kk;
      float v = 0.f;
      if (gm < p.M && gk < kend) v = ta ? ldf<T>(A + (size_t)gk * p.lda + gm) : ldf<T>(A + (size_t)gm * p.lda + gk);
      As[kk][mm] = v;
      int nn, kb;
      if (tb) { nn = idx & 63; kb = idx >> 6; } else { kb = idx & 15; nn = idx >> 4; }
      const int gn = n0 + nn, gkb = k0 + kb;
      float w = 0.f;
      if (gn < p.N && gkb < kend) w = tb ? ldf<T>(B + (size_t)gkb * p.ldb + gn) : ldf<T>(B + (size_t)gn * p.ldb + gkb);
      Bs[kb][nn] = w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i, n = n0 + tx * 4;
    f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    if (p.splitk > 1) store_partial4(p, split, m, n, v);
    else epilogue4_slow(p, m, n, v);
  }
}

__global__ void splitk_reduce_kernel(GemmArgs p) {
  const size_t total4 = ((size_t)p.M * p.N + 3) / 4;
  const size_t mn = (size_t)p.M * p.N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4;
    if ((p.N & 3) == 0) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < p.splitk; ++k) s += *(const f32x4*)(p.ws + k * mn + e);
      const int m = (int)(e / p.N), n = (int)(e - (size_t)m * p.N);
      epilogue4_slow(p, m, n, s);
    } else {
      for (int j = 0; j < 4 && e + j < mn; ++j) {
        float s = 0.f;
        for (int k = 0; k < p.splitk; ++k) s += p.ws[k * mn + e + j];
        const int m = (int)((e + j) / p.N), n = (int)((e + j) - (size_t)m * p.N);
        float* c = (float*)p.C + (size_t)m * p.ldc + n;
        *c = p.accumulate ? *c + s : s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
int g_mv_gemm_force = 0;   // test hook: 0 auto, 1 force the 128x128 kernel, 2 force the 256-row kernel
int g_mv_gemm_nj = 0;      // test hook: 0 auto, 3 / 4 force the 192- / 256-column variant
int g_mv_gemm_dbg = 0;
extern "C" void mv_set_gemm_variant(int force, int nj) { g_mv_gemm_force = force & 0xff; g_mv_gemm_nj = nj; g_mv_gemm_dbg = force >> 8; }

static inline bool aligned_to(const void* p, size_t a) { return p == nullptr || (((uintptr_t)p) % a) == 0; }

extern "C" int mv_gemm(int dtype, int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                       void* C, int ldc, int c_dtype, const float* bias, int epi, const void* R, int ldr, int r_dtype,
                       void* C2, int ldc2, void* C3, int ldc3, int c3_dtype, int splitk, float* ws, size_t ws_bytes,
                       int accumulate, float p_drop, unsigned long long drop_key, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MV_E_ARG;
  if (!mv_dtype_ok(dtype) || !mv_dtype_ok(c_dtype)) return MV_E_DTYPE;
  if (C3 && (!mv_is16(c3_dtype) || ldc3 < N)) return MV_E_DTYPE;
  if (epi < 0 || epi > MV_EPI_BIAS_RES_RELU) return MV_E_ARG;
  const bool need_bias = (epi == MV_EPI_BIAS || epi == MV_EPI_BIAS_GELU || epi == MV_EPI_BIAS_RES || epi == MV_EPI_BIAS_TANH || epi == MV_EPI_BIAS_GELU_D || epi == MV_EPI_BIAS_RELU || epi == MV_EPI_BIAS_RES_RELU);
  const bool need_r = (epi == MV_EPI_BIAS_RES || epi == MV_EPI_DGELU || epi == MV_EPI_RES || epi == MV_EPI_MUL || epi == MV_EPI_BIAS_RES_RELU);
  if (need_bias && !bias) return MV_E_ARG;
  if (need_r && (!R || !mv_dtype_ok(r_dtype))) return MV_E_ARG;
  if ((epi == MV_EPI_BIAS_GELU || epi == MV_EPI_BIAS_GELU_D) && !C2) return MV_E_ARG;
  if (lda < (ta ? M : K) || ldb < (tb ? N : K) || ldc < N) return MV_E_SHAPE;
  if (need_r && ldr < N) return MV_E_SHAPE;
  if (splitk < 0) splitk = 1;
  if (splitk > 1 || accumulate) {
    if (epi != MV_EPI_NONE || c_dtype != MV_F32 || C3) return MV_E_SHAPE;
  }
  if (splitk > 1) {
    if (!ws || ws_bytes < (size_t)splitk * M * N * sizeof(float)) return MV_E_WORKSPACE;
  }
  if (splitk == 0 && (!mv_is16(dtype) || g_mv_impl != 0)) splitk = 1;   // auto split-K only on the MFMA kernels
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.C2 = C2; p.bias = bias; p.R = R; p.C3 = C3;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldr = ldr; p.ldc3 = ldc3;
  p.c_dtype = c_dtype; p.r_dtype = r_dtype; p.c3_dtype = c3_dtype; p.epi = epi; p.accumulate = accumulate;
  p.splitk = splitk; p.ws = ws; p.dbg = g_mv_gemm_dbg;
  p.drop = mv_make_drop(epi == MV_EPI_BIAS_RES ? p_drop : 0.f, drop_key);
  if (p.drop.thr && (N & 3)) return MV_E_SHAPE;   // the mask is keyed on groups of 4 consecutive columns
  const size_t csz = (c_dtype == MV_F32) ? 16 : 8;
  const size_t rsz = (r_dtype == MV_F32) ? 16 : 8;
  p.vec_ok = ((ldc & 3) == 0) && aligned_to(C, csz) && (!C3 || (((ldc3 & 3) == 0) && aligned_to(C3, 8))) &&
             (!need_bias || aligned_to(bias, 16)) &&
             (!need_r || (((ldr & 3) == 0) && aligned_to(R, rsz))) &&
             ((epi != MV_EPI_BIAS_GELU && epi != MV_EPI_BIAS_GELU_D) || (((ldc2 & 3) == 0) && aligned_to(C2, csz)));
  const bool mfma = mv_is16(dtype) && (g_mv_impl == 0);
  const bool f16 = dtype == MV_F16;
  if (mfma && f16 && (ta || tb)) return MV_E_DTYPE;   // f16 operands exist for the forward form y = x.W^T only
  if (mfma) {
    if ((lda & 7) || (ldb & 7) || !aligned_to(A, 16) || !aligned_to(B, 16)) return MV_E_SHAPE;
    const size_t bytesA = ((size_t)((ta ? K : M) - 1) * lda + (size_t)(((ta ? M : K) + 7) & ~7)) * 2;
    const size_t bytesB = ((size_t)((tb ? K : N) - 1) * ldb + (size_t)(((tb ? N : K) + 7) & ~7)) * 2;
    if (bytesA >= 0x7fffffffULL || bytesB >= 0x7fffffffULL) return MV_E_SHAPE;
    p.bytesA = (unsigned)bytesA; p.bytesB = (unsigned)bytesB;
    // tile choice: a 256-row ring kernel when it fills the chip, in the column width (256 / 192 / 128) that needs
    // the least MFMA time over whole rounds of CUs; the 128x128 kernel for small problems
    const int tm2 = (M + 255) / 256;
    const long long t256 = (long long)tm2 * ((N + 255) / 256), t192 = (long long)tm2 * ((N + 191) / 192),
                    t128 = (long long)tm2 * ((N + 127) / 128);
    // measured on the model's shapes (profiles/r01_gemm_variants.txt): the ring kernels win for y = x.W^T and
    // dW = dy^T.x, the 128x128 register-staged kernel for dx = dy.W
    // (ring kernel with 64-deep stages for wide outputs and for dW; 128x128 register-staged for N <= 1024 and dX)
    // (768-column outputs: the 128x128 kernel at three blocks per CU wins in every layout, also for long contractions)
    const bool wide_nt = !ta && !tb && N >= 1024;
    const bool big = (g_mv_gemm_force == 2) || (g_mv_gemm_force == 0 && M >= 256 && N >= 128 && ((K & 7) == 0 || (ta && tb)) && (wide_nt || ta) &&
                                               (t128 >= 128 || (K >= 4096 && splitk != 1)));
    if (big) {
      long long sk = splitk;
      if (splitk > 1 || splitk == 0) {      // 0 = auto
        const long long tiles = t256;
        sk = 1;
        if (tiles < 256 && K >= 2048) { sk = 256 / tiles; if (sk > K / 1024) sk = K / 1024; if (sk > 16) sk = 16; if (sk < 1) sk = 1; }
        if (splitk > 1 && sk > splitk) sk = splitk;
        if (sk > 1 && (!ws || ws_bytes < (size_t)sk * M * N * sizeof(float) || epi != MV_EPI_NONE || c_dtype != MV_F32)) sk = 1;
      }
      // cost ~ rounds x tile width (1 block / CU for the 8-wave tiles, 2 blocks / CU for 256x128)
      const long long r256 = (t256 * sk + 255) / 256 * 256, r192 = (t192 * sk + 255) / 256 * 192,
                      r128 = (t128 * sk + 511) / 512 * 256;
      int variant = g_mv_gemm_nj;           // 4: 256x256, 3: 256x192, 2: 256x128
      if (variant == 0) {
        // 256x256 with 64-deep stages (whole 128-B lines per LDS-DMA row): best measured.  Weight gradients (split-K
        // units, f32 partial tiles) gain 5-8 % from the persistent form; y = x.W^T does not (profiles/r01_gemm_variants.txt)
        variant = ta ? 24 : 14;
        (void)r256; (void)r192; (void)r128;
      }
      int kchunk = (int)((K + sk - 1) / sk);
      kchunk = (kchunk + 63) / 64 * 64;
      p.kchunk = kchunk;
      p.splitk = splitk = (K + kchunk - 1) / kchunk;
      // variants: 4 = 256x256 (32-deep stages x4), 3 = 256x192, 2 = 256x128 (x3, 2 blocks/CU),
      //           14 = 256x256 with 64-deep stages x2 (128-B lines), 15 = 256x192 likewise, 12 / 13 = 256x128 with
      //           64-deep stages x2 / x3
      const int tiles = (int)t256;
      dim3 grid(tiles, splitk);
      static int n_cu = 0;
      if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
      }
#define LAUNCH_PRING(TA_, TB_, NJ_, WN_, NS_)                                                                        \
  do {                                                                                                               \
    constexpr size_t shm = (size_t)(NS_) * 2 * (16384 + ((WN_) * 16 * (NJ_) > 128 ? 16384 : 8192));                  \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_pring_kernel<TA_, TB_, NJ_, WN_, NS_>,                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                               \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    const int units = tiles * splitk;                                                                                \
    hipLaunchKernelGGL((gemm_pring_kernel<TA_, TB_, NJ_, WN_, NS_>), dim3(units < n_cu ? units : n_cu),              \
                       dim3(128 * (WN_)), shm, stream, p, units, tiles);                                             \
  } while (0)
#define LAUNCH_RING(TA_, TB_, NJ_, WN_, NS_, KS_)                                                                    \
  do {                                                                                                               \
    constexpr size_t shm = (size_t)(NS_) * (KS_) * (16384 + ((WN_) * 16 * (NJ_) > 128 ? 16384 : 8192));               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<TA_, TB_, NJ_, WN_, NS_, KS_>,                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                               \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_ring_kernel<TA_, TB_, NJ_, WN_, NS_, KS_>), grid, dim3(128 * (WN_)), shm, stream, p);    \
  } while (0)
#define LAUNCH_RING_V(TA_, TB_)                                  \
  do {                                                           \
    if (variant == 24) LAUNCH_PRING(TA_, TB_, 4, 4, 2);          \
    else LAUNCH_RING(TA_, TB_, 4, 4, 2, 2);                      \
  } while (0)
      if (!ta && !tb) {
        if (f16) {
          constexpr size_t shm = (size_t)2 * 2 * (16384 + 16384);
          static bool attr_set = false;
          if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<false, false, 4, 4, 2, 2, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            attr_set = true;
          }
          hipLaunchKernelGGL((gemm_ring_kernel<false, false, 4, 4, 2, 2, true>), grid, dim3(512), shm, stream, p);
        } else LAUNCH_RING_V(false, false);
      }
      else if (!ta && tb) LAUNCH_RING_V(false, true);
      else if (ta && tb) LAUNCH_RING_V(true, true);
      else LAUNCH_RING_V(true, false);
#undef LAUNCH_RING_V
#undef LAUNCH_RING
#undef LAUNCH_PRING
    } else {
      if (splitk == 0) {
        const int tiles = ((M + GT_BM - 1) / GT_BM) * ((N + GT_BN - 1) / GT_BN);
        splitk = 1;
        if (tiles < 512 && K >= 2048 && ws && epi == MV_EPI_NONE && c_dtype == MV_F32) {
          splitk = 768 / tiles; if (splitk > K / 1024) splitk = K / 1024; if (splitk > 16) splitk = 16; if (splitk < 1) splitk = 1;
          if (ws_bytes < (size_t)splitk * M * N * sizeof(float)) splitk = 1;
        }
      }
      int kchunk = (K + splitk - 1) / splitk;
      kchunk = (kchunk + GT_BK - 1) / GT_BK * GT_BK;
      p.kchunk = kchunk;
      p.splitk = splitk = (K + kchunk - 1) / kchunk;
      const int tiles = ((M + GT_BM - 1) / GT_BM) * ((N + GT_BN - 1) / GT_BN);
      dim3 grid(tiles, splitk), block(256);
      const size_t shm = 2 * GT_STAGE_BYTES;
#define LAUNCH_MFMA(TA_, TB_)                                                                              \
  do {                                                                                                     \
    static bool attr_set = false;                                                                          \
    if (!attr_set) {                                                                                       \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<TA_, TB_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)shm);                                                                 \
      attr_set = true;                                                                                     \
    }                                                                                                      \
    hipLaunchKernelGGL((gemm_mfma_kernel<TA_, TB_>), grid, block, shm, stream, p);                          \
  } while (0)
#define LAUNCH_MFMA_SB(TA_, TB_)                                                                           \
  do {                                                                                                     \
    static bool attr_sb = false;                                                                           \
    if (!attr_sb) {                                                                                        \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<TA_, TB_, false, true>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, GT_STAGE_BYTES);               \
      attr_sb = true;                                                                                      \
    }                                                                                                      \
    hipLaunchKernelGGL((gemm_mfma_kernel<TA_, TB_, false, true>), grid, block, GT_STAGE_BYTES, stream, p); \
  } while (0)
      // one LDS stage and three blocks per CU by default (10-15 % faster on the model's 768-column GEMMs at ~25k rows:
      // profiles/r01_gemm_variants.txt); the two-stage form stays reachable for cross-checks (mv_set_gemm_variant(., 32))
      const bool sb = g_mv_gemm_nj != 32;
      if (f16) {
        static bool attr_h = false;
        if (!attr_h) {
          (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<false, false, false, true, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GT_STAGE_BYTES);
          attr_h = true;
        }
        hipLaunchKernelGGL((gemm_mfma_kernel<false, false, false, true, true>), grid, block, GT_STAGE_BYTES, stream, p);
      }
      else if (!ta && !tb) { if (sb) LAUNCH_MFMA_SB(false, false); else LAUNCH_MFMA(false, false); }
      else if (!ta && tb) { if (sb) LAUNCH_MFMA_SB(false, true); else LAUNCH_MFMA(false, true); }
      else if (ta && tb) { if (sb) LAUNCH_MFMA_SB(true, true); else LAUNCH_MFMA(true, true); }
      else { if (sb) LAUNCH_MFMA_SB(true, false); else LAUNCH_MFMA(true, false); }
#undef LAUNCH_MFMA_SB
#undef LAUNCH_MFMA
    }
  } else {
    int kchunk = (K + splitk - 1) / splitk;
    kchunk = (kchunk + 15) / 16 * 16;
    p.kchunk = kchunk;
    p.splitk = splitk = (K + kchunk - 1) / kchunk;
    dim3 grid((N + 63) / 64, (M + 63) / 64, splitk), block(256);
    if (grid.y > 65535) return MV_E_SHAPE;
    if (dtype == MV_F32) hipLaunchKernelGGL(gemm_simple_kernel<float>, grid, block, 0, stream, p, ta, tb);
    else if (dtype == MV_F16) hipLaunchKernelGGL(gemm_simple_kernel<f16_t>, grid, block, 0, stream, p, ta, tb);
    else hipLaunchKernelGGL(gemm_simple_kernel<bf16_t>, grid, block, 0, stream, p, ta, tb);
  }
  MV_CHECK_LAUNCH();
  if (splitk > 1) {
    const size_t total4 = ((size_t)M * N + 3) / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    MV_CHECK_LAUNCH();
  }
  return MV_OK;
}

// ------------------------------------------------------------------------------------------
// Convolution as an implicit GEMM: y[(b,oy,ox), o] = sum_{ky,kx,c} x[b, oy*s-pad+ky, ox*s-pad+kx, c] * w[o, (ky*kw+kx)*C + c].
// The 128x128 kernel gathers the activation operand tap by tap while staging it (8 channels = 16 bytes per lane), so the
// [rows, kh*kw*C] patch matrix of mv_im2col is never written.  Replaces torch's conv2d inside the ResNet-50 trunk of
// models/image.py:46-55 (torchvision Bottleneck 3x3 / strided 1x1 convolutions and the 7x7 stem).
extern "C" int mv_conv2d(int dtype, const void* x, const void* w, void* y, int y_dtype, int B, int H, int W, int C, int O, int kh, int kw,
                         int stride, int pad, const float* bias, int epi, const void* R, int r_dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || O <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return MV_E_ARG;
  if (dtype != MV_BF16 || g_mv_impl != 0) return MV_E_DTYPE;              // MFMA path only; other cases: mv_im2col + mv_gemm
  if (y_dtype != MV_F32 && y_dtype != MV_BF16) return MV_E_DTYPE;
  if (epi != MV_EPI_NONE && epi != MV_EPI_BIAS && epi != MV_EPI_BIAS_RELU && epi != MV_EPI_BIAS_RES_RELU) return MV_E_ARG;
  if (epi != MV_EPI_NONE && !bias) return MV_E_ARG;
  if (epi == MV_EPI_BIAS_RES_RELU && (!R || (r_dtype != MV_F32 && r_dtype != MV_BF16))) return MV_E_ARG;
  if ((C & 7) || (C & (C - 1)) || (O & 3)) return MV_E_SHAPE;             // 16-byte channel groups, power-of-two C, vector stores
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return MV_E_SHAPE;
  const long long rows = (long long)B * Ho * Wo;
  const size_t bytesA = (size_t)B * H * W * C * 2, bytesB = (size_t)O * kh * kw * C * 2;
  if (rows > 0x7fffffffLL || bytesA >= 0x7fffffffULL || bytesB >= 0x7fffffffULL) return MV_E_SHAPE;
  if ((((uintptr_t)x) | ((uintptr_t)w)) & 15) return MV_E_SHAPE;
  GemmArgs p{};
  p.A = x; p.B = w; p.C = y;
  p.M = (int)rows; p.N = O; p.K = kh * kw * C; p.lda = p.K; p.ldb = p.K; p.ldc = O;
  p.bias = bias; p.R = R; p.r_dtype = r_dtype; p.ldr = O;
  p.c_dtype = y_dtype; p.epi = epi; p.splitk = 1; p.kchunk = (p.K + GT_BK - 1) / GT_BK * GT_BK;
  p.bytesA = (unsigned)bytesA; p.bytesB = (unsigned)bytesB;
  p.vec_ok = (((uintptr_t)y) % (y_dtype == MV_F32 ? 16 : 8)) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0) &&
             (!R || (((uintptr_t)R) % (r_dtype == MV_F32 ? 16 : 8)) == 0);
  p.drop = mv_make_drop(0.f, 0);
  p.cvH = H; p.cvW = W; p.cvC = C; p.cvKw = kw; p.cvStride = stride; p.cvPad = pad; p.cvHo = Ho; p.cvWo = Wo;
  p.cvCshift = 0;
  while ((1 << p.cvCshift) < C) ++p.cvCshift;
  const long long tiles = ((rows + GT_BM - 1) / GT_BM) * ((O + GT_BN - 1) / GT_BN);
  if (tiles > 0x7fffffffLL) return MV_E_SHAPE;
  const size_t shm = GT_STAGE_BYTES;           // one LDS stage, three blocks per CU (see gemm_mfma_kernel)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_mfma_kernel<false, false, true, true>), dim3((unsigned)tiles, 1), dim3(256), shm, stream, p);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
