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
#include "mv_gemm_common.h"

// ------------------------------------------------------------------------------------------
// MFMA kernel
#define GT_BM 128
#define GT_BN 128
#define GT_BK 64
#define GT_STAGE_BYTES 32768   // A tile 16 KiB + B tile 16 KiB

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
    constexpr int EE = (E_);                                                                                  \
    constexpr bool WIDE_E = (E_) == MV_EPI_NONE || (E_) == MV_EPI_BIAS || (E_) == MV_EPI_BIAS_GELU_D;         \
    constexpr bool WIDE_R = (E_) == MV_EPI_MUL || (E_) == MV_EPI_RES || (E_) == MV_EPI_BIAS_RES;              \
    constexpr int NJ = 4;                                                                                     \
    if (!CONV && G2_WIDE_COND(NJ)) {                                                                          \
      /* 16-bit output, whole 64-column strip: 16-byte stores of whole lines through the wave's LDS scratch (the stage is free: the K     \
         loop ended with a barrier) instead of 8-byte pieces of 16 different rows per instruction */         \
      char* scr = smem + wid * 4608;                                                                          \
      G2_EPI_WIDE(E_, 4)                                                                                      \
    } else {                                                                                                  \
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
    }                                                                                                         \
  }
  MV_EPI_SWITCH(p.epi, EPI_BODY)
#undef EPI_BODY
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
      if (p.epi == MV_EPI_NONE && p.c_dtype == MV_F32 && !p.C3 && p.vec_ok && (p.ldc & 3) == 0) {
        // the weight gradients (51 of these launches per step): one 16-byte load / store of C instead of four scalar ones
        float* c = (float*)p.C + (size_t)m * p.ldc + n;
        if (p.alpha) s *= *p.alpha;
        if (p.accumulate) s += *(const f32x4*)c;
        *(f32x4*)c = s;
      } else {
        epilogue4_slow(p, m, n, s);
      }
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
// test / experiment hooks (constants in the product library: mv_common.h)
#define g_mv_gemm_force (mv_knob(MV_KNOB_GEMM_FORCE))        // 0 auto, 1 force the 128x128 kernel, 2 force the 256-row kernel
#define g_mv_gemm_nj (mv_knob(MV_KNOB_GEMM_NJ))              // 0 auto, kernel variant
#define g_mv_gemm_dbg (mv_knob(MV_KNOB_GEMM_DBG))
#define g_mv_persistent_cus (mv_knob(MV_KNOB_PERSISTENT_CUS))

static inline bool aligned_to(const void* p, size_t a) { return p == nullptr || (((uintptr_t)p) % a) == 0; }

// Which kernel serves a 16-bit MFMA product and how many split-K slabs it would like when the caller leaves the choice to the library
// (splitk = 0) -- ONE definition, used by mv_gemm itself and by mv_gemm_workspace_bytes / mv_workspace_bytes, so that a host that is not
// hip_ops.py can size the workspace without re-deriving this from the source.
struct GemmRoute {
  bool big;            // the 256-row ring kernels (else the 128x128 kernel)
  int variant;         // ring variant: 14 = 256x256x64 two stages, 24 = its persistent form, 2 = 256x128, 10 = 320x256 (one round)
  long long tiles;     // output tiles of the chosen kernel
  long long sk_auto;   // split-K slabs wanted at splitk = 0 with an unlimited workspace (1 = none)
};
static int gemm_n_cu() {
  static const int n = [] {
    int dev = 0, v = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
    return v > 0 ? v : 256;
  }();
  return n;
}
static GemmRoute gemm_route(int ta, int tb, int M, int N, int K, int splitk, bool rows256 = false) {      // rows256: the caller needs 256-row tiles (fused column sums)
  GemmRoute r;
  const int tm2 = (M + 255) / 256;
  const long long t256 = (long long)tm2 * ((N + 255) / 256), t128 = (long long)tm2 * ((N + 127) / 128);
  // measured on the model's shapes (profiles/r01_gemm_variants.txt): the ring kernels win for y = x.W^T with wide outputs and for
  // dW = dy^T.x, the 128x128 register-staged kernel for dx = dy.W and for 768-column outputs (three blocks per CU), also for long contractions
  const bool wide_nt = !ta && !tb && N >= 1024;
  r.big = (g_mv_gemm_force == 2) || (g_mv_gemm_force == 0 && M >= 256 && N >= 128 && ((K & 7) == 0 || (ta && tb)) && (wide_nt || ta) &&
                                     (t128 >= 128 || (K >= 4096 && splitk != 1)));
  r.variant = 0;
  r.sk_auto = 1;
  // WHOLE ROUNDS of tiles (round 5).  y = x.W^T and dx = dy.W with narrow outputs (768 columns: attention output projection, FFN-down and
  // the three input gradients of a layer) over ~25,500 packed rows are 300 tiles of 256 x 256 -- 1.17 rounds of the 256 CUs, the second one
  // 44 tiles on an idle chip -- and 1,200 tiles of 128 x 128 = 1.56 rounds of that kernel's 768 slots; as 320 x 256 tiles they are 240 tiles:
  // ONE round with 94 % of the CUs busy.  That is wave quantisation, not kernel quality (a tile takes the same time whether 44 or 256 CUs
  // are busy), so the tile shape is chosen per call by (whole rounds) x (time of one round of that kernel).
  // Measured at 25,483 rows (profiles/r05_notes.txt): FFN-down 151 -> 112 us, da 156 -> 113, dx(qkv) 117 -> 87, Wo 50 -> 42, dctx 44 -> 34.
  const bool rounds_on = mv_knob(MV_KNOB_GEMM_ROUNDS) != 0 && g_mv_gemm_force == 0 && g_mv_gemm_nj == 0;
  if (rounds_on && !ta && !r.big && splitk <= 1 && M >= 2048 && N >= 256 && (N & 7) == 0 && (K & 7) == 0 && K >= 256) {
    // whole rounds x the measured time of one round (any K: the three kernels' rounds scale alike): 128 x 128 tiles on 768 slots 75,
    // 256 x 256 ring tiles on 256 CUs 85, 320 x 256 ring tiles 112 (FFN-down shape, us).  A partly filled round costs a full one.
    const int n_cu = gemm_n_cu();
    const long long tn = (N + 255) / 256, t320 = (long long)((M + 319) / 320) * tn;
    const long long s128 = (long long)((M + GT_BM - 1) / GT_BM) * ((N + GT_BN - 1) / GT_BN);
    const long long c128 = ((s128 + 3 * n_cu - 1) / (3 * n_cu)) * 75, c256 = ((t256 + n_cu - 1) / n_cu) * 85,
                    c320 = rows256 ? (1ll << 60) : ((t320 + n_cu - 1) / n_cu) * 112;
    if (c256 < c128 && c256 <= c320) { r.big = true; r.variant = 14; r.tiles = t256; return r; }
    if (c320 < c128 && c320 < c256) { r.big = true; r.variant = 10; r.tiles = t320; return r; }
  }
  // Wide y = x.W^T outputs run several rounds of tiles; the last one is partly empty.  320-row tiles when they take strictly less
  // (rounds x rows): the fused QKV projection at 25,483 rows is 900 tiles of 256 rows = 4 rounds (3.52 full) or 720 of 320 = 3 rounds.
  if (rounds_on && r.big && !ta && !tb && !rows256 && splitk <= 1 && (K & 7) == 0) {
    const int n_cu = gemm_n_cu();
    const long long tn = (N + 255) / 256, t320 = (long long)((M + 319) / 320) * tn;
    const long long c256 = ((t256 + n_cu - 1) / n_cu) * 8, c320 = ((t320 + n_cu - 1) / n_cu) * 10;
    if (c320 < c256) { r.variant = 10; r.tiles = t320; return r; }
  }
  if (r.big) {
    // 256x256 with 64-deep stages (whole 128-B lines per LDS-DMA row): best measured.  Weight gradients (split-K units, f32 partial
    // tiles) gain 5-8 % from the persistent form; y = x.W^T does not (profiles/r01_gemm_variants.txt)
    r.variant = g_mv_gemm_nj ? g_mv_gemm_nj : (ta ? 24 : 14);        // (knob: 14, 24, 10 = 320-row tiles or 2 = 256x128 tiles, 4 waves, two blocks per CU)
    if (r.variant == 10 && ta) r.variant = 24;                      // the 320-row form exists for y = x.W^T and dx = dy.W
    const bool v128 = r.variant == 2;
    r.tiles = v128 ? t128 : (r.variant == 10 ? (long long)((M + 319) / 320) * ((N + 255) / 256) : t256);
    const long long slots = v128 ? 512 : 256;
    // enough slabs to give every CU a unit, each at least 1024 deep; at most 32
    if (r.variant != 10 && r.tiles < slots && K >= 2048) { long long sk = slots / r.tiles; if (sk > K / 1024) sk = K / 1024; if (sk > 32) sk = 32; if (sk < 1) sk = 1; r.sk_auto = sk; }
  } else {
    r.tiles = (long long)((M + GT_BM - 1) / GT_BM) * ((N + GT_BN - 1) / GT_BN);
    if (r.tiles < 512 && K >= 2048) { long long sk = 768 / r.tiles; if (sk > K / 1024) sk = K / 1024; if (sk > 16) sk = 16; if (sk < 1) sk = 1; r.sk_auto = sk; }
  }
  return r;
}

extern "C" size_t mv_gemm_workspace_bytes(int dtype, int ta, int tb, int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || !mv_is16(dtype) || g_mv_impl != 0) return 0;      // split-K is chosen on the MFMA kernels only
  const GemmRoute r = gemm_route(ta, tb, M, N, K, 0);
  return r.sk_auto > 1 ? (size_t)r.sk_auto * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" size_t mv_workspace_bytes(int hidden, int intermediate, int vocab, int img_hidden, int max_rows, int max_label_rows, int max_regions) {
  if (hidden <= 0 || intermediate <= 0 || vocab <= 0 || max_rows <= 0) return 0;
  const int H = hidden, I = intermediate, R = max_label_rows > 0 ? max_label_rows : 1;
  size_t w = 0;
  auto upd = [&](size_t b) { if (b > w) w = b; };
  // the weight gradients dW[No, Ko] = dy^T . x over the rows (Engine._dW): FFN-up / FFN-down / fused QKV / attention output projection ...
  upd(mv_gemm_workspace_bytes(MV_F16, 1, 1, I, H, max_rows));
  upd(mv_gemm_workspace_bytes(MV_F16, 1, 1, H, I, max_rows));
  upd(mv_gemm_workspace_bytes(MV_F16, 1, 1, 3 * H, H, max_rows));
  upd(mv_gemm_workspace_bytes(MV_F16, 1, 1, H, H, max_rows));
  // ... the MLM transform's and the image projection's
  upd(mv_gemm_workspace_bytes(MV_F16, 1, 1, H, H, R));
  if (img_hidden > 0 && max_regions > 0) upd(mv_gemm_workspace_bytes(MV_F16, 1, 1, H, img_hidden, max_regions));
  // the tied decoder's input gradient dt[R, H] = dlogits . E contracts over the vocabulary
  upd(mv_gemm_workspace_bytes(MV_F16, 0, 1, R, H, vocab));
  return w;
}

extern "C" int mv_gemm(int dtype, int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                       void* C, int ldc, int c_dtype, const float* bias, int epi, const void* R, int ldr, int r_dtype,
                       void* C2, int ldc2, void* C3, int ldc3, int c3_dtype, int splitk, float* ws, size_t ws_bytes,
                       int accumulate, float p_drop, unsigned long long drop_key, const float* alpha_dev, float* colsum_part,
                       void* stream_) {
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
  if (alpha_dev && (epi != MV_EPI_NONE || c_dtype != MV_F32 || C3)) return MV_E_ARG;
  if (splitk == 0 && (!mv_is16(dtype) || g_mv_impl != 0)) splitk = 1;   // auto split-K only on the MFMA kernels
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.C2 = C2; p.bias = bias; p.R = R; p.C3 = C3;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldr = ldr; p.ldc3 = ldc3;
  p.c_dtype = c_dtype; p.r_dtype = r_dtype; p.c3_dtype = c3_dtype; p.epi = epi; p.accumulate = accumulate;
  p.splitk = splitk; p.ws = ws; p.dbg = g_mv_gemm_dbg; p.alpha = alpha_dev; p.csum = colsum_part;
  p.drop = mv_make_drop(epi == MV_EPI_BIAS_RES ? p_drop : 0.f, drop_key);
  if (p.drop.thr && (N & 3)) return MV_E_SHAPE;   // the mask is keyed on groups of 4 consecutive columns
  const size_t csz = (c_dtype == MV_F32) ? 16 : 8;
  const size_t rsz = (r_dtype == MV_F32) ? 16 : 8;
  p.vec_ok = ((ldc & 3) == 0) && aligned_to(C, csz) && (!C3 || (((ldc3 & 3) == 0) && aligned_to(C3, 8))) &&
             (!need_bias || aligned_to(bias, 16)) &&
             (!need_r || (((ldr & 3) == 0) && aligned_to(R, rsz))) &&
             ((epi != MV_EPI_BIAS_GELU && epi != MV_EPI_BIAS_GELU_D) || (((ldc2 & 3) == 0) && aligned_to(C2, csz)));
  p.r8_ok = need_r && mv_is16(r_dtype) && ((ldr & 7) == 0) && aligned_to(R, 16);
  p.vec8_ok = mv_is16(c_dtype) && ((ldc & 7) == 0) && aligned_to(C, 16) && (!need_bias || aligned_to(bias, 16)) &&
              (!C3 || (((ldc3 & 7) == 0) && aligned_to(C3, 16))) &&
              ((epi != MV_EPI_BIAS_GELU && epi != MV_EPI_BIAS_GELU_D) || (((ldc2 & 7) == 0) && aligned_to(C2, 16)));
  // (N itself need not be a multiple of anything: the 16-byte path covers whole 64-column strips only, the ragged last strip of e.g. the
  //  30,522-column decoder takes the per-lane path)
  {   // the 16-byte-store epilogue addresses its outputs through buffer descriptors: sizes below 2 GiB, else the 8-byte path
    const size_t bC = (size_t)M * ldc * 2, bC2 = C2 ? (size_t)M * ldc2 * 2 : 0, bC3 = C3 ? (size_t)M * ldc3 * 2 : 0;
    if (bC >= 0x7fffffffULL || bC2 >= 0x7fffffffULL || bC3 >= 0x7fffffffULL) p.vec8_ok = 0;
    p.bytesC = (unsigned)bC; p.bytesC2 = (unsigned)bC2; p.bytesC3 = (unsigned)bC3;
  }
  const bool mfma = mv_is16(dtype) && (g_mv_impl == 0);
  const bool f16 = dtype == MV_F16;
  if (mfma && f16 && ta && !tb) return MV_E_DTYPE;    // f16 operands: y = x.W^T, dx = dy.W and dW = dy^T.x
  if (mfma) {
    if ((lda & 7) || (ldb & 7) || !aligned_to(A, 16) || !aligned_to(B, 16)) return MV_E_SHAPE;
    const size_t bytesA = ((size_t)((ta ? K : M) - 1) * lda + (size_t)(((ta ? M : K) + 7) & ~7)) * 2;
    const size_t bytesB = ((size_t)((tb ? K : N) - 1) * ldb + (size_t)(((tb ? N : K) + 7) & ~7)) * 2;
    if (bytesA >= 0x7fffffffULL || bytesB >= 0x7fffffffULL) return MV_E_SHAPE;
    p.bytesA = (unsigned)bytesA; p.bytesB = (unsigned)bytesB;
    // tile choice (gemm_route): a 256-row ring kernel when it fills the chip, the 128x128 kernel for small problems, dx and 768-column outputs
    const GemmRoute route = gemm_route(ta, tb, M, N, K, splitk, colsum_part != nullptr);
    const bool big = route.big;
    if (big) {
      const int variant = route.variant;
      const bool v128 = variant == 2;
      const long long tiles_v = route.tiles;
      long long sk = splitk;
      if (splitk > 1 || splitk == 0) {      // 0 = auto
        sk = route.sk_auto;
        if (splitk > 1 && sk > splitk) sk = splitk;
        if (sk > 1 && ws) { const long long fit = (long long)(ws_bytes / ((size_t)M * N * sizeof(float))); if (sk > fit) sk = fit < 1 ? 1 : fit; }
        if (sk > 1 && (!ws || epi != MV_EPI_NONE || c_dtype != MV_F32)) sk = 1;
      }
      (void)v128;
      int kchunk = (int)((K + sk - 1) / sk);
      kchunk = (kchunk + 63) / 64 * 64;
      p.kchunk = kchunk;
      p.splitk = splitk = (K + kchunk - 1) / kchunk;
      // variants: 14 = 256x256 with 64-deep stages x2 (128-B lines), 24 = its persistent form, 2 = 256x128 (32-deep x3, 2 blocks/CU)
      const int tiles = (int)tiles_v;
      // fused column sums: only the path that owns whole 64-column strips per wave and stores 16-byte pieces computes them
      if (colsum_part && !(variant == 14 && splitk == 1 && !accumulate && p.vec8_ok && (N & 255) == 0 && mv_is16(c_dtype) &&
                           (epi == MV_EPI_NONE || epi == MV_EPI_BIAS || epi == MV_EPI_BIAS_GELU_D || ((epi == MV_EPI_MUL || epi == MV_EPI_RES) && p.r8_ok))))
        return MV_E_SHAPE;
      dim3 grid(tiles, splitk);
      const int n_cu = gemm_n_cu();
      // persistent kernels: at most g_mv_persistent_cus blocks when the host partitions the chip (mv_set_persistent_cus)
      const int n_blk = (g_mv_persistent_cus > 0 && g_mv_persistent_cus < n_cu) ? g_mv_persistent_cus : n_cu;
      const int rc_ring = mv_launch_ring(p, ta, tb, f16, variant, tiles, splitk, n_blk, stream);
      if (rc_ring != MV_OK) return rc_ring;
    } else {
      if (colsum_part) return MV_E_SHAPE;        // the 256x256 ring kernel only
      if (splitk == 0) {
        splitk = 1;
        if (route.sk_auto > 1 && ws && epi == MV_EPI_NONE && c_dtype == MV_F32) {
          splitk = (int)route.sk_auto;
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
#define LAUNCH_MFMA_H(TA_, TB_)                                                                            \
  do {                                                                                                     \
    static bool attr_h = false;                                                                            \
    if (!attr_h) {                                                                                         \
      (void)hipFuncSetAttribute((const void*)gemm_mfma_kernel<TA_, TB_, false, true, true>,                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, GT_STAGE_BYTES);               \
      attr_h = true;                                                                                       \
    }                                                                                                      \
    hipLaunchKernelGGL((gemm_mfma_kernel<TA_, TB_, false, true, true>), grid, block, GT_STAGE_BYTES, stream, p); \
  } while (0)
      if (f16) {        // f16-encoded operands: always the one-stage form
        if (!ta && !tb) LAUNCH_MFMA_H(false, false);
        else if (!ta && tb) LAUNCH_MFMA_H(false, true);
        else LAUNCH_MFMA_H(true, true);
      }
      else if (!ta && !tb) { if (sb) LAUNCH_MFMA_SB(false, false); else LAUNCH_MFMA(false, false); }
      else if (!ta && tb) { if (sb) LAUNCH_MFMA_SB(false, true); else LAUNCH_MFMA(false, true); }
      else if (ta && tb) { if (sb) LAUNCH_MFMA_SB(true, true); else LAUNCH_MFMA(true, true); }
      else { if (sb) LAUNCH_MFMA_SB(true, false); else LAUNCH_MFMA(true, false); }
#undef LAUNCH_MFMA_H
#undef LAUNCH_MFMA_SB
#undef LAUNCH_MFMA
    }
  } else {
    if (colsum_part) return MV_E_SHAPE;
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
