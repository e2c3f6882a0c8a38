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
  int M, N, K, lda, ldb, ldc, ldc2, ldr;
  int c_dtype, r_dtype, epi, accumulate, vec_ok;
  int kchunk, splitk;
  float* ws;
  unsigned bytesA, bytesB;
};

// ------------------------------------------------------------------------------------------
// fused epilogue on 4 consecutive columns (n .. n+3) of row m.
__device__ __forceinline__ void epilogue4(const GemmArgs& p, int m, int n, f32x4 v) {
  const int nv = p.N - n;
  if (m >= p.M || nv <= 0) return;
  const size_t co = (size_t)m * p.ldc + n;
  if (p.vec_ok && nv >= 4) {
    f32x4 b = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
    const int e = p.epi;
    if (e == MV_EPI_BIAS || e == MV_EPI_BIAS_GELU || e == MV_EPI_BIAS_RES || e == MV_EPI_BIAS_TANH)
      b = *(const f32x4*)(p.bias + n);
    if (e == MV_EPI_BIAS_RES || e == MV_EPI_DGELU || e == MV_EPI_RES) {
      const size_t ro = (size_t)m * p.ldr + n;
      r = (p.r_dtype == MV_F32) ? ld4<float>((const float*)p.R + ro) : ld4<bf16_t>((const bf16_t*)p.R + ro);
    }
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = v[i];
      switch (e) {
        case MV_EPI_BIAS: x += b[i]; break;
        case MV_EPI_BIAS_GELU: x += b[i]; break;
        case MV_EPI_BIAS_RES: x += b[i] + r[i]; break;
        case MV_EPI_DGELU: x *= dgelu_erf(r[i]); break;
        case MV_EPI_RES: x += r[i]; break;
        case MV_EPI_BIAS_TANH: x = tanhf(x + b[i]); break;
        default: break;
      }
      o[i] = x;
    }
    if (e == MV_EPI_BIAS_GELU) {
      const size_t c2 = (size_t)m * p.ldc2 + n;
      if (p.c_dtype == MV_F32) st4<float>((float*)p.C2 + c2, o); else st4<bf16_t>((bf16_t*)p.C2 + c2, o);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = gelu_erf(o[i]);
    }
    if (p.c_dtype == MV_F32) {
      if (p.accumulate) { f32x4 old = *(const f32x4*)((const float*)p.C + co); o += old; }
      st4<float>((float*)p.C + co, o);
    } else {
      st4<bf16_t>((bf16_t*)p.C + co, o);
    }
    return;
  }
  const int lim = nv < 4 ? nv : 4;
  for (int i = 0; i < lim; ++i) {
    float x = v[i];
    const int e = p.epi;
    float b = 0.f, r = 0.f;
    if (e == MV_EPI_BIAS || e == MV_EPI_BIAS_GELU || e == MV_EPI_BIAS_RES || e == MV_EPI_BIAS_TANH) b = p.bias[n + i];
    if (e == MV_EPI_BIAS_RES || e == MV_EPI_DGELU || e == MV_EPI_RES) r = ld_any(p.R, (size_t)m * p.ldr + n + i, p.r_dtype);
    switch (e) {
      case MV_EPI_BIAS: x += b; break;
      case MV_EPI_BIAS_GELU: x += b; break;
      case MV_EPI_BIAS_RES: x += b + r; break;
      case MV_EPI_DGELU: x *= dgelu_erf(r); break;
      case MV_EPI_RES: x += r; break;
      case MV_EPI_BIAS_TANH: x = tanhf(x + b); break;
      default: break;
    }
    if (e == MV_EPI_BIAS_GELU) {
      st_any(p.C2, (size_t)m * p.ldc2 + n + i, p.c_dtype, x);
      x = gelu_erf(x);
    }
    if (p.c_dtype == MV_F32 && p.accumulate) x += ((const float*)p.C)[co + i];
    st_any(p.C, co + i, p.c_dtype, x);
  }
}

// raw partial tile store for split-K (ws is [splitk][M][N] f32)
__device__ __forceinline__ void store_partial4(const GemmArgs& p, int split, int m, int n, f32x4 v) {
  const int nv = p.N - n;
  if (m >= p.M || nv <= 0) return;
  float* w = p.ws + ((size_t)split * p.M + m) * p.N + n;
  if ((p.N & 3) == 0 && nv >= 4) { *(f32x4*)w = v; return; }
  for (int i = 0; i < (nv < 4 ? nv : 4); ++i) w[i] = v[i];
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

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmArgs p) {
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
  const int tiles_n = (p.N + GT_BN - 1) / GT_BN;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
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
  stage_load<TA>(ra, rsA, p.bytesA, p.lda, m0, p.M, kbeg, kend, tid);
  stage_load<TB>(rb, rsB, p.bytesB, p.ldb, n0, p.N, kbeg, kend, tid);
  stage_store<TA>(ra, smem, tid);
  stage_store<TB>(rb, smem + 16384, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const char* tA = smem + (kt & 1) * GT_STAGE_BYTES;
    const char* tB = tA + 16384;
    const bool more = (kt + 1 < nk);
    if (more) {
      const int k0 = kbeg + (kt + 1) * GT_BK;
      stage_load<TA>(ra, rsA, p.bytesA, p.lda, m0, p.M, k0, kend, tid);
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* nA = smem + ((kt + 1) & 1) * GT_STAGE_BYTES;
      stage_store<TA>(ra, nA, tid);
      stage_store<TB>(rb, nA + 16384, tid);
    }
    __syncthreads();
  }

  // D[r = n][c = m]: lane holds C[m = .. + l15][n = .. + 4*lq + 0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn + j * 16 + 4 * lq;
      if (p.splitk > 1) store_partial4(p, split, m, n, acc[i][j]);
      else epilogue4(p, m, n, acc[i][j]);
    }
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
    else epilogue4(p, m, n, v);
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
      epilogue4(p, m, n, s);
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
static inline bool aligned_to(const void* p, size_t a) { return p == nullptr || (((uintptr_t)p) % a) == 0; }

extern "C" int mv_gemm(int dtype, int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                       void* C, int ldc, int c_dtype, const float* bias, int epi, const void* R, int ldr, int r_dtype,
                       void* C2, int ldc2, int splitk, float* ws, size_t ws_bytes, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MV_E_ARG;
  if (dtype != MV_F32 && dtype != MV_BF16) return MV_E_DTYPE;
  if (c_dtype != MV_F32 && c_dtype != MV_BF16) return MV_E_DTYPE;
  if (epi < 0 || epi > MV_EPI_BIAS_TANH) return MV_E_ARG;
  const bool need_bias = (epi == MV_EPI_BIAS || epi == MV_EPI_BIAS_GELU || epi == MV_EPI_BIAS_RES || epi == MV_EPI_BIAS_TANH);
  const bool need_r = (epi == MV_EPI_BIAS_RES || epi == MV_EPI_DGELU || epi == MV_EPI_RES);
  if (need_bias && !bias) return MV_E_ARG;
  if (need_r && (!R || (r_dtype != MV_F32 && r_dtype != MV_BF16))) return MV_E_ARG;
  if (epi == MV_EPI_BIAS_GELU && !C2) return MV_E_ARG;
  if (lda < (ta ? M : K) || ldb < (tb ? N : K) || ldc < N) return MV_E_SHAPE;
  if (need_r && ldr < N) return MV_E_SHAPE;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 || accumulate) {
    if (epi != MV_EPI_NONE || c_dtype != MV_F32) return MV_E_SHAPE;
  }
  if (splitk > 1) {
    if (!ws || ws_bytes < (size_t)splitk * M * N * sizeof(float)) return MV_E_WORKSPACE;
  }
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.C2 = C2; p.bias = bias; p.R = R;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldr = ldr;
  p.c_dtype = c_dtype; p.r_dtype = r_dtype; p.epi = epi; p.accumulate = accumulate;
  p.splitk = splitk; p.ws = ws;
  const size_t csz = (c_dtype == MV_F32) ? 16 : 8;
  const size_t rsz = (r_dtype == MV_F32) ? 16 : 8;
  p.vec_ok = ((ldc & 3) == 0) && aligned_to(C, csz) && (!need_bias || aligned_to(bias, 16)) &&
             (!need_r || (((ldr & 3) == 0) && aligned_to(R, rsz))) &&
             (epi != MV_EPI_BIAS_GELU || (((ldc2 & 3) == 0) && aligned_to(C2, csz)));
  const bool mfma = (dtype == MV_BF16) && (g_mv_impl == 0);
  if (mfma) {
    if ((lda & 7) || (ldb & 7) || !aligned_to(A, 16) || !aligned_to(B, 16)) return MV_E_SHAPE;
    const size_t bytesA = ((size_t)((ta ? K : M) - 1) * lda + (size_t)(((ta ? M : K) + 7) & ~7)) * 2;
    const size_t bytesB = ((size_t)((tb ? K : N) - 1) * ldb + (size_t)(((tb ? N : K) + 7) & ~7)) * 2;
    if (bytesA >= 0x7fffffffULL || bytesB >= 0x7fffffffULL) return MV_E_SHAPE;
    p.bytesA = (unsigned)bytesA; p.bytesB = (unsigned)bytesB;
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
      hipFuncSetAttribute((const void*)gemm_mfma_kernel<TA_, TB_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)shm);                                                                       \
      attr_set = true;                                                                                     \
    }                                                                                                      \
    hipLaunchKernelGGL((gemm_mfma_kernel<TA_, TB_>), grid, block, shm, stream, p);                          \
  } while (0)
    if (!ta && !tb) LAUNCH_MFMA(false, false);
    else if (!ta && tb) LAUNCH_MFMA(false, true);
    else if (ta && tb) LAUNCH_MFMA(true, true);
    else LAUNCH_MFMA(true, false);
#undef LAUNCH_MFMA
  } else {
    int kchunk = (K + splitk - 1) / splitk;
    kchunk = (kchunk + 15) / 16 * 16;
    p.kchunk = kchunk;
    p.splitk = splitk = (K + kchunk - 1) / kchunk;
    dim3 grid((N + 63) / 64, (M + 63) / 64, splitk), block(256);
    if (grid.y > 65535) return MV_E_SHAPE;
    if (dtype == MV_F32) hipLaunchKernelGGL(gemm_simple_kernel<float>, grid, block, 0, stream, p, ta, tb);
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
