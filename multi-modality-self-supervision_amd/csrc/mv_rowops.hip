// HBM-bound row kernels of the CXRBERT hot path on gfx950: LayerNorm fwd/bwd, fused
// sequence-assembly + embedding + LayerNorm fwd/bwd, fused cross-entropy (+argmax, +gradient),
// row gather/scatter, column sums, casts and the fused HF-AdamW step.
// All of them are one-wave-per-row (64 lanes x 4-element vectors) streaming kernels: the
// roofline that bounds them is HBM bandwidth, so every operand is read once, 8/16 bytes per
// lane, and every reduction stays in registers / DPP shuffles.
#include "mv_common.h"

// compile-time number of 256-column chunks per row (keeps the per-row register arrays out of scratch)
#define NC_DISPATCH(H_, CALL)                                  \
  do {                                                         \
    if ((H_) <= 256) { CALL(1); }                              \
    else if ((H_) <= 768) { CALL(3); }                         \
    else if ((H_) <= 1024) { CALL(4); }                        \
    else { CALL(8); }                                          \
  } while (0)
#define MV_MAX_H 2048

// =========================================================================================
// LayerNorm
// =========================================================================================
// y = (x-mean)*rstd*g+b ; one wave per row, H % 4 == 0, H <= 2048
template <typename TX, typename TY, int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ bta, TY* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int H, float eps,
                                                     bf16_t* __restrict__ y_bf16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const TX* xr = x + (size_t)row * H;
  f32x4 v[NC];
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c = lane * 4 + 256 * n;
    v[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < H) { v[n] = ld4<TX>(xr + c); s += v[n][0] + v[n][1] + v[n][2] + v[n][3]; }
  }
  const float mu = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    if (lane * 4 + 256 * n < H) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[n][e] - mu; q += d * d; }
    }
  }
  const float var = wave_sum(q) / (float)H;
  const float rs = 1.0f / sqrtf(var + eps);
  TY* yr = y + (size_t)row * H;
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c = lane * 4 + 256 * n;
    if (c < H) {
      const f32x4 gg = *(const f32x4*)(g + c), bb = *(const f32x4*)(bta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[n][e] - mu) * rs * gg[e] + bb[e];
      st4<TY>(yr + c, o);
      if (y_bf16) st4<bf16_t>(y_bf16 + (size_t)row * H + c, o);
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); dgamma += dy*xhat; dbeta += dy; colsum += dx
template <typename TX, typename TD, int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TD* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ g, TD* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ colsum, int M, int H,
                                                     TD* __restrict__ dx_drop, DropCfg drop, const float* __restrict__ gscale) {
  __shared__ float red[3][4][260];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  f32x4 ag[NC], ab[NC], ac[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) { ag[i] = (f32x4){0, 0, 0, 0}; ab[i] = ag[i]; ac[i] = ag[i]; }
  for (int row = blockIdx.x * 4 + wl; row < M; row += gridDim.x * 4) {
    const TD* dyr = dy + (size_t)row * H;
    const TX* xr = x + (size_t)row * H;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NC], gd[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = lane * 4 + 256 * n;
      xh[n] = (f32x4){0, 0, 0, 0}; gd[n] = xh[n];
      if (c < H) {
        const f32x4 d = ld4<TD>(dyr + c), xv = ld4<TX>(xr + c), gg = *(const f32x4*)(g + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float h_ = (xv[e] - mu) * rs;
          const float gde = gg[e] * d[e];
          xh[n][e] = h_; gd[n][e] = gde;
          s1 += gde; s2 += gde * h_;
          ag[n][e] += d[e] * h_;
          ab[n][e] += d[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    TD* dxr = dx + (size_t)row * H;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = lane * 4 + 256 * n;
      if (c < H) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gd[n][e] - s1 - xh[n][e] * s2);
        st4<TD>(dxr + c, o);
        if (dx_drop) {      // gradient w.r.t. the projection output that went through dropout before the residual add
          o = mv_drop4(o, (size_t)row * H + c, drop);
          st4<TD>(dx_drop + (size_t)row * H + c, o);
        }
        ac[n] += o;         // colsum = bias gradient of that projection
      }
    }
  }
  // cross-wave reduction of the column accumulators, one 256-column chunk at a time
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c0 = 256 * n;
    if (c0 >= H) break;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][wl][lane * 4 + e] = ag[n][e]; red[1][wl][lane * 4 + e] = ab[n][e]; red[2][wl][lane * 4 + e] = ac[n][e]; }
    __syncthreads();
    const int col = c0 + threadIdx.x;
    if (col < H) {
      const int t = threadIdx.x;
      const float gs = gscale ? *gscale : 1.0f;       // 1 / loss scale: the f32 parameter gradients are kept unscaled
      atomicAdd(dgamma + col, gs * (red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t]));
      atomicAdd(dbeta + col, gs * (red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t]));
      if (colsum) atomicAdd(colsum + col, gs * (red[2][0][t] + red[2][1][t] + red[2][2][t] + red[2][3][t]));
    }
    __syncthreads();
  }
}

// The same with the NEXT row's dy / x requested before the current row's arithmetic (raw 16-bit registers, 4 NC more): a wave keeps two
// rows of loads in flight instead of one load -> reduce -> store round trip at a time (measured: profiles/r03_notes.txt).
template <typename T> struct raw4;
template <> struct raw4<float> { typedef f32x4 type; };
template <> struct raw4<bf16_t> { typedef bf16x4 type; };
template <> struct raw4<f16_t> { typedef f16x4 type; };
template <typename T> __device__ __forceinline__ f32x4 cvt_raw4(typename raw4<T>::type r) {
  return (f32x4){(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
}
template <typename TX, typename TD, int NC, int WPB>
__global__ __launch_bounds__(64 * WPB) void ln_bwd_pf_kernel(const TD* __restrict__ dy, const TX* __restrict__ x,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ g, TD* __restrict__ dx, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ colsum, int M, int H,
                                                        TD* __restrict__ dx_drop, DropCfg drop, const float* __restrict__ gscale) {
  typedef typename raw4<TD>::type RD;
  typedef typename raw4<TX>::type RX;
  __shared__ float red[3][WPB][260];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  f32x4 ag[NC], ab[NC], ac[NC], gg[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    ag[i] = (f32x4){0, 0, 0, 0}; ab[i] = ag[i]; ac[i] = ag[i];
    const int c = lane * 4 + 256 * i;
    gg[i] = c < H ? *(const f32x4*)(g + c) : ag[i];
  }
  const int stride = gridDim.x * WPB;
  int row = blockIdx.x * WPB + wl;
  RD rd[NC]; RX rx[NC];
  float mu = 0.f, rs = 0.f;
  if (row < M) {
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = lane * 4 + 256 * n;
      if (c < H) { rd[n] = *(const RD*)(dy + (size_t)row * H + c); rx[n] = *(const RX*)(x + (size_t)row * H + c); }
    }
    mu = mean[row]; rs = rstd[row];
  }
  for (; row < M; row += stride) {
    const int nrow = row + stride;
    RD nd[NC]; RX nx[NC];
    float nmu = 0.f, nrs = 0.f;
    if (nrow < M) {
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const int c = lane * 4 + 256 * n;
        if (c < H) { nd[n] = *(const RD*)(dy + (size_t)nrow * H + c); nx[n] = *(const RX*)(x + (size_t)nrow * H + c); }
      }
      nmu = mean[nrow]; nrs = rstd[nrow];
    }
    f32x4 xh[NC], gd[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = lane * 4 + 256 * n;
      xh[n] = (f32x4){0, 0, 0, 0}; gd[n] = xh[n];
      if (c < H) {
        const f32x4 d = cvt_raw4<TD>(rd[n]), xv = cvt_raw4<TX>(rx[n]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float h_ = (xv[e] - mu) * rs;
          const float gde = gg[n][e] * d[e];
          xh[n][e] = h_; gd[n][e] = gde;
          s1 += gde; s2 += gde * h_;
          ag[n][e] += d[e] * h_;
          ab[n][e] += d[e];
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    s1 /= (float)H; s2 /= (float)H;
    TD* dxr = dx + (size_t)row * H;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = lane * 4 + 256 * n;
      if (c < H) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gd[n][e] - s1 - xh[n][e] * s2);
        st4<TD>(dxr + c, o);
        if (dx_drop) {
          o = mv_drop4(o, (size_t)row * H + c, drop);
          st4<TD>(dx_drop + (size_t)row * H + c, o);
        }
        ac[n] += o;
      }
    }
#pragma unroll
    for (int n = 0; n < NC; ++n) { rd[n] = nd[n]; rx[n] = nx[n]; }
    mu = nmu; rs = nrs;
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c0 = 256 * n;
    if (c0 >= H) break;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][wl][lane * 4 + e] = ag[n][e]; red[1][wl][lane * 4 + e] = ab[n][e]; red[2][wl][lane * 4 + e] = ac[n][e]; }
    __syncthreads();
    // the atomics on one address serialise at the memory side (about 20 ns each, measured: 1024 -> 2048 blocks costs +20 us), hence
    // few, large blocks: thread t < 768 sums column t % 256 of accumulator t / 256 over the block's waves and adds it once
    const int t = threadIdx.x & 255;
    const int col = c0 + t;
    for (int which = threadIdx.x >> 8; which < 3; which += WPB / 4) {
      if (col < H && (which < 2 || colsum)) {
        const float gs = gscale ? *gscale : 1.0f;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) a += red[which][w][t];
        atomicAdd((which == 0 ? dgamma : which == 1 ? dbeta : colsum) + col, gs * a);
      }
    }
    __syncthreads();
  }
}

#define g_mv_ln_bwd_variant (mv_knob(MV_KNOB_ROWOPS_VARIANT))     // test / experiment hook, see mv_layernorm_bwd; grid cap in bits 8..

extern "C" int mv_layernorm_fwd(int dtype, const void* x, int x_dtype, const float* gamma, const float* beta, void* y, void* y_bf16,
                                float* mean, float* rstd, int M, int H, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0 || H <= 0) return MV_E_ARG;
  if ((H & 3) || H > MV_MAX_H) return MV_E_SHAPE;
  if (y_bf16 && dtype != MV_F16) return MV_E_DTYPE;      // the second output is the bf16 copy of an f16 forward activation
  dim3 grid((M + 3) / 4), block(256);
#define LNF(NC_) hipLaunchKernelGGL((ln_fwd_kernel<TX_, TY_, NC_>), grid, block, 0, stream, (const TX_*)x, gamma, beta, (TY_*)y, mean, rstd, M, H, eps, (bf16_t*)y_bf16)
  if (dtype == MV_F32 && x_dtype == MV_F32) { typedef float TX_; typedef float TY_; NC_DISPATCH(H, LNF); }
  else if (dtype == MV_BF16 && x_dtype == MV_F32) { typedef float TX_; typedef bf16_t TY_; NC_DISPATCH(H, LNF); }
  else if (dtype == MV_BF16 && x_dtype == MV_BF16) { typedef bf16_t TX_; typedef bf16_t TY_; NC_DISPATCH(H, LNF); }
  else if (dtype == MV_F16 && x_dtype == MV_F32) { typedef float TX_; typedef f16_t TY_; NC_DISPATCH(H, LNF); }
  else if (dtype == MV_F16 && x_dtype == MV_F16) { typedef f16_t TX_; typedef f16_t TY_; NC_DISPATCH(H, LNF); }
  else return MV_E_DTYPE;
#undef LNF
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_layernorm_bwd(int dtype, const void* dy, const void* x, int x_dtype, const float* mean, const float* rstd,
                                const float* gamma, void* dx, float* dgamma, float* dbeta, float* colsum, int M, int H,
                                void* dx_drop, float p_drop, unsigned long long drop_key, const float* grad_unscale_dev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || M <= 0 || H <= 0) return MV_E_ARG;
  if ((H & 3) || H > MV_MAX_H) return MV_E_SHAPE;
  int var = g_mv_ln_bwd_variant & 0xff;                 // 0: 8 waves per block x 512 blocks (default) | 1: old kernel | 2: prefetch, 4 waves | 3: 16 waves x 256
  // (the 8- and 16-wave blocks have 128 registers per lane: enough for rows of up to 768 16-bit elements, beyond that they would spill)
  if ((var == 0 || var == 3) && !(H <= 768 && x_dtype != MV_F32)) var = 2;
  const int wpb = var == 0 ? 8 : var == 3 ? 16 : 4;
  int blocks = (M + wpb - 1) / wpb;
  const int cap = (g_mv_ln_bwd_variant >> 8) > 0 ? (g_mv_ln_bwd_variant >> 8) : (var == 0 ? 512 : var == 3 ? 256 : 1024);
  if (blocks > cap) blocks = cap;
  dim3 grid(blocks), block(64 * wpb);
  const DropCfg drop = mv_make_drop(dx_drop ? p_drop : 0.f, drop_key);
  if (dx_drop && drop.thr == 0) dx_drop = nullptr;
#define LNB_ARGS (const TD_*)dy, (const TX_*)x, mean, rstd, gamma, (TD_*)dx, dgamma, dbeta, colsum, M, H, (TD_*)dx_drop, drop, grad_unscale_dev
#define LNB(NC_)                                                                                                                  \
  do {                                                                                                                            \
    constexpr bool wide_ok__ = (NC_) <= 3 && sizeof(TX_) == 2;                                                                    \
    if constexpr (wide_ok__) {                                                                                                    \
      if (var == 3) { hipLaunchKernelGGL((ln_bwd_pf_kernel<TX_, TD_, NC_, 16>), grid, block, 0, stream, LNB_ARGS); break; }       \
      if (var == 0) { hipLaunchKernelGGL((ln_bwd_pf_kernel<TX_, TD_, NC_, 8>), grid, block, 0, stream, LNB_ARGS); break; }        \
    }                                                                                                                             \
    if (var != 1) hipLaunchKernelGGL((ln_bwd_pf_kernel<TX_, TD_, NC_, 4>), grid, block, 0, stream, LNB_ARGS);                \
    else hipLaunchKernelGGL((ln_bwd_kernel<TX_, TD_, NC_>), grid, block, 0, stream, (const TD_*)dy, (const TX_*)x, mean, rstd, gamma, (TD_*)dx, dgamma, dbeta, colsum, M, H, (TD_*)dx_drop, drop, grad_unscale_dev); \
  } while (0)
  if (dtype == MV_F32 && x_dtype == MV_F32) { typedef float TX_; typedef float TD_; NC_DISPATCH(H, LNB); }
  else if (dtype == MV_BF16 && x_dtype == MV_F32) { typedef float TX_; typedef bf16_t TD_; NC_DISPATCH(H, LNB); }
  else if (dtype == MV_BF16 && x_dtype == MV_BF16) { typedef bf16_t TX_; typedef bf16_t TD_; NC_DISPATCH(H, LNB); }
  else if (dtype == MV_BF16 && x_dtype == MV_F16) { typedef f16_t TX_; typedef bf16_t TD_; NC_DISPATCH(H, LNB); }
  else if (dtype == MV_F16 && x_dtype == MV_F32) { typedef float TX_; typedef f16_t TD_; NC_DISPATCH(H, LNB); }
  else if (dtype == MV_F16 && x_dtype == MV_F16) { typedef f16_t TX_; typedef f16_t TD_; NC_DISPATCH(H, LNB); }
  else return MV_E_DTYPE;
#undef LNB
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// =========================================================================================
// sequence assembly + embeddings + LayerNorm  (cxrbert_origin.py:114-125, :22-35)
// =========================================================================================
struct EmbArgs {
  const int64_t* cls_tok; const int64_t* txt; const int64_t* segment; const int64_t* img_pos; const int64_t* sep_tok;
  int B, N, T, H, V, maxpos, L;
  const int32_t* rowmap;   // packed rows (nullable): row r holds logical position rowmap[r] = b*L + l; n_rows of them
  int n_rows;
};
// position l of sample b -> (token id or -1 for an image region, position id, type id, region index)
// img_pos == NULL: image rows get no position embedding (args.img_postion false, cxrbert_origin.py:27-31): pos = -1 for them
__device__ __forceinline__ void emb_decode(const EmbArgs& a, int b, int l, int& tok, int& pos, int& typ, int& reg) {
  reg = -1;
  if (l == 0) { tok = (int)a.cls_tok[b]; pos = 0; typ = 0; }
  else if (l <= a.N) {
    tok = -1; reg = l - 1; typ = 0;
    if (!a.img_pos) { pos = -1; return; }
    pos = (int)a.img_pos[(size_t)b * a.N + reg];
  }
  else if (l == a.N + 1) { tok = (int)a.sep_tok[b]; pos = 0; typ = 0; }
  else { const int t = l - a.N - 2; tok = (int)a.txt[(size_t)b * a.T + t]; pos = t; typ = (int)a.segment[(size_t)b * a.T + t]; }
  // clamp so that a bad id can never fault (HF would raise an index error on the host)
  if (tok >= a.V) tok = a.V - 1;
  if (tok < -1) tok = 0;
  pos = pos < 0 ? 0 : (pos >= a.maxpos ? a.maxpos - 1 : pos);
  typ = typ < 0 ? 0 : (typ > 1 ? 1 : typ);
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbArgs a, const T* __restrict__ imgproj, const T* __restrict__ E,
                                                        const T* __restrict__ P, const T* __restrict__ Ty,
                                                        const float* __restrict__ g, const float* __restrict__ bta,
                                                        T* __restrict__ x0, float* __restrict__ pre, float* __restrict__ mean,
                                                        float* __restrict__ rstd, float eps, DropCfg drop_txt, DropCfg drop_img,
                                                        bf16_t* __restrict__ x0_bf16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n_rows) return;
  const int li = a.rowmap ? a.rowmap[row] : row;
  const int b = li / a.L, l = li - b * a.L, H = a.H;
  int tok, pos, typ, reg;
  emb_decode(a, b, l, tok, pos, typ, reg);
  const DropCfg drop = (tok < 0) ? drop_img : drop_txt;      // image rows: nn.Dropout(args.dropout_prob), cxrbert_origin.py:19
  const T* src = (tok >= 0) ? E + (size_t)tok * H : imgproj + ((size_t)b * a.N + reg) * H;
  const bool haspos = pos >= 0;
  const T* pr = P + (size_t)(haspos ? pos : 0) * H;
  const T* tr = Ty + (size_t)typ * H;
  f32x4 v[NC];
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c = lane * 4 + 256 * n;
    v[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < H) {
      const f32x4 e = ld4<T>(src + c), p = ld4<T>(pr + c), t = ld4<T>(tr + c);
      // summation order of the reference: (word|img) + position + type  (cxrbert_origin.py:29; :31 without the position)
      v[n] = haspos ? e + p + t : e + t;
      s += v[n][0] + v[n][1] + v[n][2] + v[n][3];
    }
  }
  const float mu = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    if (lane * 4 + 256 * n < H) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[n][e] - mu; q += d * d; }
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c = lane * 4 + 256 * n;
    if (c < H) {
      const f32x4 gg = *(const f32x4*)(g + c), bb = *(const f32x4*)(bta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[n][e] - mu) * rs * gg[e] + bb[e];
      if (drop.thr) o = mv_drop4(o, (size_t)row * H + c, drop);
      st4<T>(x0 + (size_t)row * H + c, o);
      if (x0_bf16) st4<bf16_t>(x0_bf16 + (size_t)row * H + c, o);
      *(f32x4*)(pre + (size_t)row * H + c) = v[n];
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbArgs a, const T* __restrict__ dx0, const float* __restrict__ pre,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ g, float* __restrict__ dE, float* __restrict__ dP,
                                                        float* __restrict__ dTy, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, T* __restrict__ dimg, int pad_id, int hot_id, DropCfg drop_txt,
                                                        DropCfg drop_img, const float* __restrict__ gscale) {
  __shared__ float red[5][4][260];
  __shared__ __attribute__((aligned(16))) float rowbuf[4][MV_MAX_H];   // per-wave row, re-read lane-contiguously for the atomics
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6, H = a.H;
  // am: the word-table gradient of ONE hot token id (hot_id: [MASK] -- 12 % of the text positions, ~2,700 rows of a B = 64 batch all adding into
  // the same 768 addresses, where float atomics serialise: embed_bwd_bench.py 248 us against 175 us with distinct ids) summed in registers
  // like the type rows and added once per block
  f32x4 ag[NC], ab[NC], at0[NC], at1[NC], am[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) { ag[i] = (f32x4){0, 0, 0, 0}; ab[i] = ag[i]; at0[i] = ag[i]; at1[i] = ag[i]; am[i] = ag[i]; }
  const int M = a.n_rows;
  const float gs = gscale ? *gscale : 1.0f;        // 1 / loss scale: table / LayerNorm gradients (f32) are kept unscaled
  for (int row = blockIdx.x * 4 + wl; row < M; row += gridDim.x * 4) {
    const int li = a.rowmap ? a.rowmap[row] : row;
    const int b = li / a.L, l = li - b * a.L;
    int tok, pos, typ, reg;
    emb_decode(a, b, l, tok, pos, typ, reg);
    const DropCfg drop = (tok < 0) ? drop_img : drop_txt;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NC], gd[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = lane * 4 + 256 * n;
      xh[n] = (f32x4){0, 0, 0, 0}; gd[n] = xh[n];
      if (c < H) {
        f32x4 d = ld4<T>(dx0 + (size_t)row * H + c);
        const f32x4 xv = *(const f32x4*)(pre + (size_t)row * H + c), gg = *(const f32x4*)(g + c);
        if (drop.thr) d = mv_drop4(d, (size_t)row * H + c, drop);      // back through the embedding dropout
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float h_ = (xv[e] - mu) * rs, gde = gg[e] * d[e];
          xh[n][e] = h_; gd[n][e] = gde; s1 += gde; s2 += gde * h_;
          ag[n][e] += d[e] * h_; ab[n][e] += d[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = lane * 4 + 256 * n;
      if (c < H) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gd[n][e] - s1 - xh[n][e] * s2);
        if (typ == 0) at0[n] += o; else at1[n] += o;
        if (tok == hot_id) am[n] += o;
        *(f32x4*)(&rowbuf[wl][c]) = o;
        if (tok < 0) st4<T>(dimg + ((size_t)b * a.N + reg) * H + c, o);
      }
    }
    // scatter-add with 64 consecutive floats per wave instruction (the shape float atomics run at full rate in);
    // the [PAD] row receives no look-up gradient (nn.Embedding padding_idx of HF BertEmbeddings)
    const bool do_e = (tok >= 0) && (tok != pad_id) && (tok != hot_id), do_p = pos >= 0;
    float* pp = dP + (size_t)(do_p ? pos : 0) * H;
    float* ep = dE + (size_t)(tok >= 0 ? tok : 0) * H;
    for (int c = lane; c < H; c += 64) {
      const float v = rowbuf[wl][c] * gs;
      if (do_p) atomicAdd(pp + c, v);
      if (do_e) atomicAdd(ep + c, v);
    }
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c0 = 256 * n;
    if (c0 >= H) break;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wl][lane * 4 + e] = ag[n][e]; red[1][wl][lane * 4 + e] = ab[n][e];
      red[2][wl][lane * 4 + e] = at0[n][e]; red[3][wl][lane * 4 + e] = at1[n][e];
      red[4][wl][lane * 4 + e] = am[n][e];
    }
    __syncthreads();
    const int col = c0 + threadIdx.x, t = threadIdx.x;
    if (col < H) {
      if (hot_id >= 0 && hot_id != pad_id) {
        const float hv = red[4][0][t] + red[4][1][t] + red[4][2][t] + red[4][3][t];
        if (hv != 0.f) atomicAdd(dE + (size_t)hot_id * H + col, gs * hv);
      }
      atomicAdd(dgamma + col, gs * (red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t]));
      atomicAdd(dbeta + col, gs * (red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t]));
      atomicAdd(dTy + col, gs * (red[2][0][t] + red[2][1][t] + red[2][2][t] + red[2][3][t]));
      atomicAdd(dTy + H + col, gs * (red[3][0][t] + red[3][1][t] + red[3][2][t] + red[3][3][t]));
    }
    __syncthreads();
  }
}

// the one token id whose look-up gradient is summed per block instead of row by row: [MASK] of the BERT vocabularies (data/dataset_origin.py:198
// writes self.vocab_stoi["[MASK]"] over 80 % of the selected tokens).  Any id is handled correctly either way; this one is only faster.
#define MV_HOT_TOKEN 103
static int emb_check(int B, int N, int T, int H, int V, int maxpos) {
  if (B <= 0 || N < 0 || T <= 0 || H <= 0 || V <= 0 || maxpos <= 0) return MV_E_ARG;
  if ((H & 3) || H > MV_MAX_H) return MV_E_SHAPE;
  return MV_OK;
}

extern "C" int mv_embed_fwd(int dtype, const int64_t* cls_tok, const int64_t* txt, const int64_t* segment, const int64_t* img_pos,
                            const int64_t* sep_tok, const void* imgproj, const void* E, const void* P, const void* Ty,
                            const float* gamma, const float* beta, void* x0, void* x0_bf16, float* pre, float* mean, float* rstd,
                            int B, int N, int T, int H, int V, int maxpos, float eps, float p_drop, float p_drop_img,
                            unsigned long long drop_key, const int32_t* rowmap, int n_rows, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!cls_tok || !txt || !segment || !sep_tok || !E || !P || !Ty || !gamma || !beta || !x0 || !pre || !mean || !rstd) return MV_E_ARG;
  if (N > 0 && !imgproj) return MV_E_ARG;          // img_pos may be NULL: no position embedding on the image rows
  int rc = emb_check(B, N, T, H, V, maxpos);
  if (rc) return rc;
  if (T > maxpos) return MV_E_SHAPE;   // text positions 0..T-1 must exist
  if (rowmap && (n_rows <= 0 || n_rows > B * (N + T + 2))) return MV_E_ARG;
  EmbArgs a{cls_tok, txt, segment, img_pos, sep_tok, B, N, T, H, V, maxpos, N + T + 2, rowmap, rowmap ? n_rows : B * (N + T + 2)};
  dim3 grid((a.n_rows + 3) / 4), block(256);
  const DropCfg drop = mv_make_drop(p_drop, drop_key), drop_img = mv_make_drop(p_drop_img, drop_key);
  if (x0_bf16 && dtype != MV_F16) return MV_E_DTYPE;
#define EMF(NC_) hipLaunchKernelGGL((embed_fwd_kernel<T_, NC_>), grid, block, 0, stream, a, (const T_*)imgproj, (const T_*)E, (const T_*)P, (const T_*)Ty, gamma, beta, (T_*)x0, pre, mean, rstd, eps, drop, drop_img, (bf16_t*)x0_bf16)
  if (dtype == MV_F32) { typedef float T_; NC_DISPATCH(H, EMF); }
  else if (dtype == MV_BF16) { typedef bf16_t T_; NC_DISPATCH(H, EMF); }
  else if (dtype == MV_F16) { typedef f16_t T_; NC_DISPATCH(H, EMF); }
  else return MV_E_DTYPE;
#undef EMF
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_embed_bwd(int dtype, const void* dx0, const float* pre, const float* mean, const float* rstd, const float* gamma,
                            const int64_t* cls_tok, const int64_t* txt, const int64_t* segment, const int64_t* img_pos,
                            const int64_t* sep_tok, float* dE, float* dP, float* dTy, float* dgamma, float* dbeta, void* dimgproj,
                            int B, int N, int T, int H, int V, int maxpos, int pad_token_id, float p_drop, float p_drop_img,
                            unsigned long long drop_key, const int32_t* rowmap, int n_rows, const float* grad_unscale_dev,
                            void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dx0 || !pre || !mean || !rstd || !gamma || !cls_tok || !txt || !segment || !sep_tok || !dE || !dP || !dTy || !dgamma || !dbeta)
    return MV_E_ARG;
  if (N > 0 && !dimgproj) return MV_E_ARG;
  int rc = emb_check(B, N, T, H, V, maxpos);
  if (rc) return rc;
  if (rowmap && (n_rows <= 0 || n_rows > B * (N + T + 2))) return MV_E_ARG;
  EmbArgs a{cls_tok, txt, segment, img_pos, sep_tok, B, N, T, H, V, maxpos, N + T + 2, rowmap, rowmap ? n_rows : B * (N + T + 2)};
  int blocks = (a.n_rows + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  dim3 grid(blocks), block(256);
  const DropCfg drop = mv_make_drop(p_drop, drop_key), drop_img = mv_make_drop(p_drop_img, drop_key);
#define EMB(NC_) hipLaunchKernelGGL((embed_bwd_kernel<T_, NC_>), grid, block, 0, stream, a, (const T_*)dx0, pre, mean, rstd, gamma, dE, dP, dTy, dgamma, dbeta, (T_*)dimgproj, pad_token_id, MV_HOT_TOKEN, drop, drop_img, grad_unscale_dev)
  if (dtype == MV_F32) { typedef float T_; NC_DISPATCH(H, EMB); }
  else if (dtype == MV_BF16) { typedef bf16_t T_; NC_DISPATCH(H, EMB); }
  else if (dtype == MV_F16) { typedef f16_t T_; NC_DISPATCH(H, EMB); }
  else return MV_E_DTYPE;
#undef EMB
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// =========================================================================================
// fused cross-entropy + argmax + gradient  (train_origin.py:62-63,120-126,133-146)
// =========================================================================================
// one block (256 threads) per row; the row is read once into registers (V <= 256*128).
#define CE_MAXPER 128
template <typename TL, typename TD, int MAXPER>
__global__ __launch_bounds__(256) void ce_kernel(const TL* __restrict__ logits, int ld, const int32_t* __restrict__ labels, int R,
                                                 int V, float* __restrict__ out, TD* __restrict__ dlogits, int ldd,
                                                 const float* __restrict__ gs_dev, float gs_host, const float* __restrict__ ls_dev) {
  __shared__ float smax[4];
  __shared__ int sarg[4];
  __shared__ float ssum[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wl = tid >> 6;
  const int label = labels[row];
  TD* drow = dlogits ? dlogits + (size_t)row * ldd : nullptr;
  if (label < 0 || label >= V) {   // ignore_index (-100): no loss, zero gradient
    if (drow) for (int c = tid; c < ldd; c += 256) stf<TD>(drow + c, 0.f);
    return;
  }
  const TL* lr = logits + (size_t)row * ld;
  float v[MAXPER];
  float mx = -INFINITY;
  int am = 0x7fffffff;
#pragma unroll
  for (int n = 0; n < MAXPER; ++n) {
    const int c = tid + 256 * n;
    v[n] = -INFINITY;
    if (c < V) {
      const float x = ldf<TL>(lr + c);
      v[n] = x;
      if (x > mx) { mx = x; am = c; }     // first maximum wins (torch.argmax tie rule)
    }
  }
  // wave + block argmax (smallest index among equal maxima)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
  }
  if (lane == 0) { smax[wl] = mx; sarg[wl] = am; }
  __syncthreads();
  mx = smax[0]; am = sarg[0];
  for (int w = 1; w < 4; ++w) if (smax[w] > mx || (smax[w] == mx && sarg[w] < am)) { mx = smax[w]; am = sarg[w]; }
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < MAXPER; ++n) { v[n] = expf(v[n] - mx); s += v[n]; }   // padding: exp(-inf) = 0
  s = wave_sum(s);
  if (lane == 0) ssum[wl] = s;
  __syncthreads();
  s = ssum[0] + ssum[1] + ssum[2] + ssum[3];
  if (tid == 0) {
    const float xl = ldf<TL>(lr + label);
    atomicAdd(out + 0, (mx + logf(s)) - xl);
    atomicAdd(out + 1, 1.0f);
    if (am == label) atomicAdd(out + 2, 1.0f);
  }
  if (drow) {
    const float gs = (gs_dev ? *gs_dev : gs_host) * (ls_dev ? *ls_dev : 1.0f);   // x loss scale (16-bit gradients)
    const float inv = gs / s;
#pragma unroll
    for (int n = 0; n < MAXPER; ++n) {
      const int c = tid + 256 * n;
      if (c < V) stf<TD>(drow + c, v[n] * inv - (c == label ? gs : 0.f));
    }
    for (int c = V + tid; c < ldd; c += 256) stf<TD>(drow + c, 0.f);
  }
}

// Vector form: every lane owns 4 consecutive columns per step (16-byte logit loads, 8/16-byte gradient stores).
// Used when ld, ldd are multiples of 4 and the bases are 16-byte aligned (the MLM head's padded [R, Vp] buffers).
template <typename TL, typename TD, int MAXV, int NT>
__global__ __launch_bounds__(NT) void ce_vec_kernel(const TL* __restrict__ logits, int ld, const int32_t* __restrict__ labels,
                                                     int R, int V, float* __restrict__ out, TD* __restrict__ dlogits, int ldd,
                                                     const float* __restrict__ gs_dev, float gs_host, const float* __restrict__ ls_dev) {
  constexpr int NWV = NT / 64;
  __shared__ float smax[NWV];
  __shared__ int sarg[NWV];
  __shared__ float ssum[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wl = tid >> 6;
  // a block walks rows blockIdx.x, + gridDim.x, ... and adds its three sums once at the end: the adds of all blocks land on the same
  // three addresses and serialise at the memory side (about 20 ns each, profiles/r03_notes.txt) -- one set per ROW cost ~70 us here
  float t_nll = 0.f, t_cnt = 0.f, t_ok = 0.f;
  for (int row = blockIdx.x; row < R; row += gridDim.x) {
  const int label = labels[row];
  TD* drow = dlogits ? dlogits + (size_t)row * ldd : nullptr;
  if (label < 0 || label >= V) {
    if (drow) for (int c = tid * 4; c < ldd; c += NT * 4) st4<TD>(drow + c, (f32x4){0.f, 0.f, 0.f, 0.f});
    continue;
  }
  const TL* lr = logits + (size_t)row * ld;
  f32x4 v[MAXV];
  float mx = -INFINITY;
  int am = 0x7fffffff;
#pragma unroll
  for (int n = 0; n < MAXV; ++n) {
    const int c = (tid + NT * n) * 4;
    v[n] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (c < V) {
      f32x4 x = ld4<TL>(lr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c + e >= V) x[e] = -INFINITY;                 // padding columns of the last vector
        if (x[e] > mx) { mx = x[e]; am = c + e; }         // first maximum wins (torch.argmax tie rule)
      }
      v[n] = x;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
  }
  if (lane == 0) { smax[wl] = mx; sarg[wl] = am; }
  __syncthreads();
  mx = smax[0]; am = sarg[0];
  for (int w = 1; w < NWV; ++w) if (smax[w] > mx || (smax[w] == mx && sarg[w] < am)) { mx = smax[w]; am = sarg[w]; }
  float s = 0.f;
  const float mxl = mx * 1.4426950408889634f;
#pragma unroll
  for (int n = 0; n < MAXV; ++n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[n][e] = fexp2(fmaf(v[n][e], 1.4426950408889634f, -mxl)); s += v[n][e]; }
  }
  s = wave_sum(s);
  if (lane == 0) ssum[wl] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) s += ssum[w];
  if (tid == 0) {
    const float xl = ldf<TL>(lr + label);
    t_nll += (mx + logf(s)) - xl;
    t_cnt += 1.0f;
    if (am == label) t_ok += 1.0f;
  }
  if (drow) {
    const float gs = (gs_dev ? *gs_dev : gs_host) * (ls_dev ? *ls_dev : 1.0f);   // x loss scale (16-bit gradients)
    const float inv = gs / s;
#pragma unroll
    for (int n = 0; n < MAXV; ++n) {
      const int c = (tid + NT * n) * 4;
      if (c < ldd) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c + e < V) ? v[n][e] * inv - ((c + e) == label ? gs : 0.f) : 0.f;
        st4<TD>(drow + c, o);
      }
    }
  }
  __syncthreads();       // smax / sarg / ssum are reused by the next row
  }
  if (tid == 0 && t_cnt > 0.f) {
    atomicAdd(out + 0, t_nll);
    atomicAdd(out + 1, t_cnt);
    if (t_ok > 0.f) atomicAdd(out + 2, t_ok);
  }
}

extern "C" int mv_ce_fwd_bwd(const void* logits, int l_dtype, int ld, const int32_t* labels, int R, int V, float* out, void* dlogits,
                             int d_dtype, int ldd, const float* grad_scale_dev, float grad_scale_host, const float* loss_scale_dev,
                             void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!logits || !labels || !out || R <= 0 || V <= 0 || ld < V) return MV_E_ARG;
  if (V > 256 * CE_MAXPER) return MV_E_SHAPE;
  if (dlogits && ldd < V) return MV_E_SHAPE;
  dim3 grid(R), block(256);
  const bool vec_ok = ((ld & 3) == 0) && (!dlogits || (ldd & 3) == 0) && ((((uintptr_t)logits) & 15) == 0) &&
                      (!dlogits || (((uintptr_t)dlogits) & 15) == 0) && V > 2048 && ldd <= 1024 * 32 && V <= 1024 * 32;
  if (vec_ok) {
    // 1024-thread blocks (32 logits per thread in registers, two blocks per CU) walking rows: more than twice the waves in flight of the
    // 256-thread form of rounds 1-2 (128 logits per thread, three blocks per CU; 373 -> 179 us alone), one set of result atomics per block
    const dim3 vgrid(R < 512 ? R : 512), vblock(1024);
#define CEV_LAUNCH(TL, TD) hipLaunchKernelGGL((ce_vec_kernel<TL, TD, 8, 1024>), vgrid, vblock, 0, stream, (const TL*)logits, ld, labels, R, V, out, (TD*)dlogits, ldd, grad_scale_dev, grad_scale_host, loss_scale_dev)
    if (l_dtype == MV_F32 && d_dtype == MV_F32) CEV_LAUNCH(float, float);
    else if (l_dtype == MV_F32 && d_dtype == MV_F16) CEV_LAUNCH(float, f16_t);
    else if (l_dtype == MV_F32 && d_dtype == MV_BF16) CEV_LAUNCH(float, bf16_t);
    else if (l_dtype == MV_BF16 && d_dtype == MV_BF16) CEV_LAUNCH(bf16_t, bf16_t);
    else if (l_dtype == MV_BF16 && d_dtype == MV_F32) CEV_LAUNCH(bf16_t, float);
    else if (l_dtype == MV_F16 && d_dtype == MV_F16) CEV_LAUNCH(f16_t, f16_t);
    else if (l_dtype == MV_F16 && d_dtype == MV_F32) CEV_LAUNCH(f16_t, float);
    else return MV_E_DTYPE;
#undef CEV_LAUNCH
    MV_CHECK_LAUNCH();
    return MV_OK;
  }
#define CE_LAUNCH(TL, TD)                                                                                              \
  do {                                                                                                                 \
    if (V <= 256 * 8) hipLaunchKernelGGL((ce_kernel<TL, TD, 8>), grid, block, 0, stream, (const TL*)logits, ld, labels, R, V, out, (TD*)dlogits, ldd, grad_scale_dev, grad_scale_host, loss_scale_dev); \
    else hipLaunchKernelGGL((ce_kernel<TL, TD, CE_MAXPER>), grid, block, 0, stream, (const TL*)logits, ld, labels, R, V, out, (TD*)dlogits, ldd, grad_scale_dev, grad_scale_host, loss_scale_dev); \
  } while (0)
  if (l_dtype == MV_F32 && d_dtype == MV_F32) CE_LAUNCH(float, float);
  else if (l_dtype == MV_F32 && d_dtype == MV_F16) CE_LAUNCH(float, f16_t);
  else if (l_dtype == MV_F32 && d_dtype == MV_BF16) CE_LAUNCH(float, bf16_t);
  else if (l_dtype == MV_BF16 && d_dtype == MV_BF16) CE_LAUNCH(bf16_t, bf16_t);
  else if (l_dtype == MV_BF16 && d_dtype == MV_F32) CE_LAUNCH(bf16_t, float);
  else if (l_dtype == MV_F16 && d_dtype == MV_F16) CE_LAUNCH(f16_t, f16_t);
  else if (l_dtype == MV_F16 && d_dtype == MV_F32) CE_LAUNCH(f16_t, float);
  else return MV_E_DTYPE;
#undef CE_LAUNCH
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// =========================================================================================
// gather / scatter rows, column sums, add, cast
// =========================================================================================
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, int lds_, const int32_t* __restrict__ rows, int R, int H,
                                   T* __restrict__ dst, int ldd) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int sr = rows[r];
  T* d = dst + (size_t)r * ldd;
  if (sr < 0) {        // "no such row" (e.g. a position dropped by the packed-row plan): defined output, never a wild read
    for (int c = lane * 4; c < H; c += 256) st4<T>(d + c, (f32x4){0.f, 0.f, 0.f, 0.f});
    return;
  }
  const T* s = src + (size_t)sr * lds_;
  for (int c = lane * 4; c < H; c += 256) st4<T>(d + c, ld4<T>(s + c));
}
template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, int lds_, const int32_t* __restrict__ rows, int R, int H,
                                    T* __restrict__ dst, int ldd, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R || rows[r] < 0) return;      // negative row index = no such row: nothing is written
  const T* s = src + (size_t)r * lds_;
  T* d = dst + (size_t)rows[r] * ldd;
  for (int c = lane * 4; c < H; c += 256) {
    f32x4 v = ld4<T>(s + c);
    if (accumulate) v += ld4<T>(d + c);
    st4<T>(d + c, v);
  }
}

extern "C" int mv_gather_rows(int dtype, const void* src, int lds_, const int32_t* rows, int R, int H, void* dst, int ldd, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !rows || !dst || R <= 0 || H <= 0) return MV_E_ARG;
  if ((H & 3) || (lds_ & 3) || (ldd & 3)) return MV_E_SHAPE;
  dim3 grid((R + 3) / 4), block(256);
  if (dtype == MV_F32) hipLaunchKernelGGL(gather_rows_kernel<float>, grid, block, 0, stream, (const float*)src, lds_, rows, R, H, (float*)dst, ldd);
  else if (dtype == MV_BF16) hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)src, lds_, rows, R, H, (bf16_t*)dst, ldd);
  else if (dtype == MV_F16) hipLaunchKernelGGL(gather_rows_kernel<f16_t>, grid, block, 0, stream, (const f16_t*)src, lds_, rows, R, H, (f16_t*)dst, ldd);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}
extern "C" int mv_scatter_rows(int dtype, const void* src, int lds_, const int32_t* rows, int R, int H, void* dst, int ldd,
                               int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !rows || !dst || R <= 0 || H <= 0) return MV_E_ARG;
  if ((H & 3) || (lds_ & 3) || (ldd & 3)) return MV_E_SHAPE;
  dim3 grid((R + 3) / 4), block(256);
  if (dtype == MV_F32) hipLaunchKernelGGL(scatter_rows_kernel<float>, grid, block, 0, stream, (const float*)src, lds_, rows, R, H, (float*)dst, ldd, accumulate);
  else if (dtype == MV_BF16) hipLaunchKernelGGL(scatter_rows_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)src, lds_, rows, R, H, (bf16_t*)dst, ldd, accumulate);
  else if (dtype == MV_F16) hipLaunchKernelGGL(scatter_rows_kernel<f16_t>, grid, block, 0, stream, (const f16_t*)src, lds_, rows, R, H, (f16_t*)dst, ldd, accumulate);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// out[n] += sum_m x[m,n].  A block owns a strip of 128 (f32) / 256 (bf16) columns: 32 lanes x one 16-byte vector
// per row, 8 row-lanes walking a slice of the rows, LDS reduction over the row-lanes, one atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ldx, int M, int N, float* __restrict__ out,
                                                     const float* __restrict__ gscale) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[8][32 * VEC + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + tx) * VEC;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  const bool vec = (col + VEC <= N) && ((ldx % VEC) == 0) && ((((uintptr_t)x) & 15) == 0);
  if (col < N) {
    for (int r = r0 + ty; r < r1; r += 8) {
      const T* p = x + (size_t)r * ldx + col;
      if (vec) {
        if constexpr (sizeof(T) == 2) {
          typedef __attribute__((ext_vector_type(8))) T t8;
          const t8 v = *(const t8*)p;
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += (float)v[e];
        } else {
          const f32x4 v = *(const f32x4*)p;
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += v[e];
        }
      } else {
        for (int e = 0; e < VEC && col + e < N; ++e) acc[e] += ldf<T>(p + e);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[ty][tx * VEC + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < 32 * VEC; c += 256) {
    const int gc = blockIdx.x * 32 * VEC + c;
    if (gc < N) {
      float s_ = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s_ += red[k][c];
      atomicAdd(out + gc, gscale ? s_ * *gscale : s_);
    }
  }
}
__global__ void zero_f32_kernel(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

extern "C" int mv_colsum(int dtype, const void* x, int ldx, int M, int N, float* out, int accumulate, const float* grad_unscale_dev,
                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || M <= 0 || N <= 0 || ldx < N) return MV_E_ARG;
  if (!accumulate) {
    hipLaunchKernelGGL(zero_f32_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, out, (size_t)N);
    MV_CHECK_LAUNCH();
  }
  const int strip = (dtype == MV_F32) ? 128 : 256;
  const int xb = (N + strip - 1) / strip;
  int ysplit = (M + 255) / 256;
  const int want = (2048 + xb - 1) / xb;          // ~2048 blocks in flight
  if (ysplit > want) ysplit = want;
  if (ysplit < 1) ysplit = 1;
  dim3 grid(xb, ysplit), block(256);
  if (dtype == MV_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, stream, (const float*)x, ldx, M, N, out, grad_unscale_dev);
  else if (dtype == MV_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)x, ldx, M, N, out, grad_unscale_dev);
  else if (dtype == MV_F16) hipLaunchKernelGGL(colsum_kernel<f16_t>, grid, block, 0, stream, (const f16_t*)x, ldx, M, N, out, grad_unscale_dev);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// out[n] += [*gscale] * sum_p part[p, n]: folds the per-block partial column sums that the attention backward (and the dz GEMM's
// epilogue) write next to their outputs -- a bias gradient without a second pass over the [rows, N] matrix.
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, int P, int ld, int N, float* __restrict__ out,
                                                              const float* __restrict__ gscale) {
  // grid (ceil(N/64), row slices): 64 columns x 4 row-lanes per block, every row slice adds its share atomically -- the matrix is
  // a few MB, so the point is many short independent load chains, not bandwidth
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int per = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(P, p0 + per);
  float s_ = 0.f;
  if (n < N)
    for (int p = p0 + ty; p < p1; p += 4) s_ += part[(size_t)p * ld + n];
  red[ty][tx] = s_;
  __syncthreads();
  if (ty == 0 && n < N) {
    const float t = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
    atomicAdd(out + n, gscale ? t * *gscale : t);
  }
}
extern "C" int mv_colsum_partials(const float* part, int P, int ld, int N, float* out, const float* grad_unscale_dev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!part || !out || P <= 0 || N <= 0 || ld < N) return MV_E_ARG;
  const int xb = (N + 63) / 64;
  int ys = (512 + xb - 1) / xb;                 // ~512 blocks
  if (ys > (P + 7) / 8) ys = (P + 7) / 8;       // at least 8 rows per slice
  if (ys < 1) ys = 1;
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(xb, ys), dim3(256), 0, stream, part, P, ld, N, out, grad_unscale_dev);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// keep-mask of the counter-based dropout for linear indices 0..n-1 (test / inspection utility)
__global__ void dropout_mask_kernel(uint8_t* __restrict__ out, size_t n, DropCfg d) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned h = mv_hash32((unsigned)(i >> 1), d.k0, d.k1);
    out[i] = (d.thr == 0 || mv_keep(h, (int)(i & 1), d.thr)) ? 1 : 0;
  }
}
extern "C" int mv_dropout_mask(float p_drop, unsigned long long drop_key, size_t n, uint8_t* keep, float* scale_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!keep || n == 0) return MV_E_ARG;
  const DropCfg d = mv_make_drop(p_drop, drop_key);
  if (scale_out) *scale_out = d.inv_keep;     // host pointer: 1 / (1 - thr/65536)
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, stream, keep, n, d);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ c, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    st4<T>(c + 4 * i, ld4<T>(a + 4 * i) + ld4<T>(b + 4 * i));
}
extern "C" int mv_add(int dtype, const void* a, const void* b, void* c, size_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !b || !c || n == 0) return MV_E_ARG;
  if (n & 3) return MV_E_SHAPE;
  const size_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (dtype == MV_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)a, (const float*)b, (float*)c, n4);
  else if (dtype == MV_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)c, n4);
  else if (dtype == MV_F16) hipLaunchKernelGGL(add_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, (const f16_t*)a, (const f16_t*)b, (f16_t*)c, n4);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// out = dy * gelu'(z)   (mode 0)   |   out = dy * (1 - y*y)   (mode 1: tanh backward, z holds y)
template <typename T>
__global__ void dact_kernel(const T* __restrict__ dy, const T* __restrict__ z, T* __restrict__ out, size_t n4, int mode) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 d = ld4<T>(dy + 4 * i), zz = ld4<T>(z + 4 * i);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = mode == 0 ? d[e] * dgelu_erf(zz[e]) : d[e] * (1.0f - zz[e] * zz[e]);
    st4<T>(out + 4 * i, o);
  }
}
extern "C" int mv_dact(int dtype, int mode, const void* dy, const void* z, void* out, size_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dy || !z || !out || n == 0) return MV_E_ARG;
  if ((n & 3) || (mode != 0 && mode != 1)) return MV_E_SHAPE;
  const size_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (dtype == MV_F32) hipLaunchKernelGGL(dact_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)dy, (const float*)z, (float*)out, n4, mode);
  else if (dtype == MV_BF16) hipLaunchKernelGGL(dact_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)z, (bf16_t*)out, n4, mode);
  else if (dtype == MV_F16) hipLaunchKernelGGL(dact_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, (const f16_t*)dy, (const f16_t*)z, (f16_t*)out, n4, mode);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// 2-D cast with leading dimensions; columns cols..ldd-1 of dst are zero-filled
template <typename TS, typename TDs>
__global__ void cast2d_kernel(const TS* __restrict__ s, long long lds_, TDs* __restrict__ d, long long ldd, int rows, int cols) {
  const int r = blockIdx.y;
  const TS* sr = s + (size_t)r * lds_;
  TDs* dr = d + (size_t)r * ldd;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ldd; c += gridDim.x * blockDim.x)
    stf<TDs>(dr + c, c < cols ? ldf<TS>(sr + c) : 0.f);
}
extern "C" int mv_cast2d(const void* src, int src_dtype, long long lds_, void* dst, int dst_dtype, long long ldd, int rows, int cols,
                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || rows <= 0 || cols <= 0 || lds_ < cols || ldd < cols) return MV_E_ARG;
  if (rows > 65535 * 16) return MV_E_SHAPE;
  int bx = (int)((ldd + 255) / 256);
  if (bx > 32) bx = 32;
  for (int r0 = 0; r0 < rows; r0 += 65535) {
    const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
    dim3 grid(bx, nr), block(256);
    const char* sp = (const char*)src + (size_t)r0 * lds_ * mv_dtype_size(src_dtype);
    char* dp = (char*)dst + (size_t)r0 * ldd * mv_dtype_size(dst_dtype);
    if (src_dtype == MV_F32 && dst_dtype == MV_BF16) hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), grid, block, 0, stream, (const float*)sp, lds_, (bf16_t*)dp, ldd, nr, cols);
    else if (src_dtype == MV_BF16 && dst_dtype == MV_F32) hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), grid, block, 0, stream, (const bf16_t*)sp, lds_, (float*)dp, ldd, nr, cols);
    else if (src_dtype == MV_F32 && dst_dtype == MV_F32) hipLaunchKernelGGL((cast2d_kernel<float, float>), grid, block, 0, stream, (const float*)sp, lds_, (float*)dp, ldd, nr, cols);
    else if (src_dtype == MV_BF16 && dst_dtype == MV_BF16) hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), grid, block, 0, stream, (const bf16_t*)sp, lds_, (bf16_t*)dp, ldd, nr, cols);
    else if (src_dtype == MV_F32 && dst_dtype == MV_F16) hipLaunchKernelGGL((cast2d_kernel<float, f16_t>), grid, block, 0, stream, (const float*)sp, lds_, (f16_t*)dp, ldd, nr, cols);
    else if (src_dtype == MV_F16 && dst_dtype == MV_F32) hipLaunchKernelGGL((cast2d_kernel<f16_t, float>), grid, block, 0, stream, (const f16_t*)sp, lds_, (float*)dp, ldd, nr, cols);
    else if (src_dtype == MV_F16 && dst_dtype == MV_F16) hipLaunchKernelGGL((cast2d_kernel<f16_t, f16_t>), grid, block, 0, stream, (const f16_t*)sp, lds_, (f16_t*)dp, ldd, nr, cols);
    else return MV_E_DTYPE;
    MV_CHECK_LAUNCH();
  }
  return MV_OK;
}

// dst[c, r] = src[r, c]: 64x64 tiles through LDS (65-element pitch), both sides coalesced.  Used once per optimizer step
// to keep k-contiguous copies of the weights whose input-gradient GEMM would otherwise read them contraction-major.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, long long lds_, T* __restrict__ dst, long long ldd,
                                                        int rows, int cols) {
  __shared__ T tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty * 16 + i][tx] = src[(size_t)r * lds_ + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty * 16 + i, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * ldd + r] = tile[tx][ty * 16 + i];
  }
}
extern "C" int mv_transpose(int dtype, const void* src, long long lds_, void* dst, long long ldd, int rows, int cols, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || rows <= 0 || cols <= 0 || lds_ < cols || ldd < rows) return MV_E_ARG;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64), block(256);
  if (grid.y > 65535) return MV_E_SHAPE;
  if (dtype == MV_BF16 || dtype == MV_F16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd, rows, cols);   // 2-byte elements move as bits
  else if (dtype == MV_F32) hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, stream, (const float*)src, lds_, (float*)dst, ldd, rows, cols);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

template <typename TS, typename TDs>
__global__ void cast_kernel(const TS* __restrict__ s, TDs* __restrict__ d, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    st4<TDs>(d + 4 * i, ld4<TS>(s + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) stf<TDs>(d + n4 * 4 + threadIdx.x, ldf<TS>(s + n4 * 4 + threadIdx.x));
}
extern "C" int mv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, size_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || n == 0) return MV_E_ARG;
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (src_dtype == MV_F32 && dst_dtype == MV_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, stream, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == MV_BF16 && dst_dtype == MV_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == MV_F32 && dst_dtype == MV_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, stream, (const float*)src, (float*)dst, n);
  else if (src_dtype == MV_BF16 && dst_dtype == MV_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, n);
  else if (src_dtype == MV_F32 && dst_dtype == MV_F16) hipLaunchKernelGGL((cast_kernel<float, f16_t>), dim3(blocks), dim3(256), 0, stream, (const float*)src, (f16_t*)dst, n);
  else if (src_dtype == MV_F16 && dst_dtype == MV_F32) hipLaunchKernelGGL((cast_kernel<f16_t, float>), dim3(blocks), dim3(256), 0, stream, (const f16_t*)src, (float*)dst, n);
  else if (src_dtype == MV_F16 && dst_dtype == MV_BF16) hipLaunchKernelGGL((cast_kernel<f16_t, bf16_t>), dim3(blocks), dim3(256), 0, stream, (const f16_t*)src, (bf16_t*)dst, n);
  else if (src_dtype == MV_BF16 && dst_dtype == MV_F16) hipLaunchKernelGGL((cast_kernel<bf16_t, f16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, (f16_t*)dst, n);
  else if (src_dtype == MV_F16 && dst_dtype == MV_F16) hipLaunchKernelGGL((cast_kernel<f16_t, f16_t>), dim3(blocks), dim3(256), 0, stream, (const f16_t*)src, (f16_t*)dst, n);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// =========================================================================================
// fused HF AdamW over the flat parameter buffer  (train_origin.py:60,131)
// =========================================================================================
// Device-resident state of the dynamic loss scale of the f16-gradient path (f32 [8]); written only by mv_scaler_update:
//   [0] loss scale S the NEXT backward multiplies its loss gradients by      [1] 1 / S (what the f32 gradient writers multiply by)
//   [2] clean steps since S last changed      [3] 1.0 = the step just finished overflowed: the optimizer must skip it
//   [4] number of optimizer steps applied so far (t of the bias correction)   [5] number of skipped steps
//   [6] non-finite gradient elements counted since the last update (mv_count_nonfinite adds, mv_scaler_update clears)
#define MV_LS_SCALE 0
#define MV_LS_INV 1
#define MV_LS_GOOD 2
#define MV_LS_SKIP 3
#define MV_LS_T 4
#define MV_LS_NSKIP 5
#define MV_LS_NONFINITE 6

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, f16_t* __restrict__ shadow16,
                                                    size_t n, float step_size, float b1, float b2, float eps, float lr_wd, float gscale,
                                                    const float* __restrict__ state, float lr, int correct_bias) {
  if (state) {
    if (state[MV_LS_SKIP] != 0.f) return;             // overflowed step: parameters, moments and shadows stay as they are
    // t = steps applied so far + this one (mv_scaler_update has already counted it); HF's bias correction, in double like the host
    const double t = (double)state[MV_LS_T];
    step_size = correct_bias ? (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t))) : lr;
  }
  const size_t n4 = n / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pp = *(const f32x4*)(p + 4 * i), gg = *(const f32x4*)(g + 4 * i), mm = *(const f32x4*)(m + 4 * i), vv = *(const f32x4*)(v + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * gscale;
      mm[e] = b1 * mm[e] + (1.0f - b1) * ge;
      vv[e] = b2 * vv[e] + (1.0f - b2) * ge * ge;
      float x = pp[e] - step_size * (mm[e] / (sqrtf(vv[e]) + eps));
      x -= lr_wd * x;
      pp[e] = x;
    }
    *(f32x4*)(p + 4 * i) = pp; *(f32x4*)(m + 4 * i) = mm; *(f32x4*)(v + 4 * i) = vv;
    if (shadow) st4<bf16_t>(shadow + 4 * i, pp);
    if (shadow16) st4<f16_t>(shadow16 + 4 * i, pp);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float ge = g[i] * gscale;
    const float mm = b1 * m[i] + (1.0f - b1) * ge, vv = b2 * v[i] + (1.0f - b2) * ge * ge;
    float x = p[i] - step_size * (mm / (sqrtf(vv) + eps));
    x -= lr_wd * x;
    p[i] = x; m[i] = mm; v[i] = vv;
    if (shadow) shadow[i] = (bf16_t)x;
    if (shadow16) shadow16[i] = (f16_t)x;
  }
}

extern "C" int mv_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, void* shadow_f16, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, int correct_bias, float grad_scale,
                             const float* scaler_state, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!p || !g || !m || !v || n == 0 || (step < 1 && !scaler_state)) return MV_E_ARG;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return MV_E_SHAPE;
  if ((shadow_bf16 && (((uintptr_t)shadow_bf16) & 7)) || (shadow_f16 && (((uintptr_t)shadow_f16) & 7))) return MV_E_SHAPE;
  double ss = lr;
  if (correct_bias && !scaler_state) ss = ss * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow_bf16, (f16_t*)shadow_f16, n, (float)ss, beta1, beta2, eps,
                     lr * weight_decay, grad_scale, scaler_state, lr, correct_bias);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// =========================================================================================
// dynamic loss scale of the f16-gradient path (no counterpart in the fp32 reference: its gradients never leave f32)
// =========================================================================================
// counter += number of elements of x that are inf or nan (one pass at HBM speed over the flat gradient)
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ x, size_t n, float* __restrict__ counter) {
  const size_t n4 = n / 4;
  int bad = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 v = *(const u32x4*)(x + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) bad += ((v[e] & 0x7f800000u) == 0x7f800000u) ? 1 : 0;      // exponent all ones: inf or nan
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const unsigned u = __float_as_uint(x[n4 * 4 + threadIdx.x]);
    bad += ((u & 0x7f800000u) == 0x7f800000u) ? 1 : 0;
  }
  if (__any(bad != 0)) {
    const float s = wave_sum((float)bad);
    if ((threadIdx.x & 63) == 0) atomicAdd(counter, s);
  }
}
extern "C" int mv_count_nonfinite(const float* x, size_t n, float* counter, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !counter || n == 0) return MV_E_ARG;
  if (((uintptr_t)x) & 15) return MV_E_SHAPE;
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3(blocks), dim3(256), 0, stream, x, n, counter);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

__global__ void scaler_update_kernel(float* __restrict__ st, int growth_interval, float growth, float backoff, float max_scale, float min_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float scale = st[MV_LS_SCALE], good = st[MV_LS_GOOD];
  if (st[MV_LS_NONFINITE] > 0.f) {
    st[MV_LS_SKIP] = 1.f;
    st[MV_LS_NSKIP] += 1.f;
    scale = fmaxf(scale * backoff, min_scale);
    good = 0.f;
  } else {
    st[MV_LS_SKIP] = 0.f;
    st[MV_LS_T] += 1.f;
    good += 1.f;
    if (growth_interval > 0 && good >= (float)growth_interval) { scale = fminf(scale * growth, max_scale); good = 0.f; }
  }
  st[MV_LS_SCALE] = scale;
  st[MV_LS_INV] = 1.0f / scale;
  st[MV_LS_GOOD] = good;
  st[MV_LS_NONFINITE] = 0.f;
}
extern "C" int mv_scaler_update(float* state, int growth_interval, float growth, float backoff, float max_scale, float min_scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!state || growth < 1.f || backoff <= 0.f || backoff > 1.f || min_scale <= 0.f || max_scale < min_scale) return MV_E_ARG;
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, stream, state, growth_interval, growth, backoff, max_scale, min_scale);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
