// Region-feature extractor support: the ResNet-50 trunk of models/image.py:46-69 runs as NHWC
// matrices through mv_gemm (a 1x1 convolution IS a GEMM over [B*H*W, C]; 3x3 / 7x7 / strided ones go through the patch
// gather below), with BatchNorm + ReLU (+ residual) as one row-wise kernel.  The reference never back-propagates into
// the CNN (cxrbert_origin.py:66-70 unfreezes `children()[5:]` of a module that has a single child), so only the forward
// exists -- in both BatchNorm modes: batch statistics (model.train()) and running statistics (eval()).
// All kernels here are HBM-bound byte movers: coalesced 16-byte accesses along the channel dimension, no LDS tiling.
#include "mv_common.h"

namespace {

// [B, C, H, W] f32 -> [B, H, W, Cp] (Cp >= C, pad channels zero); one thread per output pixel-channel group
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W, int Cp) {
  const size_t n = (size_t)B * H * W * Cp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const size_t p = i / Cp;                 // b*H*W + y*W + x
    const size_t hw = (size_t)H * W;
    const size_t b = p / hw, yx = p - b * hw;
    const float v = c < C ? src[(b * C + c) * hw + yx] : 0.f;
    stf<T>(dst + i, v);
  }
}

// patch gather: dst[(b, oy, ox), (ky, kx, c)] = src[b, oy*s - pad + ky, ox*s - pad + kx, c] (0 outside), row pitch ldk >= kh*kw*C
// (columns kh*kw*C .. ldk-1 are zeroed: the GEMM requires a zero-padded contraction tail).  VEC = channels per thread.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C,
                                                     int Ho, int Wo, int kh, int kw, int stride, int pad, int ldk) {
  const int kc = kh * kw * C;
  const int groups = (ldk + VEC - 1) / VEC;            // VEC-wide column groups per output row
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int gcol = (int)(i % groups);
    const size_t row = i / groups;
    const int col = gcol * VEC;
    const int ox = (int)(row % Wo);
    const int oy = (int)((row / Wo) % Ho);
    const size_t b = row / ((size_t)Wo * Ho);
    T* out = dst + row * (size_t)ldk + col;
    if (VEC == 8) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (col < kc) {                                   // C % 8 == 0: the 8 columns share (ky, kx)
        const int tap = col / C, c = col - tap * C;
        const int ky = tap / kw, kx = tap - ky * kw;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const u32x4*)(src + ((b * H + iy) * (size_t)W + ix) * C + c);
      }
      *(u32x4*)out = v;
    } else {
      for (int e = 0; e < VEC && col + e < ldk; ++e) {
        const int cc = col + e;
        float v = 0.f;
        if (cc < kc) {
          const int tap = cc / C, c = cc - tap * C;
          const int ky = tap / kw, kx = tap - ky * kw;
          const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ldf<T>(src + ((b * H + iy) * (size_t)W + ix) * C + c);
        }
        stf<T>(out + e, v);
      }
    }
  }
}

// per-column sum and sum of squares over the rows of x [rows, C] (f32 accumulators, atomics into stats[2][C]);
// block = 16 column groups of 4 (64 columns) x 16 row lanes, each block walks a slab of rows, two rows in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* __restrict__ x, int ldx, int rows, int C, float* __restrict__ stats,
                                                        int rows_per_block) {
  __shared__ f32x4 s1[16][16], s2[16][16];
  const int cg = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cg * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, q0 = a0, a1 = a0, q1 = a0;
  if (c < C) {
    int r = r0 + ry;
    for (; r + 16 < r1; r += 32) {
      const f32x4 v0 = ld4<T>(x + (size_t)r * ldx + c), v1 = ld4<T>(x + (size_t)(r + 16) * ldx + c);
      a0 += v0; q0 += v0 * v0; a1 += v1; q1 += v1 * v1;
    }
    if (r < r1) { const f32x4 v0 = ld4<T>(x + (size_t)r * ldx + c); a0 += v0; q0 += v0 * v0; }
  }
  s1[ry][cg] = a0 + a1; s2[ry][cg] = q0 + q1;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int col = blockIdx.x * 64 + threadIdx.x, g = threadIdx.x >> 2, e = threadIdx.x & 3;
    if (col < C) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { a += s1[i][g][e]; q += s2[i][g][e]; }
      atomicAdd(stats + col, a);
      atomicAdd(stats + C + col, q);
    }
  }
}

// BatchNorm statistics from the column sums: mean, rstd of this batch (biased variance), and -- when running buffers are
// given -- the momentum update of the running estimates with the UNBIASED variance (nn.BatchNorm2d train())
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int C, float rows, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                   float* __restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float m = stats[c] / rows;
  const float var = fmaxf(stats[C + c] / rows - m * m, 0.f);
  mean[c] = m;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (rows / fmaxf(rows - 1.f, 1.f));
  }
}

// y = (x - mean) * rstd * gamma + beta (+ residual) (-> ReLU); 4 channels per thread
template <typename TX, typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const TX* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const T* __restrict__ res, T* __restrict__ y, size_t rows, int C, int relu) {
  const int c4n = C / 4;
  const size_t total = rows * c4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const size_t off = (i / c4n) * (size_t)C + c;
    const f32x4 v = ld4<TX>(x + off), m = *(const f32x4*)(mean + c), r = *(const f32x4*)(rstd + c), g = *(const f32x4*)(gamma + c),
                bt = *(const f32x4*)(beta + c);
    f32x4 o = (v - m) * r * g + bt;
    if (res) o += ld4<T>(res + off);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    st4<T>(y + off, o);
  }
}

// 3x3 / stride 2 / pad 1 max pooling over NHWC (torchvision ResNet stem), 8 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C,
                                                           int Ho, int Wo) {
  const int cg = C / 4;
  const size_t total = (size_t)B * Ho * Wo * cg;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cg) * 4;
    const size_t p = i / cg;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const size_t b = p / ((size_t)Wo * Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          const f32x4 v = ld4<T>(x + ((b * H + iy) * (size_t)W + ix) * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
      }
    st4<T>(y + p * (size_t)C + c, m);
  }
}

inline int grid_for(size_t n, int per_block = 256, int cap = 16384) {
  size_t b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

extern "C" int mv_nchw_to_nhwc(const float* src, void* dst, int dst_dtype, int B, int C, int H, int W, int Cp, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) return MV_E_ARG;
  const size_t n = (size_t)B * H * W * Cp;
  if (dst_dtype == MV_BF16) nchw_to_nhwc_kernel<bf16_t><<<grid_for(n), 256, 0, stream>>>(src, (bf16_t*)dst, B, C, H, W, Cp);
  else if (dst_dtype == MV_F32) nchw_to_nhwc_kernel<float><<<grid_for(n), 256, 0, stream>>>(src, (float*)dst, B, C, H, W, Cp);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_im2col(int dtype, const void* src, int B, int H, int W, int C, int kh, int kw, int stride, int pad, void* dst,
                         int ldk, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return MV_E_ARG;
  if (ldk < kh * kw * C) return MV_E_SHAPE;
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return MV_E_SHAPE;
  const bool vec = dtype == MV_BF16 && (C & 7) == 0 && (ldk & 7) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  const size_t rows = (size_t)B * Ho * Wo;
  if (dtype == MV_BF16) {
    if (vec) im2col_kernel<bf16_t, 8><<<grid_for(rows * (ldk / 8)), 256, 0, stream>>>((const bf16_t*)src, (bf16_t*)dst, B, H, W, C, Ho, Wo, kh, kw, stride, pad, ldk);
    else im2col_kernel<bf16_t, 4><<<grid_for(rows * ((ldk + 3) / 4)), 256, 0, stream>>>((const bf16_t*)src, (bf16_t*)dst, B, H, W, C, Ho, Wo, kh, kw, stride, pad, ldk);
  } else if (dtype == MV_F32) {
    im2col_kernel<float, 4><<<grid_for(rows * ((ldk + 3) / 4)), 256, 0, stream>>>((const float*)src, (float*)dst, B, H, W, C, Ho, Wo, kh, kw, stride, pad, ldk);
  } else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_col_stats(int dtype, const void* x, int ldx, int rows, int C, float* stats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !stats || rows <= 0 || C <= 0 || ldx < C) return MV_E_ARG;
  if ((C & 3) || (ldx & 3)) return MV_E_SHAPE;
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * 2 * (size_t)C, stream);
  if (e != hipSuccess) return (int)e;
  int slabs = (rows + 511) / 512;
  if (slabs > 2048) slabs = 2048;
  const int rpb = (rows + slabs - 1) / slabs;
  dim3 grid((C + 63) / 64, (rows + rpb - 1) / rpb), block(256);
  if (dtype == MV_BF16) col_stats_kernel<bf16_t><<<grid, block, 0, stream>>>((const bf16_t*)x, ldx, rows, C, stats, rpb);
  else if (dtype == MV_F32) col_stats_kernel<float><<<grid, block, 0, stream>>>((const float*)x, ldx, rows, C, stats, rpb);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_bn_finalize(const float* stats, int C, long long rows, float eps, float momentum, float* mean, float* rstd,
                              float* running_mean, float* running_var, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!stats || !mean || !rstd || C <= 0 || rows <= 0) return MV_E_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return MV_E_ARG;
  bn_finalize_kernel<<<(C + 255) / 256, 256, 0, stream>>>(stats, C, (float)rows, eps, momentum, mean, rstd, running_mean, running_var);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_bn_act(int dtype, const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         const void* residual, void* y, long long rows, int C, int relu, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !mean || !rstd || !gamma || !beta || !y || rows <= 0 || C <= 0) return MV_E_ARG;
  if (C & 3) return MV_E_SHAPE;
  const size_t n = (size_t)rows * (C / 4);
  if (dtype == MV_BF16 && x_dtype == MV_BF16) bn_act_kernel<bf16_t, bf16_t><<<grid_for(n), 256, 0, stream>>>((const bf16_t*)x, mean, rstd, gamma, beta, (const bf16_t*)residual, (bf16_t*)y, (size_t)rows, C, relu);
  else if (dtype == MV_BF16 && x_dtype == MV_F32) bn_act_kernel<float, bf16_t><<<grid_for(n), 256, 0, stream>>>((const float*)x, mean, rstd, gamma, beta, (const bf16_t*)residual, (bf16_t*)y, (size_t)rows, C, relu);
  else if (dtype == MV_F32 && x_dtype == MV_F32) bn_act_kernel<float, float><<<grid_for(n), 256, 0, stream>>>((const float*)x, mean, rstd, gamma, beta, (const float*)residual, (float*)y, (size_t)rows, C, relu);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_maxpool3x3s2(int dtype, const void* x, void* y, int B, int H, int W, int C, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MV_E_ARG;
  if (C & 3) return MV_E_SHAPE;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t n = (size_t)B * Ho * Wo * (C / 4);
  if (dtype == MV_BF16) maxpool3x3s2_kernel<bf16_t><<<grid_for(n), 256, 0, stream>>>((const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo);
  else if (dtype == MV_F32) maxpool3x3s2_kernel<float><<<grid_for(n), 256, 0, stream>>>((const float*)x, (float*)y, B, H, W, C, Ho, Wo);
  else return MV_E_DTYPE;
  MV_CHECK_LAUNCH();
  return MV_OK;
}
