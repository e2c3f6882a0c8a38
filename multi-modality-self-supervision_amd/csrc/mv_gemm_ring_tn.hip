// 256-row LDS-DMA GEMM kernels, operand layout: dW = dy^T.x (both operands contraction-major).  See mv_gemm_ring.h.
#include "mv_gemm_ring.h"

int mv_launch_ring_tn(const GemmArgs& p, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream) {
  dim3 grid(tiles, splitk);
  if (f16) {
    if (variant == 2) LAUNCH_RING(true, true, 4, 2, 3, 1, true);
    else if (variant == 24) LAUNCH_PRING(true, true, 4, 4, 2, true);
    else LAUNCH_RING(true, true, 4, 4, 2, 2, true);
  } else {
    if (variant == 24) LAUNCH_PRING(true, true, 4, 4, 2, false);
    else LAUNCH_RING(true, true, 4, 4, 2, 2, false);
  }
  return MV_OK;
}
