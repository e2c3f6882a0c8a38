// 256-row LDS-DMA GEMM kernels, operand layout: dx = dy.W (weight contraction-major).  See mv_gemm_ring.h.
#include "mv_gemm_ring.h"

int mv_launch_ring_nn(const GemmArgs& p, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream) {
  dim3 grid(tiles, splitk);
  if (f16) {
    if (variant == 10) LAUNCH_RING_MI(false, true, 4, 4, 2, 2, true, 10);               // 320 x 256 tiles (one round of CUs: gemm_route)
    else LAUNCH_RING(false, true, 4, 4, 2, 2, true);
  } else {
    if (variant == 24) LAUNCH_PRING(false, true, 4, 4, 2, false);
    else if (variant == 10) LAUNCH_RING_MI(false, true, 4, 4, 2, 2, false, 10);
    else LAUNCH_RING(false, true, 4, 4, 2, 2, false);
  }
  return MV_OK;
}
