// 256-row LDS-DMA GEMM kernels, operand layout dW = dy^T.x, with 32-deep stages x4 (three stages = 96 KiB in flight instead of one
// 64-KiB stage): the weight-gradient GEMMs stream both operands from HBM once (no reuse across K), so what bounds a K-step is the
// fetch latency that the ring can cover.  See mv_gemm_ring.h.
#include "mv_gemm_ring.h"

int mv_launch_ring_tn4(const GemmArgs& p, bool f16, int tiles, int splitk, hipStream_t stream) {
  dim3 grid(tiles, splitk);
  if (f16) LAUNCH_RING(true, true, 4, 4, 4, 1, true);
  else LAUNCH_RING(true, true, 4, 4, 4, 1, false);
  return MV_OK;
}
