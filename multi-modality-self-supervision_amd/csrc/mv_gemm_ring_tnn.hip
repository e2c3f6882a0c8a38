// 256-row LDS-DMA GEMM kernels, operand layout: A^T.B^T (bf16 only; not used by the training engine).  See mv_gemm_ring.h.
#include "mv_gemm_ring.h"

int mv_launch_ring_tnn(const GemmArgs& p, bool f16, int variant, int tiles, int splitk, int n_cu, hipStream_t stream) {
  dim3 grid(tiles, splitk);
  if (f16) return MV_E_DTYPE;
  {
    if (variant == 24) LAUNCH_PRING(true, false, 4, 4, 2, false);
    else LAUNCH_RING(true, false, 4, 4, 2, 2, false);
  }
  return MV_OK;
}
