#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// each wave adds 64 consecutive floats; `passes` blocks hit the same addresses (like key-blocks adding their dQ partials)
__global__ void atom_f32(float* dst, size_t n, int passes) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t per = n;            // elements per pass
  const size_t idx = i % per;
  if (i < per * passes) atomicAdd(dst + idx, 1.0f);
}
__global__ void atom_f32x4(float* dst, size_t n, int passes) {   // each lane adds 4 consecutive floats (16 B), like an f32x4 accumulator row
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t per = n / 4;
  const size_t idx = (i % per) * 4;
  if (i < per * passes) {
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dst + idx + e, 1.0f);
  }
}
__global__ void plain_store(float* dst, size_t n, int passes) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n * passes) dst[i % n] = 1.0f;
}
int main() {
  const size_t n = 25483ull * 768;     // one layer's dQ
  float* d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 3; ++which) for (int passes : {1, 4}) {
    const size_t work = (which == 1 ? n / 4 : n) * passes;
    dim3 grid((unsigned)((work + 255) / 256));
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      if (which == 0) atom_f32<<<grid, 256>>>(d, n, passes);
      else if (which == 1) atom_f32x4<<<grid, 256>>>(d, n, passes);
      else plain_store<<<grid, 256>>>(d, n, passes);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s passes=%d: %.1f us  (%.1f G elem/s)\n", which == 0 ? "atomicAdd f32 lane-contiguous" : which == 1 ? "atomicAdd 4 x f32 per lane   " : "plain store                  ", passes, ms * 1e3, n * passes / ms / 1e6);
  }
  return 0;
}
