// The 256-row LDS-DMA GEMM kernels (ring and persistent ring); see mv_gemm_common.h / mv_gemm.hip.  Templates only: the
// instantiations and their launchers live in mv_gemm_ring_{nt,nn,tn,tnn}.hip, one translation unit per operand layout.
#pragma once
#include "mv_gemm_common.h"

// One ring stage's MFMAs for the layouts with a contraction-major operand (fragments through `ds_read_b64_tr_b16`).  hipcc reads each
// A fragment right before the MFMAs that consume it (register pressure), so every group of NJ MFMAs starts with a full LDS round
// trip that the second wave of the SIMD only partly covers (the loop alone ran at 57-64 % of the MFMA rate, profiles/r03_notes.txt).
// Here the next A fragment is requested BEFORE the current group's MFMAs (4 more registers); the scheduling fences pin that order.
template <bool TA, bool TB, bool BP512, int NJ, int KS, bool F16>
__device__ __forceinline__ void g2_stage_mma_tr(const char* tA, const char* tB, int wm, int wn, int l15, int lq, f32x4 (&acc)[8][NJ]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 fb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[j] = g2_frag<TB, BP512, KS>(tB, wn + j * 16, l15, lq, ks);
    bf16x8 cur = g2_frag<TA, true, KS>(tA, wm, l15, lq, ks);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bf16x8 nxt = cur;
      if (i + 1 < 8) nxt = g2_frag<TA, true, KS>(tA, wm + (i + 1) * 16, l15, lq, ks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mma16<F16>(fb[j], cur, acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }
}

// MI = 16-row accumulator groups per wave: 8 -> the 256-row tile; 10 -> a 320-row tile (2 x 160 rows per wave pair, 160 accumulator registers,
// 72 KiB stages).  Why 320: an output of 768 columns over ~25,500 packed rows is 300 tiles of 256 x 256 = 1.17 rounds of the 256 CUs (the second
// round runs 44 tiles on an otherwise idle chip), and 1.56 rounds of 768 slots on the 128 x 128 kernel; as 320 x 256 tiles it is 240 tiles: ONE
// round with 94 % of the CUs busy (mv_gemm.hip: gemm_route picks the row count that makes a single round).
template <bool TA, bool TB, int NJ, int WN, int NSTAGE, int KS, bool F16 = false, int MI = 8>
__global__ __launch_bounds__(128 * WN, 2) void gemm_ring_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 2 * WN;                      // waves per block
  constexpr int BN = WN * 16 * NJ;
  constexpr bool BP512 = BN > 128;                // pitch of a contraction-major B image
  constexpr int BKS = G2_BK * KS;                 // contraction depth of one stage (32 or 64)
  constexpr int BM = 32 * MI;                     // rows of the tile: two wave rows of 16 * MI
  constexpr int A_BYTES = BM * 64 * KS;
  constexpr int B_BYTES = (BP512 ? 16384 : 8192) * KS;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int LPS = (A_BYTES + B_BYTES) / 1024 / NW;   // LDS-DMA instructions per wave per stage
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int wm = (wid / WN) * (16 * MI), wn = (wid % WN) * (16 * NJ);

  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, in = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  int tm, tn;
  {
    // row tiles per group of the rasterisation (debug bits 32 / 64 / 128: 4 / 16 / 12 instead of 8 -- timing experiments only)
    const int GM = (p.dbg & 32) ? 4 : (p.dbg & 64) ? 16 : (p.dbg & 128) ? 12 : 8;
    const int per_group = GM * tiles_n;
    const int group = bid / per_group, rem = bid - group * per_group;
    const int gm = min(GM, tiles_m - group * GM);
    tm = group * GM + rem % gm;
    tn = rem / gm;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int split = blockIdx.y;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nst = (kend - kbeg + BKS - 1) / BKS;

  const dma_rsrc_t rsA = dma_rsrc(p.A, p.bytesA);
  const dma_rsrc_t rsB = dma_rsrc(p.B, p.bytesB);
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define G2_ISSUE(S_)                                                                                        \
  do {                                                                                                      \
    char* st__ = smem + ((S_) % NSTAGE) * STAGE;                                                            \
    const int k0__ = kbeg + (S_) * BKS;                                                                     \
    g2_issue<TA, true, A_BYTES / 1024, NW, KS>(rsA, p.bytesA, p.lda, m0, p.M, BM, k0__, kend, st__, wid, lane, p.dbg); \
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
    _Pragma("unroll") for (int i = 0; i < MI; ++i) FA_[i] = g2_frag<TA, true, KS>(tA__, wm + i * 16, l15, lq, KS_);   \
  } while (0)
#define G2_FRAGS(FA_, FB_, S_) G2_FRAGS_K(FA_, FB_, S_, 0)
#define G2_MMA(FA_, FB_)                                                                         \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                               \
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
      if constexpr (KS == 2 && !TA && !TB && MI == 8) {
        // all 24 fragment reads of the 64-deep stage are issued before its first MFMA: the MFMAs then wait on a
        // counted lgkmcnt that only the first reads hold up, instead of a read-wait-MFMA ping-pong per 2 fragments
        bf16x8 fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
        G2_FRAGS_K(fa0, fb0, s, 0);
        G2_FRAGS_K(fa1, fb1, s, 1);
        __builtin_amdgcn_sched_barrier(0);
        G2_MMA(fa0, fb0);
        G2_MMA(fa1, fb1);
      } else {       // (g2_stage_mma_tr spills in this kernel -- 3-4x slower, measured; the persistent form below takes it)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 fa[MI], fb[NJ];
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
    for (int i = 0; i < MI; ++i)
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
  constexpr int G2_NI = MI;                       // 16-row groups per wave the epilogue walks
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
template <bool TA, bool TB, int NJ, int WN, int NSTAGE, bool F16 = false>
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

  const dma_rsrc_t rsA = dma_rsrc(p.A, p.bytesA);
  const dma_rsrc_t rsB = dma_rsrc(p.B, p.bytesB);

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
      if constexpr (TA && TB) {      // (the NN / TNN forms spill with the extra fragment: measured 2.3x slower)
        g2_stage_mma_tr<TA, TB, BP512, NJ, KS, F16>(tA, tB, wm, wn, l15, lq, acc);
      } else {
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
            for (int j = 0; j < NJ; ++j) acc[i][j] = mma16<F16>(fb[j], fa[i], acc[i][j]);
        }
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
    constexpr int G2_NI = 8;
    if (p.splitk > 1) { G2_EPI_BODY(-1) }
    else { MV_EPI_SWITCH(p.epi, G2_EPI_BODY) }
    // whole tile inside the matrix and vector stores: every one of the 32 row-group stores above was issued
    const bool full = (m0 + G2_BM <= p.M) && (n0 + BN <= p.N) && ((p.N & 3) == 0) && (p.splitk > 1 || p.vec_ok);
    // (16-bit outputs leave in 16-byte pieces, 16 stores per output: below the bound, so the next unit's first wait drains them)
    epi_ops = (full && (p.splitk > 1 || p.c_dtype == MV_F32)) ? EPI_OPS : 0;
  }
}



#define LAUNCH_PRING(TA_, TB_, NJ_, WN_, NS_, F16_)                                                                  \
  do {                                                                                                               \
    constexpr size_t shm = (size_t)(NS_) * 2 * (16384 + ((WN_) * 16 * (NJ_) > 128 ? 16384 : 8192));                  \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_pring_kernel<TA_, TB_, NJ_, WN_, NS_, F16_>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                               \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    const int units = tiles * splitk;                                                                                \
    hipLaunchKernelGGL((gemm_pring_kernel<TA_, TB_, NJ_, WN_, NS_, F16_>), dim3(units < n_cu ? units : n_cu),        \
                       dim3(128 * (WN_)), shm, stream, p, units, tiles);                                             \
  } while (0)
#define LAUNCH_RING_MI(TA_, TB_, NJ_, WN_, NS_, KS_, F16_, MI_)                                                      \
  do {                                                                                                               \
    constexpr size_t shm = (size_t)(NS_) * (KS_) * (2048 * (MI_) + ((WN_) * 16 * (NJ_) > 128 ? 16384 : 8192));        \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<TA_, TB_, NJ_, WN_, NS_, KS_, F16_, MI_>,              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                               \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_ring_kernel<TA_, TB_, NJ_, WN_, NS_, KS_, F16_, MI_>), grid, dim3(128 * (WN_)), shm, stream, p); \
  } while (0)
#define LAUNCH_RING(TA_, TB_, NJ_, WN_, NS_, KS_, F16_) LAUNCH_RING_MI(TA_, TB_, NJ_, WN_, NS_, KS_, F16_, 8)
