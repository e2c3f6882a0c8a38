// Fused-mask multi-head attention for the CXRBERT hot path on gfx950.
//
// Replaces HF BertSelfAttention's  softmax(q.k^T/sqrt(dh) + (1-mask)*-10000) . v  and its
// backward (spec: Downstream_task/report_generation_and_vqa/sc/pytorch_pretrained_bert/
// model.py:301-320) together with CXRBertEncoder.get_extended_attn_mask
// (models/cxrbert_origin.py:75-85).  The [B,L,L] int64 mask of data/dataset_origin.py:138-176
// is packed once per batch into bits + a per-64x64-tile class (mask_pack kernels) and the
// additive -10000 is applied on the fly; the [B,A,L,L] score tensor never exists in HBM.
//
// MFMA kernels (bf16, dh = 64), all on v_mfma_f32_32x32x16_bf16, flash-style:
//   forward : one wave = 32 queries ON THE LANES (S^T = K.Q^T, O^T = V^T.P^T) so that the
//             running max / sum / rescale are lane-local; P^T goes from the accumulator
//             straight into the next MFMA's B operand (no LDS round trip).
//   dQ      : same orientation (dS^T is lane-local in q, dQ^T = K^T.dS^T).
//   dK,dV   : one wave = 32 keys on the lanes (S = Q.K^T, dV^T = dO^T.P, dK^T = Q^T.dS).
// K/V (or Q/dO) tiles of 64 rows x 64 dh live in LDS in ONE image that serves both the
// row reads (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16), conflict-free for both.
// The VALU kernels below them are the exact-fp32 path and the on-GPU cross-check.
#include "mv_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
// Timing experiments on the forward kernel (profiles/tools/attn_ablate.sh builds variant libraries with -DATT_ABL=<bits>; results are
// WRONG by construction): 1 no per-tile barrier, 2 no exponentials, 4 dropout masks from constants instead of scalar loads,
// 8 no cross-half maximum, 16 no P.V products, 32 no Q.K products, 64 no LDS-DMA (tiles never loaded).  0 in the product build.
#ifndef ATT_ABL
#define ATT_ABL 0
#endif
#ifndef ATT_EARLY_MASKS      // forward: 0 = the select masks are loaded where they are used, 1 = the first half's at the top of the tile, 2 = both
#define ATT_EARLY_MASKS 0
#endif
// bit 4096: phase profile -- every wave reads the shader clock at the phase boundaries of its tile loop, thread 0 of each block adds the
// phase totals to g_att_prof[kernel][phase] (0 prologue, 1 wait for the tile's LDS-DMA, 2 barrier, 3 tile body, 4 issue of the next tile,
// 5 epilogue, 7 = number of blocks); mv_debug_attn_prof() returns and clears them (profiles/tools/attn_phase.py).
#if ATT_ABL & 4096
__device__ unsigned long long g_att_prof[3][8];
__device__ unsigned long long g_att_tile[3][8];       // inside the tile body (forward: 0 score MFMAs, 1 row maximum + shuffle, 2 exponentials + sums, 3 dropout selects, 4 P.V)
#define PROF_DECL unsigned long long pt_ = clock64(), pacc_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long ptile_[6] = {0, 0, 0, 0, 0, 0}
#define PROF_TILE_ARGS , unsigned long long (&ptile_)[6]
#define PROF_TILE_PASS , ptile_
#define PROF_T0 unsigned long long tt_ = clock64()
#define PROF_T(i, dep) do { asm volatile("s_nop 0" ::"v"(dep)); const unsigned long long n_ = clock64(); ptile_[i] += n_ - tt_; tt_ = n_; } while (0)
#define PROF_MARK(i) do { const unsigned long long n_ = clock64(); pacc_[i] += n_ - pt_; pt_ = n_; } while (0)
#define PROF_FLUSH(k) do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 6; ++i_) { atomicAdd(&g_att_prof[k][i_], pacc_[i_]); atomicAdd(&g_att_tile[k][i_], ptile_[i_]); } atomicAdd(&g_att_prof[k][7], 1ull); } } while (0)
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_FLUSH(k)
#define PROF_TILE_ARGS
#define PROF_TILE_PASS
#define PROF_T0
#define PROF_T(i, dep)
#endif
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define MASK_ADD (-10000.0f)

// =========================================================================================
// mask packing
// =========================================================================================
__global__ void mask_pack_kernel(const int64_t* __restrict__ mask, int ndim, int B, int L, int W, uint32_t* __restrict__ bits) {
  // one wave per (b, i) row
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= B * L) return;
  const int b = wave / L, i = wave - b * L;
  const int64_t* row = (ndim == 3) ? mask + ((size_t)b * L + i) * L : mask + (size_t)b * L;
  uint32_t* out = bits + ((size_t)b * L + i) * W;
  for (int j0 = 0; j0 < L; j0 += 64) {
    const int j = j0 + lane;
    const bool v = (j < L) && (row[j] != 0);
    const unsigned long long bal = __ballot(v);
    if (lane == 0) {
      out[j0 >> 5] = (uint32_t)bal;
      if ((j0 >> 5) + 1 < W) out[(j0 >> 5) + 1] = (uint32_t)(bal >> 32);
    }
  }
}

// one wave per (b, tq): class of every 64x64 tile of that query-tile row
__global__ void mask_tileinfo_kernel(const uint32_t* __restrict__ bits, int B, int L, int W, int T, uint8_t* __restrict__ info) {
  const int b = blockIdx.x / T, tq = blockIdx.x - b * T;
  const int lane = threadIdx.x;
  const int i = tq * 64 + lane;
  const bool rv = i < L;
  const uint32_t* row = bits + ((size_t)b * L + (rv ? i : 0)) * W;
  bool row_any = false;
  unsigned long long all0_m = 0, all1_m = 0;  // per-tile flags (T <= 64)
  for (int tk = 0; tk < T; ++tk) {
    const int w0 = 2 * tk;
    unsigned long long w = 0;
    if (rv) {
      w = row[w0];
      if (w0 + 1 < W) w |= ((unsigned long long)row[w0 + 1]) << 32;
    }
    const int ncol = min(64, L - tk * 64);
    const unsigned long long full = (ncol == 64) ? ~0ull : ((1ull << ncol) - 1ull);
    w &= full;
    row_any |= (w != 0);
    const bool a0 = __all(!rv || w == 0);
    const bool a1 = __all(!rv || w == full);
    if (a0) all0_m |= (1ull << tk);
    if (a1) all1_m |= (1ull << tk);
  }
  const bool rows_ok = __all(!rv || row_any);
  if (lane < T) {
    const int tk = lane;
    uint8_t c = 2;
    if ((all1_m >> tk) & 1) c = 1;
    else if (((all0_m >> tk) & 1) && rows_ok) c = 0;
    info[((size_t)b * T + tq) * T + tk] = c;
  }
}

// bits straight from per-sample descriptors {family, n2, vl}: the closed forms of SURVEY Appendix B
// (data/dataset_origin.py:138-176).  family: 0 full, 1 s2s, 2 BAR, 3 non-cross, 4 1-D (== full as a [L,L] matrix)
__global__ void mask_build_kernel(const int32_t* __restrict__ desc, int B, int L, int W, uint32_t* __restrict__ bits) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= B * L) return;
  const int b = wave / L, i = wave - b * L;
  const int fam = desc[3 * b], n2 = desc[3 * b + 1], vl = desc[3 * b + 2];
  uint32_t* out = bits + ((size_t)b * L + i) * W;
  for (int j0 = 0; j0 < L; j0 += 64) {
    const int j = j0 + lane;
    bool v;
    switch (fam) {
      case 1: v = (j < n2) || (i >= n2 && j >= n2 && j <= i); break;
      case 2: v = (i < n2) || (j < n2) || (j <= i); break;
      case 3: v = (i < n2) == (j < n2); break;
      default: v = j < vl; break;
    }
    v = v && (j < L);
    const unsigned long long bal = __ballot(v);
    if (lane == 0) {
      out[j0 >> 5] = (uint32_t)bal;
      if ((j0 >> 5) + 1 < W) out[(j0 >> 5) + 1] = (uint32_t)(bal >> 32);
    }
  }
}

extern "C" int mv_mask_build(const int32_t* desc, int B, int L, uint32_t* bits, uint8_t* tileinfo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!desc || !bits || !tileinfo || B <= 0 || L <= 0) return MV_E_ARG;
  const int W = (L + 31) / 32, T = (L + 63) / 64;
  if (T > 64) return MV_E_SHAPE;
  const long long waves = (long long)B * L;
  hipLaunchKernelGGL(mask_build_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, desc, B, L, W, bits);
  MV_CHECK_LAUNCH();
  hipLaunchKernelGGL(mask_tileinfo_kernel, dim3(B * T), dim3(64), 0, stream, bits, B, L, W, T, tileinfo);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_mask_pack(const int64_t* mask, int mask_ndim, int B, int L, uint32_t* bits, uint8_t* tileinfo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!mask || !bits || !tileinfo || B <= 0 || L <= 0) return MV_E_ARG;
  if (mask_ndim != 2 && mask_ndim != 3) return MV_E_SHAPE;   // NotImplementedError in cxrbert_origin.py:80-81
  const int W = (L + 31) / 32, T = (L + 63) / 64;
  if (T > 64) return MV_E_SHAPE;
  const long long waves = (long long)B * L;
  hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, mask, mask_ndim, B, L, W, bits);
  MV_CHECK_LAUNCH();
  hipLaunchKernelGGL(mask_tileinfo_kernel, dim3(B * T), dim3(64), 0, stream, bits, B, L, W, T, tileinfo);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// =========================================================================================
// MFMA kernels (bf16, dh = 64)
// =========================================================================================
struct AttnArgs {
  const bf16_t* qkv; const bf16_t* ctx; const bf16_t* dctx; bf16_t* out; bf16_t* dqkv;
  bf16_t* out2;   // forward, f16 operands only (nullable): bf16 copy of the context for the backward's gradient products
  const uint32_t* bits; const uint8_t* info; float* lse; const float* lse_in; const float* delta; float* delta_out;
  int B, L, A, H, W, T;
  float scale;
  unsigned bytes_qkv, bytes_ctx;
  unsigned bytes_stat, bytes_bits;     // sizes of lse / delta ([B,A,L] f32) and of the mask words, for the LDS-DMA descriptors
  // attention-probability dropout (drop_on): the mask is a tensor of precomputed keep-bits (mv_attn_dropmask): per (b*A + h,
  // 32-query block, 64-key tile) 32 x u64, word 16*kk + r, bit l = keep(query 32*qb + (l & 31), key 64*kt + 32*kk + acc_row(r, l >> 5))
  // -- i.e. the 64-lane select mask of accumulator register r of the forward / dQ kernels (queries on the lanes), and for the
  // dK/dV kernel (keys on the lanes) one dword per key holding the bits of 32 queries.  No kernel hashes: one select per score
  // element (the kernels are bound by vector-instruction issue).  Survivors are scaled by inv_keep.
  int drop_on;
  float inv_keep;
  const uint32_t* dropbits;
  unsigned bytes_dbits;
  int NQB, NKT;     // 32-query blocks / 64-key tiles per (b, h) in dropbits
  // packed rows (nullable): sample b owns rows cu[b] .. cu[b+1]-1 of qkv / ctx / dctx / out / dqkv, i.e. only its first
  // cu[b+1]-cu[b] positions exist; mask words, lse, delta and the dropout counter keep their logical [B, L] indexing
  const int32_t* cu;
  // query limit (nullable): only the first qlim[b] rows of sample b are QUERIES (every row is a key).  The forward leaves the context /
  // lse of the other rows untouched, the backward writes zero dQ rows for them and ignores their dctx rows.  Used for the last encoder
  // layer, whose rows are reordered so that the rows the heads consume come first (mv_tail_perm).
  const int32_t* qlim;
  int order;        // att_block(): block -> (row block, head, sample) order
};
// ---- precomputed dropout keep-bits -------------------------------------------------------------------------------------------
typedef unsigned long long u64x8 __attribute__((ext_vector_type(8)));
// 16 select masks (one 32-key half of a tile) from a wave-uniform address.  The pointer is made provably uniform (readfirstlane) and
// read through the constant address space, so hipcc emits two s_load_dwordx16 that IT tracks: it may issue them ahead of the tile's
// last MFMAs and waits where the first select needs them (round 3 first had both loads and their wait inside one asm statement, i.e.
// a full scalar-memory round trip in front of every 32-key half: forward -2..5 %, backward unchanged).
__device__ __forceinline__ void sload_masks16(const void* p, u64x8& m0, u64x8& m1) {
  const unsigned long long v = (unsigned long long)(uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  typedef const u64x8 __attribute__((address_space(4))) cmask_t;
  cmask_t* q = (cmask_t*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  m0 = q[0];
  m1 = q[1];
}
// Selects by a 64-lane mask held in an SGPR pair: one VALU instruction (v_cndmask_b32 with an SGPR condition).  hipcc pads no hazard
// for an asm statement, so the VGPR inputs must never be the direct result of an MFMA (18 wait states) or of a transcendental
// (1 wait state): sel_set_else() is fed with fma results, sel_lane_after() names a second value `after` that a compiler-scheduled
// VALU instruction computed FROM x, which orders the statement behind that instruction.
__device__ __forceinline__ float sel_set_else(float if_set, float if_clear, unsigned long long m) {
  float o;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(if_clear), "v"(if_set), "s"(m));
  return o;
}
__device__ __forceinline__ float sel_lane_after(float x, float after, unsigned long long m) {
  float o;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(x), "s"(m), "v"(after));
  return o;
}
// one bit of the keep-bits for any (query, key): the plain VALU kernels' view of the same tensor
__device__ __forceinline__ bool attn_keep_bit(const unsigned long long* db, int NQB, int NKT, size_t bh, int q, int k) {
  const int k32 = k & 31, r = ((k32 >> 3) << 2) | (k32 & 3), hf = (k32 >> 2) & 1;
  const unsigned long long w = db[((bh * (size_t)NQB + (q >> 5)) * (size_t)NKT + (k >> 6)) * 32 + ((k >> 5) & 1) * 16 + r];
  return (w >> ((q & 31) + 32 * hf)) & 1ull;
}
__device__ __forceinline__ float and_bits(float x, int t) { return __uint_as_float(__float_as_uint(x) & (unsigned)t); }
__device__ __forceinline__ size_t dbits_block(const AttnArgs& a, size_t bh, int qb, int kt) {      // first u64 of a (32 x 64) block
  return ((bh * (size_t)a.NQB + qb) * (size_t)a.NKT + kt) * 32;
}

// One thread per 64-bit word.  The word's 64 keep decisions are 64 independent PL-bit uniforms compared with the threshold
// thr = round(p * 2^PL), evaluated bit-sliced: PL bit-planes of 64 random bits each (two 32-bit hashes of a counter), most
// significant plane first -- lt collects the lanes already known to be below the threshold, eq those still equal to its prefix.
// drop = (x < thr), so P(drop) = thr / 2^PL.  No cross-lane work; the kernel is bound by the hashes' quarter-rate multiplies
// (profiles/r03_notes.txt: 2 * PL hashes per word, time proportional to PL -- until round 5 cut the planes after the eighth, see below).
template <int PL>
__global__ __launch_bounds__(256) void attn_dropmask_kernel(unsigned k0, unsigned k1, unsigned thr16, int A, int L, int NQB, int NKT,
                                                            const int32_t* __restrict__ cu, unsigned long long* __restrict__ out, size_t nwords) {
  // thread blocks of one (sample, head) are spread over the dispatch order (see att_block: its later query blocks are the empty ones)
  const unsigned nbh = (unsigned)(nwords / ((size_t)NQB * NKT * 32)), per = gridDim.x / nbh;
  const unsigned lb = (per * nbh == gridDim.x) ? (blockIdx.x % nbh) * per + blockIdx.x / nbh : blockIdx.x;
  const size_t w = (size_t)lb * 256 + threadIdx.x;
  if (w >= nwords) return;
  const size_t blk = w >> 5;
  const int kt = (int)(blk % NKT), qb = (int)((blk / NKT) % NQB);
  const int b = (int)(blk / ((size_t)NKT * NQB * A));
  const int Lv = cu ? cu[b + 1] - cu[b] : L;
  if (qb * 32 >= Lv || kt * 64 >= Lv) return;        // no kernel ever looks at a block without an existing query or key
  const unsigned base = (unsigned)w * 32u;
  unsigned long long lt = 0ull, eq = ~0ull;
  // PL > 8: only the 8 most significant planes are evaluated bit-sliced.  After them a lane is still undecided with probability 2^-8
  // (its high byte equals the threshold's): 0.25 lanes per word on average, and those few draw their remaining PL - 8 bits one lane at a
  // time from single extra hashes (counters below the planes').  Same distribution -- P(drop) = thr / 2^PL exactly -- for 16 + ~2 hashes
  // per word instead of 2 * PL (round 5: 36 -> 23 us per layer at PL = 16).
  constexpr int SL = PL > 8 ? PL - 8 : 0;          // planes left to the serial tail
#pragma unroll
  for (int j = PL - 1; j >= SL; --j) {
    const unsigned long long x = ((unsigned long long)mv_hash32(base + 2 * j, k0, k1) << 32) | mv_hash32(base + 2 * j + 1, k0, k1);
    const unsigned long long tb = 0ull - (unsigned long long)((thr16 >> j) & 1u);      // all ones where the threshold has this bit
    lt |= eq & ~x & tb;
    eq &= x ^ ~tb;
  }
  if constexpr (SL > 0) {
    const unsigned lo = thr16 & ((1u << SL) - 1u);
    constexpr int PER = 32 / SL;                    // lanes served by one hash
    unsigned r = 0, ctr = 0;
    int have = 0;
    while (eq) {
      const int l = __builtin_ctzll(eq);
      eq &= eq - 1ull;
      if (have == 0) { r = mv_hash32(base + ctr, k0, k1); ++ctr; have = PER; }
      const unsigned x = r & ((1u << SL) - 1u);
      r >>= SL;
      --have;
      if (x < lo) lt |= 1ull << l;
    }
  }
  out[w] = ~lt;
}

// dual-use LDS image of a [64 rows][64 x bf16] tile: 128-B rows, 16-B chunk index XORed with f(row)
__device__ __forceinline__ int att_f(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int att_off(int r, int ch) { return r * 128 + ((ch ^ att_f(r)) << 4); }

// rows of a 32x32 accumulator: element `reg` of lane-half h is row (reg&3) + 8*(reg>>2) + 4*h
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// stage one [64][64] tile (rows row0.. of a [*, ld] bf16 matrix starting at column col0) into regs
__device__ __forceinline__ void tile_load(u32x4 (&reg)[2], __amdgpu_buffer_rsrc_t rs, unsigned bytes, size_t rowbase,
                                          int row0, int nrows, int ld, int col0, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 3, ch = idx & 7;
    const bool ok = (row0 + r) < nrows;
    const unsigned off = (unsigned)(((rowbase + row0 + r) * (size_t)ld + col0 + ch * 8) * 2);
    reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : bytes, 0, 0);
  }
}
__device__ __forceinline__ void tile_store(const u32x4 (&reg)[2], char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    *(u32x4*)(tile + att_off(idx >> 3, idx & 7)) = reg[i];
  }
}
// The same tile, HBM -> LDS directly (LDS-DMA, `buffer_load ... lds`, 16 B per lane): 8 pieces of 1 KiB = 8 rows each; wave
// `wid` of NW issues pieces wid, wid + NW, ...  The LDS image of a piece is lane-linear, so the chunk swizzle of att_off() is
// applied to the per-lane SOURCE address; rows past `nrows` are zero-filled (out-of-range buffer offset).  No registers hold
// the tile, so a ring of several stages can be in flight: the kernels below keep NS - 1 tiles ahead of the one they compute on
// and retire them with counted `s_waitcnt vmcnt(N)` + one raw s_barrier per tile.
// The LDS-DMA itself is issued from inline assembly.  Through the builtin the compiler tracks the transfer as a pending LDS
// write and, having no alias information for `ds_read_b64_tr_b16`, puts `s_waitcnt vmcnt(0)` in front of the first transposed
// fragment read of every tile -- the whole ring drained once per tile, so only the transfer issued last was ever ahead.  The
// kernels below order the ring themselves (counted vmcnt + s_barrier before a stage is read, see att_wait_stage).
// (dma_rsrc_t, dma_rsrc, lds_dma16, lds_dma4: mv_common.h)
template <int NW>
__device__ __forceinline__ void tile_dma(dma_rsrc_t rs, unsigned bytes, size_t rowbase, int row0, int nrows, int ld,
                                         int col0, char* tile, int wid, int lane) {
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) {
    const int pc = wid + NW * i;
    const int r = pc * 8 + (lane >> 3), ch = (lane & 7) ^ att_f(r);
    const bool ok = (row0 + r) < nrows;
    const unsigned off = (unsigned)(((rowbase + row0 + r) * (size_t)ld + col0 + ch * 8) * 2);
    lds_dma16(rs, (MV_LDS void*)(tile + pc * 1024), ok ? off : bytes);
  }
}
// Where a ring iteration requests the next tile: right after the barrier (0) or after its own tile body (1).  Inside a tile body
// the compiler may wait for one of ITS memory operations -- a scratch reload of a spilled register (dK/dV kernel), the mask words
// of a mixed-class tile (fwd, dQ) -- with vmcnt(0); vmcnt retires in order, so that wait also waits for every LDS-DMA issued
// before it.  Issued late, the youngest transfer in flight at that point is a whole tile old instead of a few instructions.
#ifndef ATT_ISSUE_LATE
#define ATT_ISSUE_LATE 1
#endif
template <int N> __device__ __forceinline__ void att_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `younger` stages (of PPW LDS-DMA instructions per wave each) are still in flight
template <int PPW>
__device__ __forceinline__ void att_wait_stage(int younger) {
  if (younger <= 0) att_wait_vmcnt<0>();
  else if (younger == 1) att_wait_vmcnt<PPW>();
  else if (younger == 2) att_wait_vmcnt<2 * PPW>();
  else att_wait_vmcnt<3 * PPW>();
}

// row-read fragment  X[idx = base + (lane&31)][k = 16*s + 8*h + j]
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int base, int s, int l31, int h) {
  return *(const bf16x8*)(tile + att_off(base + l31, 2 * s + h));
}
// transposed fragment  A[row = cbase + (lane&31)][k = tile row rb + 8*(j>>2) + 4*h + (j&3)]
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int cbase, int rb, int lane) {
  const int li = lane & 15, g = lane >> 4, h = g >> 1;
  const int col = cbase + 16 * (g & 1) + 4 * (li & 3);
  const int r = rb + 4 * h + (li >> 2);
  const int sub = (li & 1) * 8;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(tile + att_off(r, col >> 3) + sub));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(tile + att_off(r + 8, col >> 3) + sub));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// The same two fragment reads with the lane-dependent part of att_off() taken out of the tile loop.  Per lane and tile-invariant:
//   rk = (lane&31)*128 + ((h ^ f0) << 4) + ((f >> 1) << 5)        f = att_f(lane & 31): bits 5-6 of rk ARE the chunk swizzle of s = 0
//   tr = (4h' + q)*128 + ((q >> 1) << 6) + ((g & 1) << 5) + ((c0 ^ h') << 4) + 8 (li & 1)
// so that, for a tile at LDS byte address T (a multiple of 128 B -- the XORs touch bits 4-6 only, and T + rk / T + tr carry nothing into them),
//   frag_row(T, base, s)      = ((T + rk) ^ (s << 5)) + 128 base                                 base = 0 / 32
//   frag_tr (T, 32 dt, rb) lo = ((T + tr) ^ (dt << 6)) + 128 rb,   hi = (... ^ 32) + 128 rb + 1024
// i.e. one add per tile and one XOR per (s) / (dt, half); everything else is an instruction immediate.  Before, hipcc rebuilt every
// address from the (per tile opaque, see dkv_tile) lane index: ~6 VALU instructions per ds_read, a quarter of the forward's issue slots.
struct FragLane { unsigned rk, tr; };
__device__ __forceinline__ FragLane frag_lane(int lane) {
  const int l31 = lane & 31, h = lane >> 5, f = att_f(l31);
  const int li = lane & 15, g = lane >> 4, hh = g >> 1, q = li >> 2, c0 = (li & 3) >> 1;
  FragLane o;
  o.rk = (unsigned)(l31 * 128 + (((h ^ f) & 1) << 4) + ((f >> 1) << 5));
  o.tr = (unsigned)((4 * hh + q) * 128 + ((q >> 1) << 6) + ((g & 1) << 5) + ((c0 ^ hh) << 4) + (li & 1) * 8);
  return o;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(MV_LDS const char*)p; }
// tk = lds_addr(tile) + FragLane::rk
__device__ __forceinline__ bf16x8 frag_row_x(unsigned tk, int base, int s) {
  return *(MV_LDS const bf16x8*)(uintptr_t)((tk ^ (unsigned)(s << 5)) + 128u * base);
}
// tt = lds_addr(tile) + FragLane::tr
__device__ __forceinline__ bf16x8 frag_tr_x(unsigned tt, int dt, int rb) {
  const unsigned t = tt ^ (unsigned)(dt << 6);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(uintptr_t)(t + 128u * rb));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MV_LDS bf16x4*)(uintptr_t)((t ^ 32u) + 128u * rb + 1024u));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s2) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16_t)v[8 * s2 + j];
  return r;
}
// 32x32x16 MFMA / accumulator packing in the operand encoding of the kernel: F16 = f16 bit patterns carried in bf16x8
template <bool F16>
__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ bf16x8 pack8t(const f32x16& v, int s2) {
  if constexpr (F16) {
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (f16_t)v[8 * s2 + j];
    return __builtin_bit_cast(bf16x8, r);
  } else {
    return pack8(v, s2);
  }
}
template <bool F16>
__device__ __forceinline__ float frag_elem(bf16x8 v, int j) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16x8, v)[j];
  else return (float)v[j];
}
template <bool F16>
__device__ __forceinline__ void store4_16(bf16_t* p, float a, float b, float c, float d) {
  const f32x4 v = {a, b, c, d};
  if constexpr (F16) st4<f16_t>((f16_t*)p, v);
  else st4<bf16_t>(p, v);
}
__device__ __forceinline__ bf16x8 load_rowfrag_global(__amdgpu_buffer_rsrc_t rs, unsigned bytes, size_t row, bool ok, int ld,
                                                      int col) {
  const unsigned off = (unsigned)((row * (size_t)ld + col) * 2);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : bytes, 0, 0);
  return __builtin_bit_cast(bf16x8, v);
}

// tile classes of the block's two 64-row tiles, fetched ONCE (lane t holds tile t) and turned into wave-uniform bit
// masks: the K loop then walks set bits instead of issuing dependent global loads per tile
struct TileMasks { unsigned long long need, w_nz, w_is1; };
__device__ __forceinline__ TileMasks load_tile_masks_q(const uint8_t* info, int b, int T, int ta, int tq_wave, int lane) {
  uint8_t c0 = 0, c1 = 0;
  if (lane < T) {
    c0 = info[((size_t)b * T + ta) * T + lane];
    if (ta + 1 < T) c1 = info[((size_t)b * T + ta + 1) * T + lane];
  }
  const uint8_t cw = (tq_wave == ta) ? c0 : c1;
  TileMasks m;
  m.need = __ballot(c0 != 0 || c1 != 0);
  m.w_nz = __ballot(cw != 0);
  m.w_is1 = __ballot(cw == 1);
  return m;
}
__device__ __forceinline__ TileMasks load_tile_masks_k(const uint8_t* info, int b, int T, int ka, int tk_wave, int lane) {
  uint8_t c0 = 0, c1 = 0;
  if (lane < T) {
    c0 = info[((size_t)b * T + lane) * T + ka];
    if (ka + 1 < T) c1 = info[((size_t)b * T + lane) * T + ka + 1];
  }
  const uint8_t cw = (tk_wave == ka) ? c0 : c1;
  TileMasks m;
  m.need = __ballot(c0 != 0 || c1 != 0);
  m.w_nz = __ballot(cw != 0);
  m.w_is1 = __ballot(cw == 1);
  return m;
}
__device__ __forceinline__ int next_tile(unsigned long long need, int after, int n) {   // first set bit > after, else n
  const unsigned long long rem = (after >= 63) ? 0ull : (need >> (after + 1));
  return rem ? after + 1 + (int)__builtin_ctzll(rem) : n;
}

// ---- forward --------------------------------------------------------------------------------
// Block -> (128-row block xb, head, sample).  Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), each XCD has its own
// 4 MiB L2, and the row blocks of one (sample, head) pair all stream the SAME K / V rows (forward, dQ) or Q / dO rows (dK/dV): 128 KiB at
// L = 512.  Three orders have existed:
//   blockIdx as is (rounds 1-2): the row block is the fastest index, so the pair's four row blocks go to four DIFFERENT XCDs -- and XCD k
//     only ever sees row block k % 4, nearly empty for the last block of a packed sample: two XCDs idled.
//   order 0 (rounds 3-5, the default): the row block is the SLOWEST index.  Even load; the row blocks of a pair run a whole wave of ~768
//     resident blocks apart, so every block fetches its K / V over the fabric (profiles/r04_attn_pmc.json: 2.97x the algorithmic
//     bytes, L2 hit 16-28 %) -- which is NOT what bounds these kernels:
//   order 1 (round 5 experiment, debug knob attn_order): flat = 8 * (group * nxb + xb) + x, pair = 8 * group + x.  The nxb row blocks of
//     a pair are 8 apart in dispatch order -- consecutive blocks of ONE XCD, resident together -- so the pair's K / V is fetched into
//     that L2 once.  Measured in one process against order 0 (profiles/tools/attn_order_ab.py, profiles/r05_notes.txt): bidirectional
//     packed forward 82.5 -> 81.2 us, backward 257 -> 266 us; BAR forward 97 -> 116, backward 293 -> 353; seq2seq backward 190 -> 246; whole
//     step 24.22 -> 24.36 ms.  With row-dependent masks the blocks resident together then differ in length by up to 4x and the launch
//     ends on a tail of long blocks; with uniform masks nothing is gained either: the kernels wait on dependent latency inside a tile
//     (DESIGN.md 8), not on fabric bandwidth.  Pairs beyond the last multiple of 8 fall back to the row-block-fastest walk.
//   Placement is a speed matter only: nothing depends on it for correctness (outputs are bit-identical in both orders).
__device__ __forceinline__ void att_block(const AttnArgs& a, int& xb, int& head, int& b) {
  const int flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int nxb = gridDim.x, nhb = gridDim.y * gridDim.z;
  int pair;
  if (a.order == 0) {
    xb = flat / nhb;
    pair = flat - xb * nhb;
  } else {
    const int full = (nhb >> 3) << 3;               // pairs covered by whole groups of 8
    if (flat < full * nxb) {
      const int x = flat & 7, k = flat >> 3;
      const int g = k / nxb;
      xb = k - g * nxb;
      pair = 8 * g + x;
    } else {
      const int r = flat - full * nxb;
      pair = full + r / nxb;
      xb = r - (pair - full) * nxb;
    }
  }
  b = pair / gridDim.y;
  head = pair - b * gridDim.y;
}
// A wave's [32 rows][64 columns] 16-bit output tile leaves as WHOLE 128-byte row segments.  The tile sits in two 32x32 accumulators with
// the rows on the lanes (lane (l31, h), register reg of acc[dt]: row l31, column 32 dt + acc_row(reg, h)), so a direct store writes 8 bytes of
// 32 different rows per instruction and every 128-byte line of the output is written in eight pieces by eight instructions: measured on
// the forward (profiles/r04_notes.txt) the epilogue alone cost 25 of the kernel's 100 us.  Here the tile goes through a private
// 32 x 144-byte LDS patch (ds_write_b64 per 4 columns, ds_read_b128 per 8): 16 bytes per lane, 8 full rows per store instruction.
// `rows_ok`: rows [0, rows_ok) of the tile exist in the output; `patch` is this wave's own 4608-byte region (callers barrier first).
#define ATT_PATCH_BYTES 4608
template <bool F16>
__device__ __forceinline__ void store_rows_tile(char* patch, const f32x16 (&acc)[2], float scale, bool lane_ok, bf16_t* g_row0, size_t ld,
                                                int rows_ok, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
  // The patch is written and read by ONE wave (different lanes touch the same bytes).  The hardware completes a wave's LDS operations in
  // order; the COMPILER is told so explicitly -- a wavefront-scope fence pair around a wave barrier (no instruction: a scheduling and
  // memory-ordering point) before the writes (a previous use of the same patch: dK then dV) and between the writes and the reads --
  // instead of relying on it failing to prove the addresses disjoint.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {acc[dt][4 * g] * scale, acc[dt][4 * g + 1] * scale, acc[dt][4 * g + 2] * scale, acc[dt][4 * g + 3] * scale};
      if (!lane_ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};        // (a select, not a product: the lane's accumulators may hold anything)
      bf16_t* dst = (bf16_t*)(patch + l31 * 144) + 32 * dt + 8 * g + 4 * h;
      if constexpr (F16) st4<f16_t>((f16_t*)dst, v);
      else st4<bf16_t>(dst, v);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (lane >> 3) + 8 * i, c = lane & 7;
    const u32x4 v = *(const u32x4*)(patch + r * 144 + 16 * c);
    if (r < rows_ok) *(u32x4*)(g_row0 + (size_t)r * ld + 8 * c) = v;
  }
}

// One 64-key tile of the forward for a wave's 32 queries.  PLAIN: every entry of the tile is visible and inside the sample (class 1, no
// ragged tail): no mask words, no bounds, the scale folded into the exponent's fma.  DROP: attention dropout on.  Compile-time, so each
// variant is straight-line code (the run-time form merged its paths through 32 register copies per tile).
template <bool PLAIN, bool DROP, bool F16>
__device__ __forceinline__ void fwd_tile(const AttnArgs& a, unsigned tk, unsigned tv, const bf16x8 (&qf)[4], f32x16 (&o)[2], float& m2,
                                         float& lsum, const uint32_t* myw, bool q_ok, int k0, int Lv, int cls, float c2, int h,
                                         const unsigned long long* mp PROF_TILE_ARGS) {
  PROF_T0;
  // the 64 select masks of the tile (two scalar loads of 128 bytes per 32-key half) are requested FIRST: a scalar-cache miss costs a round
  // trip to L2 / HBM (measured: 3,500 shader cycles per tile when requested where they are used), the score MFMAs and the softmax hide it
  u64x8 dm[2][2];
#if ATT_EARLY_MASKS
  if (DROP) {
#if !(ATT_ABL & 4)
    sload_masks16(mp, dm[0][0], dm[0][1]);
#if ATT_EARLY_MASKS >= 2
    sload_masks16(mp + 16, dm[1][0], dm[1][1]);
#endif
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
#endif
  f32x16 st[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int i = 0; i < 16; ++i) st[kk][i] = 0.f;
#if ATT_ABL & 32
#pragma unroll
    for (int i = 0; i < 16; ++i) st[kk][i] = frag_elem<F16>(qf[i & 3], i >> 2) * (float)(k0 + i);
#else
#pragma unroll
    for (int s = 0; s < 4; ++s) st[kk] = mma32<F16>(frag_row_x(tk, 32 * kk, s), qf[s], st[kk]);
#endif
  }
  PROF_T(0, st[1][15]);
  float mx = -INFINITY;
  if (PLAIN) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kk][r]);
    mx *= c2;
  } else {
    const bool tail = (k0 + 64 > Lv);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint32_t w = 0xffffffffu;
      if (cls != 1) { const int wi = (k0 >> 5) + kk; w = (q_ok && wi < a.W) ? myw[wi] : 0xffffffffu; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = acc_row(r, h);
        float v = st[kk][r] * c2;
        if (cls != 1) v += ((w >> kr) & 1u) ? 0.f : MASK_ADD * LOG2E;
        if (tail && (k0 + 32 * kk + kr >= Lv)) v = -INFINITY;
        st[kk][r] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
#if !(ATT_ABL & 8)
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#endif
  const float mn = fmaxf(m2, mx);
  PROF_T(1, mn);
  const float alpha = fexp2(m2 - mn);
  m2 = mn;
  f32x2 ps2 = {0.f, 0.f};            // two partial sums: v_pk_add_f32, half the add instructions
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
#if ATT_ABL & 2
      const float p0 = fmaf(st[kk][r], c2, -mn), p1 = fmaf(st[kk][r + 1], c2, -mn);
#else
      const float p0 = PLAIN ? fexp2(fmaf(st[kk][r], c2, -mn)) : fexp2(st[kk][r] - mn);
      const float p1 = PLAIN ? fexp2(fmaf(st[kk][r + 1], c2, -mn)) : fexp2(st[kk][r + 1] - mn);
#endif
      st[kk][r] = p0;
      st[kk][r + 1] = p1;
      ps2 += (f32x2){p0, p1};
    }
  const float ps = ps2[0] + ps2[1];
  lsum = lsum * alpha + ps;          // the normaliser sums the UNdropped probabilities
  PROF_T(2, lsum);
  if (DROP) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u64x8 m0 = dm[kk][0], m1 = dm[kk][1];
#if !(ATT_ABL & 4)
      if (ATT_EARLY_MASKS < 2 && (kk == 1 || !ATT_EARLY_MASKS)) sload_masks16(mp + 16 * kk, m0, m1);
#endif
#if ATT_ABL & 4
#pragma unroll
      for (int r = 0; r < 8; ++r) { m0[r] = 0xfff7ffffffffefffull ^ (unsigned long long)(k0 + r); m1[r] = ~m0[r] | 0xffff0000ffffull; }
#endif
#pragma unroll
      for (int r = 0; r < 8; ++r) {       // `ps` was computed from these values by compiler-scheduled adds: see sel_lane_after
        st[kk][r] = sel_lane_after(st[kk][r], ps, m0[r]);
        st[kk][8 + r] = sel_lane_after(st[kk][8 + r], ps, m1[r]);
      }
    }
  }
  PROF_T(3, st[1][15]);
  // Unconditional: 16 packed multiplies.  Skipping them while the running maximum does not move (almost every tile) costs MORE -- the two
  // paths leave the accumulators in different registers and hipcc pays the merge with 32-64 register copies on the skipping path.
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = pack8t<F16>(st[kk], s2);
#if ATT_ABL & 16
      asm volatile("" ::"v"(pf));
      (void)tv;
#else
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = mma32<F16>(frag_tr_x(tv, dt, 32 * kk + 16 * s2), pf, o[dt]);
#endif
    }
  PROF_T(4, o[1][15]);
}

#define FWD_NS 3      // 48 KiB of LDS per block: three blocks per CU
template <bool F16>
__global__ __launch_bounds__(256, 3) void attn_fwd_mfma_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(128))) char smem[];   // (fragment addressing XORs bits 4-6 of tile base + lane word: bases are multiples of 128)   // FWD_NS stages x (K 8 KiB + V 8 KiB)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int xb, head, b;
  PROF_DECL;
#if ATT_ABL & 256
  if (a.B > 0) return;            // dispatch cost of the grid alone
#endif
  att_block(a, xb, head, b);
  const int L = a.L, H = a.H, ld = 3 * a.H, T = a.T;
  const int qb0 = xb * 128, q0 = qb0 + wid * 32, q = q0 + l31;
  // The tile classes depend on the block index alone: requested BEFORE the row plan (cu, qlim), not behind it -- a block's prologue is a chain
  // of dependent memory round trips (arguments -> row plan -> Q rows / tile classes -> first K / V tiles: 12,600 shader cycles, a fifth of
  // the block's life, profiles/r04_notes.txt), this takes one link out
  const TileMasks tmk = load_tile_masks_q(a.info, b, T, qb0 >> 6, min(q0 >> 6, T - 1), lane);
  const int Lv = a.cu ? a.cu[b + 1] - a.cu[b] : L;             // positions of this sample that exist as rows
  const int Lq = a.qlim ? min(Lv, a.qlim[b]) : Lv;             // ... and those that are queries
  if (qb0 >= Lq) return;
  const bool wave_on = q0 < Lq, q_ok = q < Lq;
  const size_t rowbase = a.cu ? (size_t)a.cu[b] : (size_t)b * L;
  const size_t lrow = (size_t)b * L;                            // logical row base (mask words)
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.qkv, 0, a.bytes_qkv, 0x00020000);
  const dma_rsrc_t rsq = dma_rsrc(a.qkv, a.bytes_qkv);

  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = load_rowfrag_global(rs, a.bytes_qkv, rowbase + q, q_ok, ld, head * 64 + 16 * s + 8 * h);

  const float c2 = a.scale * LOG2E;
  float m2 = -INFINITY, lsum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }

  const int nkt = (Lv + 63) / 64;
  // K/V ring: the needed key tiles (set bits of tmk.need) are requested FWD_NS - 1 ahead of the one being computed on
  int cur = next_tile(tmk.need, -1, nkt);
  int iss = cur, issued = 0, done = 0;
  auto issue = [&]() {
    char* st_ = smem + (issued % FWD_NS) * 16384;
#if !(ATT_ABL & 64)
    tile_dma<4>(rsq, a.bytes_qkv, rowbase, iss * 64, Lv, ld, H + head * 64, st_, wid, lane);
    tile_dma<4>(rsq, a.bytes_qkv, rowbase, iss * 64, Lv, ld, 2 * H + head * 64, st_ + 8192, wid, lane);
#endif
    ++issued;
    iss = next_tile(tmk.need, iss, nkt);
  };
#pragma unroll
  for (int i = 0; i < FWD_NS - 1; ++i)
    if (iss < nkt) issue();
  const uint32_t* myw = a.bits + (lrow + (q_ok ? q : 0)) * a.W;
  const FragLane fl = frag_lane(lane);            // two registers held across the loop: the lane's part of every fragment address
  const unsigned smem_a = lds_addr(smem);
#if ATT_ABL & 512
  asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "s"(tmk.need), "v"(fl.rk));
  if (a.B > 0) { asm volatile("s_waitcnt vmcnt(0)"); return; }      // prologue only
#endif
#if ATT_ABL & 1024
  asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "s"(tmk.need), "v"(fl.rk));
  asm volatile("s_waitcnt vmcnt(0)");
  cur = nkt;                                      // prologue + epilogue, no tile loop
#endif
  PROF_MARK(0);
  while (cur < nkt) {
    att_wait_stage<4>(issued - done - 1);           // this wave's pieces of tile `cur` have landed ...
    PROF_MARK(1);
#if !(ATT_ABL & 1)
    __builtin_amdgcn_s_barrier();                   // ... and everybody's; everybody is done reading the slot refilled next
#endif
    PROF_MARK(2);
    __builtin_amdgcn_sched_barrier(0);
    if (!ATT_ISSUE_LATE && iss < nkt) issue();
    const unsigned tKa = smem_a + (unsigned)(done % FWD_NS) * 16384u;
    const int cls = !wave_on ? 0 : (((tmk.w_is1 >> cur) & 1) ? 1 : (((tmk.w_nz >> cur) & 1) ? 2 : 0));
    if (wave_on && cls != 0) {
      const int k0 = cur * 64;
      // (the two lane words are made opaque per tile: hipcc otherwise hoists every derived address out of the loop and spills them)
      unsigned rk_ = fl.rk, tr_ = fl.tr;
      int h_ = h;                   // (also the half-wave index: the masked variants' 32 per-register bit masks derive from it)
      asm volatile("" : "+v"(rk_), "+v"(tr_), "+v"(h_));
      const unsigned tk = tKa + rk_, tv = tKa + 8192u + tr_;
      const bool plain = (cls == 1) && (k0 + 64 <= Lv);
      const unsigned long long* mp = a.drop_on ? (const unsigned long long*)a.dropbits + dbits_block(a, (size_t)b * a.A + head, q0 >> 5, cur) : nullptr;
#define FWD_CALL(P_, D_) fwd_tile<P_, D_, F16>(a, tk, tv, qf, o, m2, lsum, myw, q_ok, k0, Lv, cls, c2, h_, mp PROF_TILE_PASS)
      switch ((plain ? 0 : 1) | (a.drop_on ? 2 : 0)) {       // one wave-uniform dispatch per tile
        case 0: FWD_CALL(true, false); break;
        case 1: FWD_CALL(false, false); break;
        case 2: FWD_CALL(true, true); break;
        default: FWD_CALL(false, true); break;
      }
#undef FWD_CALL
    }
#if ATT_ABL & 4096
    asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][15]));      // the tile's last MFMAs have delivered
#endif
    PROF_MARK(3);
    if (ATT_ISSUE_LATE && iss < nkt) issue();
    cur = next_tile(tmk.need, cur, nkt);
    ++done;
    PROF_MARK(4);
  }
  // epilogue: whole rows through a per-wave LDS patch (every wave is past its last tile: the K / V stages are free)
  __builtin_amdgcn_s_barrier();
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = (a.drop_on ? a.inv_keep : 1.0f) / ltot;
  const int rows_ok = Lq - q0;                      // (<= 0 for a wave without queries: nothing is stored)
  char* patch = smem + wid * ATT_PATCH_BYTES;
  store_rows_tile<F16>(patch, o, inv, q_ok, a.out + (rowbase + q0) * (size_t)H + head * 64, (size_t)H, rows_ok, lane);
  if constexpr (F16) {
    if (a.out2) store_rows_tile<false>(patch, o, inv, q_ok, a.out2 + (rowbase + q0) * (size_t)H + head * 64, (size_t)H, rows_ok, lane);
  }
  if (q_ok && h == 0) a.lse[((size_t)b * a.A + head) * L + q] = (m2 + log2f(ltot)) * LN2;
#if ATT_ABL & 4096
  asm volatile("s_waitcnt vmcnt(0)");
#endif
  PROF_MARK(5);
  PROF_FLUSH(0);
}

// ---- backward: dQ ----------------------------------------------------------------------------
// One 64-key tile of the dQ pass for a wave's 32 queries.  MASKED: the tile needs its mask words (class 2; a ragged
// TAIL tile is always run as masked), DROP: attention dropout is on -- compile-time, so the per-element loops are
// straight-line code the scheduler can interleave with the MFMAs.
template <bool MASKED, bool DROP, bool TAIL, bool F16>
__device__ __forceinline__ void dq_tile(const AttnArgs& a, unsigned tk, unsigned tv, unsigned tkt, const bf16x8 (&qf)[4], const bf16x8 (&dof)[4],
                                        f32x16 (&dq)[2], const uint32_t* myw, bool q_ok, int q, int b, int head, int k0, int Lv,
                                        float lse2, float dlt, float c2, int h, const unsigned long long* mp) {
  // tk / tv: row-read bases of the K / V tiles (lds address + FragLane::rk), tkt: transposed-read base of the K tile (+ FragLane::tr)
  const float dlt_s = dlt * a.scale;                          // ds = p * (dp' - delta) * scale, with the scale folded in
  const float dscale = DROP ? a.inv_keep * a.scale : a.scale;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    f32x16 st, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = mma32<F16>(frag_row_x(tk, 32 * kk, s), qf[s], st);
      dp = mma32<F16>(frag_row_x(tv, 32 * kk, s), dof[s], dp);
    }
    uint32_t w = 0xffffffffu;
    if (MASKED) { const int wi = (k0 >> 5) + kk; w = (q_ok && wi < a.W) ? myw[wi] : 0xffffffffu; }
    u64x8 m0, m1;
    if (DROP) sload_masks16(mp + 16 * kk, m0, m1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = acc_row(r, h);
      float pv;
      if (!MASKED) {
        pv = fexp2(fmaf(st[r], c2, -lse2));
      } else {
        const float v = fmaf(st[r], c2, ((w >> kr) & 1u) ? 0.f : MASK_ADD * LOG2E);
        pv = fexp2(v - lse2);
      }
      if (TAIL) pv = (k0 + 32 * kk + kr >= Lv) ? 0.f : pv;
      // ds = p * (keep ? dp * inv_keep * scale - delta * scale : -delta * scale): the select takes the fma's result, never dp[r]
      // itself (an MFMA result feeding an asm statement would not get its wait states)
      float t = fmaf(dp[r], dscale, -dlt_s);
      if (DROP) t = sel_set_else(t, -dlt_s, r < 8 ? m0[r & 7] : m1[r & 7]);
      st[r] = pv * t;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 dsf = pack8t<F16>(st, s2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        dq[dt] = mma32<F16>(frag_tr_x(tkt, dt, 32 * kk + 16 * s2), dsf, dq[dt]);
    }
  }
}

#define DQ_NS 4       // 64 KiB of LDS per block, two blocks per CU (register-limited)
template <bool F16>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_mfma_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(128))) char smem[];   // (fragment addressing XORs bits 4-6 of tile base + lane word: bases are multiples of 128)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int xb, head, b;
  att_block(a, xb, head, b);
  const int L = a.L, H = a.H, ld = 3 * a.H, T = a.T;
  const int qb0 = xb * 128, q0 = qb0 + wid * 32, q = q0 + l31;
  PROF_DECL;
  const TileMasks tmk = load_tile_masks_q(a.info, b, T, qb0 >> 6, min(q0 >> 6, T - 1), lane);      // before the row plan: see the forward
  const int Lv = a.cu ? a.cu[b + 1] - a.cu[b] : L;
  const int Lq = a.qlim ? min(Lv, a.qlim[b]) : Lv;
  if (qb0 >= Lv) return;
  const size_t rowbase = a.cu ? (size_t)a.cu[b] : (size_t)b * L;
  if (qb0 >= Lq) {        // rows that are keys only: their dQ is zero (thread t: row t / 2, 32 of the head's 64 columns)
    const int r = qb0 + (tid >> 1);
    if (r < Lv) {
      bf16_t* z = a.dqkv + (rowbase + r) * (size_t)ld + head * 64 + 32 * (tid & 1);
#pragma unroll
      for (int c = 0; c < 32; c += 4) store4_16<F16>(z + c, 0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const bool wave_on = q0 < Lq, q_ok = q < Lq;
  const size_t lrow = (size_t)b * L;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.qkv, 0, a.bytes_qkv, 0x00020000);
  const dma_rsrc_t rsq = dma_rsrc(a.qkv, a.bytes_qkv);
  __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dctx, 0, a.bytes_ctx, 0x00020000);

  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = load_rowfrag_global(rs, a.bytes_qkv, rowbase + q, q_ok, ld, head * 64 + 16 * s + 8 * h);
    dof[s] = load_rowfrag_global(rsd, a.bytes_ctx, rowbase + q, q_ok, H, head * 64 + 16 * s + 8 * h);
  }
  const size_t sidx = ((size_t)b * a.A + head) * L + (q_ok ? q : 0);
  const float lse2 = q_ok ? a.lse_in[sidx] * LOG2E : INFINITY;
  // delta[q] = sum_d dO[q,d] * O[q,d], computed here (each lane holds 32 of the 64 d of its query) and published for
  // the dK/dV kernel that runs next
  float dlt = 0.f;
  {
    __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)a.ctx, 0, a.bytes_ctx, 0x00020000);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 of = load_rowfrag_global(rsc, a.bytes_ctx, rowbase + q, q_ok, H, head * 64 + 16 * s + 8 * h);
#pragma unroll
      for (int j = 0; j < 8; ++j) dlt = fmaf(frag_elem<F16>(of, j), frag_elem<F16>(dof[s], j), dlt);
    }
    dlt += __shfl_xor(dlt, 32, 64);
    if (q_ok && h == 0) a.delta_out[sidx] = dlt;
  }
  const float c2 = a.scale * LOG2E;
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }

  const int nkt = (Lv + 63) / 64;
  int cur = next_tile(tmk.need, -1, nkt);
  int iss = cur, issued = 0, done = 0;
  auto issue = [&]() {
    char* st_ = smem + (issued % DQ_NS) * 16384;
    tile_dma<4>(rsq, a.bytes_qkv, rowbase, iss * 64, Lv, ld, H + head * 64, st_, wid, lane);
    tile_dma<4>(rsq, a.bytes_qkv, rowbase, iss * 64, Lv, ld, 2 * H + head * 64, st_ + 8192, wid, lane);
    ++issued;
    iss = next_tile(tmk.need, iss, nkt);
  };
#pragma unroll
  for (int i = 0; i < DQ_NS - 1; ++i)
    if (iss < nkt) issue();
  const uint32_t* myw = a.bits + (lrow + (q_ok ? q : 0)) * a.W;
  const FragLane fl = frag_lane(lane);
  const unsigned smem_a = lds_addr(smem);
  PROF_MARK(0);
  while (cur < nkt) {
    att_wait_stage<4>(issued - done - 1);
    PROF_MARK(1);
    __builtin_amdgcn_s_barrier();
    PROF_MARK(2);
    __builtin_amdgcn_sched_barrier(0);
    if (!ATT_ISSUE_LATE && iss < nkt) issue();
    const unsigned tKa = smem_a + (unsigned)(done % DQ_NS) * 16384u;
    const int cls = !wave_on ? 0 : (((tmk.w_is1 >> cur) & 1) ? 1 : (((tmk.w_nz >> cur) & 1) ? 2 : 0));
    if (wave_on && cls != 0) {
      const int k0 = cur * 64;
      const bool tail = (k0 + 64 > Lv);
      // fragment addresses: the lane's two words (opaque per tile: nothing derived from them is hoisted and spilled) + the tile's address
      unsigned rk_ = fl.rk, tr_ = fl.tr;
      int h_ = h;
      asm volatile("" : "+v"(rk_), "+v"(tr_), "+v"(h_));
      const unsigned tk = tKa + rk_, tv = tKa + 8192u + rk_, tkt = tKa + tr_;
      // one wave-uniform dispatch per tile: the element loops below contain no branches
      const unsigned long long* mp = a.drop_on ? (const unsigned long long*)a.dropbits + dbits_block(a, (size_t)b * a.A + head, q0 >> 5, cur) : nullptr;
      const int variant = (cls == 1 ? 0 : 1) | (tail ? 2 : 0) | (a.drop_on ? 4 : 0);
#define DQ_CALL(M_, D_, T_) dq_tile<M_, D_, T_, F16>(a, tk, tv, tkt, qf, dof, dq, myw, q_ok, q, b, head, k0, Lv, lse2, dlt, c2, h_, mp)
      switch (variant) {
        case 0: DQ_CALL(false, false, false); break;
        case 1: DQ_CALL(true, false, false); break;
        case 2: case 3: DQ_CALL(true, false, true); break;
        case 4: DQ_CALL(false, true, false); break;
        case 5: DQ_CALL(true, true, false); break;
        default: DQ_CALL(true, true, true); break;
      }
#undef DQ_CALL
    }
#if ATT_ABL & 4096
    asm volatile("s_nop 0" ::"v"(dq[0][0]), "v"(dq[1][15]));
#endif
    PROF_MARK(3);
    if (ATT_ISSUE_LATE && iss < nkt) issue();
    cur = next_tile(tmk.need, cur, nkt);
    ++done;
    PROF_MARK(4);
  }
  // epilogue: whole rows through a per-wave LDS patch; a key-only row inside a query tile (a.qlim) gets a zero dQ row
  __builtin_amdgcn_s_barrier();
  store_rows_tile<F16>(smem + wid * ATT_PATCH_BYTES, dq, 1.0f, q_ok, a.dqkv + (rowbase + q0) * (size_t)ld + head * 64, (size_t)ld, Lv - q0,
                       lane);
#if ATT_ABL & 4096
  asm volatile("s_waitcnt vmcnt(0)");
#endif
  PROF_MARK(5);
  PROF_FLUSH(1);
}

// ---- backward: dK, dV --------------------------------------------------------------------------
// LDS stage layout: Q tile 8 KiB | dO tile 8 KiB | lse2[64] f32 | delta[64] f32 | words[64][4] u32
#define KV_STAGE (8192 + 8192 + 256 + 256 + 1024 + 1024)     // ... | keep-bit dwords [4 waves][2 query halves][32 keys] u32
static_assert(KV_STAGE % 128 == 0 && ATT_PATCH_BYTES % 128 == 0, "frag_row_x / frag_tr_x XOR bits 4-6 of (tile base + lane word): every tile base must be a multiple of 128 bytes");
// One 64-query tile of the dK/dV pass for a wave's 32 keys (key on the lane).  MASKED: class-2 tile (mask words from LDS);
// DROP: attention dropout on -- compile-time, so the element loops are branch-free.
template <bool MASKED, bool DROP, bool F16>
__device__ __forceinline__ void dkv_tile(const AttnArgs& a, const char* tQ, const char* tD, const float* s_lse, const float* s_dl,
                                         const uint32_t* s_w, const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], f32x16 (&dk)[2],
                                         f32x16 (&dv)[2], int b, int head, int cur, int key, int wid, float c2, int lane_in, const uint32_t* s_db,
                                         const FragLane& fl) {
  // The lane index is made opaque per tile: hipcc otherwise hoists every lane-dependent LDS offset of the 48 fragment reads out of the tile
  // loop (a dozen registers), runs out at the 256-register cap of two waves per SIMD and SPILLS them -- each reload inside a tile body was a
  // scratch load whose `s_waitcnt vmcnt(0)` also drained the LDS-DMA ring (8-11 per tile; SQ_WAIT_ANY 50 % of the wave cycles).  Recomputing
  // the offsets costs a few VALU instructions per tile.
  int lane = lane_in;
  unsigned rk_ = fl.rk, tr_ = fl.tr;       // the lane's part of every fragment address (FragLane): two registers across the tile loop
  asm volatile("" : "+v"(lane), "+v"(rk_), "+v"(tr_));
  const int l31 = lane & 31, h = lane >> 5;
  const unsigned tq = lds_addr(tQ) + rk_, td = lds_addr(tD) + rk_, tqt = lds_addr(tQ) + tr_, tdt = lds_addr(tD) + tr_;
  const float dscale = DROP ? a.inv_keep * a.scale : a.scale;
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    f32x16 sc, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sc[i] = 0.f; dp[i] = 0.f; }
    {
      // the Q / dO fragments one step ahead of their MFMAs and no further: hipcc otherwise requests all eight up front (32 registers)
      // and spills elsewhere -- every reload of a spilled register is a scratch load whose `vmcnt(0)` also drains the LDS-DMA ring
      bf16x8 qc = frag_row_x(tq, 32 * qq, 0), dc = frag_row_x(td, 32 * qq, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 qn = qc, dn = dc;
        if (s + 1 < 4) { qn = frag_row_x(tq, 32 * qq, s + 1); dn = frag_row_x(td, 32 * qq, s + 1); }
        __builtin_amdgcn_sched_barrier(0);
        sc = mma32<F16>(qc, kf[s], sc);
        dp = mma32<F16>(dc, vf[s], dp);
        __builtin_amdgcn_sched_barrier(0);
        qc = qn; dc = dn;
      }
    }
    f32x16& pv = dp;                          // P (dropped-out) overwrites dP element by element
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int qr0 = 32 * qq + 8 * g + 4 * h;      // 4 consecutive query rows of this register quad
      f32x4 l4 = *(const f32x4*)(s_lse + qr0);      // natural-log row statistics as the forward wrote them
      f32x4 d4 = *(const f32x4*)(s_dl + qr0);
      l4 *= LOG2E;
      d4 *= a.scale;
      // keep-bits: this key's dword holds the bits of the 32 queries of the half; element r is query acc_row(r, h) = (r&3) + 8*(r>>2) + 4*h
      int wsh = 0;
      if (DROP) wsh = (int)(s_db[wid * 64 + qq * 32 + l31] >> (4 * h));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float p;                              // rows q >= Lv: dO row and delta are zero, so whatever p is, nothing is added
        if (!MASKED) {
          p = fexp2(fmaf(sc[r], c2, -l4[e]));
        } else {
          const uint32_t w = s_w[(qr0 + e) * 4 + wid];
          p = fexp2(fmaf(sc[r], c2, ((w >> l31) & 1u) ? 0.f : MASK_ADD * LOG2E) - l4[e]);
        }
        if (DROP) {
          const int t = __builtin_amdgcn_sbfe(wsh, e + 8 * g, 1);      // 0 or all ones
          sc[r] = p * fmaf(and_bits(dp[r], t), dscale, -d4[e]);
          pv[r] = and_bits(p * a.inv_keep, t);
        } else {
          sc[r] = p * fmaf(dp[r], dscale, -d4[e]);
          pv[r] = p;
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = pack8t<F16>(pv, s2);
      const bf16x8 dsf = pack8t<F16>(sc, s2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dv[dt] = mma32<F16>(frag_tr_x(tdt, dt, 32 * qq + 16 * s2), pf, dv[dt]);
        dk[dt] = mma32<F16>(frag_tr_x(tqt, dt, 32 * qq + 16 * s2), dsf, dk[dt]);
      }
    }
  }
}
#define DKV_NS 4      // 70 KiB of LDS per block, two blocks per CU (register-limited)
template <bool F16>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(128))) char smem[];   // (fragment addressing XORs bits 4-6 of tile base + lane word: bases are multiples of 128)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int xb, head, b;
  att_block(a, xb, head, b);
  const int L = a.L, H = a.H, ld = 3 * a.H, T = a.T;
  const int kb0 = xb * 128, k0w = kb0 + wid * 32, key = k0w + l31;
  PROF_DECL;
  const TileMasks tmk = load_tile_masks_k(a.info, b, T, kb0 >> 6, min(k0w >> 6, T - 1), lane);      // before the row plan: see the forward
  const int Lv = a.cu ? a.cu[b + 1] - a.cu[b] : L;
  if (kb0 >= Lv) return;
  const bool wave_on = k0w < Lv, k_ok = key < Lv;
  const size_t rowbase = a.cu ? (size_t)a.cu[b] : (size_t)b * L;
  const size_t lrow = (size_t)b * L;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.qkv, 0, a.bytes_qkv, 0x00020000);
  const dma_rsrc_t rsq = dma_rsrc(a.qkv, a.bytes_qkv);
  __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dctx, 0, a.bytes_ctx, 0x00020000);

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = load_rowfrag_global(rs, a.bytes_qkv, rowbase + key, k_ok, ld, H + head * 64 + 16 * s + 8 * h);
    vf[s] = load_rowfrag_global(rs, a.bytes_qkv, rowbase + key, k_ok, ld, 2 * H + head * 64 + 16 * s + 8 * h);
  }
  const float c2 = a.scale * LOG2E;
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

  const int Lq = a.qlim ? min(Lv, a.qlim[b]) : Lv;        // rows that are queries: the others arrive as zero rows with zero statistics
  const int nqt = (Lq + 63) / 64;
  const int ka = kb0 >> 6;
  const int kw0 = kb0 >> 5;           // first of the block's 4 mask words
  const size_t sbase = ((size_t)b * a.A + head) * L;

  // Q / dO tiles, their softmax statistics and this key block's mask words all arrive by LDS-DMA into a ring of DKV_NS
  // stages (rows past Lv are zero-filled: a zero dO row and a zero delta make that row contribute nothing to dK or dV)
  const dma_rsrc_t rsdo = dma_rsrc(a.dctx, a.bytes_ctx);
  const dma_rsrc_t rsl = dma_rsrc(a.lse_in, a.bytes_stat), rsdl = dma_rsrc(a.delta, a.bytes_stat), rsw = dma_rsrc(a.bits, a.bytes_bits);
  const bool use_db = a.drop_on != 0;
  const dma_rsrc_t rsdb = dma_rsrc(a.dropbits, a.bytes_dbits);
  // this lane's keep-bit dword inside a (32-query x 64-key) block: key half kk, then the accumulator order of the key (2*r + h)
  const int kq32 = (lane & 31), kkw = (k0w >> 5) & 1;
  const unsigned db_in_block = (unsigned)(kkw * 32 + 2 * (((kq32 >> 3) << 2) | (kq32 & 3)) + ((kq32 >> 2) & 1));
  int cur = next_tile(tmk.need, -1, nqt);
  int iss = cur, issued = 0, done = 0;
  auto issue = [&]() {
    char* st_ = smem + (issued % DKV_NS) * KV_STAGE;
    tile_dma<4>(rsq, a.bytes_qkv, rowbase, iss * 64, Lq, ld, head * 64, st_, wid, lane);
    tile_dma<4>(rsdo, a.bytes_ctx, rowbase, iss * 64, Lq, H, head * 64, st_ + 8192, wid, lane);
    {   // 64 rows x 4 mask words: 256 dwords, 64 per wave
      const int idx = wid * 64 + lane, qq = iss * 64 + (idx >> 2), wi = kw0 + (idx & 3);
      const bool ok = qq < Lq && wi < a.W;
      const unsigned off = (unsigned)(((lrow + qq) * (size_t)a.W + wi) * 4);
      lds_dma4(rsw, (MV_LDS void*)(st_ + 16384 + 512 + wid * 256), ok ? off : a.bytes_bits);
    }
    {   // lse (even waves) / delta (odd waves) of the 64 query rows; issued by every wave so that the counted waits are uniform
      const int qi = iss * 64 + lane;
      const unsigned off = (unsigned)((sbase + qi) * 4);
      if (wid & 1) lds_dma4(rsdl, (MV_LDS void*)(st_ + 16384 + 256), qi < Lq ? off : a.bytes_stat);
      else lds_dma4(rsl, (MV_LDS void*)(st_ + 16384), qi < Lq ? off : a.bytes_stat);
    }
    if (use_db) {   // keep-bit dwords of this wave's 32 keys for the tile's two 32-query halves (lane >> 5)
      const size_t blk = dbits_block(a, (size_t)b * a.A + head, 2 * iss + (lane >> 5), k0w >> 6);
      const unsigned off = (unsigned)((blk * 2 + db_in_block) * 4);
      lds_dma4(rsdb, (MV_LDS void*)(st_ + 16384 + 512 + 1024 + wid * 256), (k0w < Lv && (2 * iss + (lane >> 5)) * 32 < Lq) ? off : a.bytes_dbits);
    }
    ++issued;
    iss = next_tile(tmk.need, iss, nqt);
  };
#pragma unroll
  for (int i = 0; i < DKV_NS - 1; ++i)
    if (iss < nqt) issue();
  const FragLane fl = frag_lane(lane);
  PROF_MARK(0);
  while (cur < nqt) {
    if (use_db) att_wait_stage<7>(issued - done - 1);
    else att_wait_stage<6>(issued - done - 1);
    PROF_MARK(1);
    __builtin_amdgcn_s_barrier();
    PROF_MARK(2);
    __builtin_amdgcn_sched_barrier(0);
    if (!ATT_ISSUE_LATE && iss < nqt) issue();
    const char* st = smem + (done % DKV_NS) * KV_STAGE;
    const uint32_t* s_db = (const uint32_t*)(st + 16384 + 512 + 1024);
    const char* tQ = st;
    const char* tD = st + 8192;
    const float* s_lse = (const float*)(st + 16384);
    const float* s_dl = (const float*)(st + 16384 + 256);
    const uint32_t* s_w = (const uint32_t*)(st + 16384 + 512);
    const int cls = !wave_on ? 0 : (((tmk.w_is1 >> cur) & 1) ? 1 : (((tmk.w_nz >> cur) & 1) ? 2 : 0));
    if (wave_on && cls != 0) {
      const int variant = (cls == 1 ? 0 : 1) | (use_db ? 2 : 0);        // one wave-uniform dispatch per tile
#define DKV_CALL(M_, D_) dkv_tile<M_, D_, F16>(a, tQ, tD, s_lse, s_dl, s_w, kf, vf, dk, dv, b, head, cur, key, wid, c2, lane, s_db, fl)
      switch (variant) {
        case 0: DKV_CALL(false, false); break;
        case 1: DKV_CALL(true, false); break;
        case 2: DKV_CALL(false, true); break;
        default: DKV_CALL(true, true); break;
      }
#undef DKV_CALL
    }
#if ATT_ABL & 4096
    asm volatile("s_nop 0" ::"v"(dk[0][0]), "v"(dv[1][15]));
#endif
    PROF_MARK(3);
    if (ATT_ISSUE_LATE && iss < nqt) issue();
    cur = next_tile(tmk.need, cur, nqt);
    ++done;
    PROF_MARK(4);
  }
  // epilogue: whole rows through a per-wave LDS patch (dK, then dV through the same patch)
  __builtin_amdgcn_s_barrier();
  bf16_t* krow0 = a.dqkv + (rowbase + k0w) * (size_t)ld + H + head * 64;
  store_rows_tile<F16>(smem + wid * ATT_PATCH_BYTES, dk, 1.0f, k_ok, krow0, (size_t)ld, Lv - k0w, lane);
  store_rows_tile<F16>(smem + wid * ATT_PATCH_BYTES, dv, 1.0f, k_ok, krow0 + H, (size_t)ld, Lv - k0w, lane);
#if ATT_ABL & 4096
  asm volatile("s_waitcnt vmcnt(0)");
#endif
  PROF_MARK(5);
  PROF_FLUSH(2);
}
#if ATT_ABL & 4096
extern "C" int mv_debug_attn_prof(unsigned long long* out48) {       // variant libraries only: totals since the last call, then cleared
  unsigned long long z[24] = {0};
  hipError_t e = hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_att_prof), sizeof(z));
  if (e == hipSuccess) e = hipMemcpyFromSymbol(out48 + 24, HIP_SYMBOL(g_att_tile), sizeof(z));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_att_prof), z, sizeof(z));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_att_tile), z, sizeof(z));
  return (int)e;
}
#endif

// =========================================================================================
// plain VALU kernels (any dtype, dh <= 128): exact-fp32 path and cross-check
// =========================================================================================
template <typename T>
struct SArgs {
  const T* qkv; const T* ctx; const T* dctx; T* out; T* dqkv;
  const uint32_t* bits; float* lse; const float* lse_in; float* delta;
  int B, L, A, H, W, dh;
  float scale;
  const unsigned long long* dropbits;   // nullable = dropout off; layout of AttnArgs::dropbits
  float inv_keep;
  int NQB, NKT;
};

// delta[b,h,q] = sum_d dctx*ctx   (one wave per (b,q,h))
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ ctx, const T* __restrict__ dctx, float* __restrict__ delta, int B, int L,
                                  int A, int H, int dh) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= B * L * A) return;
  const int hd = wave % A, row = wave / A;   // row = b*L + q
  const T* c = ctx + (size_t)row * H + hd * dh;
  const T* d = dctx + (size_t)row * H + hd * dh;
  float s = 0.f;
  for (int i = lane; i < dh; i += 64) s += ldf<T>(c + i) * ldf<T>(d + i);
  s = wave_sum(s);
  if (lane == 0) {
    const int b = row / L, q = row - b * L;
    delta[((size_t)b * A + hd) * L + q] = s;
  }
}

// one wave per (b, h, q)
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_simple_kernel(SArgs<T> a) {
  __shared__ float sq[4][128];
  __shared__ float sp[4][64];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  const long long wave = (long long)blockIdx.x * 4 + wl;
  if (wave >= (long long)a.B * a.A * a.L) return;
  const int q = (int)(wave % a.L);
  const int hd = (int)((wave / a.L) % a.A);
  const int b = (int)(wave / ((long long)a.L * a.A));
  const int ld = 3 * a.H, dh = a.dh;
  const size_t rb = (size_t)b * a.L;
  const T* qp = a.qkv + (rb + q) * ld + hd * dh;
  for (int i = lane; i < dh; i += 64) sq[wl][i] = ldf<T>(qp + i);
  float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;
  const uint32_t* wrow = a.bits + (rb + q) * a.W;
  for (int k0 = 0; k0 < a.L; k0 += 64) {
    const int k = k0 + lane;
    float s = -INFINITY;
    if (k < a.L) {
      const T* kp = a.qkv + (rb + k) * ld + a.H + hd * dh;
      float acc = 0.f;
      for (int i = 0; i < dh; ++i) acc = fmaf(sq[wl][i], ldf<T>(kp + i), acc);
      s = acc * a.scale + (((wrow[k >> 5] >> (k & 31)) & 1u) ? 0.f : MASK_ADD);
    }
    const float mx = wave_max(s);
    const float mn = fmaxf(m, mx);
    const float alpha = expf(m - mn);
    const float p = (k < a.L) ? expf(s - mn) : 0.f;
    l = l * alpha + wave_sum(p);
    m = mn;
    float pd = p;
    if (a.dropbits && k < a.L) pd = attn_keep_bit(a.dropbits, a.NQB, a.NKT, (size_t)b * a.A + hd, q, k) ? p * a.inv_keep : 0.f;
    sp[wl][lane] = pd;
    o0 *= alpha; o1 *= alpha;
    const int kn = min(64, a.L - k0);
    for (int j = 0; j < kn; ++j) {
      const T* vp = a.qkv + (rb + k0 + j) * ld + 2 * a.H + hd * dh;
      const float pj = sp[wl][j];
      if (lane < dh) o0 = fmaf(pj, ldf<T>(vp + lane), o0);
      if (lane + 64 < dh) o1 = fmaf(pj, ldf<T>(vp + lane + 64), o1);
    }
  }
  T* op = a.out + (rb + q) * a.H + hd * dh;
  if (lane < dh) stf<T>(op + lane, o0 / l);
  if (lane + 64 < dh) stf<T>(op + lane + 64, o1 / l);
  if (lane == 0) a.lse[((size_t)b * a.A + hd) * a.L + q] = m + logf(l);
}

// dQ: one wave per (b,h,q)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_simple_kernel(SArgs<T> a) {
  __shared__ float sq[4][128];
  __shared__ float sdo[4][128];
  __shared__ float sds[4][64];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  const long long wave = (long long)blockIdx.x * 4 + wl;
  if (wave >= (long long)a.B * a.A * a.L) return;
  const int q = (int)(wave % a.L);
  const int hd = (int)((wave / a.L) % a.A);
  const int b = (int)(wave / ((long long)a.L * a.A));
  const int ld = 3 * a.H, dh = a.dh;
  const size_t rb = (size_t)b * a.L;
  const T* qp = a.qkv + (rb + q) * ld + hd * dh;
  const T* dop = a.dctx + (rb + q) * a.H + hd * dh;
  for (int i = lane; i < dh; i += 64) { sq[wl][i] = ldf<T>(qp + i); sdo[wl][i] = ldf<T>(dop + i); }
  const size_t si = ((size_t)b * a.A + hd) * a.L + q;
  const float lse = a.lse_in[si], dl = a.delta[si];
  const uint32_t* wrow = a.bits + (rb + q) * a.W;
  float g0 = 0.f, g1 = 0.f;
  for (int k0 = 0; k0 < a.L; k0 += 64) {
    const int k = k0 + lane;
    float ds = 0.f;
    if (k < a.L) {
      const T* kp = a.qkv + (rb + k) * ld + a.H + hd * dh;
      const T* vp = kp + a.H;
      float acc = 0.f, dp = 0.f;
      for (int i = 0; i < dh; ++i) { acc = fmaf(sq[wl][i], ldf<T>(kp + i), acc); dp = fmaf(sdo[wl][i], ldf<T>(vp + i), dp); }
      const float s = acc * a.scale + (((wrow[k >> 5] >> (k & 31)) & 1u) ? 0.f : MASK_ADD);
      if (a.dropbits) dp = attn_keep_bit(a.dropbits, a.NQB, a.NKT, (size_t)b * a.A + hd, q, k) ? dp * a.inv_keep : 0.f;
      ds = expf(s - lse) * (dp - dl) * a.scale;
    }
    sds[wl][lane] = ds;
    const int kn = min(64, a.L - k0);
    for (int j = 0; j < kn; ++j) {
      const T* kp = a.qkv + (rb + k0 + j) * ld + a.H + hd * dh;
      const float dj = sds[wl][j];
      if (lane < dh) g0 = fmaf(dj, ldf<T>(kp + lane), g0);
      if (lane + 64 < dh) g1 = fmaf(dj, ldf<T>(kp + lane + 64), g1);
    }
  }
  T* gp = a.dqkv + (rb + q) * ld + hd * dh;
  if (lane < dh) stf<T>(gp + lane, g0);
  if (lane + 64 < dh) stf<T>(gp + lane + 64, g1);
}

// dK, dV: one wave per (b,h,key)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_simple_kernel(SArgs<T> a) {
  __shared__ float sk[4][128];
  __shared__ float sv[4][128];
  __shared__ float sds[4][64];
  __shared__ float spp[4][64];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  const long long wave = (long long)blockIdx.x * 4 + wl;
  if (wave >= (long long)a.B * a.A * a.L) return;
  const int k = (int)(wave % a.L);
  const int hd = (int)((wave / a.L) % a.A);
  const int b = (int)(wave / ((long long)a.L * a.A));
  const int ld = 3 * a.H, dh = a.dh;
  const size_t rb = (size_t)b * a.L;
  const T* kp = a.qkv + (rb + k) * ld + a.H + hd * dh;
  const T* vp = kp + a.H;
  for (int i = lane; i < dh; i += 64) { sk[wl][i] = ldf<T>(kp + i); sv[wl][i] = ldf<T>(vp + i); }
  float gk0 = 0.f, gk1 = 0.f, gv0 = 0.f, gv1 = 0.f;
  const size_t sb = ((size_t)b * a.A + hd) * a.L;
  for (int q0 = 0; q0 < a.L; q0 += 64) {
    const int q = q0 + lane;
    float ds = 0.f, p = 0.f;
    if (q < a.L) {
      const T* qp = a.qkv + (rb + q) * ld + hd * dh;
      const T* dop = a.dctx + (rb + q) * a.H + hd * dh;
      float acc = 0.f, dp = 0.f;
      for (int i = 0; i < dh; ++i) { acc = fmaf(sk[wl][i], ldf<T>(qp + i), acc); dp = fmaf(sv[wl][i], ldf<T>(dop + i), dp); }
      const uint32_t w = a.bits[(rb + q) * a.W + (k >> 5)];
      const float s = acc * a.scale + (((w >> (k & 31)) & 1u) ? 0.f : MASK_ADD);
      p = expf(s - a.lse_in[sb + q]);
      float keepf = 1.0f;
      if (a.dropbits) keepf = attn_keep_bit(a.dropbits, a.NQB, a.NKT, (size_t)b * a.A + hd, q, k) ? a.inv_keep : 0.f;
      ds = p * (keepf * dp - a.delta[sb + q]) * a.scale;
      p *= keepf;
    }
    sds[wl][lane] = ds;
    spp[wl][lane] = p;
    const int qn = min(64, a.L - q0);
    for (int j = 0; j < qn; ++j) {
      const T* qp = a.qkv + (rb + q0 + j) * ld + hd * dh;
      const T* dop = a.dctx + (rb + q0 + j) * a.H + hd * dh;
      const float dj = sds[wl][j], pj = spp[wl][j];
      if (lane < dh) { gk0 = fmaf(dj, ldf<T>(qp + lane), gk0); gv0 = fmaf(pj, ldf<T>(dop + lane), gv0); }
      if (lane + 64 < dh) { gk1 = fmaf(dj, ldf<T>(qp + lane + 64), gk1); gv1 = fmaf(pj, ldf<T>(dop + lane + 64), gv1); }
    }
  }
  T* gk = a.dqkv + (rb + k) * ld + a.H + hd * dh;
  T* gv = gk + a.H;
  if (lane < dh) { stf<T>(gk + lane, gk0); stf<T>(gv + lane, gv0); }
  if (lane + 64 < dh) { stf<T>(gk + lane + 64, gk1); stf<T>(gv + lane + 64, gv1); }
}

// =========================================================================================
// host
// =========================================================================================
static inline size_t dropbits_words(int B, int L, int A) { return (size_t)B * A * ((L + 31) / 32) * ((L + 63) / 64) * 64; }   // uint32 count
// bits per uniform of mv_attn_dropmask (8, 12 or 16): P(drop) = thr / 2^planes with thr = round(p * 2^planes); survivors are scaled
// by 2^planes / (2^planes - thr).  p = 0.1: 16 -> 0.100006, 12 -> 0.100098, 8 -> 0.101563.
#define g_mv_attn_planes (mv_knob(MV_KNOB_ATTN_PLANES))
static inline unsigned attn_thr16(float p) {          // threshold at the current plane count
  const int full = 1 << g_mv_attn_planes;
  int t = (int)(p * (float)full + 0.5f);
  return (unsigned)(t < 1 ? 1 : (t > full - 1 ? full - 1 : t));
}
static inline float attn_inv_keep(float p) {
  const float full = (float)(1 << g_mv_attn_planes);
  return p > 0.f ? full / (full - (float)attn_thr16(p)) : 1.0f;
}
// p_drop > 0 needs the keep-bits (64-byte aligned: scalar loads of 64-byte pieces; 32-bit buffer offsets)
static inline int dropbits_check(float p_drop, const uint32_t* dropbits, int B, int L, int A) {
  if (p_drop <= 0.f) return MV_OK;
  if (p_drop >= 1.f || !dropbits) return MV_E_ARG;
  if ((((uintptr_t)dropbits) & 63) || dropbits_words(B, L, A) * 4 >= 0x7fffffffULL) return MV_E_SHAPE;
  return MV_OK;
}

extern "C" int mv_attn_dropmask(float p_drop, unsigned long long drop_key, int B, int L, int A, const int32_t* cu, uint32_t* dropbits,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dropbits || B <= 0 || L <= 0 || A <= 0 || p_drop <= 0.f || p_drop >= 1.f) return MV_E_ARG;
  const size_t nwords = dropbits_words(B, L, A) / 2;         // 64-bit words
  if (nwords * 32 >= (1ull << 32)) return MV_E_SHAPE;         // 32 hash counters per word, 32-bit counters
  const int rc = dropbits_check(p_drop, dropbits, B, L, A);
  if (rc) return rc;
#define DM_LAUNCH(PL_)                                                                                                                  \
  hipLaunchKernelGGL(attn_dropmask_kernel<PL_>, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, stream, (unsigned)(drop_key & 0xffffffffULL), \
                     (unsigned)(drop_key >> 32), attn_thr16(p_drop), A, L, (L + 31) / 32, (L + 63) / 64, cu, (unsigned long long*)dropbits, nwords)
  if (g_mv_attn_planes == 8) DM_LAUNCH(8);
  else if (g_mv_attn_planes == 12) DM_LAUNCH(12);
  else DM_LAUNCH(16);
#undef DM_LAUNCH
  MV_CHECK_LAUNCH();
  return MV_OK;
}

template <typename T>
static int launch_simple_fwd(const void* qkv, const uint32_t* bits, void* ctx, float* lse, int B, int L, int A, int dh, float p_drop,
                             const uint32_t* dropbits, hipStream_t stream) {
  SArgs<T> s{};
  s.dropbits = p_drop > 0.f ? (const unsigned long long*)dropbits : nullptr; s.inv_keep = attn_inv_keep(p_drop);
  s.NQB = (L + 31) / 32; s.NKT = (L + 63) / 64;
  s.qkv = (const T*)qkv; s.out = (T*)ctx; s.bits = bits; s.lse = lse;
  s.B = B; s.L = L; s.A = A; s.H = A * dh; s.W = (L + 31) / 32; s.dh = dh; s.scale = 1.0f / sqrtf((float)dh);
  const long long waves = (long long)B * A * L;
  hipLaunchKernelGGL(attn_fwd_simple_kernel<T>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, s);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_attn_fwd(int dtype, const void* qkv, const uint32_t* bits, const uint8_t* tileinfo, void* ctx, void* ctx_bf16,
                           float* lse, int B, int L, int A, int dh, float p_drop, const uint32_t* dropbits,
                           const int32_t* cu, int total_rows, const int32_t* qlim, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !bits || !tileinfo || !ctx || !lse || B <= 0 || L <= 0 || A <= 0 || dh <= 0) return MV_E_ARG;
  if (!mv_dtype_ok(dtype)) return MV_E_DTYPE;
  if (ctx_bf16 && dtype != MV_F16) return MV_E_DTYPE;     // the second context output is the bf16 copy of an f16 forward
  const int rcd = dropbits_check(p_drop, dropbits, B, L, A);
  if (rcd) return rcd;
  const int H = A * dh;
  if (mv_is16(dtype) && g_mv_impl == 0) {
    if (dh != 64) return MV_E_SHAPE;
    if (cu && (total_rows <= 0 || total_rows > B * L)) return MV_E_ARG;
    const size_t nrow = cu ? (size_t)total_rows : (size_t)B * L;
    const size_t bq = nrow * 3 * H * 2;
    if (bq >= 0x7fffffffULL || (((uintptr_t)qkv) & 15) || (((uintptr_t)ctx) & 7) || (((uintptr_t)ctx_bf16) & 7)) return MV_E_SHAPE;
    AttnArgs a{};
    a.cu = cu; a.qlim = qlim; a.order = mv_knob(MV_KNOB_ATTN_ORDER);
    a.qkv = (const bf16_t*)qkv; a.out = (bf16_t*)ctx; a.out2 = (bf16_t*)ctx_bf16; a.bits = bits; a.info = tileinfo; a.lse = lse;
    a.B = B; a.L = L; a.A = A; a.H = H; a.W = (L + 31) / 32; a.T = (L + 63) / 64;
    a.scale = 1.0f / sqrtf((float)dh);
    a.bytes_qkv = (unsigned)bq;
    a.drop_on = p_drop > 0.f; a.inv_keep = attn_inv_keep(p_drop);
    a.dropbits = a.drop_on ? dropbits : nullptr; a.bytes_dbits = (unsigned)(dropbits_words(B, L, A) * 4);
    a.NQB = (L + 31) / 32; a.NKT = (L + 63) / 64;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_NS * 16384);
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_NS * 16384);
      attr = true;
    }
    if (dtype == MV_F16) hipLaunchKernelGGL(attn_fwd_mfma_kernel<true>, dim3((L + 127) / 128, A, B), dim3(256), FWD_NS * 16384, stream, a);
    else hipLaunchKernelGGL(attn_fwd_mfma_kernel<false>, dim3((L + 127) / 128, A, B), dim3(256), FWD_NS * 16384, stream, a);
    MV_CHECK_LAUNCH();
    return MV_OK;
  }
  if (dh > 128 || cu || qlim) return MV_E_SHAPE;       // packed rows / query limits: MFMA kernels only
  if (dtype == MV_F16) {                         // VALU cross-check of the f16 forward (mv_set_impl(1))
    int rc = launch_simple_fwd<f16_t>(qkv, bits, ctx, lse, B, L, A, dh, p_drop, dropbits, stream);
    if (rc == MV_OK && ctx_bf16) rc = mv_cast(ctx, MV_F16, ctx_bf16, MV_BF16, (size_t)B * L * H, stream_);
    return rc;
  }
  return dtype == MV_F32 ? launch_simple_fwd<float>(qkv, bits, ctx, lse, B, L, A, dh, p_drop, dropbits, stream)
                         : launch_simple_fwd<bf16_t>(qkv, bits, ctx, lse, B, L, A, dh, p_drop, dropbits, stream);
}

template <typename T>
static int launch_simple_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, const uint32_t* bits,
                             void* dqkv, float* delta, int B, int L, int A, int dh, float p_drop, const uint32_t* dropbits, hipStream_t stream) {
  SArgs<T> s{};
  s.dropbits = p_drop > 0.f ? (const unsigned long long*)dropbits : nullptr; s.inv_keep = attn_inv_keep(p_drop);
  s.NQB = (L + 31) / 32; s.NKT = (L + 63) / 64;
  s.qkv = (const T*)qkv; s.ctx = (const T*)ctx; s.dctx = (const T*)dctx; s.dqkv = (T*)dqkv; s.bits = bits;
  s.lse_in = lse; s.delta = delta;
  s.B = B; s.L = L; s.A = A; s.H = A * dh; s.W = (L + 31) / 32; s.dh = dh; s.scale = 1.0f / sqrtf((float)dh);
  const long long waves = (long long)B * A * L;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  hipLaunchKernelGGL(attn_delta_kernel<T>, dim3(blocks), dim3(256), 0, stream, s.ctx, s.dctx, delta, B, L, A, s.H, dh);
  MV_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dq_simple_kernel<T>, dim3(blocks), dim3(256), 0, stream, s);
  MV_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkv_simple_kernel<T>, dim3(blocks), dim3(256), 0, stream, s);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_attn_bwd(int dtype, const void* qkv, const void* ctx, const void* dctx, const float* lse, const uint32_t* bits,
                           const uint8_t* tileinfo, void* dqkv, float* delta, int B, int L, int A, int dh, float p_drop,
                           const uint32_t* dropbits, const int32_t* cu, int total_rows, const int32_t* qlim, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !ctx || !dctx || !lse || !bits || !tileinfo || !dqkv || !delta || B <= 0 || L <= 0 || A <= 0 || dh <= 0)
    return MV_E_ARG;
  if (!mv_dtype_ok(dtype)) return MV_E_DTYPE;
  const int rcd = dropbits_check(p_drop, dropbits, B, L, A);
  if (rcd) return rcd;
  const int H = A * dh;
  if (mv_is16(dtype) && g_mv_impl == 0) {
    if (dh != 64) return MV_E_SHAPE;
    if (cu && (total_rows <= 0 || total_rows > B * L)) return MV_E_ARG;
    const size_t nrow = cu ? (size_t)total_rows : (size_t)B * L;
    const size_t bq = nrow * 3 * H * 2, bc = nrow * H * 2;
    if (bq >= 0x7fffffffULL || (((uintptr_t)qkv) & 15) || (((uintptr_t)dctx) & 15) || (((uintptr_t)dqkv) & 7)) return MV_E_SHAPE;
    AttnArgs a{};
    a.cu = cu; a.qlim = qlim; a.order = mv_knob(MV_KNOB_ATTN_ORDER);
    a.qkv = (const bf16_t*)qkv; a.ctx = (const bf16_t*)ctx; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
    a.bits = bits; a.info = tileinfo; a.lse_in = lse; a.delta = delta; a.delta_out = delta;
    a.B = B; a.L = L; a.A = A; a.H = H; a.W = (L + 31) / 32; a.T = (L + 63) / 64;
    a.scale = 1.0f / sqrtf((float)dh);
    a.bytes_qkv = (unsigned)bq; a.bytes_ctx = (unsigned)bc;
    a.bytes_stat = (unsigned)((size_t)B * A * L * 4); a.bytes_bits = (unsigned)((size_t)B * L * a.W * 4);
    a.drop_on = p_drop > 0.f; a.inv_keep = attn_inv_keep(p_drop);
    a.dropbits = a.drop_on ? dropbits : nullptr; a.bytes_dbits = (unsigned)(dropbits_words(B, L, A) * 4);
    a.NQB = (L + 31) / 32; a.NKT = (L + 63) / 64;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_NS * 16384);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_NS * KV_STAGE);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_NS * 16384);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_NS * KV_STAGE);
      attr = true;
    }
    dim3 grid((L + 127) / 128, A, B);
    if (dtype == MV_F16) hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<true>, grid, dim3(256), DQ_NS * 16384, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<false>, grid, dim3(256), DQ_NS * 16384, stream, a);
    MV_CHECK_LAUNCH();
    if (dtype == MV_F16) hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<true>, grid, dim3(256), DKV_NS * KV_STAGE, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<false>, grid, dim3(256), DKV_NS * KV_STAGE, stream, a);
    MV_CHECK_LAUNCH();
    return MV_OK;
  }
  if (dh > 128 || cu || qlim) return MV_E_SHAPE;
  if (dtype == MV_F16) return launch_simple_bwd<f16_t>(qkv, ctx, dctx, lse, bits, dqkv, delta, B, L, A, dh, p_drop, dropbits, stream);
  return dtype == MV_F32 ? launch_simple_bwd<float>(qkv, ctx, dctx, lse, bits, dqkv, delta, B, L, A, dh, p_drop, dropbits, stream)
                         : launch_simple_bwd<bf16_t>(qkv, ctx, dctx, lse, bits, dqkv, delta, B, L, A, dh, p_drop, dropbits, stream);
}
