// On-device assembly of one pretraining mini-batch from raw token ids: MLM corruption, label
// layout, [SEP]/[PAD] placement, segment ids, mask descriptors and the compact labelled-row index the fused MLM head
// consumes.  Replaces the per-sample python of data/dataset_origin.py:102-135 and random_word (:183-209); the
// [B,L,L] int64 mask the reference builds at :138-176 is never materialised (mv_mask_build consumes the descriptors).
// Integer work, a few hundred KiB per batch: one block per sample, no LDS tiling needed.
#include "mv_common.h"

namespace {

constexpr int TOK_PAD = 0, TOK_SEP = 102, TOK_MASK = 103;
constexpr int NT = 256;

// draws of random_word's two random sources for token (b, i): u = random.random() stand-in on a 24-bit grid,
// r = random.randrange(vocab) stand-in.  Counter-based, so any (b, i) can be regenerated independently.
__global__ void mlm_draws_kernel(unsigned k0, unsigned k1, int n, int vocab, float* u, int32_t* rnd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned h0 = mv_hash32(2u * (unsigned)i, k0, k1), h1 = mv_hash32(2u * (unsigned)i + 1u, k0, k1);
  u[i] = (float)(h0 >> 8) * (1.0f / 16777216.0f);
  rnd[i] = (int32_t)(((unsigned long long)h1 * (unsigned long long)vocab) >> 32);
}

__device__ __forceinline__ int block_sum(int v, int* s_red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < NT / 64; ++w) t += s_red[w];
  return t;
}

// The selection rule of random_word, evaluated in double like the python it replaces:
//   p < 0.15 -> selected; p/0.15 < 0.8 -> [MASK]; < 0.9 -> random id; else keep.   returns 0 none, 1 mask, 2 random, 3 keep
__device__ __forceinline__ int mlm_action(float uf) {
  double p = (double)uf;
  if (!(p < 0.15)) return 0;
  p /= 0.15;
  return p < 0.8 ? 1 : (p < 0.9 ? 2 : 3);
}

// one block per sample
__global__ __launch_bounds__(NT) void mlm_corrupt_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ lengths,
                                                         const float* __restrict__ u, const int32_t* __restrict__ rnd,
                                                         const int32_t* __restrict__ family, int N, int S,
                                                         int64_t* __restrict__ input_txt, int64_t* __restrict__ segment,
                                                         int64_t* __restrict__ txt_labels, int32_t* __restrict__ n_ids,
                                                         int32_t* __restrict__ desc, int32_t* __restrict__ counts) {
  __shared__ int s_red[NT / 64];
  const int b = blockIdx.x, T = S + 1, L = S + N + 3, n2 = N + 2;
  const int len = min(max(lengths[b], 0), S);
  const int64_t* id = ids + (size_t)b * S;
  const float* ub = u + (size_t)b * S;
  const int32_t* rb = rnd + (size_t)b * S;
  int64_t* txt = input_txt + (size_t)b * T;
  int64_t* lab = txt_labels + (size_t)b * L;
  int mine = 0;
  for (int i = threadIdx.x; i < L; i += NT) {
    // labels: [-100]*(N+2) | text labels | -100 for the text [SEP] and the pads   (dataset_origin.py:108-131)
    int64_t lv = -100;
    const int t = i - n2;
    if (t >= 0 && t < T) {
      int64_t tok = TOK_PAD;
      if (t < len) {
        const int act = mlm_action(ub[t]);
        tok = id[t];
        if (act) { lv = tok; ++mine; }
        if (act == 1) tok = TOK_MASK;
        else if (act == 2) tok = rb[t];
      } else if (t == len) {
        tok = TOK_SEP;
      }
      txt[t] = tok;
      segment[(size_t)b * T + t] = 1;            // dataset_origin.py:129: 1 over all T positions, pads included
    }
    lab[i] = lv;
  }
  const int total = block_sum(mine, s_red);
  if (threadIdx.x == 0) {
    int c = total;
    if (total == 0 && len > 0) {                 // "at least one mask" (dataset_origin.py:204-207)
      lab[n2] = id[0];
      txt[0] = TOK_MASK;
      c = 1;
    }
    counts[b] = c;
    n_ids[b] = len + 1;
    if (desc) {
      desc[3 * b + 0] = family ? family[b] : 0;
      desc[3 * b + 1] = n2;
      desc[3 * b + 2] = n2 + len + 1;
    }
  }
}

// ordered (row-major over [B, L]) compaction of the labelled positions; one block per sample
__global__ __launch_bounds__(NT) void label_index_kernel(const int64_t* __restrict__ txt_labels, const int32_t* __restrict__ counts,
                                                         int B, int L, int32_t* __restrict__ rows, int32_t* __restrict__ lids,
                                                         int32_t* __restrict__ n_labels) {
  __shared__ int s_red[NT / 64];
  __shared__ int s_wave[NT / 64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int part = 0;
  for (int i = threadIdx.x; i < b; i += NT) part += counts[i];
  int base = block_sum(part, s_red);
  if (b == B - 1 && threadIdx.x == 0) *n_labels = base + counts[b];
  const int64_t* lab = txt_labels + (size_t)b * L;
  for (int i0 = 0; i0 < L; i0 += NT) {
    const int i = i0 + threadIdx.x;
    const int64_t v = i < L ? lab[i] : -100;
    const bool on = v != -100;
    const unsigned long long bal = __ballot(on);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wid] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wid; ++w) off += s_wave[w];
    if (on) {
      rows[off + before] = b * L + i;
      lids[off + before] = (int32_t)v;
    }
    for (int w = 0; w < NT / 64; ++w) base += s_wave[w];
  }
}

// Packed-row plan: sample b keeps its first vl[b] positions (everything after its text [SEP] is padding that no valid
// query can see in the full / seq2seq / 1-D mask families).  cu = exclusive prefix sums of vl; rowmap[cu[b] + p] = b*L + p;
// inv[b*L + p] = cu[b] + p for p < vl[b], -1 for the dropped positions.
__global__ __launch_bounds__(NT) void pack_plan_kernel(const int32_t* __restrict__ desc, int B, int L, int32_t* __restrict__ cu,
                                                       int32_t* __restrict__ rowmap, int32_t* __restrict__ inv) {
  __shared__ int s_red[NT / 64];
  const int b = blockIdx.x;
  int part = 0;
  for (int i = threadIdx.x; i < b; i += NT) part += min(max(desc[3 * i + 2], 0), L);
  const int base = block_sum(part, s_red);
  const int vl = min(max(desc[3 * b + 2], 0), L);
  if (threadIdx.x == 0) {
    cu[b] = base;
    if (b == B - 1) cu[B] = base + vl;
  }
  for (int p = threadIdx.x; p < L; p += NT) {
    if (p < vl) rowmap[base + p] = b * L + p;
    inv[(size_t)b * L + p] = p < vl ? base + p : -1;
  }
}

// Row order of the last encoder layer: within every sample the rows the heads consume (`sel`: labelled rows + first rows) come first,
// in ascending order, then the others.  One block per sample; wave 0 assigns positions with ballots (stable partition).
#define TP_MAXL 2048
__global__ __launch_bounds__(NT) void tail_perm_kernel(const int32_t* __restrict__ cu, const int32_t* __restrict__ sel, int n_sel,
                                                       int32_t* __restrict__ perm, int32_t* __restrict__ newpos,
                                                       int32_t* __restrict__ qlim, int32_t* __restrict__ sel_new) {
  __shared__ unsigned char flag[TP_MAXL];
  const int b = blockIdx.x, lo = cu[b], n = cu[b + 1] - lo, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < n; i += NT) flag[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_sel; i += NT) {
    const int r = sel[i] - lo;
    if (r >= 0 && r < n) flag[r] = 1;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    int pos = 0;
    for (int pass = 0; pass < 2; ++pass) {
      for (int c = 0; c < n; c += 64) {
        const int r = c + lane;
        const bool f = r < n && (flag[r] != 0) == (pass == 0);
        const unsigned long long m = __ballot(f);
        if (f) {
          const int p = pos + __popcll(m & ((1ull << lane) - 1ull));
          perm[lo + p] = lo + r;
          newpos[lo + r] = lo + p;
        }
        pos += __popcll(m);
      }
      if (pass == 0 && lane == 0) qlim[b] = pos;
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int i = threadIdx.x; i < n_sel; i += NT) {
    const int r = sel[i] - lo;
    if (r >= 0 && r < n) sel_new[i] = newpos[lo + r];
  }
}

}  // namespace

extern "C" int mv_pack_plan(const int32_t* desc, int B, int L, int32_t* cu, int32_t* rowmap, int32_t* inv, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!desc || !cu || !rowmap || !inv || B <= 0 || L <= 0) return MV_E_ARG;
  if ((long long)B * L > 0x7fffffffLL) return MV_E_SHAPE;
  pack_plan_kernel<<<B, NT, 0, stream>>>(desc, B, L, cu, rowmap, inv);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_tail_perm(const int32_t* cu, int B, int L, const int32_t* sel, int n_sel, int32_t* perm, int32_t* newpos,
                           int32_t* qlim, int32_t* sel_new, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!cu || !sel || !perm || !newpos || !qlim || !sel_new || B <= 0 || L <= 0 || n_sel <= 0) return MV_E_ARG;
  if (L > TP_MAXL) return MV_E_SHAPE;
  tail_perm_kernel<<<B, NT, 0, stream>>>(cu, sel, n_sel, perm, newpos, qlim, sel_new);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_mlm_draws(unsigned long long key, int B, int S, int vocab, float* u, int32_t* rnd, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!u || !rnd || B <= 0 || S <= 0 || vocab <= 0) return MV_E_ARG;
  const int n = B * S;
  mlm_draws_kernel<<<(n + 255) / 256, 256, 0, stream>>>((unsigned)(key & 0xffffffffULL), (unsigned)(key >> 32), n, vocab, u, rnd);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_mlm_corrupt(const int64_t* ids, const int32_t* lengths, const float* u, const int32_t* rnd,
                              const int32_t* family, int B, int N, int S, int64_t* input_txt, int64_t* segment,
                              int64_t* txt_labels, int32_t* n_ids, int32_t* desc, int32_t* counts, int32_t* label_rows,
                              int32_t* label_ids, int32_t* n_labels, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ids || !lengths || !u || !rnd || !input_txt || !segment || !txt_labels || !n_ids || !counts) return MV_E_ARG;
  if (B <= 0 || N < 0 || S <= 0) return MV_E_ARG;
  if ((label_rows || label_ids || n_labels) && !(label_rows && label_ids && n_labels)) return MV_E_ARG;
  if ((long long)B * (S + N + 3) > 0x7fffffffLL) return MV_E_SHAPE;
  mlm_corrupt_kernel<<<B, NT, 0, stream>>>(ids, lengths, u, rnd, family, N, S, input_txt, segment, txt_labels, n_ids, desc, counts);
  MV_CHECK_LAUNCH();
  if (label_rows) {
    label_index_kernel<<<B, NT, 0, stream>>>(txt_labels, counts, B, S + N + 3, label_rows, label_ids, n_labels);
    MV_CHECK_LAUNCH();
  }
  return MV_OK;
}
