/*
 * medvill.h -- C ABI of the MI355X (gfx950) hot-path library for MedViLL / CXRBERT
 * cross-modal BERT pretraining (libmedvill_hip.so).
 *
 * The reference (reonaledo/Multi-modality-Self-supervision) is pure Python and has no
 * FFI layer: its boundary is the Python object protocol
 *     CXRBERT(config, args).forward(cls_tok, input_txt, attn_mask, segment, input_img, sep_tok)
 *         -> (mlm_logits[B,L,V], itm_logits[B,2])          models/cxrbert_origin.py:132-149
 *     CXRBERT_Trainer(args, train_dataloader, test_dataloader).train(epoch) / .save(epoch, path)
 *                                                            models/train_origin.py:19-20,70,254
 * which `multi-modality-self-supervision_amd/` mirrors in Python.  Underneath that
 * mirror every piece of arithmetic goes through the entry points below; each one
 * names the reference code whose work it replaces.  (INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; all data pointers are DEVICE pointers owned
 *     by the caller (torch's allocator); no allocation inside, workspaces are passed in.
 *   - every call takes the hipStream_t to enqueue on (as void*) and never synchronises.
 *   - return 0 = ok; negative = invalid argument / unsupported shape (MV_E_*);
 *     positive = hipError_t of a failed launch.  Nothing throws.
 *   - re-entrant, no global mutable state: which kernel serves a call follows from the call's arguments alone.  (The kernel-forcing
 *     knobs that tests and timing experiments use exist only in libmedvill_hip_dbg.so: include/medvill_debug.h.)
 *   - dtype: MV_F32 = exact fp32 path (plain VALU kernels; parity at 1e-3 and below),
 *            MV_BF16 / MV_F16 = 16-bit storage, fp32 accumulate, MFMA kernels (the fast path).
 *   - matrices are row-major; "ld*" are leading dimensions in ELEMENTS.
 */
#ifndef MEDVILL_H_
#define MEDVILL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_ABI_VERSION 6

/* MV_F16 is the encoding of the 16-bit path (weights' f16 shadow, stored activations, gradients): 11 significand bits
 * instead of bf16's 8 at the same MFMA rate.  At BERT-base depth bf16-encoded forward operands cannot meet the 1e-2 logit
 * tolerance (the weight rounding alone gives 1.4e-2, profiles/r02_bf16_error.txt).  Gradients in f16 need a LOSS SCALE
 * (f16 has 5 exponent bits): the loss gradients are multiplied by S where they enter the 16-bit chain (mv_ce_fwd_bwd),
 * every kernel that writes an f32 PARAMETER gradient multiplies by 1/S again (grad_unscale_dev of mv_gemm / mv_colsum /
 * mv_layernorm_bwd / mv_embed_bwd), so the flat f32 gradient always holds true values; S lives on the device and is
 * adapted by mv_scaler_update from mv_count_nonfinite's overflow count (an overflowed step is skipped by mv_adamw_step).
 * The round-2 form remains available: f16 forward operands with MV_BF16 gradient operands, for which kernels that produce
 * a forward activation write it twice from one accumulator (C3 / y_bf16 / ctx_bf16 / x0_bf16 below). */
enum { MV_F32 = 0, MV_BF16 = 1, MV_F16 = 2 };

enum {
  MV_OK = 0,
  MV_E_ARG = -1,      /* null pointer / non-positive size */
  MV_E_SHAPE = -2,    /* shape or alignment the kernels do not support */
  MV_E_DTYPE = -3,
  MV_E_WORKSPACE = -4, /* workspace too small */
  MV_E_NO_RCCL = -5,   /* mv_comm_*: librccl could not be loaded at run time */
  MV_E_COMM_BASE = -1000 /* mv_comm_*: a failed RCCL call returns MV_E_COMM_BASE - ncclResult_t (-1001, -1002, ...): negative like every
                            MV_E_* code, disjoint from hipError_t (positive, some of them above 1000) */
};

/* GEMM epilogues (mv_gemm `epi`) */
enum {
  MV_EPI_NONE = 0,      /* C = A.B                                                        */
  MV_EPI_BIAS = 1,      /* C = A.B + bias[n]                                              */
  MV_EPI_BIAS_GELU = 2, /* Z = A.B + bias -> C2 (pre-activation, kept for backward);
                           C = gelu_erf(Z)          cxrbert_origin.py:176-181 / HF BertIntermediate */
  MV_EPI_BIAS_RES = 3,  /* C = A.B + bias[n] + R[m,n]   (HF BertSelfOutput / BertOutput before LayerNorm) */
  MV_EPI_DGELU = 4,     /* C = (A.B) * gelu_erf'(R[m,n])   (backward of MV_EPI_BIAS_GELU; R = saved Z)  */
  MV_EPI_RES = 5,       /* C = A.B + R[m,n]             (backward: add the residual-branch gradient)    */
  MV_EPI_BIAS_TANH = 6, /* C = tanh(A.B + bias)         HF BertPooler, cxrbert_origin.py:130            */
  MV_EPI_BIAS_GELU_D = 7, /* Z = A.B + bias; C = gelu_erf(Z); C2 = gelu_erf'(Z): the derivative shares the forward's exp and
                             reciprocal, so the backward GEMM only multiplies (MV_EPI_MUL) and Z itself is never stored */
  MV_EPI_MUL = 8,       /* C = (A.B) * R[m,n]           (backward of MV_EPI_BIAS_GELU_D; R = saved gelu')  */
  MV_EPI_BIAS_RELU = 9,      /* C = max(A.B + bias[n], 0)            convolution + folded BatchNorm + ReLU (region encoder, eval()) */
  MV_EPI_BIAS_RES_RELU = 10  /* C = max(A.B + bias[n] + R[m,n], 0)   ... + the bottleneck's identity branch                          */
};

int mv_abi_version(void);
const char* mv_build_info(void);
/* A HIP stream whose kernels run only on the CUs of `mask_words` (n_words x 32 bits; hipExtStreamCreateWithCUMask; on a multi-XCD
 * device bit i is CU i / n_xcd of XCD i % n_xcd): lets a host partition the chip between the streams of a step. */
int mv_stream_create_cumask(const uint32_t* mask_words, int n_words, void** stream_out);
int mv_stream_destroy(void* stream);

/* ---- dense projections --------------------------------------------------------------------
 * Replaces every nn.Linear on the path and its autograd backward:
 *   cxrbert_origin.py:16,24 (image projection), HF BertSelfAttention/BertSelfOutput/
 *   BertIntermediate/BertOutput/BertPooler (call sites cxrbert_origin.py:72-73,126-130),
 *   cxrbert_origin.py:214,228-237 (MLM transform + tied decoder), :170-173 (ITM).
 * C[M,N] = epi( opA(A)[M,K] . opB(B)[K,N] ):
 *   ta = 0: A stored [M,K] (lda >= K);  ta = 1: A stored [K,M] (lda >= M)
 *   tb = 0: B stored [N,K] (ldb >= K) -- an nn.Linear weight;  tb = 1: B stored [K,N] (ldb >= N)
 * so  y = x.W^T            is (ta=0, tb=0, A=x,  B=W)
 *     dx = dy.W            is (ta=0, tb=1, A=dy, B=W)
 *     dW = dy^T.x          is (ta=1, tb=1, A=dy, B=x)
 * dtype applies to A and B (MV_F16 for every form but ta = 1, tb = 0); c_dtype to C (and C2); r_dtype to R; bias is always f32.
 * C3 (nullable, leading dimension ldc3): a second copy of C in the 16-bit encoding c3_dtype (see MV_F16 above).
 * 16-bit operands need 16-byte aligned bases and lda, ldb multiples of 8; a contraction length
 * K that is not a multiple of 8 is allowed only when the k-contiguous operand's rows are
 * zero-padded up to the next multiple of 8.
 * splitk = 0: let the library pick (up to 32 slices, as many as ws holds; without ws 1 is used).
 * splitk > 1: the K range is cut in `splitk` slices whose partial tiles go to `ws`
 * (>= splitk*M*N floats) and are summed by a second kernel; only with MV_EPI_NONE and an f32 C.
 * accumulate != 0: C += result (f32 C, MV_EPI_NONE only).
 * p_drop > 0 (MV_EPI_BIAS_RES only, N % 4 == 0): C = dropout(A.B + bias) + R -- the hidden-state dropout
 * of HF BertSelfOutput / BertOutput; mask = mv_dropout_mask(p_drop, drop_key) over index m*N + n.
 * alpha_dev (nullable; MV_EPI_NONE with an f32 C only): C = *alpha_dev * (A.B) -- the weight gradients of the f16-gradient
 * path are un-scaled (1 / loss scale, read from the device) where they are written.
 * colsum_part (nullable): f32 [2*ceil(M/256)][N] -- the kernel also writes the column sums of C, one partial row per 128-row
 * tile half, so a bias gradient needs no second pass over C (fold with mv_colsum_partials).  Only where the 256x256 MFMA
 * kernel runs with 16-bit C, N % 256 == 0, 16-byte aligned outputs, no split-K (else MV_E_SHAPE: use mv_colsum).  */
int mv_gemm(int dtype, int ta, int tb, int M, int N, int K,
            const void* A, int lda, const void* B, int ldb,
            void* C, int ldc, int c_dtype,
            const float* bias, int epi,
            const void* R, int ldr, int r_dtype,
            void* C2, int ldc2,
            void* C3, int ldc3, int c3_dtype,
            int splitk, float* ws, size_t ws_bytes, int accumulate,
            float p_drop, unsigned long long drop_key, const float* alpha_dev, float* colsum_part, void* stream);

/* Split-K workspace sizes (SURVEY 8b's mv_workspace_bytes): what a host needs to allocate without reading the dispatch code.
 *   mv_gemm_workspace_bytes  bytes of `ws` with which mv_gemm(splitk = 0) takes the split-K choice it prefers for this product
 *                            (slabs x M x N x 4; 0 = this call never splits: f32 data, a wide / large product, K < 2048).  A smaller
 *                            workspace is legal: mv_gemm then takes as many slabs as fit.
 *   mv_workspace_bytes       the largest such workspace any mv_gemm call of a pretraining step asks for at this geometry (weight
 *                            gradients of the four encoder projections, of the MLM transform and of the image projection over
 *                            `max_rows` / `max_label_rows` / `max_regions` rows; the tied decoder's input gradient over the
 *                            vocabulary).  One workspace PER STREAM that issues split-K products concurrently.  Pure functions. */
size_t mv_gemm_workspace_bytes(int dtype, int ta, int tb, int M, int N, int K);
size_t mv_workspace_bytes(int hidden, int intermediate, int vocab, int img_hidden, int max_rows, int max_label_rows, int max_regions);

/* ---- attention masks ----------------------------------------------------------------------
 * Replaces CXRBertEncoder.get_extended_attn_mask (cxrbert_origin.py:75-85): instead of an
 * fp16 additive [B,1,L,L] tensor the int64 0/1 mask the Dataset built
 * (data/dataset_origin.py:138-176) is packed once per batch into
 *   bits     uint32 [B, L, W]   W = ceil(L/32); bit (j&31) of word j>>5 = mask[b,i,j]
 *   tileinfo uint8  [B, TQ, TK] TQ = TK = ceil(L/64); per 64x64 tile: 0 = every entry masked and
 *                               every query row of the tile sees at least one key somewhere
 *                               (safe to skip), 1 = every in-range entry visible, 2 = mixed
 * mask_ndim = 3: mask is [B,L,L];  mask_ndim = 2: mask is [B,L] (the `attn_1d` / retrieval form,
 * cxrbert_origin.py:76-77), broadcast over query rows.  Masked entries contribute the
 * reference's additive -10000.0 (not -inf) inside the kernels.                                 */
int mv_mask_pack(const int64_t* mask, int mask_ndim, int B, int L,
                 uint32_t* bits, uint8_t* tileinfo, void* stream);

/* Same outputs built on the device from per-sample descriptors instead of a materialised [B,L,L] int64 matrix
 * (the reference ships 2 MB / sample at L = 512 from its DataLoader workers, dataset_origin.py:138-176):
 * desc int32 [B,3] = {family, n2, vl}; family 0 full (j < vl), 1 seq2seq, 2 BAR, 3 non-cross, 4 1-D (j < vl);
 * n2 = num_image_embeds + 2; vl = n2 + #text ids incl. [SEP].  Closed forms: SURVEY.md Appendix B.             */
int mv_mask_build(const int32_t* desc, int B, int L, uint32_t* bits, uint8_t* tileinfo, void* stream);

/* ---- on-device mini-batch assembly (MLM corruption, labels, descriptors) ---------------------------
 * Replaces the per-sample python of data/dataset_origin.py:102-135 and CXRDataset.random_word (:183-209).
 * mv_mlm_draws: the two random sources of random_word for every token, from a counter-based hash of `key`:
 *   u   f32   [B,S]  stand-in for random.random(), on a 2^-24 grid in [0,1)
 *   rnd int32 [B,S]  stand-in for random.randrange(vocab)
 * mv_mlm_corrupt consumes draws (these or the caller's own) and raw, un-corrupted ids:
 *   ids     int64 [B,S]   token ids; row b is valid for t < lengths[b] (1 <= lengths[b] <= S; out-of-range is clamped)
 *   family  int32 [B]     attention-mask family per sample for `desc` (see mv_mask_build); NULL = 0 (full)
 * outputs, in the reference's batch protocol (dataset_origin.py:181):
 *   input_txt  int64 [B,T]  T = S+1: corrupted ids, [SEP]=102 at t = lengths[b], [PAD]=0 after
 *   segment    int64 [B,T]  all 1
 *   txt_labels int64 [B,L]  L = S+N+3: -100 except the selected text positions (offset N+2), which carry the original id
 *   n_ids      int32 [B]    lengths[b] + 1
 *   desc       int32 [B,3]  {family, N+2, N+2+n_ids} for mv_mask_build / the attention kernels (nullable)
 *   counts     int32 [B]    labels per sample (scratch the index pass reads)
 *   label_rows / label_ids int32 [>= B*S], n_labels int32 [1] (device): row-major compact index of the labelled
 *                           positions (row = b*L + i) and their ids -- all three NULL to skip.
 * Selection rule per valid token, evaluated in double like the python: u < 0.15 selects; then u/0.15 < 0.8 -> [MASK]=103,
 * < 0.9 -> rnd, else unchanged; a sample with no selection gets token 0 masked and labelled (":204-207").        */
int mv_mlm_draws(unsigned long long key, int B, int S, int vocab, float* u, int32_t* rnd, void* stream);
int mv_mlm_corrupt(const int64_t* ids, const int32_t* lengths, const float* u, const int32_t* rnd, const int32_t* family,
                   int B, int N, int S, int64_t* input_txt, int64_t* segment, int64_t* txt_labels, int32_t* n_ids,
                   int32_t* desc, int32_t* counts, int32_t* label_rows, int32_t* label_ids, int32_t* n_labels, void* stream);

/* HOST function (no device work, no stream): every entry of a materialised reference mask (int64 [B,L,L], or [B,L] for the 1-D
 * family; data/dataset_origin.py:138-176) against the closed form of its descriptor desc[b] = {family, n2, vl} -- the predicates of
 * mv_mask_build, a 32-column word at a time, `threads` host threads split by sample.  *first_mismatch = -1 when every entry agrees,
 * else the linear index (b*L + i)*L + j (b*L + j for 2-D masks) of the first entry that differs.  CXRBERT_Trainer runs it one batch
 * ahead of the step on a worker thread: the descriptors it derives from probe entries are proven bit for bit on EVERY batch without
 * moving the 134 MB matrix over PCIe.  Both pointers are HOST pointers. */
int mv_mask_verify_host(const int64_t* mask, int mask_ndim, const int32_t* desc, int B, int L, int threads, long long* first_mismatch);

/* ---- gradient exchange: RCCL over xGMI (one process per GPU) -----------------------------------------------
 * Replaces nn.DataParallel of models/train_origin.py:53-55 for a host that is not torch (medvill_amd itself reaches the SAME
 * library through torch.distributed's nccl backend, dist.py: a torch process keeps one communicator).  RCCL is loaded at run time
 * (dlopen): MV_E_NO_RCCL when it is absent; a failed RCCL call returns MV_E_COMM_BASE + ncclResult_t.
 *   mv_comm_unique_id        rank 0 makes the 128-byte id, the caller hands it to every rank (any transport)
 *   mv_comm_init             communicator of `world` ranks on the calling thread's current device
 *   mv_comm_allreduce_async  in-place SUM all-reduce of `count` elements (MV_F32 / MV_F16 / MV_BF16), enqueued on `stream`:
 *                            call it per gradient bucket as soon as the backward has finished the bucket, on a side stream
 *   mv_comm_wait             `stream` (the compute stream, before the optimizer) waits -- on the device -- for every collective
 *                            issued so far through this communicator; the host never blocks
 *   mv_comm_destroy                                                                                                    */
int mv_comm_unique_id(void* id128);
int mv_comm_init(void** comm_out, int rank, int world, const void* id128);
int mv_comm_allreduce_async(void* comm, void* buf, size_t count, int dtype, void* stream);
int mv_comm_wait(void* comm, void* stream);
int mv_comm_destroy(void* comm);

/* ---- packed rows (padding removal) -------------------------------------------------------------------
 * In the full, seq2seq and 1-D mask families no valid query can see a position after the sample's text [SEP]
 * (SURVEY.md Appendix B), and those positions carry no label: their rows contribute exactly nothing to the loss,
 * to any gradient or to any valid position's hidden state.  The encoder may therefore run on the valid rows only.
 * mv_pack_plan turns the mask descriptors into the row plan:
 *   cu     int32 [B+1]   exclusive prefix sums of vl[b] = desc[b][2] (clamped to L); cu[B] = number of packed rows
 *   rowmap int32 [>= cu[B]] (capacity B*L)  logical flat position b*L + p of every packed row
 *   inv    int32 [B*L]   packed row of logical position b*L + p, or -1 when it was dropped
 * Consumers: mv_embed_fwd/bwd take (rowmap, n_rows) and read / write packed rows; mv_attn_fwd/bwd take (cu, total_rows)
 * and address qkv / ctx / dctx / dqkv by packed row while mask words, lse, delta and the dropout counter keep their
 * logical [B, L] indexing (so a packed run draws the same attention-dropout masks as the padded one).  All four accept
 * NULL for the dense [B*L] layout.  Packed attention exists for the bf16 MFMA kernels only.                      */
int mv_pack_plan(const int32_t* desc, int B, int L, int32_t* cu, int32_t* rowmap, int32_t* inv, void* stream);
/* Row order of the LAST encoder layer (round 3): attention is equivariant under a reordering of a sample's rows, and in the full / 1-D
 * mask families the mask of a packed sample does not depend on the order either (every row sees every row).  Only the rows the heads
 * consume (`sel`, packed row indices: labelled rows and each sample's first row) are needed as QUERIES of the last layer, so that layer
 * runs on rows reordered with those first and its attention kernels stop after them (qlim of mv_attn_fwd / mv_attn_bwd):
 *   perm    int32 [cu[B]]  packed row (old order) held by row i of the new order
 *   newpos  int32 [cu[B]]  its inverse
 *   qlim    int32 [B]      number of consumed rows of sample b (they are rows cu[b] .. cu[b]+qlim[b]-1 of the new order, ascending old order)
 *   sel_new int32 [n_sel]  newpos[sel[i]]
 * L: the logical sequence length (an upper bound of every sample's row count, <= 2048).  */
int mv_tail_perm(const int32_t* cu, int B, int L, const int32_t* sel, int n_sel, int32_t* perm, int32_t* newpos, int32_t* qlim,
                 int32_t* sel_new, void* stream);

/* ---- fused-mask multi-head attention --------------------------------------------------------
 * Replaces HF BertSelfAttention's scores/softmax/context (spec:
 * Downstream_task/report_generation_and_vqa/sc/pytorch_pretrained_bert/model.py:301-320):
 *   ctx[b,i,h,:] = sum_j softmax_j( q.k/sqrt(dh) + (1-mask)*-10000 ) v
 * qkv is the fused projection output [B*L, 3H] (q | k | v, heads contiguous inside each).
 * lse [B, A, L] (f32) = log-sum-exp of each score row, kept for the backward.
 * dh must be 64 (bf16 MFMA path) or <= 128 (f32 path).
 * p_drop > 0: attention-probability dropout (HF BertSelfAttention.dropout): ctx = dropout(softmax).v; the mask is the
 * tensor of keep-bits `dropbits` written by mv_attn_dropmask (required then), survivors scaled by 1 / (1 - P(drop)) with the
 * P(drop) of that generator: the forward and both backward kernels spend one select per score element (they are bound by vector-instruction
 * issue, not by MFMA, so nothing is hashed inside them).
 * dtype MV_F16: qkv and ctx in the f16 encoding; ctx_bf16 (nullable, MV_F16 only) receives the bf16 copy of ctx that
 * the backward pairs with bf16 gradients.
 * qlim (nullable, int32 [B]): only the first qlim[b] rows of sample b are queries (every row is a key); the context / lse of the other
 * rows are left untouched.  MFMA kernels only.  */
int mv_attn_fwd(int dtype, const void* qkv, const uint32_t* bits, const uint8_t* tileinfo,
                void* ctx, void* ctx_bf16, float* lse, int B, int L, int A, int dh,
                float p_drop, const uint32_t* dropbits, const int32_t* cu, int total_rows, const int32_t* qlim, void* stream);

/* The attention-probability dropout mask of one layer as keep-bits, laid out as the kernels' select masks:
 * dropbits uint32 [B*A][ceil(L/32)][ceil(L/64)][64] (64-byte aligned), i.e. per (head, 32-query block, 64-key tile) 32 x uint64;
 * word 16*kk + r, bit l <-> query 32*qb + (l & 31), key 64*kt + 32*kk + (r&3) + 8*(r>>2) + 4*(l>>5); bit = 1 keeps the entry.
 * Every entry is an independent Bernoulli draw with P(drop) = round(p_drop * 2^n) / 2^n (n = 16: 0.1 -> 0.100006; the debug library has a knob for 12 / 8), a counter-based
 * hash of (drop_key, word index): the same key reproduces the same mask.  Blocks whose queries or keys all lie beyond the
 * sample's packed length (cu, nullable) are not written and never read.  Depends on the key only: it can run ahead of the
 * forward on another stream.  */
int mv_attn_dropmask(float p_drop, unsigned long long drop_key, int B, int L, int A, const int32_t* cu, uint32_t* dropbits,
                     void* stream);

/* dqkv [B*L,3H] from dctx [B*L,H]; `delta` is a [B,A,L] f32 scratch (rowsum(dctx*ctx)).  dtype (MV_F32, MV_BF16 or
 * MV_F16) is the encoding of qkv, ctx, dctx and dqkv alike.  qlim (nullable): as in mv_attn_fwd; the dctx rows past a sample's
 * limit are ignored (taken as zero) and their dQ rows are written as zeros. */
int mv_attn_bwd(int dtype, const void* qkv, const void* ctx, const void* dctx, const float* lse,
                const uint32_t* bits, const uint8_t* tileinfo,
                void* dqkv, float* delta, int B, int L, int A, int dh,
                float p_drop, const uint32_t* dropbits, const int32_t* cu, int total_rows, const int32_t* qlim, void* stream);

/* ---- LayerNorm ------------------------------------------------------------------------------
 * y = (x-mean)/sqrt(var+eps)*gamma+beta over the last dim (HF LayerNorm eps=1e-12 in the
 * encoder; TF-style BertLayerNorm eps=1e-5 of cxrbert_origin.py:189-202 in the MLM head).
 * x_dtype: dtype of x (the fused-residual GEMM writes f32); y in `dtype`; y_bf16 (nullable, dtype MV_F16 only): bf16
 * copy of y for the backward's weight-gradient product.  mean, rstd: f32 [M], kept for backward.  */
int mv_layernorm_fwd(int dtype, const void* x, int x_dtype, const float* gamma, const float* beta,
                     void* y, void* y_bf16, float* mean, float* rstd, int M, int H, float eps, void* stream);

/* dx (dtype) = LN backward of dy (dtype) w.r.t. x; dgamma/dbeta (f32 [H]) are ACCUMULATED
 * (atomically) -- zero them first; colsum (f32 [H], nullable) accumulates sum_m dx[m,:]
 * (the bias gradient of the projection that produced x).
 * dx_drop (nullable): second output = dx * dropout_mask / (1 - p), the gradient w.r.t. the projection
 * output when x = dropout(projection) + residual (mask index m*H + c); colsum then sums dx_drop.
 * grad_unscale_dev (nullable): *grad_unscale_dev multiplies what is added to dgamma / dbeta / colsum (1 / loss scale). */
int mv_layernorm_bwd(int dtype, const void* dy, const void* x, int x_dtype, const float* mean,
                     const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta,
                     float* colsum, int M, int H,
                     void* dx_drop, float p_drop, unsigned long long drop_key, const float* grad_unscale_dev, void* stream);

/* ---- sequence assembly + embeddings ---------------------------------------------------------
 * Replaces CXRBertEncoder.forward's else-branch assembly (cxrbert_origin.py:114-125),
 * ImageBertEmbeddings.forward (:22-35) and the four HF BertEmbeddings calls: writes
 *   x0[b] = LN( [ E[cls]+Ty[0]+P[0] | imgproj[b,n]+P[img_pos[b,n]]+Ty[0] | E[sep]+Ty[0]+P[0] |
 *               E[txt[b,t]]+Ty[segment[b,t]]+P[t] ] )         L = N + T + 2
 * straight into the concatenated [B,L,H] buffer.  imgproj = feats.Wi^T + bi comes from mv_gemm.
 * Tables E, P, Ty and imgproj are in `dtype`; gamma/beta f32.  `pre` (f32 [B,L,H]) keeps the
 * pre-LayerNorm sums for the backward.  x0_bf16 (nullable, dtype MV_F16 only): bf16 copy of x0.
 * img_pos NULL: the image rows get NO position embedding (args.img_postion false, cxrbert_origin.py:27-31).
 * p_drop: dropout of the text / [CLS] / [SEP] rows (HF hidden_dropout_prob); p_drop_img: of the image rows
 * (ImageBertEmbeddings' nn.Dropout(args.dropout_prob), cxrbert_origin.py:19); same key and element index.  */
int mv_embed_fwd(int dtype, const int64_t* cls_tok, const int64_t* txt, const int64_t* segment,
                 const int64_t* img_pos, const int64_t* sep_tok, const void* imgproj,
                 const void* E, const void* P, const void* Ty, const float* gamma, const float* beta,
                 void* x0, void* x0_bf16, float* pre, float* mean, float* rstd,
                 int B, int N, int T, int H, int V, int maxpos, float eps,
                 float p_drop, float p_drop_img, unsigned long long drop_key, const int32_t* rowmap, int n_rows, void* stream);

/* Backward of the above: LN backward of dx0, scatter-add (f32 atomics) into dE [V,H], dP, dTy,
 * dgamma, dbeta (all ACCUMULATED) and write d(imgproj) [B,N,H] in `dtype`.  Row `pad_token_id`
 * of dE receives no look-up gradient (HF BertEmbeddings builds nn.Embedding(..., padding_idx=
 * pad_token_id = 0)); pass -1 for none.  grad_unscale_dev (nullable): factor on everything added to the f32 gradients. */
int mv_embed_bwd(int dtype, const void* dx0, const float* pre, const float* mean, const float* rstd,
                 const float* gamma, const int64_t* cls_tok, const int64_t* txt, const int64_t* segment,
                 const int64_t* img_pos, const int64_t* sep_tok,
                 float* dE, float* dP, float* dTy, float* dgamma, float* dbeta, void* dimgproj,
                 int B, int N, int T, int H, int V, int maxpos, int pad_token_id,
                 float p_drop, float p_drop_img, unsigned long long drop_key, const int32_t* rowmap, int n_rows,
                 const float* grad_unscale_dev, void* stream);

/* ---- losses + step metrics ------------------------------------------------------------------
 * Replaces nn.CrossEntropyLoss(ignore_index=-100) on mlm.transpose(1,2) and
 * nn.CrossEntropyLoss() on the ITM logits (train_origin.py:62-63,120-126) plus the argmax
 * metrics of train_origin.py:133-146, fused with the loss gradient.
 * logits [R, ld] (l_dtype f32 or bf16); labels int32 [R] (-100 = ignored row);
 * out[0] += sum of nll over labelled rows, out[1] += #labelled rows, out[2] += #rows whose
 * argmax == label (f32 accumulators, zero them first).
 * dlogits (nullable; d_dtype) = (softmax - onehot) * grad_scale for labelled rows, 0 otherwise;
 * columns V..ldd-1 are zero-filled (so dlogits can feed mv_gemm with K = V).
 * grad_scale is read from the device (*grad_scale_dev) when non-null (so 1/n_labelled_global
 * can be produced by an all-reduce without a host sync), else grad_scale_host; loss_scale_dev
 * (nullable) multiplies it by the loss scale of the f16-gradient path (device state [0]).       */
int mv_ce_fwd_bwd(const void* logits, int l_dtype, int ld, const int32_t* labels, int R, int V,
                  float* out, void* dlogits, int d_dtype, int ldd,
                  const float* grad_scale_dev, float grad_scale_host, const float* loss_scale_dev, void* stream);

/* ---- row gather / scatter (labelled-row compaction for the MLM head) ------------------------
 * dst[i,:] = src[rows[i],:]  /  dst[rows[i],:] (+)= src[i,:]; rows int32 [R].  A negative rows[i] means "no such
 * row" (a position dropped by mv_pack_plan): gather writes zeros, scatter writes nothing.          */
int mv_gather_rows(int dtype, const void* src, int lds, const int32_t* rows, int R, int H,
                   void* dst, int ldd, void* stream);
int mv_scatter_rows(int dtype, const void* src, int lds, const int32_t* rows, int R, int H,
                    void* dst, int ldd, int accumulate, void* stream);

/* out[n] (+)= [*grad_unscale_dev] * sum_m x[m,n]  (bias gradients). x [M,N] in dtype, out f32.   */
int mv_colsum(int dtype, const void* x, int ldx, int M, int N, float* out, int accumulate, const float* grad_unscale_dev,
              void* stream);
/* out[n] += [*grad_unscale_dev] * sum_{p < P} part[p*ld + n]: folds the partial column sums written by mv_gemm (colsum_part). */
int mv_colsum_partials(const float* part, int P, int ld, int N, float* out, const float* grad_unscale_dev, void* stream);

/* elementwise c = a + b (dtype), n elements */
int mv_add(int dtype, const void* a, const void* b, void* c, size_t n, void* stream);

/* out = dy * gelu_erf'(z) (mode 0; backward of the MLM transform's activation,
 * cxrbert_origin.py:176-181,216) or out = dy * (1 - z*z) (mode 1; tanh backward of the pooler,
 * z holds the tanh OUTPUT).  n elements, n % 4 == 0. */
int mv_dact(int dtype, int mode, const void* dy, const void* z, void* out, size_t n, void* stream);

/* 2-D cast: dst[r, 0..cols) = src[r, 0..cols), dst[r, cols..ldd) = 0 (pads a [rows, V] gradient
 * to a leading dimension the MFMA GEMM accepts). */
int mv_cast2d(const void* src, int src_dtype, long long lds, void* dst, int dst_dtype, long long ldd,
              int rows, int cols, void* stream);

/* dst[c, r] = src[r, c] for r < rows, c < cols (leading dimensions lds >= cols, ldd >= rows; same dtype both sides).
 * Utility (the training engine keeps one transposed copy per layer: the FFN-down weights, for dz as an NT-form GEMM). */
int mv_transpose(int dtype, const void* src, long long lds, void* dst, long long ldd, int rows, int cols, void* stream);

/* dst(dst_dtype) = src(src_dtype), n elements */
int mv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, size_t n, void* stream);

/* ---- region-feature extractor: ResNet-50 trunk pieces (models/image.py:46-69) ------------------------------
 * The trunk runs in NHWC: an activation is the matrix [B*H*W, C] and every convolution is mv_gemm over it -- directly
 * for 1x1 / stride 1, after mv_im2col otherwise (weights re-laid as [Cout, kh*kw*Cin], (ky, kx, c) order).  BatchNorm
 * (+ residual)(+ ReLU) is mv_bn_act, with batch statistics from mv_col_stats in train() mode or the running ones in
 * eval().  Forward only: the reference never trains the CNN (cxrbert_origin.py:66-70 unfreezes nothing).
 *   mv_nchw_to_nhwc : f32 [B,C,H,W] -> dst_dtype [B,H,W,Cp], channels C..Cp-1 zero
 *   mv_im2col       : src [B,H,W,C] -> dst [B*Ho*Wo, ldk], column (ky*kw + kx)*C + c = src[b, oy*stride-pad+ky, ox*stride-pad+kx, c]
 *                     (0 outside the image; columns kh*kw*C..ldk-1 zero), Ho = (H + 2*pad - kh)/stride + 1
 *   mv_col_stats    : stats f32 [2, C] = per-column sum and sum of squares over the rows of x [rows, C] (C, ldx % 4 == 0)
 *   mv_bn_finalize  : mean / rstd of the batch from those sums (biased variance), plus nn.BatchNorm2d's momentum update of
 *                     running_mean / running_var (unbiased variance) when they are given (both or neither)
 *   mv_bn_act       : y = (x - mean) * rstd * gamma + beta (+ residual) (max 0 when relu != 0); C % 4 == 0; x may be f32
 *                     while y / residual are bf16
 *   mv_maxpool3x3s2 : 3x3 / stride 2 / pad 1 max pooling, NHWC                                                      */
int mv_nchw_to_nhwc(const float* src, void* dst, int dst_dtype, int B, int C, int H, int W, int Cp, void* stream);
int mv_im2col(int dtype, const void* src, int B, int H, int W, int C, int kh, int kw, int stride, int pad, void* dst, int ldk,
              void* stream);
/* y[(b,oy,ox), o] = sum_{ky,kx,c} x[b, oy*stride-pad+ky, ox*stride-pad+kx, c] * w[o, (ky*kw + kx)*C + c]  (NHWC x, zero padding):
 * the convolution as an implicit GEMM on the MFMA kernel -- the activation operand is gathered tap by tap while it is
 * staged, no patch matrix is materialised.  bf16 x / w, y in y_dtype; C a power of two >= 8, O % 4 == 0.
 * epi: MV_EPI_NONE, MV_EPI_BIAS, MV_EPI_BIAS_RELU or MV_EPI_BIAS_RES_RELU (R [rows, O], leading dimension O).          */
int mv_conv2d(int dtype, const void* x, const void* w, void* y, int y_dtype, int B, int H, int W, int C, int O, int kh, int kw,
              int stride, int pad, const float* bias, int epi, const void* R, int r_dtype, void* stream);
int mv_col_stats(int dtype, const void* x, int ldx, int rows, int C, float* stats, void* stream);
int mv_bn_finalize(const float* stats, int C, long long rows, float eps, float momentum, float* mean, float* rstd,
                   float* running_mean, float* running_var, void* stream);
int mv_bn_act(int dtype, const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
              const void* residual, void* y, long long rows, int C, int relu, void* stream);
int mv_maxpool3x3s2(int dtype, const void* x, void* y, int B, int H, int W, int C, void* stream);

/* ---- dropout mask (inspection / tests) -----------------------------------------------------------
 * The kernels above never store dropout masks: they regenerate them from a counter-based hash of
 * (drop_key, linear element index).  keep[i] = 1 if element i survives; an element is dropped with
 * probability round(p_drop*65536)/65536 and survivors are scaled by *scale_out (HOST pointer, nullable).
 * These are the masks of the hidden-state sites (embeddings, both projections); the attention-probability site's mask is
 * the keep-bit tensor of mv_attn_dropmask. */
int mv_dropout_mask(float p_drop, unsigned long long drop_key, size_t n, uint8_t* keep, float* scale_out,
                    void* stream);

/* ---- optimizer ------------------------------------------------------------------------------
 * HF transformers.optimization.AdamW (<= 4.x) as called at train_origin.py:60,131:
 *   m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps);
 *   p -= lr*wd*p            (eps added BEFORE the bias correction, decoupled decay)
 * over one flat f32 buffer of n elements (all parameters live in one flat buffer);
 * g is multiplied by grad_scale first; shadow_bf16 / shadow_f16 (nullable) receive the 16-bit copies of
 * the updated parameters that the MFMA kernels read.
 * scaler_state (nullable; the f32 [8] device state of mv_scaler_update): when given, the step is SKIPPED entirely if
 * state[3] != 0 (the backward overflowed) and t of the bias correction is state[4] instead of `step`.   */
int mv_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, void* shadow_f16, size_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  int correct_bias, float grad_scale, const float* scaler_state, void* stream);

/* ---- dynamic loss scale of the f16-gradient path -------------------------------------------
 * No counterpart in the reference (its gradients never leave f32; train_origin.py:129-131): f16 gradient operands need
 * the loss scaled into f16's range.  state f32 [8] on the device: [0] scale S (read by mv_ce_fwd_bwd as loss_scale_dev),
 * [1] 1/S (read as grad_unscale_dev / alpha_dev), [2] clean steps since S changed, [3] skip flag, [4] optimizer steps
 * applied, [5] steps skipped, [6] non-finite count.  Initialise to {S0, 1/S0, 0, 0, 0, 0, 0, 0}.
 *   mv_count_nonfinite: counter[0] += #elements of x (f32 [n], 16-byte aligned) that are inf or nan
 *   mv_scaler_update  : count > 0 -> skip = 1, S = max(S*backoff, min_scale); else skip = 0, t += 1 and after
 *                       growth_interval clean steps S = min(S*growth, max_scale); clears the count.        */
int mv_count_nonfinite(const float* x, size_t n, float* counter, void* stream);
int mv_scaler_update(float* state, int growth_interval, float growth, float backoff, float max_scale, float min_scale,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDVILL_H_ */
