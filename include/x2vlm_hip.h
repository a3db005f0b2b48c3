/* x2vlm_hip.h — C ABI of libx2vlm_hip.so: the MI355X (gfx950) kernels of the X^2-VLM pre-training step.
 *
 * The reference (zengyan-97/X2-VLM) is 100 % Python and has no FFI layer: every entry point below replaces
 * an implicit ATen / cuBLAS / cuDNN call made by the cited reference line (forward AND its autograd
 * backward).  A maintainer binds them with ctypes (INTEGRATION.md shows the stub); x2-vlm_amd/_lib.py is
 * that binding, x2-vlm_amd/kernels.py the tensor-level wrappers.
 *
 * Conventions
 *   - extern "C", POD arguments only: device pointers as void* / typed pointers, sizes as int / long,
 *     the HIP stream as void* (hipStream_t).  No torch types.
 *   - returns 0 on success, < 0 on error (-1 bad argument, -2 launch failure); x2_last_error() gives the
 *     message (thread-local).  Nothing is launched when an argument check fails.
 *   - the caller owns all memory; kernels never allocate, never synchronise the host; every call is
 *     asynchronous and ordered on the stream passed in; distinct streams may be used concurrently.
 *   - bf16 tensors are raw uint16 bit patterns; "ld*" are leading dimensions in ELEMENTS.
 *   - gradients documented as "+=" are added onto caller-initialised memory; since ABI v12 no entry point the training step uses adds
 *     with atomics (every sum has a fixed order: two runs give the same bits); the two that still can - x2_gemm_nt(colsum != NULL) and
 *     x2_gemm_tn_grouped on its 128x128 kernel with split > 1 and no workspace - say so below.
 */
#ifndef X2VLM_HIP_H
#define X2VLM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* x2_last_error(void);
int x2_abi_version(void);          /* == 14 */
int x2_device_cus(void);           /* compute units of the current HIP device, 0 if none */
int x2_tune(int key, int value);   /* kernel-variant knobs for probes/ and tests (0 = automatic); keys listed in csrc/gemm.hip.
                                    * key 12 = compute units every tile plan leaves to RCCL's channel kernels (world > 1);
                                    * key 14 = 1: attention backward as two kernels even where a one-pass kernel applies (3: only the long form 3 off);
                                    * key 15 = NT ping-pong kernel (32x32x16 MFMAs): 0 automatic, 1 never, 3 .. 6 always, at 32 x value rows;
                                    * key 2 (work-skipping ablation bits) is refused unless the library was built with -DX2_PROBE */
int x2_tune_get(int key);          /* current value of a knob, -1 for an unknown key (bench.py reports the non-default ones) */

/* ---- dense contractions (csrc/gemm.hip) -------------------------------------------------------------
 * F.linear of beit2.py:131 (fused qkv), :160 (proj), :62/:66 (MLP); xbert.py:338-350 (Q/K/V, cross K/V from
 * image tokens), :428 (attention output), :497 (intermediate), :512 (output), :798 (MLM transform),
 * :822 (tied decoder); nn.Conv2d patch embedding beit2.py:225,231 (as a GEMM over patch rows).
 *
 * C[M,N] = epilogue(A[M,K] . B[N,K]^T), bf16 operands, fp32 MFMA accumulation.  K % 64 == 0, N % 8 == 0.
 *   v = acc + bias[n]
 *   act == 1: aux[m,n] = bf16(v); v = gelu(v)        (erf form)   -> fc1 / intermediate forward
 *   act == 2: v *= gelu'(aux[m,n])                                 -> dgrad through the GELU
 *   act == 0 and aux != NULL: aux[m,n] = bf16(v)                   -> value before layer scale (for dgamma)
 *   v *= dropout(v)   (drop_thr16 != 0: hidden dropout, xbert.py:429, 513; counter-based, see below)
 *   v *= gamma[n]; v *= rowscale[m]; v += resid[m,n]               -> x + drop_path(gamma_1 * proj(...)) (beit2.py:206-207),
 *                                                                     dropout(dense(x)) + residual (xbert.py:430,514)
 * Dropout everywhere in this ABI: element e of a site is dropped iff u16(hash(e >> 1 ^ seed), e & 1) < thr16
 * (thr16 = round(p * 65536), 0 = off); survivors are multiplied by `scale`.  The backward regenerates the mask
 * from the same (thr16, seed, scale); kernels.dropout_keep() is the host mirror.
 * drop_epoch (every dropout-capable entry point): NULL, or a device word holding a step counter that is mixed into the
 * seed (seed' = hash(seed + 0x9E3779B1 * *drop_epoch)): a hipGraph-captured step, whose kernel arguments are frozen,
 * increments the word once per replay and so draws new masks on every step.
 *   C = out_f32 ? float : bf16.  Forward linears pass B = W (N x K); input gradients pass B = W^T. */
int x2_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
               const float* bias, const float* gamma, const float* resid, int ldr, void* aux, int ldaux,
               int act, int out_f32, unsigned drop_thr16, unsigned drop_seed, float drop_scale, const unsigned* drop_epoch,
               const float* rowscale, float* colsum, void* stream);      /* colsum[n] += sum_m C[m,n] (fused bias gradient), NULL = off */

/* C = (A . B^T) x GELU'(aux) -> bf16 plus the column sums of C as partial rows colparts[r][N], r < *nrows_out (2 per row tile
 * of the launch; colparts must hold 2 * ceil(M / 64) rows); the caller adds them with x2_reduce_partials(_multi)(colparts,
 * *nrows_out, 1, N, out).  Input gradient through the MLP's GELU + bias gradient of its first linear in one kernel
 * (autograd of beit2.py:62-66, xbert.py:497: GeluBackward + the bias reduction of AddmmBackward). */
int x2_gemm_nt_dgelu_colparts(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                              const void* aux, int ldaux, float* colparts, int* nrows_out, void* stream);
/* The same product with a split contraction, for few output tiles and a long K: C[M,N] (fp32, no epilogue) =
 * A[M,K] . B[N,K]^T.  The input gradient of the tied MLM decoder (reference: autograd of the F.linear behind
 * xbert.py:822, dt[R,768] = dlogits[R,30528] . E) is 36-72 output tiles of 477 contraction steps - a serial chain on a
 * seventh of the chip.  Slice s of the contraction writes its partial product to ws[s*M*N ...]; a second kernel adds the
 * slices in a fixed order (deterministic, no atomics).  slices: 0 = chosen to put two workgroups on every CU with at
 * least 8 contraction steps each; clamped to ws_floats / (M*N); ws = NULL or one slice: a single launch straight into C. */
int x2_gemm_nt_splitk(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int slices, float* ws, long ws_floats, void* stream);

/* Weight gradients of one layer in one launch: for each problem  dW[N,K] (+)= dY[Mc,N]^T . X[Mc,K]  (fp32 out).
 * problems: count (<= 8) rows of 11 int64 {dY, X, dW, Mc, N, K, ld_dY, ld_X, ld_dW, n_ld, k_ld}; n_ld / k_ld are
 * the readable widths of dY / X rows (>= N / K, multiples of 8).  accumulate: dW += instead of dW =.
 * Contraction lengths that are all multiples of 64 run on 256x256 tiles (one workgroup per CU); split = slices of the
 * contraction per tile (0 = chosen to fill the CUs), whose partial tiles pass through ws (split * tiles256 * 65536
 * floats; ws = NULL: never split) and are added in a fixed order; an explicit split > 1 with a workspace takes this path whatever the tile
 * count.  Otherwise 128x128 tiles, where split > 1 (without a workspace) adds with fp32 atomics and requires accumulate. */
int x2_gemm_tn_grouped(const int64_t* problems, int count, int accumulate, int split, float* ws, long ws_floats,
                       void* stream);

/* ---- fused attention, head dim 64 (csrc/attention.hip) ----------------------------------------------
 * beit2.py:135-159 (q*scale, QK^T, + relative_position_bias, softmax, PV); xbert.py:364-409 (QK^T/sqrt(d),
 * + additive mask, softmax, PV) for self-attention and for cross-attention to image tokens (xbert.py:345-348).
 * Scores/probabilities never reach HBM.  kv_idx lets several query batches share one K/V batch.
 * Field order is the ABI (mirrored by ctypes.Structure in x2-vlm_amd/_lib.py). */
typedef struct X2AttnArgs {
  const void *Q, *K, *V, *O, *dO;            /* bf16; O/dO: backward inputs                               */
  void *Out, *dQ, *dK, *dV, *dS;             /* bf16 outputs; dS [B][H][Lq][ds_ld] optional (bias grad)   */
  float *LSE, *Delta;                        /* [B][H][Lq] fp32: log2-domain logsumexp; rowsum(dO*O)      */
  long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;            /* batch / row strides, elements        */
  long dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, do_bs, do_rs;
  int B, Bkv, H, Lq, Lk;
  float scale;
  const float* bias;  int bias_ld;           /* [H][Lq][bias_ld]  additive, bias_ld % 64 == 0             */
  const float* biasT; int biasT_ld;          /* [H][Lk][biasT_ld] transposed copy (backward)              */
  const float* mask;  int mask_ld;           /* [B][mask_ld] additive per key, mask_ld % 64 == 0          */
  const int* kv_idx;                         /* [B] query batch -> K/V batch, NULL = identity             */
  const int* seq_off; const int* seq_ids;    /* CSR inverse of kv_idx (backward dK/dV), NULL = identity   */
  int ds_ld;
  unsigned drop_thr16, drop_seed; float drop_scale;   /* dropout on the probabilities (xbert.py:399);
                                                          element = ((b*H + h)*Lq + q) * round_up(Lk,64) + key */
  int dbg;                                           /* 0; ablation switches for probes/bench_attn.py */
  int head_dim;                                      /* hidden / heads of the caller: must be 64, checked */
  const unsigned* drop_epoch;                        /* device step counter mixed into drop_seed, or NULL */
  int grid_nx, grid_ny, grid_nz, grid_map;           /* written by the library (XCD-aware workgroup order of the bias kernels): pass 0 */
  int phase;                                         /* x2_attn_bwd: 0 = dQ (+dS, Delta) then dK/dV; 1 = the dQ half only; 2 = the dK/dV half only
                                                        (reads the Delta a phase-1 call wrote): the K/V-side gradients on another stream */
  float* ws; long ws_floats;                         /* x2_attn_bwd (ABI v13): fp32 scratch of B * H * ceil(Lq / 128) * 8192 floats for the long one-pass
                                                        backward (form 3 below), or NULL / smaller: such a geometry runs the two kernels */
  float* colsum_ws;                                  /* x2_attn_bwd (ABI v14), honoured by forms 1 and 3 only (ask x2_attn_bwd_one_pass), or NULL: [B][2][H*64] fp32,
                                                        per sequence the column sums of the stored dQ (k = 0) and dV (k = 1) rows - summed over B
                                                        (x2_reduce_partials, nblk = B, nk = 2) they are the q / v bias gradient of the fused qkv
                                                        projection (beit2.py:129-131: qkv_bias = cat(q_bias, 0, v_bias)) without a pass over [M, 3D] */
} X2AttnArgs;
int x2_attn_fwd(const X2AttnArgs* args, void* stream);
int x2_attn_bwd(const X2AttnArgs* args, void* stream);   /* dQ (+dS, Delta) then dK/dV; no atomics */
/* Which backward a phase-0 x2_attn_bwd call with these arguments runs: 0 = two kernels (dQ, then dK/dV), 1 = one pass, one workgroup per
 * (sequence, head) (64 < Lq, Lk <= 208, no K/V sharing, no probability dropout: the BEiT-2 blocks, beit2.py:135-159), 2 = one pass, one
 * workgroup per (shared K/V batch, head) (Lq <= 128, Lk <= 208, no bias: the cross-attention of xbert.py:322-415), 3 = one pass, one workgroup
 * per (sequence, head) that walks 256-key parts x 128-query chunks (208 < Lq <= 640, 208 < Lk <= 768, no K/V sharing, no probability
 * dropout, `ws` given, biasT_ld >= 128 ceil(Lq / 128): the BEiT-2 blocks of X2VLM-large at 384 px, N = 577).  A caller that
 * would put the dK/dV half (phase 2) on another stream asks first: in one pass there is no such half. */
int x2_attn_bwd_one_pass(const X2AttnArgs* args);

/* ---- row-wise kernels (csrc/rowwise.hip) ------------------------------------------------------------
 * nn.LayerNorm: beit2.py:175,181,411 (eps 1e-6); xbert.py:214,422,506,796 (1e-12); xvlm.py:166 (1e-5).
 * period > 0: rows are the non-cls tokens of a (B, period+1, D) tensor (row r -> r + r/period + 1), used for
 * fc_norm over patches (beit2.py:409-411). */
int x2_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, float* mean,
                     float* rstd, int rows, int D, float eps, int period, unsigned drop_thr16, unsigned drop_seed,
                     float drop_scale, const unsigned* drop_epoch, void* stream);   /* drop: dropout on the LN output (xbert.py:215) */
/* g = LN'(mask_in(dy)); dx = dres + g (fp32); dx_bf16 = mask_out(g); dw += , db += ; dcol += column sums of mask_out(g)
 * (= gradient and bias gradient of the linear whose dropped output was added to the residual before this LN).
 * post != 0 (dy bf16, period 0, no output mask): the by-products describe the FINAL output instead - dx_bf16 =
 * post_rowscale[row] * dx (post_rowscale NULL = 1), dcol += its column sums: what the layer scale below this LayerNorm needs
 * of its incoming gradient (x2_layerscale_finish) */
int x2_layernorm_bwd(const void* dy /* fp32, or bf16 when dy_is_bf16 */, int dy_is_bf16, const float* x, const float* mean, const float* rstd, const float* w,
                     const float* dres, float* dx, void* dx_bf16, float* dw, float* db, float* dcol, int rows, int D,
                     int period, unsigned in_thr16, unsigned in_seed, float in_scale, unsigned out_thr16,
                     unsigned out_seed, float out_scale, const unsigned* drop_epoch, float* ws /* [ceil(rows/16)][3][D] */,
                     int defer, int post, const float* post_rowscale, void* stream);
/* Column reductions are two-stage (per-workgroup partial rows in the caller's workspace `ws`, then a deterministic
 * add): fp32 atomics measured ~43 G adds/s on MI355X, slower than the HBM traffic of these kernels. */
int x2_colsum_bf16(const void* y, float* out, int M, int N, int ld, float* ws /* [ceil(M/64)][N] */, int defer, void* stream);
/* defer != 0: only stage 1 (partials into ws); the caller finishes with x2_reduce_partials, possibly on another stream:
 * out_k[c] += sum_blk ws[blk][k][c], k < nk <= 3 */
int x2_reduce_partials(const float* part, int nblk, int nk, int width, float* o0, float* o1, float* o2, void* stream);
/* `count` such reductions in one launch (all parameter-gradient sums of one layer's backward), nk <= 4 here;
 * desc: count rows of 8 int64 {part, nblk, nk, width, o0, o1, o2, o3} (a null output skips that partial row) */
int x2_reduce_partials_multi(const int64_t* desc, int count, void* stream);
/* backward of x + r[m] * gamma * u, u = A . W^T + b (beit2.py:206-207; r = DropPath factor or 1) without a pass over u: with
 * dX' = r * dX the input gradient is dX' . (diag(gamma) W) (x2_cast_transpose_multi folds gamma into the W^T copy), the weight-
 * gradient GEMM computes G = dX'^T . A, and x2_layerscale_finish turns G and cs = colsum(dX') into
 *   dgamma += rowdot(G, W) + b * cs,  dbias += gamma * cs,  G <- diag(gamma) G  (= dW).
 * desc: count rows of 9 int64 {G [N][K] fp32, W [N][K] fp32 master weight, b or 0, gamma, cs, dgamma, dbias or 0, N, K}.
 * dX' (bf16) and cs come from x2_layernorm_bwd (post = 1) or from x2_rowscale_cast_colsum. */
int x2_layerscale_finish(const int64_t* desc, int count, void* stream);
int x2_rowscale_cast_colsum(const float* dx, const float* rowscale, void* dx_bf16, float* colsum, int M, int D,
                            float* ws /* [ceil(M/32)][D] */, int defer, void* stream);
int x2_cast_bf16(const float* src, void* dst, long n, void* stream);
int x2_cast_transpose_bf16(const float* src, void* dst, void* dstT, int R, int C, int ldt, void* stream);
/* bf16 (W, W^T) copies of many fp32 weights in one launch (what apex O1's per-call weight casts amount to, done once per
 * optimizer step): desc = count rows of 8 int64 {src, dst, dstT, R, C, ldt, roff, tscale}: src [R][C] fp32 -> rows roff.. of
 * dst [*][C] and columns roff.. of dstT [C][ldt]; R, C, ldt, roff multiples of 4; tscale (fp32 [R]) or 0: the transposed
 * copy holds tscale[r] * src[r][c] (a layer scale folded into the weight of the input-gradient GEMM) */
int x2_cast_transpose_multi(const int64_t* desc, int count, void* stream);
/* many small fp32 vectors packed in one launch (stacked q/k/v biases): desc = count rows {src or 0 = zeros, dst, n} */
int x2_copy_f32_multi(const int64_t* desc, int count, void* stream);
/* PatchEmbed input rows (beit2.py:225-232): image (B,3,R,R) -> bf16 [B*(R/ps)^2][3*ps*ps] */
int x2_patchify(const float* image, void* cols, int B, int R, int ps, void* stream);
/* torch.cat((cls_tokens, x), 1) (beit2.py:385-387) and its backward */
int x2_assemble_tokens(const float* patch, const float* cls, float* x, int B, int P, int D, void* stream);
int x2_assemble_tokens_bwd(const float* dx, void* dpatch_bf16, float* dcls, int B, int P, int D, void* stream);
/* token 0 <- (weighted) mean of patch tokens: avgpool beit2.py:413-416, region pooling beit2.py:430-436 */
int x2_pool_tokens(float* x, const float* w, int B, int P, int D, int bwd, void* stream);
/* relative_position_bias_table[relative_position_index] x scale -> [H][N][ld] (+ transposed), beit2.py:138-144.  scale = 1, or
 * log2(e) for a bias in the unit the attention kernels' softmax works in: set bit 4 (16) of X2AttnArgs.dbg with such a bias
 * (no mask) and a score costs one fma instead of four VALU operations */
int x2_relpos_bias(const float* table, const long* index, const long* indexT /* index^T or NULL */, float* bias, float* biasT, int N, int H, int ld, int ldT,
                   float scale, void* stream);
/* dtable[index[i][j]][h] += sum_b dS[b][h][i][j] (dS bf16 [B][H][N][ld]); the index arrives as its CSR inverse
 * (inv_off [T+1], inv_pos = i*ld + j); ws: slices*H*N*ld floats (batch-slice sums, then a gather: no atomics).
 * Reads columns j < 8 ceil(N / 8) of dS only: the one-pass attention backward leaves columns >= 16 ceil(N / 16) of its dS stream unwritten. */
int x2_relpos_bias_bwd(const void* dS, const int* inv_off, const int* inv_pos, float* dtable, int B, int N, int H, int ld,
                       int T, float* ws, int slices, void* stream);

/* ---- embeddings, heads, losses (csrc/heads.hip) -----------------------------------------------------
 * BertEmbeddings xbert.py:205-213 (word + position + token-type 0); backward scatter-adds. */
int x2_embed_fwd(const long* ids, const float* word, const float* pos, const float* type0, float* out, int R, int L,
                 int D, void* stream);
/* dword[ids[r]] += g[r], dpos[r % L] += g[r], dtype0 += sum_r g[r]: every sum in a fixed order (no atomics: rows with the same token id are added in
 * ascending r by one workgroup; position totals pass through `scratch`, min(L, R) * D floats).  D <= 4096. */
int x2_embed_bwd(const long* ids, const float* g, float* dword, float* dpos, float* dtype0, int R, int L, int D,
                 float* scratch, void* stream);
/* torch.gather of sequences / masked positions (xbert.py:1588-1589, xvlm.py:866-884) and its backward */
int x2_gather_rows(const float* src, const int* idx, float* dst, void* dst_bf16, int R, long len, void* stream);
/* dst[d][:] = sum of src[r][:] over idx[r] == d, all D rows of dst written (zeros where nothing points); no atomics; R <= 8192 */
int x2_scatter_rows(const float* src, const int* idx, float* dst, int R, int D, long len, void* stream);
/* small fp32 linear with arbitrary strides: vision_proj / text_proj (xvlm.py:785-792), similarity matrices
 * (xvlm.py:807, 831-832), last layers of itm_head / bbox_head (xvlm.py:163-169) and their backward */
int x2_linear_f32(const float* A, const float* B, float* C, const float* bias, const float* alpha_ptr, float alpha,
                  int M, int N, int K, long sam, long sak, long sbn, long sbk, long ldc, int accumulate,
                  int ksplit /* K slices; > 1: partial tiles through ws, added in slice order */,
                  float* ws /* ksplit * M * N floats when ksplit > 1 */, void* stream);
int x2_l2norm(const float* x, const float* dy, float* out, int R, int D, int bwd, void* stream);   /* F.normalize */
/* F.cross_entropy / CrossEntropyLoss(ignore_index=-100): xvlm.py:812-813, 899; xbert.py:1660-1661.
 * out2 = {mean loss, number of counted rows}; backward writes (softmax - onehot) * gscale * g / count. */
int x2_ce_fwd(const float* logits, long ld, const long* labels, int R, int C, float* lse, float* loss_row, float* out2,
              void* stream);
int x2_ce_bwd(const float* logits, long ld, const long* labels, const float* lse, const float* g, const float* stat,
              float gscale, int R, int C, float* dl_f32, void* dl_bf16, long ldd, void* stream);
/* The MLM head's cross-entropy over the tied decoder with the logits kept in the GEMM accumulators (reference:
 * BertLMPredictionHead.decoder = nn.Linear(hidden, vocab) followed by CrossEntropyLoss, xbert.py:822, 1653-1661; there
 * the [R, V] logits are materialised in fp16 and up-cast for the loss).  X [R, Hd] bf16 = transformed masked rows,
 * E [Vp, Hd] bf16 = word embeddings padded to Vp % 64 == 0 rows, bias [Vp], labels [R] (< 0: ignored), V = vocabulary.
 *   x2_mlm_ce_fwd : part[R][Vp/64][2] = (max, sum exp(z - max)) of every 64-column chunk of z = X.E^T + bias (columns
 *                   < V only), zlab[r] = z[r][label[r]]
 *   x2_ce_combine : lse[r] from the chunks of row r, loss_row[r] = lse - zlab (0 for ignored rows),
 *                   out2 = {mean loss, counted rows}
 *   x2_mlm_ce_bwd : dl[R][ldd] bf16 = (exp(z - lse[r]) - [c == label]) * gscale * g[0] / stat[1], z recomputed by the same
 *                   GEMM; ignored rows and columns >= V are written as 0 */
int x2_mlm_ce_fwd(const void* X, const void* E, const float* bias, const long* labels, int R, int Vp, int V, int Hd,
                  int ldx, int lde, float* part, float* zlab, void* stream);
int x2_ce_combine(const float* part, int chunks, const float* zlab, const long* labels, int R, float* lse, float* loss_row,
                  float* out2, void* stream);
int x2_mlm_ce_bwd(const void* X, const void* E, const float* bias, const long* labels, const float* lse, const float* g,
                  const float* stat, float gscale, int R, int Vp, int V, int Hd, int ldx, int lde, void* dl_bf16, long ldd,
                  void* stream);
/* MLM text masking of a padded batch on the device (csrc/masking.hip): dataset/pretrain_dataset.py:59-130 (TextMaskingGenerator.__call__) and
 * :242-275 (ImageTextJsonDataset.preprocess: masked_ids, padding) - host work of the reference's data-loader workers.
 *   text_ids, text_atts [B, L] int64 (captions left-aligned, atts = 1 on the caption, first token = cls_id); is_subword [vocab] bytes, 1 where the
 *   token text starts with '##' (WordPiece continuation); outputs int64: text_ids_masked [B, L] (pad_id beyond the caption), masked_pos [B, max_masks]
 *   (pad 0), masked_ids [B, max_masks] (original ids, pad pad_mask = -100).  2 <= L <= 512.
 *   Random draws come from a stream of 32-bit words per caption, consumed in the reference's draw order (rand() < p: word < ceil(p 2^32);
 *   randint(a, b): a + ((word (b - a + 1)) >> 32); shuffle: random.shuffle's loop with j = (word (i + 1)) >> 32): words != NULL: caption b reads
 *   words[b * words_ld + k] (injected, parity tests: at most 4 L + 64 are consumed); words == NULL: word k = hash(seed', b, k) with
 *   seed' = epoch ? hash(seed + 0x9E3779B1 * *epoch) : seed (a hipGraph-captured step increments *epoch per replay: new masks every step).
 *   The order of the reported positions is the iteration order of the CPython set the reference collects them in (restated in the kernel). */
int x2_mask_tokens(const long* text_ids, const long* text_atts, int B, int L, const unsigned char* is_subword, int vocab,
                   const unsigned* words, int words_ld, unsigned seed, const unsigned* epoch, double mask_prob, int max_masks,
                   double skipgram_prb, int skipgram_size, int mask_whole_word, long cls_id, long mask_id, long pad_id, long pad_mask,
                   long* text_ids_masked, long* masked_pos, long* masked_ids, void* stream);
/* hard negatives, xvlm.py:828-857: softmax(sim)+1e-5 with the diagonal (or same-group entries) zeroed, one
 * inverse-CDF draw per row from u[b] in [0,1); replaces 2*B torch.multinomial(...).item() host syncs */
int x2_sample_negatives(const float* sim, int n, const long* group, const float* u, int* out, void* stream);
/* additive key mask of BertModel (get_extended_attention_mask: neg = -10000; invert_attention_mask: -1e9; xbert.py:1105-1160):
 * out[s][l] = (1 - atts[s][l]) * neg, l < L; 0 in the pad columns L..Lp-1 (the attention kernels read [S][Lp], Lp % 64 == 0) */
int x2_additive_mask(const long* atts, float* out, int S, int L, int Lp, float neg, void* stream);
/* CSR "K/V batch -> query sequences using it" from kv[S] (values in [0, Bi)): off[Bi+1], order[S] (stable counting sort);
 * the seq_off / seq_ids tables of X2AttnArgs for rows that share an image's K/V (the 4-pass fusion batch) */
int x2_kv_csr(const int* kv, int S, int Bi, int* off, int* order, void* stream);
/* row tables of the 4B-row fusion batch the step runs instead of the reference's four fusion passes (models/model_pretrain.py:44-62,
 * xvlm.py:859-899): row q * B + b = (text b, image b) | (text b, image ineg[b]) | (text tneg[b], image b) | (masked text b, image b);
 * t_idx[r] = row of the 2B-row [clean ; masked] text batch, kv[r] = image, atts_out[r] = text mask, enc_out[r] = image mask.
 * with_match == 0: the masked rows only (ret_match_loss=False).  Replaces torch.arange / cat / index launches. */
int x2_tail_index(const int* ineg, const int* tneg, const long* text_atts /* [B][L] */, const long* image_atts /* [Bi][T] */, int B, int L,
                  int T, int with_match, int* t_idx, int* kv, long* atts_out /* [R][L] */, long* enc_out /* [R][T] */, void* stream);
/* timm drop_path of the BEiT blocks (beit2.py:205-207, rates linspace(0, 0.1, depth) beit2.py:314): per-row keep / (1 - rate[l])
 * factors out[depth][2 branches][B * T], one Bernoulli per (block, branch, sample) hashed from (seed [, *epoch]) */
int x2_droppath_rows(const float* rates, unsigned seed, const unsigned* epoch, int depth, int B, int T, float* out, void* stream);
/* video path, xvlm.py:627-645 ('avgpool'): forward out[Bc][T][D] = mean_f (x[Bc * F][T][D] + pos[F][D]) (pos may be NULL);
 * backward (bwd != 0): out = dx[Bc * F][T][D] = dy / F per frame, dpos[F][D] = column sums of dy / F (NULL: not wanted) */
int x2_frame_mean(const float* x, const float* pos, const float* dy, float* out, float* dpos, int Bc, int F, int T, int D, int bwd,
                  void* stream);
int x2_gelu_f32(const float* x, const float* dy, float* out, long n, void* stream);               /* nn.GELU, xvlm.py:167 */
int x2_colsum_f32(const float* x, float* out, int M, int N, void* stream);

/* ---- optimizer (csrc/optim.hip) -- "next" row of the scope table ------------------------------------------
 * optim.py:26-104 (transformers==4.12.5 AdamW, eps 1e-8, betas (0.9,0.98), correct_bias) and the global-norm clip of
 * accelerators/apex_ddp_accelerator.py:99-102, as two multi-tensor launches.  table: ntensors records
 * {float* p; const float* g (NULL = no gradient); float* m; float* v; long n; int group; int blk0; float stepscale; int pad}
 * in device memory, blk0 = prefix sum of ceil(n / 16384); nblocks = total; stepscale = sqrt(1 - b2^t) / (1 - b1^t) with t the
 * tensor's own step count (HF AdamW keeps state["step"] per parameter).  out2 = {total norm, min(1, max_norm / (norm + 1e-6))}. */
int x2_grad_norm(const void* table, int ntensors, int nblocks, float max_norm, float* partial, float* out2, void* stream);
int x2_adamw_multi(const void* table, int ntensors, int nblocks, const float* lr, const float* wd, int ngroups, float b1,
                   float b2, float eps, const float* clip2 /* out2 of x2_grad_norm or NULL */, void* stream);

/* ---- data-parallel communication (csrc/comm.hip) -----------------------------------------------------------
 * RCCL over xGMI behind the C ABI, for hosts that are not PyTorch (the Python host may use either these or
 * torch.distributed's "nccl" backend, which is the same RCCL).  Replaces accelerators/apex_ddp_accelerator.py:57-66
 * (NCCL init), :70-77 (per-tensor broadcast), :80-97 (apex DDP: flat all-reduce + average) and models/xvlm.py:140-160
 * (ITC all_gather; its backward keeps the local slice and needs no collective).  One communicator per process, one
 * process per GPU.  librccl.so.1 is loaded on first use (the library itself has no link-time RCCL dependency).
 * Every collective is enqueued on `stream` (the caller's communication stream) and, if done_event (a hipEvent_t) is
 * non-NULL, records it behind the collective.  dtype: 0 = fp32, 1 = bf16. */
int x2_comm_unique_id(void* out128);                                     /* rank 0; share the 128 bytes out of band */
int x2_comm_init(const void* id128, int rank, int world, void** comm_out);  /* collective; current HIP device = this rank's GPU */
int x2_comm_info(void* comm, int* rank, int* world);
int x2_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, int average, void* done_event, void* stream);
int x2_comm_allgather(void* comm, const void* send, void* recv, long count_per_rank, int dtype, void* done_event, void* stream);
int x2_comm_broadcast(void* comm, void* buf, long count, int dtype, int root, void* done_event, void* stream);
int x2_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
