/* libfbl -- C ABI of the MI355X (gfx950) kernels behind FrozenBiLM's masked-LM hot path.
 *
 * The reference (antoyang/FrozenBiLM) has no FFI: its hot path is stock ATen ops called from
 * model/deberta.py + model/adapter.py.  This header is therefore the boundary the build inserts BELOW the
 * reference's Python call boundary: every entry point replaces a group of ATen calls, cited
 * per function as "ref: file:line".  Conventions:
 *   - extern "C", plain device pointers + sizes, no torch types; `stream` is a hipStream_t passed as void*.
 *   - every function only enqueues work on `stream` (graph-capturable: no malloc/free/sync inside) and returns
 *     0 on success, a positive hipError_t, or a negative FBL_ERR_* code for argument errors.  The library owns no
 *     stream and reads no environment variable.  Two GEMM entry points take an optional caller-provided `aux_stream`
 *     (see fbl_gemm_bf16_nt): work put there is forked from and joined back into `stream` by events inside the call,
 *     so after the call everything still depends only on `stream` -- also under stream capture.  Those events (one
 *     pair per (stream, aux_stream) pair, created on first use on the calling thread's current device, mutex-guarded)
 *     are the only process state of the library; calls on different streams may come from different threads.
 *   - bf16 tensors are raw uint16 storage (torch.bfloat16), fp32 are float, indices int64/int32 as stated.
 *   - row-major; `ld*` are row strides in ELEMENTS.
 */
#ifndef FBL_H
#define FBL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FBL_ERR_SHAPE (-1)
#define FBL_ERR_ALIGN (-2)
#define FBL_ERR_ARG (-3)

/* Dropout seeds.  Every entry point with a dropout site takes `uint64_t seed` and, right behind it, `const uint64_t*
 * seed_dev` (device pointer or NULL): the mask of element e is a pure function of (seed + *seed_dev, e) -- seed alone when
 * seed_dev is NULL -- so a backward launch given the same pair regenerates the forward's mask.  The device word exists for
 * launch graphs: `seed` is a kernel argument and frozen into a captured launch, the word is read when the kernel runs, so
 * advancing it between replays gives every replay fresh masks. */

enum {
  FBL_ACT_NONE = 0,
  FBL_ACT_GELU = 1,
  FBL_ACT_RELU = 2,
  FBL_ACT_GELU_GRAD = 3 /* out = gelu(v); out_pre receives gelu'(v) (not v): backward is then a plain multiply */
};
enum {
  FBL_AUX_NONE = 0,
  FBL_AUX_ADD_F32 = 1,        /* out = act(..) + aux_f32[m,n]                       (residual / grad accumulate) */
  FBL_AUX_ADD_BF16 = 2,       /* out = act(..) + aux_bf16[m,n]                                                    */
  FBL_AUX_MUL_DGELU_BF16 = 3, /* out = (..) * gelu'(aux_bf16[m,n])   (backward of deberta.py:310-313)            */
  FBL_AUX_MUL_POS_BF16 = 4,   /* out = (..) * (aux_bf16[m,n] > 0)    (backward of adapter.py:39 ReLU[+dropout])  */
  FBL_AUX_MUL_BF16 = 5,       /* out = (..) * aux_bf16[m,n]          (aux = gelu' saved by FBL_ACT_GELU_GRAD)    */
  FBL_AUX_ADAPTER_TAIL = 6    /* internal to fbl_adapter_up_resid_fwd; fbl_gemm_bf16_nt rejects it                */
};

/* Bumped whenever the exported interface changes (6: this header -- fbl_dropout_sum_f32, fbl_zero, fbl_heads_to_rows_bf16, fbl_gt_tilemask added; tile masks on the shear pass and the k-skipping GEMM); the ctypes binding refuses any other value. */
int fbl_abi_version(void);

/* C[M,N] = epi(alpha * A[M,K] . B[N,K]^T): bf16 MFMA, fp32 accumulate.  K % 64 == 0, lda/ldb % 8 == 0.
 * epi: v = alpha*acc + bias[n]; v *= rowscale[m]; pre = v; v = act(v); v = aux-op(v); -> out_f32 / out_bf16 (/ out_pre).
 * batch > 1: strided batch (strides in elements).  splitk > 1: ACCUMULATE mode, out_f32 += A.B^T with the K range
 * split over `splitk` workgroup sets; partial tiles go to `splitk_ws` (>= batch*splitk*M*roundup(N,4) floats) and are
 * folded deterministically by a second tiny kernel; with splitk_ws == NULL (or too small) they are atomicAdd-ed.
 * kskip_len != NULL (split-K / accumulate mode only): K consists of samples of kskip_steps 64-wide k-steps each and step j of
 * sample b is known to be all zero in A when 64*j >= kskip_len[b] (int32, device): such steps are neither read nor
 * multiplied -- the rows of G^T beyond a sample's last valid position (fbl_disent_attn_bwd_shear leaves them unwritten).
 * kskip_tilemask != NULL (with kskip_len; uint32 per 64-wide k-step, device): bit t set <=> rows [128t, 128t+128) of A can be
 * non-zero in that k-step; (tile, step) pairs whose bit is clear are neither read nor multiplied (fbl_gt_tilemask: the rows of
 * G^T outside the window a 64-row step can reach, which fbl_disent_attn_bwd_shear then does not even zero-fill).
 * a_kblock_stride > 0: A is k-blocked, A[m][k] lives at m*lda + (k/32)*a_kblock_stride + k%32 (the G^T layout written
 * by fbl_disent_attn_bwd_shear, lda = 32); 0 = plain K-contiguous rows.
 * aux_stream (may be NULL): a second stream of the caller on the same device.  A multi-round problem runs its last,
 * partly filled round as small tiles; with an aux stream those are launched there first, concurrently with the big tiles
 * (fork / join by events inside this call).  NULL: they simply precede the big tiles on `stream`.
 * ref: every nn.Linear on the path -- model/deberta.py:255,311,329,757-765,847-853,994,1545,1550;
 *      model/adapter.py:38,42; conv1d deberta.py:397 (as K=3H GEMM); and their autograd dX/dW. */
int fbl_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, const float* bias,
                     const float* rowscale, float alpha, int act, int aux_kind, const void* aux, int64_t ld_aux,
                     float* out_f32, void* out_bf16, void* out_pre_bf16, int64_t ldc, int batch, int64_t strideA,
                     int64_t strideB, int64_t strideC, int64_t strideAux, int64_t strideBias, int splitk,
                     float* splitk_ws, int64_t splitk_ws_floats, int64_t a_kblock_stride, const int32_t* kskip_len,
                     int kskip_steps, const uint32_t* kskip_tilemask, void* stream, void* aux_stream);

/* Host-side query, no launch: the kernel fbl_gemm_bf16_nt uses for a plain (K-contiguous, no k-skip) problem of this
 * shape: 8 = the 8-phase 256x256 / 224x256 kernel (gemm8_kernel; remainder rows of a multi-round problem run as 64x128
 * tiles), 2 = the 2-stage 128x128 / 64x128 kernel.  bench.py attributes launches to the dominant kernel with it. */
int fbl_gemm_plan(int M, int N, int K, int batch, int splitk);

/* Adapter down-projection with the whole bottleneck non-linearity in the GEMM epilogue:
 *   z[M, A] = dropout_p( relu( x[M,K] . Wd[A,K]^T + bd ) )        (bf16 MFMA, fp32 accumulate, bf16 out)
 * Dropout is the counter-based one of fbl_dropout_bf16: element (m, a) is keyed by (seed, m*ldz + a), dropped elements
 * are exactly 0, kept ones scaled by 1/(1-p); p_drop == 0 -> plain ReLU.  K % 64 == 0, ldx/ldw % 8 == 0.
 * The backward needs no mask: d/dz passes where z > 0, scaled by 1/(1-p) (FBL_AUX_MUL_POS_BF16 with alpha).
 * ref: model/adapter.py:38-41 (down -> ReLU -> nn.Dropout). */
int fbl_adapter_down_fwd(const void* x_bf16, int64_t ldx, const void* wd_bf16, int64_t ldw, int M, int A, int K,
                         const float* bias, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* z_bf16, int64_t ldz,
                         void* stream);

/* A dense layer and the down-projection of the adapter behind it as ONE GEMM.  wm = [W ; Wd.W] ([N1 + A, K] bf16: the
 * lower A rows hold the adapter's down-projection composed with the dense weight -- the caller rebuilds them, with
 * fbl_gemm_bf16_nt, whenever Wd changes), bias_m = [b ; Wd.b + bd] ([N1 + A] fp32):
 *   y[M, N1] = x.W^T + b  -> y_f32 and/or y_bf16 (row stride ldy);   z[M, A] = dropout_p(relu(x.(Wd.W)^T + Wd.b + bd)) -> z_bf16.
 * z equals fbl_adapter_down_fwd(y) up to bf16 rounding of the operands (y is not rounded to bf16 on the way), dropout
 * keyed by (seed, m*ldz + a).  N1 % 64 == 0 (FBL_ERR_ARG otherwise: the segment boundary must not cut a wave's column
 * range; the 256-wide tiles are only used when N1 % 256 == 0), K % 64 == 0, ldx/ldw % 8 == 0.
 * ref: model/deberta.py:255-257, 329-331 (dense -> adapter) + model/adapter.py:38-41. */
int fbl_dense_adapter_down_fwd(const void* x_bf16, int64_t ldx, const void* wm_bf16, int64_t ldw, int M, int N1, int A,
                               int K, const float* bias_m, float* y_f32, void* y_bf16, int64_t ldy, float p_drop,
                               uint64_t seed, const uint64_t* seed_dev, void* z_bf16, int64_t ldz, void* stream,
                               void* aux_stream);

/* Everything between an adapter's bottleneck and the LayerNorm statistics behind it, as the epilogue of the up-projection:
 *   t[M, H] = dropout_p( x[M,H] + z[M,A] . Wu[H,A]^T + bu ) + resid[M,H]                      (fp32 out, row stride ldt)
 * x: the adapter's input (the dense layer's output, bf16); resid either plain fp32 (r_stats == NULL: r_t[m,n], row stride
 * ld_r) or in LayerNorm-normalised form ((r_t[m,n] - r_stats[2m]) * r_stats[2m+1] * r_gamma[n] + r_beta[n]) * r_rowmask[m]
 * (r_rowmask int32 or NULL) -- the representation fbl_ln_fwd leaves behind (out_t, out_stats).  Dropout keys as in
 * fbl_ln_fwd: element (m, n) by (seed, m*H + n), so fbl_ln_bwd regenerates the same mask.  t is the pre-norm tensor of the
 * following LayerNorm: fbl_ln_fwd(y = t, no dropout, no residual, out_t = NULL) completes the block.  The adapter output
 * never reaches HBM on its own.  A % 64 == 0 (K of the GEMM), H % 4 == 0, ldz/ldw/ldx % 8 == 0, ldt/ld_r % 4 == 0.
 * ref: model/adapter.py:42-45 (up + residual), model/deberta.py:258-259, 332-333 (dropout; LayerNorm(. + input_tensor)). */
int fbl_adapter_up_resid_fwd(const void* z_bf16, int64_t ldz, const void* wu_bf16, int64_t ldw, int M, int H, int A,
                             const float* bias_u, const void* x_bf16, int64_t ldx, float p_drop, uint64_t seed,
                             const uint64_t* seed_dev, const float* r_t, int64_t ld_r, const float* r_stats, const float* r_gamma,
                             const float* r_beta, const int32_t* r_rowmask, float* out_t, int64_t ldt, void* stream);

/* Weight and bias gradients of a GROUP of bottleneck adapters of one shape, accumulated (+=) by ONE launch:
 *     dWu[o][H,A] += sum_s dy_s^T . z_s      dWd[o][A,H] += sum_s dz_s^T . x_s      dbd[o][A] += sum_s colsum(dz_s)
 * over the segments s in [seg_first[o], seg_first[o+1]) of adapter o (one segment per execution of the adapter in the
 * forward pass; seg_first: HOST array of n_adapters + 1 ints, seg_first[0] = 0).  dy_s, x_s: bf16 [N,H] (row strides
 * ld_dy[o], ld_x[o]: HOST arrays, one entry per adapter -- its segments share them; operands may be column slices of wider
 * buffers); z_s = dropout(relu(down(x_s))), dz_s = grad of the bottleneck pre-activation: bf16 [N,Ap], Ap = A rounded up to 64
 * (<= 256), columns A..Ap-1 zero.  The pointer tables (dy_bf16 .. x_bf16: one entry per segment; dWu, dWd, dbd: one entry
 * per adapter, a table or an entry may be NULL) are HOST arrays of DEVICE pointers, read during the call.  Every output
 * tile has one writer that walks all rows: no workspace, deterministic; the more adapters per call, the better the chip is
 * filled (2 x ceil(H/64) workgroups per adapter).  Gradients are contiguous fp32.  Strides and H multiples of 8.
 * (up.bias's gradient colsum(dy) comes out of fbl_ln_bwd's `dysum`.)
 * ref: autograd of model/adapter.py:38-42 (down, ReLU, dropout, up). */
#define FBL_ADW_MAX_ADAPTERS 16
#define FBL_ADW_MAX_SEGMENTS 24
int fbl_adapter_bwd_dw(int n_adapters, const int32_t* seg_first, const void* const* dy_bf16, const void* const* z_bf16,
                       const void* const* dz_bf16, const void* const* x_bf16, const int64_t* ld_dy, const int64_t* ld_z,
                       const int64_t* ld_dz, const int64_t* ld_x, int N, int H, int A, int Ap, float* const* dWu,
                       float* const* dWd, float* const* dbd, void* stream);

/* out_f32[M,N] += sum_k A[k,m] * B[k,n]: both operands row-major bf16 ([K,M] and [K,N]), contraction over ROWS, so the
 * trainable-weight gradients dW = X^T . dY need no transposed copies in HBM.  Split-K with deterministic workspace fold
 * (splitk_ws >= splitk*M*roundup(N,4) floats, required).  M, N, lda, ldb multiples of 8; K arbitrary.
 * ref: autograd dW of model/adapter.py:38,42 and of linear_video (model/deberta.py:994). */
int fbl_gemm_bf16_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, float* out_f32,
                         int64_t ldc, int splitk, float* splitk_ws, int64_t splitk_ws_floats, void* stream);

/* t0[b, s, :] = s < T ? vproj[b*T + s, :] : E[ids[b, s-T], :]     (fp32).  vproj may be NULL (T = 0).
 * ref: model/deberta.py:1012-1016 (word_embeddings + cat with linear_video output). */
int fbl_embed_gather(const int64_t* ids, const float* E, const float* vproj, int B, int T, int L, int H, float* out_t,
                     void* stream);

/* Input side.  feats: packed fp16 [sum n_rows, F] (the per-video CLIP feature files back to back), clip b starts at row
 * row_off[b] and has n_rows[b] rows.  out[b, j, :] (fp32) = clip row (j*n)/T when n > T, row j when j < n <= T, zeros
 * otherwise; video_len[b] = min(n, T); video_mask[b, j] = j < video_len[b] (int64, either may be NULL).  F % 8 == 0.
 * ref: datasets/videotext_dataset.py:23-43 (uniform subsample / zero pad) + util/misc.py:6-11 (get_mask). */
int fbl_video_stage_f16(const void* feats_f16, const int64_t* row_off, const int32_t* n_rows, int B, int T, int F,
                        float* out_f32, int64_t* video_len, int64_t* video_mask, void* stream);

/* Masked-LM corruption in place on the device: tokens whose id is not in special_ids[n_special] are selected with
 * probability mlm_probability; labels = original id where selected, -100 elsewhere; 80 % of the selected become
 * mask_token_id, 10 % a uniform id in [0, vocab_size), 10 % stay.  Counter-based RNG keyed by (seed, 4*index + draw).
 * ref: util/misc.py:14-56 (same distribution; the reference draws from torch's CPU generator, so bit parity with it is
 * provided by the host implementation frozenbilm_amd.util.misc.mask_tokens, not by this kernel). */
int fbl_mask_tokens(int64_t* ids, int64_t* labels, int64_t n, const int64_t* special_ids, int n_special,
                    float mlm_probability, int64_t mask_token_id, int64_t vocab_size, uint64_t seed, void* stream);

/* Fused  t = dropout(y) + residual ;  out = LayerNorm(t) * gamma + beta [* rowmask]     (one wave per row).
 *  y: fp32 [N, ldy] or NULL.  dropout: p in [0,1), element (row*H+col) keyed by `seed` (counter-based, regenerated in bwd).
 *  residual, either  r_plain fp32 [N,H]  or the "normalised form" of a previous LayerNorm output:
 *     (r_t - mean)*rstd*r_gamma + r_beta [* r_rowmask]  with (mean,rstd) = r_stats[row*2..]  -- the fp32 residual stream
 *     is never materialised, only t and the row statistics are.
 *  writes out_t fp32 [N,H] (pre-LN, saved for backward), out_stats fp32 [N,2], out_bf16 [N,H] (GEMM operand),
 *  out_f32 optional.  H % 64 == 0, H <= 2048.
 * ref: model/deberta.py:258-259, 332-333 (dropout + residual + LayerNorm), :1043-1054 (embeddings), :405-417 (conv). */
int fbl_ln_fwd(const float* y, int64_t ldy, float p_drop, uint64_t seed, const uint64_t* seed_dev, const float* r_plain, const float* r_t,
               const float* r_stats, const float* r_gamma, const float* r_beta, const int32_t* r_rowmask,
               const float* gamma, const float* beta, float eps, const int32_t* rowmask, float* out_t,
               float* out_stats, void* out_bf16, float* out_f32, int N, int H, void* stream);

/* out = LN-normalised-form(t, stats, gamma, beta)[*rowmask] + add_bcast[row % S]  -> fp32 and/or bf16.
 * ref: model/deberta.py:1392 (z_states += hidden_states) and materialising hidden_states on request. */
int fbl_ln_materialize(const float* t, const float* stats, const float* gamma, const float* beta,
                       const int32_t* rowmask, const float* add_bcast, int S, float* out_f32, void* out_bf16, int N,
                       int H, void* stream);

/* Backward of fbl_ln_fwd.  dout fp32 [N,H] = grad of the LN output (after rowmask).
 *  out_dt fp32 [N,H]: grad wrt t (= grad of the residual branch).  out_dy_bf16/out_dy_f32: grad wrt y (dropout mask
 *  regenerated from seed), optional.  dgamma/dbeta [H] and dysum [H] (= column sums of dy: the bias gradient of the
 *  layer that produced y) are ACCUMULATED (+=) deterministically via `ws` (fbl_ln_bwd_ws_floats(H) floats); each may
 *  be NULL.  ld_dy_bf16: row stride of out_dy_bf16 in elements (0 = H; a multiple of 8): dy can land in the first H columns
 *  of a wider operand buffer ([dy | dz] of the folded adapter backward).
 * ref: autograd of torch.nn.LayerNorm + XDropout.backward (model/deberta.py:185-190). */
int64_t fbl_ln_bwd_ws_floats(int H);
int fbl_ln_bwd(const float* dout, const int32_t* rowmask, const float* t, const float* stats, const float* gamma,
               float p_drop, uint64_t seed, const uint64_t* seed_dev, float* out_dt, void* out_dy_bf16, float* out_dy_f32, float* dgamma,
               float* dbeta, float* dysum, float* ws, int N, int H, int64_t ld_dy_bf16, void* stream);

/* out[n, k*H + c] = x[b, s+k-1, c] (0 outside the sequence), n = b*S+s, k in {0,1,2}: im2col for the 3-tap conv.
 * ref: model/deberta.py:396-400 (Conv1d k=3 pad=1 over the sequence axis). */
int fbl_im2col3(const void* x_bf16, void* out_bf16, int B, int S, int H, void* stream);
/* backward of im2col3: dx[b,s,c] (+)= sum_k dcol[b, s-k+1, k*H + c]   (fp32 in, fp32 out, accumulate flag) */
int fbl_col2im3(const float* dcol, float* dx, int B, int S, int H, int accumulate, void* stream);

/* y = gelu(dropout(c)) elementwise, fp32 -> fp32 (+ optional bf16).  ref: model/deberta.py:403. */
int fbl_dropout_gelu_fwd(const float* c, float p_drop, uint64_t seed, const uint64_t* seed_dev, float* out_f32, int64_t n,
                         void* stream);
/* dc = dy * gelu'(dropout(c)) * dropscale ; writes bf16 (GEMM operand) */
int fbl_dropout_gelu_bwd(const float* dy, const float* c, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* out_bf16, float* out_f32,
                         int64_t n, void* stream);

/* out_bf16[c, r] = in[r, c] for r < rows, 0 for rows <= r < rows_pad; in is fp32 (in_is_bf16=0) or bf16.
 * Produces the K-contiguous operands of the dW = X^T.dY contractions (contraction over rows). */
int fbl_transpose_to_bf16(const void* in, int in_is_bf16, int64_t ld_in, int rows, int cols, void* out_bf16,
                          int64_t rows_pad, void* stream);

/* count equally-shaped contiguous bf16 matrices [rows, cols] at element offsets src_off[i] of src are written
 * transposed ([cols, rows]) at dst_off[i] of dst; offsets are int64 arrays in device memory.  One launch rebuilds the
 * W^T operands of all adapters (autograd of model/adapter.py:38,42) after an optimizer step. */
int fbl_transpose_batched_bf16(const void* src_bf16, const int64_t* src_off, void* dst_bf16, const int64_t* dst_off,
                               int count, int rows, int cols, void* stream);

/* out[c] += sum_r in[r, c]  (bias gradients).  in fp32 or bf16.  ws: fbl_colsum_ws_floats(cols) floats. */
int64_t fbl_colsum_ws_floats(int cols);
int fbl_colsum(const void* in, int in_is_bf16, int64_t ld_in, int rows, int cols, float* out, float* ws,
               void* stream);

/* Per-head transpose: vt[h*out_sh + b*out_sb + d*out_sd + s] = V[b*S+s, h*64+d] (s < S), 0 for S <= s < Sp.
 * V has row stride ldv (e.g. 3H inside the fused QKV buffer).  Gives the P.V / dS.K / dS^T.Q MFMAs their
 * position-contiguous operands; with (sh, sd, sb) = (64*B*Sp, B*Sp, Sp) the rows are also the K-contiguous operands of
 * the per-head position-table gradient GEMMs. */
int fbl_head_transpose(const void* v_bf16, int64_t ldv, void* vt_bf16, int B, int S, int Sp, int nh, int64_t out_sh,
                       int64_t out_sb, int64_t out_sd, void* stream);

/* Fused disentangled self-attention forward (one launch per layer execution):
 *   score[i,j] = scale*(Q_i.K_j + Q_i.PK[idx(i-j)] + K_j.PQ[idx(i-j)]),  masked softmax (masked -> 0, fully masked
 *   rows -> 0), attention-prob dropout, ctx = P.V.   head_dim = 64, S <= 512.
 *   q/k/v: bf16 rows b*S+s, head h at column h*64 (row strides ldq/ldk/ldv) -- V is consumed row-major, its transposed
 *   MFMA fragments come from ds_read_b64_tr_b16;
 *   pk/pq bf16 [2*span, ldp]; relidx int16 [2S-1]: relidx[d+S-1] = clamp(bucket(d)+span, 0, 2span-1);
 *   mask int32 [B,S]; klen int32 [B] (optional): last valid position + 1 -- tiles beyond it are exactly zero and are
 *   skipped; border int32 [B] (optional): a permutation of the samples, the order in which they are dispatched
 *   (longest first balances the ragged batch over the CUs; the results do not depend on it);
 *   lin_span: |d| < lin_span => relidx[d+S-1] = relidx[S-1] + d (the identity buckets: position_buckets/2; 0 if
 *   unknown) -- tile pairs inside that band take index-table-free addressing;
 *   out ctx bf16 [B*S, ldo]; lse fp32 [B,nh,S] (log-sum-exp of the scaled scores, +inf for empty rows).
 *   row0 int32 [B+1] (optional; needs klen): PACKED rows -- a ragged batch stored without its trailing padding rows.
 *   Sample b then owns the q / k / v / ctx rows [row0[b], row0[b+1]), which are its positions 0 .. row0[b+1]-row0[b]-1
 *   (>= klen[b]: every position up to the last valid one has a row; rows a sample does not have are neither read nor
 *   written).  mask, lse and the scratch tensors of the backward keep their padded [B, S(p)] indexing.  The rows that
 *   exist receive exactly the values of the padded layout.  The same argument on the three backward entry points below
 *   (fbl_attn_bwd_prep: q / k / dO / O; fbl_disent_attn_bwd_ds: q / k / v / dO / dV; fbl_disent_attn_bwd_shear: out).
 *   psave / msave (optional, both or neither; training): the forward also leaves what the backward would otherwise
 *   recompute -- psave bf16 [B,nh,Sp,Sp]: psave[b,h,i,j] = exp2(k2*(s_ij - m)) with s the unscaled score, k2 = scale*log2(e)
 *   and m the running row maximum when key tile j/64 was processed (BEFORE dropout, exactly 0 for masked keys; the SIGN BIT
 *   carries the dropout decision of the pair: set = dropped, so the backward does not regenerate the mask); msave fp32
 *   [B,nh,Sp/64,S]: msave[b,h,j/64,i] = k2*m.  P_ij = psave_ij * exp2(msave - lse_i*log2(e)).  Only the tile pairs the forward
 *   visits (both tile indices below ceil(klen/64)) are written -- fbl_disent_attn_bwd_dsp reads exactly those.
 * ref: model/deberta.py:717-818 (forward), :820-947 (disentangled_attention_bias), :100-138 (XSoftmax). */
int fbl_disent_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                        const void* pk, const void* pq, int64_t ldp, const int16_t* relidx,
                        const int32_t* mask, const int32_t* klen, const int32_t* border, float scale, float p_drop,
                        uint64_t seed, const uint64_t* seed_dev, void* ctx, int64_t ldo, float* lse, int B, int S, int Sp, int nh, int span2,
                        int lin_span, const int32_t* row0, void* psave, float* msave, void* stream);

/* The attention probabilities of fbl_disent_attn_fwd, materialised on request (output_attentions=True; never on the hot
 * path): probs[b, h, i, j] fp32 [B, nh, S, S] = exp(score[i,j] - lse[b,h,i]) with the lse the fused forward stored, exactly 0
 * for masked pairs and masked query rows; q/k/pk/pq/relidx/mask as in fbl_disent_attn_fwd (eval mode: no dropout).
 * ref: model/deberta.py:789-818 (return_att), :544-560 (the encoder's attentions tuple). */
int fbl_disent_attn_probs(const void* q, const void* k, int64_t ldq, const void* pk, const void* pq, int64_t ldp,
                          const int16_t* relidx, const int32_t* mask, const float* lse, float scale, float* probs, int B,
                          int S, int nh, void* stream);

/* Backward of fbl_disent_attn_fwd, three launches (ref: autograd of model/deberta.py:717-947, XSoftmax.backward
 * :134-138, XDropout.backward :185-190):
 *  fbl_attn_rowdot:            Dv[b,h,i] = dO_i . O_i  (per head).
 *  fbl_disent_attn_bwd_ds:     recomputes P; writes dV (bf16, into a row-major buffer), dS and dS^T (bf16 [B,nh,Sp,Sp],
 *                              dS = P*(dP - Dv)*scale, exactly 0 where masked / padded; with klen only the
 *                              [klen x klen] corner (rounded up to 64) is written and read).
 *  fbl_disent_attn_bwd_shear:  neg=0: out = dQ = dS.K + G1.PK,  G1[i,r] = sum_{j: idx(i-j)=r} dS[i,j]
 *                              neg=1: out = dK = dS^T.Q + G2.PQ, G2[j,r] = sum_{i: idx(i-j)=r} dS[i,j]
 *                              X = dS / dS^T; YT = transposed K / Q (fbl_head_transpose strides); PT = transposed
 *                              PK / PQ [nh][64][span2]; also writes GT = G^T (bf16), the A operand of the position-table
 *                              gradient GEMM dPK[h] = G1T[h] . QT[h]^T (dPQ: G2T, KT), k-blocked so that every
 *                              workgroup writes one contiguous block: GT[h][b][t][r][32] with t = row/32, r in
 *                              [0, gt_rcnt) standing for table row gt_rmin + r (the range of relidx; others are 0).
 *                              lin_span: |i-j| < lin_span => relidx is injective there (identity buckets, = position_buckets/2;
 *                              0 if unknown): those entries are scattered with plain LDS stores instead of atomics.
 * With klen given, G^T blocks (32 rows) of 64-row steps that start beyond klen[b] are left UNWRITTEN: their consumer
 * (fbl_gemm_bf16_nt with kskip_len = klen) never reads them. */
int fbl_attn_rowdot(const void* dO, const void* O, int64_t ld, float* out, int B, int S, int nh, void* stream);
/* Position-table gradients of E layer executions in one launch, straight from the dS / dS^T tensors of fbl_disent_attn_bwd_ds(p):
 *   neg = 0: out[e][h][r][d] = dPK = sum_b sum_{(i,j): relidx(i-j) = rmin + r} dS[i,j] * Q[b*S+i, h*64+d]     (X = dS,   Y = q)
 *   neg = 1: out[e][h][r][d] = dPQ = sum_b sum_{(i,j): relidx(i-j) = rmin + r} dS[i,j] * K[b*S+j, h*64+d]     (X = dS^T, Y = k)
 * X, Y: HOST arrays of E device pointers (X[e]: bf16 [B,nh,Sp,Sp]; Y[e]: bf16 rows of stride ldy, packed by row0 if given);
 * dlo / dcnt int16 [rcnt] (device): table row rmin + r collects the deltas i-j in [dlo[r], dlo[r] + dcnt[r]) (relidx is
 * monotone: the inverse of the index vector), dcnt_max = max(dcnt) (host value); klen as in fbl_disent_attn_bwd_ds (only the written corner of X is read);
 * out fp32 [E, nh, rcnt, 64], fully written (no accumulation, no workspace, bit-reproducible).  When this entry point is used
 * fbl_disent_attn_bwd_shear may be called with GT = NULL (no G^T is written).
 * ref: autograd of model/deberta.py:870-918 (c2p / p2c gathers) and :847-853 (the position projections' inputs). */
int fbl_attn_pos_grad(int neg, const void* const* X, const void* const* Y, int64_t ldy, const int16_t* dlo, const int16_t* dcnt,
                      int dcnt_max, const int32_t* klen, const int32_t* row0, float* out, int E, int B, int S, int Sp, int nh,
                      int rcnt, void* stream);
/* mask[b*(Sp/64) + j] (uint32): which 128-row tiles of the gt_rcnt rows of G^T (row 0 = table row gt_rmin) the 64-row k-step j of
 * sample b can touch (neg as in fbl_disent_attn_bwd_shear; klen optional).  A function of the lengths and the relative-index map only:
 * computed once per backward pass and handed to every shear launch (gt_tilemask: rows outside the marked tiles are not written)
 * and to the position-table products (fbl_gemm_bf16_nt kskip_tilemask: not read).  ref: the index arithmetic of
 * model/deberta.py:870-918 (c2p / p2c gather ranges). */
int fbl_gt_tilemask(const int16_t* relidx, const int32_t* klen, int B, int S, int Sp, int span2, int neg, int gt_rmin, int gt_rcnt,
                    uint32_t* mask, void* stream);
/* The preparation of one attention backward as ONE launch: QT / KT = fbl_head_transpose of q / k (head-major
 * [nh,64,B,Sp]), PQT / PKT = the same of the position projections ([nh,64,span2]), Dv = fbl_attn_rowdot(dO, O), and the
 * position tables EXPANDED by the relative-index map for the fused key-major pass (fbl_disent_attn_bwd_dspk):
 *   PQX[h][t][d] = pq[relidx[clamp(t - Sp + S - 1, 0, 2S-2)]][h*64 + d],  t in [0, 2 Sp)  (t - Sp = delta = i - j),
 * bf16 [nh,2*Sp,64]; PKX the same of pk.  Every output but Dv is optional (NULL: not produced); PQX / PKX need relidx.
 * ref: transpose_for_scores model/deberta.py:712-715 (position-contiguous operand copies), XSoftmax.backward :134-138 (D),
 * the c2p / p2c gathers :870-918 (the index map the expansion applies once per table instead of once per score). */
int fbl_attn_bwd_prep(const void* q, const void* k, int64_t ldq, const void* pq, const void* pk, int64_t ldp, const void* dO,
                      const void* O, int64_t ldo, void* QT, void* KT, void* PQT, void* PKT, float* Dv, const int16_t* relidx,
                      void* PQX, void* PKX, int B, int S, int Sp, int nh, int span2, const int32_t* row0, void* stream);
int fbl_disent_attn_bwd_ds(const void* q, const void* k, const void* v, int64_t ldq, const void* dO, int64_t ldo,
                           const void* pk, const void* pq, int64_t ldp, const int16_t* relidx, const int32_t* mask, const int32_t* klen,
                           const int32_t* border, const float* lse, const float* Dv,
                           float scale, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* dV, int64_t lddv, void* dS, void* dST, int B,
                           int S, int Sp, int nh, int span2, int lin_span, const int32_t* row0, void* stream);
/* fbl_disent_attn_bwd_ds without the recomputation: P and the dropout mask come from the psave / msave a training forward left
 * (see fbl_disent_attn_fwd; p_drop gives the scale of the kept pairs, the seed arguments are not read); same outputs (dV, dS,
 * dS^T), q / k / position tables not needed.
 * ref: autograd of model/deberta.py:789-818 (softmax, dropout, context), XSoftmax.backward :134-138, XDropout.backward :185-190. */
int fbl_disent_attn_bwd_dsp(const void* psave, const float* msave, const void* v, int64_t ldv, const void* dO, int64_t ldo,
                            const int32_t* klen, const int32_t* border, const float* lse, const float* Dv, float scale,
                            float p_drop, uint64_t seed, const uint64_t* seed_dev, void* dV, int64_t lddv, void* dS, void* dST,
                            int B, int S, int Sp, int nh, const int32_t* row0, void* stream);
int fbl_disent_attn_bwd_shear(int neg, const void* X, const void* YT, int64_t y_sh, int64_t y_sb, int64_t y_sd,
                              const void* PT, const int16_t* relidx, const int32_t* klen, const int32_t* border,
                              void* out, int64_t ldout,
                              void* GT, int gt_rmin, int gt_rcnt, int lin_span, int B, int S, int Sp, int nh,
                              int span2, const int32_t* row0, const uint32_t* gt_tilemask, void* stream);
/* The query-major half in Toeplitz form: dQ = dS.K + G1.PK (what fbl_disent_attn_bwd_shear(neg = 0) computes) from the dS of
 * fbl_disent_attn_bwd_ds / _dsp / _dspk, k bf16 rows, and pkx = the PKX of fbl_attn_bwd_prep -- no index table, no scatter, no
 * transposed copies of K / PK.  klen / border / row0 (packed rows of k and dQ) as above.
 * ref: autograd of model/deberta.py:756-765 (QK^T) and of the c2p term :870-894. */
int fbl_disent_attn_bwd_dq(const void* dS, const void* k, int64_t ldk, const void* pkx, const int32_t* klen, const int32_t* border,
                           void* dQ, int64_t lddq, int B, int S, int Sp, int nh, const int32_t* row0, void* stream);
/* fbl_disent_attn_bwd_dsp + the key-major shear pass in one kernel: besides dV, dS and dS^T it forms
 *   dK = dS^T.Q + G2.PQ  (what fbl_disent_attn_bwd_shear(neg = 1) computes from dS^T) -- the second term as a Toeplitz product
 * against pqx = the PQX of fbl_attn_bwd_prep (no index table, no scatter); q bf16 rows like v.  dS^T is not read back.
 * ref: autograd of model/deberta.py:789-818 and of the p2c term :896-918. */
int fbl_disent_attn_bwd_dspk(const void* psave, const float* msave, const void* q, int64_t ldq, const void* v, int64_t ldv,
                             const void* dO, int64_t ldo, const void* pqx, const int32_t* klen, const int32_t* border,
                             const float* lse, const float* Dv, float scale, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                             void* dK, int64_t lddk, void* dV, int64_t lddv, void* dS, void* dST, int B, int S, int Sp, int nh,
                             const int32_t* row0, void* stream);

/* Cross entropy over rows with label != -100 (mean reduction).  logits fp32 [N, ldv], labels int64 [N].
 * loss_sum_cnt[0] += sum of row losses, [1] += count (a fixed-order fold: reproducible bit for bit); row_lse [N] fp32 out.
 * ref: model/deberta.py:1483-1488. */
int fbl_ce_fwd(const float* logits, int64_t ldv, const int64_t* labels, int N, int V, float* row_lse,
               float* loss_sum_cnt, void* stream);
/* dlogits_bf16[r, :] = (softmax(logits[rows[r]]) - onehot) * gscale * (gscale_dev ? gscale_dev[0] : 1) / count,
 * zero-padded to Vp columns; rows int32 [R] = indices of labelled rows.  gscale_dev (device scalar, may be NULL) lets the
 * incoming loss gradient stay on the GPU: no host synchronisation at the start of the backward. */
int fbl_ce_bwd_rows(const float* logits, int64_t ldv, const int64_t* labels, const int32_t* rows, int R, int V,
                    int Vp, const float* row_lse, const float* loss_sum_cnt, float gscale, const float* gscale_dev,
                    void* dlogits_bf16, void* stream);

/* gathers rows: out_bf16[r, :] = in_bf16[rows[r], :] ; scatter-add fp32: out[rows[r], :] += in[r, :] */
int fbl_gather_rows_bf16(const void* in, int64_t ld, const int32_t* rows, int R, int cols, void* out, void* stream);
int fbl_scatter_rows_f32(const float* in, const int32_t* rows, int R, int cols, float* out, int64_t ld, void* stream);

/* Fused multi-tensor Adam over ONE flat fp32 buffer (all trainable params are views into it) with the global-norm
 * clip folded in: g *= min(1, max_norm/(norm+1e-6)) where norm = sqrt(sumsq[0]).
 * ref: main.py:82-84 (clip_grad_norm_ + torch.optim.Adam.step, betas (0.9,0.95), eps 1e-8, wd 0). */
int64_t fbl_sumsq_ws_floats(void);
/* out[0] += sum x^2, reproducible bit for bit (per-block partials in ws[>= fbl_sumsq_ws_floats()], folded in index order) */
int fbl_sumsq(const float* x, int64_t n, float* out_sumsq, float* ws, void* stream);
int fbl_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, const float* sumsq, float max_norm, float grad_scale,
                  void* stream);

/* elementwise helpers */
int fbl_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* counter-based dropout (element i keyed by seed): fp32 in -> fp32 and/or bf16 out (in-place allowed);
 * bf16 in place.  ref: StableDropout / XDropout model/deberta.py:171-217, nn.Dropout model/adapter.py:41. */
int fbl_dropout_f32(const float* in, float p_drop, uint64_t seed, const uint64_t* seed_dev, float* out_f32, void* out_bf16, int64_t n,
                    void* stream);
int fbl_dropout_bf16(void* inout_bf16, float p_drop, uint64_t seed, const uint64_t* seed_dev, int64_t n, void* stream);
/* out[i] = sum over the n_slices slices s of dropout_{seeds[s]}(x[s*n + i]) (element i of every slice keyed by (seeds[s], key0 + i):
 * the keys of fbl_dropout_f32 on a tensor of which the slice is the part starting at element key0; p_drop == 0: plain sum),
 * slices added in index order.  `seeds` is a HOST array of n_slices <=
 * FBL_DROPSUM_MAX_SLICES values (copied into the launch).  The gradients of the shared relative-position table of all layer
 * executions, each through the mask its forward drew, folded in one pass.  ref: autograd of model/deberta.py:779 (pos_dropout)
 * summed over the 24 + 2 executions that share `rel_embeddings` (:507-575, :1382-1412). */
#define FBL_DROPSUM_MAX_SLICES 64
/* p[0 .. bytes) = 0 on `stream`: the zero fills of the step (gradient accumulators, split-K targets, the loss accumulator).  A
 * kernel launch like every other entry point (16-byte stores, byte stores for an unaligned head / tail) and NOT hipMemsetAsync:
 * a captured step then consists of kernel nodes only, ordered like eager launches (the memset-node version produced non-finite
 * gradients in about one replay of 500). */
int fbl_zero(void* p, int64_t bytes, void* stream);
/* dst[e][r][h*64 + c] (bf16, row stride ld_dst elements, ld_dst >= nh*64) = src[e][h][r][c] (fp32, contiguous [E][nh][rows][64]):
 * per-head results of a strided-batch GEMM -> the [rows, heads*64] operand of the next one (position-table gradients,
 * autograd of model/deberta.py:847-853).  16-byte aligned pointers, ld_dst % 8 == 0. */
int fbl_heads_to_rows_bf16(const float* src, void* dst_bf16, int E, int nh, int rows, int64_t ld_dst, void* stream);
int fbl_dropout_sum_f32(const float* x, int64_t n, int64_t key0, int n_slices, const uint64_t* seeds, float p_drop,
                        const uint64_t* seed_dev, float* out_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif
