/* libdesta_hip.so — C ABI of the MI355X-native DeSTA2.5-Audio training hot path.
 *
 * The reference (voidful/DeSTA2.5-Audio) has no FFI boundary of its own: every FLOP of
 * `DeSTA25AudioModel.forward` + HF-Trainer backward/clip/Adafactor runs inside torch / transformers
 * calls.  This header is the boundary the build defines underneath the reference's Python surface
 * (SURVEY.md §8b): each entry point names the reference call site(s) (file:line, `TF:` =
 * transformers 5.15) whose arithmetic it replaces.  `INTEGRATION.md` shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, raw DEVICE pointers + explicit sizes; no torch / HIP types in signatures;
 *     `stream` is a `hipStream_t` passed as `void*` (0 = default stream);
 *   - every call only ENQUEUES work on `stream` (no allocation, no host sync: graph-capturable);
 *   - return value: 0 = ok, <0 = error (DESTA_EINVAL -1, DESTA_ELAUNCH -2); nothing throws across
 *     the ABI; `desta_last_error()` returns a thread-local message for the last failure;
 *   - state: the data path has none besides what the caller passes in (buffers, descriptors, plans, stream).  The only
 *     process-wide variables are (a) the thread-local error text, (b) TUNING / DIAGNOSTIC switches for A/B runs
 *     (`desta_gemm_set_option`, `desta_gemm_force_variant`, `desta_attention_set_concurrent_bwd`,
 *     `desta_gemm_last_kernel`): defaults are what every test and benchmark runs, a data path never needs to touch them,
 *     and they are not synchronised (set them before launching work from several threads, or not at all), and (c) one
 *     internal side stream + two events per DEVICE for the attention backward's dQ / dK,dV fork (created on first use,
 *     joined before the call returns to the caller's stream);
 *   - caller owns every buffer including workspaces (`*_workspace_bytes` helpers say how much);
 *   - bf16 tensors are `uint16_t` bit patterns; "rows" are always the slow index, row-major.
 */
#ifndef DESTA_HIP_H
#define DESTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DESTA_ABI_VERSION 7

int desta_abi_version(void);
/* sizeof of the descriptor structs as this library was compiled (0 = desta_gemm_desc, 1 = desta_attn_desc,
 * 2 = desta_opt_plan): a binding checks its own struct layouts against these before the first call. */
size_t desta_sizeof_desc(int which);
const char* desta_last_error(void);

/* Optional per-device context.  No data-path call takes it: the path is stateless (see "state" above).  What
 * it owns is item (c) of that list, which the library otherwise makes at first use and keeps until the process ends:
 * `desta_create` checks that `device` is a gfx950 part (the only code objects in the library), makes it current and creates
 * the internal stream / events up front (so a later hipGraph capture never meets a stream creation); `desta_destroy`
 * drains and releases them.  `desta_handle_last_error` = `desta_last_error()` of the calling thread, copied into the handle. */
typedef struct desta_context* desta_handle;
int desta_create(int device, desta_handle* out);
int desta_destroy(desta_handle h);
int desta_handle_info(desta_handle h, int* device, int* compute_units, char* arch, size_t arch_bytes);
const char* desta_handle_last_error(desta_handle h);

/* ------------------------------------------------------------------------------------------
 * Dense contraction  C[M,N] = epi(alpha * A[M,K] · B[N,K]^T)   (bf16 in, fp32 accumulate, MFMA)
 *   epi(v) = act(v + bias[n]) + residual[m,n];  optional copy of (v + bias) before act -> preact.
 *   `batch` > 1 runs independent problems offset by the stride_* fields (stride 0 = shared).
 *   Rows of A may overlap (lda < K): used for the zero-copy im2col of the Whisper conv stem.
 * Replaces every nn.Linear / nn.Conv1d matmul (+ bias, GELU, residual add) on the path:
 *   modeling_desta25.py:563-564 (conv1/conv2+gelu), :606 (proj);
 *   TF:models/whisper/modeling_whisper.py:279-330,379-413; TF:models/bert/modeling_bert.py:354-416;
 *   TF:models/llama/modeling_llama.py:163-176,230-281,480 (q/k/v/o, gate/up/down, lm_head);
 *   and their autograd backward (dX: B = transposed weight copy; dW: A = dY^T, B = X^T).
 * Requirements: K % 64 == 0, N % 4 == 0, lda/ldb % 8 == 0, 16-byte aligned bases. */
typedef struct desta_gemm_desc {
    const void* A; const void* B; void* C;
    int M, N, K, batch;
    int64_t lda, ldb, ldc;
    int64_t stride_a, stride_b, stride_c;
    const float* bias;                 /* [N] fp32 or NULL */
    const void* residual;              /* [M,N] bf16 or fp32, or NULL */
    int64_t ldr, stride_r;
    int residual_f32;
    int act;                           /* 0 = none, 1 = GELU(erf), 2 = SwiGLU fwd, 3 = SwiGLU bwd (see aux),   */
                                       /* 4 = decode SwiGLU (M <= 16): B = [2N,K] gate rows then up rows,      */
                                       /*     C[M,N] (bf16) = silu(gate) * up                                   */
    int out_f32;                       /* C dtype: 0 = bf16, 1 = fp32 */
    void* preact;                      /* optional bf16 [M,N] */
    int64_t ldp, stride_p;
    float alpha;
    void* aux;                         /* act 2 / act 3: fused SwiGLU over the BLOCKED gate|up layout (ABI 6): 64-column  */
    int64_t ld_aux;                    /*   block b of the projection = gate_{32b..32b+31} | up_{32b..32b+31} (weight rows    */
                                       /*   permuted once at load; N % 64 == 0).  act 2 (forward): C[M,N] keeps the bf16      */
                                       /*   projection, aux[M,N/2] (bf16, out) = silu(gate) * up in plain column order.       */
                                       /* act 3 (backward): v = d(act)[M,N] in plain column order, aux[M,2N] (bf16, in) = the */
                                       /*   saved blocked gate|up, C[M,2N] (bf16, ldc = row length) = d(gate|up), blocked.    */
                                       /*   Replaces TF:models/llama/modeling_llama.py:163-176 (LlamaMLP act_fn(gate) * up)   */
                                       /*   and its autograd; same rounding points as desta_swiglu_fwd / _bwd.                */
    float dropout_p;                   /* > 0: inverted dropout of act(acc+bias) BEFORE the residual add, mask */
    uint64_t dropout_seed;             /*   = desta_dropout_mask(seed, m*N + n) (BertSelfOutput/BertOutput, p=0.1) */
    void* workspace;                   /* optional fp32 scratch for the split-K tail (NULL = never split);   */
    size_t workspace_bytes;            /* 64 MiB + 4 KiB covers every shape (<= 256 slabs of 256x256 fp32);   */
                                       /* its LAST 4 KiB are the arrival tickets of the in-kernel K-slice     */
                                       /* reduction: zero when first handed over, self-resetting afterwards;  */
                                       /* one workspace per stream (two concurrent GEMMs must not share it)   */
    int trans_a, trans_b;              /* != 0: the operand is stored transposed, A as [K,M] (lda = row stride), B as   */
                                       /*   [K,N]: autograd's dW = dY^T X (both) and dX = dY W (trans_b) without a        */
                                       /*   materialised transpose; M resp. N must be a multiple of 8                     */
    const float* a_rms_weight;         /* decode path (M <= 16, M*2K <= 65536 B, act 0 or 4): A holds the  */
    float a_rms_eps;                   /*   un-normalised rows; RMSNorm(A; weight, eps) is applied on the fly   */
                                       /*   (LlamaRMSNorm fused into the projection that consumes it). NULL = off */
    /* (ABI 4) rotary embedding in the epilogue of the fused q|k|v projection (apply_rotary_pos_emb,                  */
    /* TF:models/llama/modeling_llama.py:120-160): output columns [0, rope_cols) are rotated in ADJACENT PAIRS          */
    /* (2i, 2i+1) of each rope_head_dim-wide head by the angle of (position rope_pos[m], frequency i):                  */
    /*   y[2i] = x[2i] c - x[2i+1] s,  y[2i+1] = x[2i+1] c + x[2i] s,  (c, s) = rope_cos_sin[(pos * hd/2 + i) * 2 ..].    */
    /* HF pairs column i with i + hd/2 (rotate_half); the caller stores the frozen q / k weight rows of every head in   */
    /* the order 0, hd/2, 1, hd/2 + 1, ... so that the pair is adjacent — q.k is invariant under a permutation of the    */
    /* head dimension applied to both.  The fp32 accumulator is rotated and rounded ONCE (the reference rounds the      */
    /* projection to bf16, rotates, rounds again).  NULL = off.  Needs act 0, no bias / residual / dropout, bf16 output. */
    const float* rope_cos_sin;         /* [positions][rope_head_dim / 2][2] fp32 */
    const int32_t* rope_pos;           /* [M] position of every output row */
    int rope_cols, rope_head_dim;
} desta_gemm_desc;
int desta_gemm_bf16_nt(const desta_gemm_desc* d, void* stream);
/* tuning / tests only: 0 = automatic tile choice, 1 = force 128x128, 2 / 3 / 4 = force the 256x256 kernel with the
 * lockstep / staggered / staggered-persistent four-phase schedule, 6 / 7 = staggered / lockstep two-phase schedule,
 * 8 = two-phase schedule in the persistent tile walk */
int desta_gemm_force_variant(int variant);
/* kernel family the most recent desta_gemm_bf16_nt call launched: 1 = 128x128, 2 = 256x256 (+ split-K fix-up), 3 = skinny
 * (bench.py attributes its HIP-event timings to the dominant kernel with this) */
int desta_gemm_last_kernel(void);
int desta_gemm_set_persistent(int on);   /* 1: the automatic choice uses the persistent kernel when a block owns > 1 item */
/* A/B switches of the automatic choice: option 0 = persistent, 1 = staggered, 2 = skinny (M <= 16) kernel variant
 * (0 auto, else COLS*10 + U: 162 164 322 641), 3 = persistent grid size of the skinny kernel (default 512 = 2 blocks per CU),
 * 4 = two-phase schedule of the 256x256 kernel (default 1; 0 = the four-phase schedule), 5 = K-slices of tail tiles reduced inside
 * the GEMM launch instead of by the fix-up launch (default 0), 6 = four-slot software-pipelined ring form of the 128x128 kernel:
 * 0 never, 1 (default) when the grid leaves one block per CU (<= 256 tiles), 2 always; 10 = the two-phase 256x256 kernel stops its
 * half-tile stream at the last K-tile (default 1; 0 = re-load dead slots as rounds 1-3 did); bit-identical results in every setting.
 * Options 7, 8 (round 3's de-synchronised start) and 9 (round 4's polynomial GELU) are accepted and ignored: both measured equal and
 * their code slowed every tile of the kernel (profiles/r04_gemm_regression_bisect.log). */
int desta_gemm_set_option(int option, int value);

/* ------------------------------------------------------------------------------------------
 * Global-norm clip + Adafactor over a flat fp32 arena.
 * Replaces `clip_grad_norm_(params, max_grad_norm)` (TF:trainer.py:1780-1782) followed by
 * `Adafactor.step` (TF:optimization.py:1203-1294; scale_parameter=False, relative_step=False,
 * beta1=None, eps=(eps1,.), clip_threshold, decay_rate folded into beta2t = 1 - step^decay_rate by
 * the caller).  `params`/`grads` are the arena; tensors >= 2-D are factored over their last two
 * dims (leading dims = `batch`), 1-D tensors keep a full second-moment vector.  All tables are
 * DEVICE arrays built once by the caller:
 *   tensors  int64 [n_tensors][8] : arena offset, batch, rows, cols, row_state_off, col_state_off,
 *                                   first unit, number of units   (offsets in floats, multiples of 4)
 *   units    int32 [n_units][4]   : tensor index, batch index, first row, number of rows (64 on the chunk path; any count for a
 *                                   tensor with ragged rows, ABI 7)
 *   unit_col_off int64 [n_units]  : offset of the unit's column partial sums in the workspace
 *   vecs     int64 [n_vec][3]     : arena offset, length, state offset
 * `state` holds exp_avg_sq_row / exp_avg_sq_col / exp_avg_sq.  workspace[0] = pre-clip global
 * grad norm, workspace[1] = clip coefficient after the call. */
typedef struct desta_opt_plan {
    const int64_t* tensors; const float* tensor_wd; int n_tensors;
    const int32_t* units; const int64_t* unit_col_off; int n_units;
    const int64_t* vecs; const float* vec_wd; int n_vec;
    int64_t sum_rows, sum_cols;      /* total factored row / column state entries */
    int max_batch, max_cols;
    /* ABI 3: work items of the update kernels and of the factor kernel.
     * chunks [n_chunks][4] = (tensor, batch index, first element inside the [rows, cols] matrix, count <= 16384), the chunks of
     * one tensor contiguous, tensors in DESCENDING arena order (the update passes walk the arena backwards, from the end the
     * statistics pass has just streamed through the Infinity Cache); ten_chunks [n_tensors][2] = (first chunk, chunk count);
     * fin [n_fin][3] = (tensor, batch index, part): part 0 = the row factors, k >= 1 = columns [256 (k-1), 256 k);
     * group_bounds: HOST array [n_groups + 1] of chunk indices cutting the chunk list at tensor boundaries into groups of
     * <= 64 MB of gradients (the re-read of a group's gradients is then served by the Infinity Cache). */
    const int32_t* chunks; const int32_t* ten_chunks; int n_chunks, max_chunks_per_tensor;
    const int32_t* fin; int n_fin;
    int64_t colpart_floats;
    int cols_multiple_of_4;          /* every factored tensor WITH CHUNKS has cols % 4 == 0 (16-B row accesses in the chunk kernels) */
    const int32_t* group_bounds; int n_groups;
    /* ABI 7: factored tensors with ragged rows (cols % 4 != 0, e.g. a Conv1d weight [out, in, 5]) carry NO chunks
     * (ten_chunks = (0, 0)) and are updated by the unit-based kernels: ragged_units = HOST array [n_ragged][2] of
     * (first unit, unit count), one entry per such tensor.  n_ragged == 0 with cols_multiple_of_4 == 0 is the ABI <= 6
     * behaviour (EVERY tensor on the unit-based kernels). */
    const int32_t* ragged_units; int n_ragged;
} desta_opt_plan;
size_t desta_adafactor_workspace_floats(int n_units, int n_vec, int64_t sum_rows, int64_t sum_cols,
                                        int64_t colpart_floats);      /* ABI <= 2 layout, kept for old callers */
size_t desta_adafactor_workspace_floats_v3(const desta_opt_plan* plan, int64_t colpart_floats);
int desta_clip_adafactor_step(const desta_opt_plan* plan, float* params, const float* grads, float* state,
                              float* workspace, float lr, float beta2t, float eps1, float clip_threshold,
                              float max_grad_norm, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whisper log-mel front end: wave [batch, n_samples] f32 (row stride wave_stride) ->
 * out [batch, n_mels, 3000] f32.  Replaces WhisperFeatureExtractor.__call__ at
 * desta/trainer/data/simple_dataset.py:239-243 and modeling_desta25.py:1570
 * (TF:models/whisper/feature_extraction_whisper.py:135-168, TF:audio_utils.py:638-729).
 * `tables` = device copy of the buffer desta_logmel_fill_tables() writes on the host
 * (hann window | cos | sin | slaney filter bank [201][n_mels] | non-zero DFT-bin range [n_mels][2] of every filter: an opaque
 * blob of desta_logmel_table_floats(n_mels) floats); workspace: batch*94 floats. */
size_t desta_logmel_table_floats(int n_mels);
size_t desta_logmel_workspace_floats(int batch);
int desta_logmel_fill_tables(int n_mels, float* host_out);
int desta_logmel_f32(const float* wave, int batch, int n_samples, int64_t wave_stride, const float* tables,
                     int n_mels, float* out, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-wise normalisation.  One wave per row, fp32 statistics, 16-byte accesses; cols % 8 == 0.
 * LayerNorm: nn.LayerNorm at TF:models/whisper/modeling_whisper.py:392,402 (eps 1e-5),
 *   TF:models/bert/modeling_bert.py:296,350 (eps 1e-12), modeling_desta25.py:166; x is fp32 or bf16,
 *   outputs bf16 and/or fp32; stats [rows][2] = (mean, rstd) saved for backward.
 * layernorm_bwd: dx (fp32 and/or bf16) and dgamma/dbeta (written, or added when accumulate != 0);
 *   cols <= 4096; workspace from desta_layernorm_bwd_workspace_floats.
 * RMSNorm: LlamaRMSNorm TF:models/llama/modeling_llama.py:53-67 (bf16 in/out, fp32 weight copy);
 *   rmsnorm_bwd returns dx = dres + d(norm) (dres may be NULL), frozen weight (no dweight). */
int desta_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, float eps, int rows,
                        int cols, void* y_bf16, float* y_f32, float* stats, void* stream);
size_t desta_layernorm_bwd_workspace_floats(int rows, int cols);
int desta_layernorm_bwd(const void* dy, int dy_f32, const void* x, int x_f32, const float* gamma,
                        const float* stats, int rows, int cols, float* dx_f32, void* dx_bf16, float* dgamma,
                        float* dbeta, int accumulate, float* workspace, void* stream);
int desta_rmsnorm_fwd(const void* x, const float* weight, float eps, int rows, int cols, void* y, float* rstd,
                      void* stream);
int desta_rmsnorm_bwd(const void* dy, const void* x, const float* weight, const float* rstd, const void* dres,
                      int rows, int cols, void* dx, void* stream);

/* Column sums of a bf16 [rows, cols] matrix (bias gradients of nn.Linear), fixed-order two-stage. */
size_t desta_colsum_workspace_floats(int rows, int cols);
int desta_colsum_bf16(const void* x, int rows, int cols, int64_t ld, float* out, int accumulate,
                      float* workspace, void* stream);

/* Rotary embedding applied IN PLACE to the first (n_q_heads + n_kv_heads) heads of a fused
 * [rows, ld] bf16 q|k|v buffer; position = row % seq (position_ids = arange(S) for every row,
 * TF:models/llama/modeling_llama.py:386-389); cos_sin = fp32 [seq][2][head_dim/2].
 * With q_norm_w/k_norm_w != NULL the Qwen3 per-head RMSNorm (TF:models/qwen3/modeling_qwen3.py:237-257)
 * runs first (forward) / is differentiated (backward, needs the saved pre-norm q|k in pre_norm).
 * backward != 0 applies the transposed rotation to gradients. head_dim 64 or 128.
 * pos_shift (int32 [rows/seq] or NULL): position = max(0, row % seq + pos_shift[row / seq]) — generate() derives
 * position_ids from the attention mask (prompt: -left_pad; decode step: cache_len - left_pad).
 * s_major_batch > 0: the token grid is stored position-major (row = s * s_major_batch + b) instead of batch-major
 * (row = b * seq + s) — the training layout, in which "all positions >= s0" is one contiguous row range. */
int desta_rope(void* buf, int64_t ld, int rows, int seq, int n_q_heads, int n_kv_heads, int head_dim,
               const float* cos_sin, const float* q_norm_w, const float* k_norm_w, float eps,
               const void* pre_norm, int64_t ld_pre, int backward, const int32_t* pos_shift, int s_major_batch, void* stream);

/* SwiGLU on a fused [rows, 2*inter] gate|up buffer (LlamaMLP, TF:models/llama/modeling_llama.py:163-176),
 * GELU'(erf) for the Q-Former FFN backward, and small layout helpers. */
int desta_swiglu_fwd(const void* gate_up, void* act, int64_t rows, int inter, void* stream);
int desta_swiglu_bwd(const void* gate_up, const void* dact, void* dgate_up, int64_t rows, int inter, void* stream);
int desta_gelu_bwd(const void* preact, const void* dact, void* dpre, int64_t n, void* stream);
int desta_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);
int desta_add_f32(float* y, const float* x, int64_t n, void* stream);
/* out[c][r] = in[r][c] as bf16; out row length ld_out >= rows, tail zero-filled (K padding for dW GEMMs) */
int desta_transpose_to_bf16(const void* in, int in_f32, int64_t ld_in, int rows, int cols, void* out,
                            int64_t ld_out, void* stream);
/* mel [B, n_mels, T] fp32 -> channel-last rows [B, T+2, c_pad] bf16 (rows 0 and T+1 stay as the caller
 * zeroed them): the A operand of the zero-copy im2col conv1 GEMM (modeling_desta25.py:563). */
int desta_mel_to_rows(const float* mel, int batch, int n_mels, int frames, int c_pad, void* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding gather + audio splice: out[r] = src_row[r] >= 0 ? table[src_row[r]]
 *                                                           : audio_rows[-(src_row[r]+1)]
 * `src_row` (int32 [rows], built by the caller from input_ids / start positions / transcription
 * ids) restates `embed_tokens(input_ids)` + the slice assignment of cat(audio_features,
 * transcription_embeddings) at modeling_desta25.py:1009-1041.  gather_rows is its backward
 * (rows of dL/d inputs_embeds at the audio positions -> dL/d audio_features). */
int desta_embed_gather(const void* table, const void* audio_rows, const int32_t* src_row, int rows, int hidden,
                       void* out, void* stream);
int desta_gather_rows_bf16(const void* in, const int32_t* idx, int rows, int hidden, void* out, void* stream);

/* ForCausalLMLoss (TF:loss/loss_utils.py:49-71): labels int64 [batch, seq] UNSHIFTED (the shift by one
 * and the -100 padding of the last position happen inside), logits bf16 [batch*seq, ld] upcast to fp32;
 * loss[0] = mean CE over targets != -100.  With write_grad != 0 the logits buffer is overwritten by
 * dloss/dlogits (bf16).  workspace: desta_ce_workspace_floats(batch, seq). */
size_t desta_ce_workspace_floats(int batch, int seq);
int desta_causal_lm_loss(void* logits, int64_t ld, const int64_t* labels, int batch, int seq, int vocab,
                         float* loss, float* workspace, int write_grad, void* stream);

/* Connector tap mix (modeling_desta25.py:600-604): x fp32 [taps][batch*prompt][d], layer_weights
 * fp32 [prompt][taps]; out[b,k,:] = sum_j softmax(layer_weights[k,:])_j x[j,b,k,:]; and its backward. */
int desta_tap_mix_fwd(const float* x, const float* layer_weights, int taps, int batch, int prompt, int d,
                      float* out, void* stream);
int desta_tap_mix_bwd(const float* x, const float* layer_weights, const float* dout, int taps, int batch,
                      int prompt, int d, float* dx, float* dlayer_weights, void* stream);

/* ------------------------------------------------------------------------------------------
 * ORCA hybrid connector / deep injection (ABI 6).  The ORCA branch shares the GEMM,
 * attention and LayerNorm entry points with the qformer_1 path; these are the row-wise pieces it adds
 * (/root/reference/desta/models/modeling_desta25.py, line numbers per entry).  All bf16 streams, fp32 arithmetic.
 *   desta_orca_local_mix      :336-343  out[r,:] = sum_l softmax(layer_weights)[l] x[l,r,:]; x [taps][rows][d], taps <= 8
 *   desta_orca_rope           :22-95, :422-438  rotation of the WHOLE hidden vector (pairs i, i + hidden/2) by the angle
 *                             (t / position_scale) theta^(-i / (hidden/2)); round_cos_sin: cos / sin rounded to bf16 first (bf16 model)
 *   desta_orca_gate_residual  :456-490  hidden[m,:] += sigmoid(gate_hidden[m,:] . gate_w2 + gate_b2[0]) * cross[m,:], in place;
 *                             gate_hidden = GELU(Linear(hidden)) [rows, gate_width] from a GEMM; gate_out (fp32 [rows]) optional
 *   desta_orca_sim_loss       :1174-1198  partials[b * nx + i] = sum_j (xhat_i . yhat_j - [subtract_identity and i == j])^2 over
 *                             L2-normalised rows; y rows picked through y_index[ny] (NULL: 0..ny-1) out of y_rows per batch entry
 *   desta_orca_align          :460-486  out[e] = 1 - cos(mean_t audio[e,t,:], mean_{s0 <= s < s1} hidden[row, s, :]),
 *                             spans[e] = (text row, s0, s1); hidden addressed as row * batch_stride + s * row_stride */
/* Backward pieces (autograd of the modules above; gradients of the trainer's total loss = LM loss + sum of the ORCA losses):
 *   desta_orca_gate_residual_bwd  d_cross[m,:] = gate[m] d_out[m,:];  d_gate_pre[m] = (d_out[m,:] . cross[m,:]) gate[m] (1 - gate[m])
 *   desta_orca_gate_mlp_bwd       d_preact = d_gate_pre (x) w2 * gelu'(preact);  d_w2 = sum_m d_gate_pre[m] gate_hidden[m,:];  d_b2 = sum_m d_gate_pre[m]
 *                                 (workspace: 64 x (gate_width + 1) floats, ABI 7: row slices summed in a fixed order)
 *   desta_orca_align_bwd          d_hidden[row, s, :] += coef * d(1 - cos)/d(mean over the span) / span length   (audio pooled under no_grad, :461-462)
 *   desta_orca_rope_bwd           rotation by the negative angle of the fp32 gradient of the rotated tokens, ADDED to d_first (tokens [0, n_first)
 *                                 of every clip: the global tokens under orca_global_cross_attn) resp. d_rest (the local tokens)
 *   desta_orca_col2im_add         gradient of the Conv1d's im2col view: d_padded[b,t,:] = sum of the d_col windows that cover row t (bf16)
 *   desta_orca_local_mix_bwd      d(local_layer_weights) through the softmax (workspace: 256 x 32 floats; taps <= 32)
 *   desta_orca_sim_loss_bwd       d_x += gradient of coef * sum_ij (xhat_i . yhat_j - [identity])^2 w.r.t. the un-normalised rows of x (fp32, added;
 *                                 rows of x / d_x and of y picked through optional index lists; ny <= 128) */
int desta_orca_local_mix(const void* x, const float* layer_weights, int taps, int64_t rows, int d, void* out, void* stream);
int desta_orca_rope(const void* x, void* y, int batch, int tokens, int hidden, float theta, float position_scale, int round_cos_sin, void* stream);
int desta_orca_gate_residual(void* hidden, int64_t ld_hidden, const void* cross, const void* gate_hidden, const float* gate_w2, const float* gate_b2,
                             int64_t rows, int hidden_size, int gate_width, float* gate_out, void* stream);
int desta_orca_sim_loss(const void* x, const void* y, const int32_t* y_index, int batch, int nx, int ny, int64_t y_rows, int hidden,
                        int subtract_identity, float* partials, void* stream);
int desta_orca_align(const void* audio, int tokens, const void* hidden, int64_t hidden_row_stride, int64_t hidden_batch_stride, int hidden_size,
                     const int32_t* spans, int n_spans, float* out, void* stream);
int desta_orca_gate_residual_bwd(const void* d_out, int64_t ld, const void* cross, const float* gate, int64_t rows, int hidden_size, void* d_cross,
                                 float* d_gate_pre, void* stream);
int desta_orca_gate_mlp_bwd(const float* d_gate_pre, const void* gate_preact, const void* gate_hidden, const float* gate_w2, int64_t rows, int gate_width,
                            void* d_preact, float* d_w2, float* d_b2, float* workspace, void* stream);
int desta_orca_align_bwd(const void* audio, int tokens, const void* hidden, int64_t hidden_row_stride, int64_t hidden_batch_stride, int hidden_size,
                         const int32_t* spans, int n_spans, float coef, void* d_hidden, int64_t d_row_stride, int64_t d_batch_stride, void* stream);
int desta_orca_rope_bwd(const float* d_rotated, int batch, int tokens, int hidden, float theta, float position_scale, int round_cos_sin, int n_first,
                        float* d_first, float* d_rest, void* stream);
int desta_orca_col2im_add(const void* d_col, int batch, int tokens_out, int tokens_padded, int hidden, int kernel, int stride, void* d_padded, void* stream);
int desta_orca_local_mix_bwd(const void* d_out, const void* x, const float* layer_weights, int taps, int64_t rows, int d, float* d_layer_weights,
                             float* workspace, void* stream);
int desta_orca_sim_loss_bwd(const void* x, const int32_t* x_index, int64_t x_rows, const void* y, const int32_t* y_index, int64_t y_rows, int batch, int nx,
                            int ny, int hidden, int subtract_identity, float coef, float* d_x, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flash-style attention, forward and backward (bf16 operands, fp32 softmax, MFMA 32x32x16).
 * Element (b, s, h, d) of a tensor X lives at X + b*x_batch_stride + s*x_row_stride + h*head_dim + d,
 * so q/k/v may be slices of one fused projection buffer.  GQA: kv head = q head / (n_q/n_kv).
 * Mask: key j is visible to query i iff j < seq_k, j >= kv_start[b] (left padding; NULL = 0) and,
 * when causal, j <= i + (seq_k - seq_q).  softmax(scale * q.k).  Rows with no visible key give 0.
 * lse (fp32 [batch][n_q_heads][seq_q], log2 domain) is written by fwd and read by bwd.
 * Replaces: Whisper self-attention TF:models/whisper/modeling_whisper.py:241-357 (q pre-scaled by
 * hd^-0.5 == scale), BERT eager self/cross attention TF:models/bert/modeling_bert.py:100-293,
 * Llama/Qwen3 attention TF:models/llama/modeling_llama.py:179-281 with the mask of
 * `create_causal_mask` (:391), and their autograd backward.  bwd: dK/dV may both be NULL
 * (Whisper states carry no gradient, modeling_desta25.py:594). */
typedef struct desta_attn_desc {
    const void* Q; const void* K; const void* V; void* O;
    const void* dO; void* dQ; void* dK; void* dV;
    float* lse;
    int64_t q_batch_stride, q_row_stride, k_batch_stride, k_row_stride, v_batch_stride, v_row_stride;
    int64_t o_batch_stride, o_row_stride, do_batch_stride, do_row_stride;
    int64_t dq_batch_stride, dq_row_stride, dk_batch_stride, dk_row_stride, dv_batch_stride, dv_row_stride;
    int batch, n_q_heads, n_kv_heads, seq_q, seq_k, head_dim;
    int causal;
    const int32_t* kv_start;
    float scale;
    float dropout_p;                   /* attention-probability dropout (head_dim 64 only), same mask in fwd and bwd */
    uint64_t dropout_seed;
    const float* rope_cos_sin;         /* optional (ABI 4), bwd only: [seq][head_dim / 2][2] fp32 (cos, sin).  Q and K are the rotary-  */
                                       /* embedded projections in the adjacent-pair layout of desta_gemm_desc.rope_*: dQ (row q: position */
                                       /* q) and dK (row k: position k) are rotated back by the transposed rotation before they are       */
                                       /* stored, i.e. they come out as gradients of the projection OUTPUTS.  Needs the 8-wave dQ path   */
                                       /* (seq_q >= 128, no dropout, 16-byte aligned dQ / dK / dV).  NULL = off.                         */
    float* O_f32;                      /* optional (ABI 4): fp32 copy of O written by fwd, indexed with the o_* strides.  bwd then   */
                                       /* takes delta = rowsum(dO * O) from the UNROUNDED output: with delta from the bf16-rounded O */
                                       /* the error of delta is coherent over the keys of a row (dS_err = P * eps_q) and, where the   */
                                       /* softmax is flat over 1500 encoder frames, puts 10-13 % error on the cross-attention query   */
                                       /* weight gradient (eager attention in the reference sums P * dP itself: no such term).        */
    int dkv_transposed;                /* optional (ABI 5), bwd on the one-query-tile path only (head_dim 64, seq_q <= 64, seq_k >= 256, */
    int64_t dkv_t_ld;                  /* seq_k % 4 == 0, no GQA): dK and dV are written TRANSPOSED, as [n_heads * 64][dkv_t_ld] bf16    */
    float* dkv_bias_grad;              /* matrices with element (h * 64 + d, b * seq_k + key) — the operand layout of the K / V          */
                                       /* projection's weight-gradient GEMM (no transpose pass); dk_* / dv_* strides are ignored.       */
                                       /* dkv_bias_grad (optional): fp32 [2][n_heads * 64] = sums over batch and keys of dK | dV (the   */
                                       /* projection's bias gradients), overwritten.                                                    */
} desta_attn_desc;
int desta_attention_fwd(const desta_attn_desc* d, void* stream);
/* floats of `workspace` for desta_attention_bwd: delta [batch][heads][seq_q] and, for seq_q <= 64 (one query tile: the one-pass
 * dQ / dK / dV kernel), up to 4 fp32 dQ partials of 64 x 64 and bias-gradient partials of 2 x 64 per (batch, head), summed in
 * fixed order (no atomics). */
size_t desta_attention_bwd_workspace_floats(int batch, int n_q_heads, int seq_q);
int desta_attention_bwd(const desta_attn_desc* d, float* workspace, void* stream);
/* D = 128 backward: run the dQ kernel on an internal side stream next to dK/dV (fork after delta, join on `stream`);
 * 1 = on (default), 0 = everything on `stream`.  Results are identical either way. */
int desta_attention_set_concurrent_bwd(int on);
/* Process-wide kernel selection switches (A/B measurements; results agree to rounding either way):
 *   which 0: forward for seq_q >= 128 without dropout on the 8-wave kernel (1, default) or the 4-wave kernel (0);
 *   which 1: head_dim 64 non-causal 8-wave forward at two blocks per CU (1) or one (0, default);
 *   which 2: 1 = the 8-wave forward WITH waves 4-7 half a tile behind waves 0-3 (LLM / Whisper shapes only; default 0);
 *   which 3: 1 = backward on round 2's path (separate delta launch, 4-wave dQ kernel on a side stream beside dK / dV) instead
 *            of the 8-wave dQ kernel that computes delta itself (default 0; not available with rope_cos_sin);
 *   which 4: 1 (default) = one query tile (seq_q <= 64, head_dim 64, seq_k >= 256, no GQA, dK / dV requested): dQ, dK, dV in ONE
 *            pass over K / V; 0 = the separate dQ and dK / dV kernels;
 *   which 5: 1 = causal head_dim-128 dK / dV on 64-key blocks of two waves (640 shorter work items instead of 320; measured
 *            SLOWER on the LLM shape, 207 vs 194 us: default 0). */
int desta_attention_set_option(int which, int value);

/* layer_prompts[j].expand(B,-1,-1) for all taps at once (modeling_desta25.py:589): prompts fp32
 * [taps][n], n = prompt_size*d -> rows [(taps*batch)][n] in fp32 and bf16; prompt_grad sums the
 * gradient back over the batch. */
int desta_prompt_expand(const float* prompts, int taps, int batch, int64_t n, float* x_f32, void* x_bf16, void* stream);
int desta_prompt_grad(const float* dx, int taps, int batch, int64_t n, float* dprompts, void* stream);

/* ------------------------------------------------------------------------------------------
 * Counter-based dropout (stateless: element `i` of stream `seed` is kept iff rng32(seed, i) >= p*2^32),
 * used by the GEMM epilogue, the attention kernels (i = ((b*H + h)*seq_q + q)*seq_k + k) and:
 *   desta_dropout_bf16      y[r][c] = keep(r*cols + c) ? x[r][c] / (1-p) : 0   (backward of the epilogue dropout)
 *   desta_dropout_mask_u8   materialises the keep mask (tests / debugging)
 * Replaces nn.Dropout(hidden_dropout_prob) / attention_probs dropout of the Q-Former
 * (TF:models/bert/modeling_bert.py:150-160, 296, 350; BertConfig defaults 0.1, modeling_desta25.py:156). */
int desta_dropout_bf16(const void* x, void* y, int rows, int cols, int64_t ld, float p, uint64_t seed, void* stream);
int desta_dropout_mask_u8(uint64_t seed, int64_t n, float p, uint8_t* out, void* stream);

/* lm_head on target rows only.  `ForCausalLMLoss` (TF:loss/loss_utils.py:49-71) ignores every row whose shifted label is
 * -100, so the logits of those rows (context, audio span, last position: 20 % of the synthetic batch, more on real data)
 * and their zero gradients need not be computed.  desta_target_rows lists the rows of the [batch*seq] grid that carry a
 * target (idx, in order; count[0] = n, count[1] = the first POSITION s that carries a target in any sequence, seq if none:
 * `count` is int32[2] since ABI 5) and lays their targets out as compact_labels[0] = -100, [1 + i] = target of compact
 * row i, [1 + n] = -100 (room for batch*seq + 2 entries): desta_causal_lm_loss(batch = 1, seq = n + 1) on a compact
 * [n + 1, vocab] logits buffer then gives the same loss / gradients as the full grid.  desta_scatter_rows_bf16 is the
 * inverse of desta_gather_rows_bf16 (out[idx[i]] = in[i]). */
int desta_target_rows(const int64_t* labels, int batch, int seq, int32_t* idx, int64_t* compact_labels, int32_t* count,
                      int s_major, void* stream);   /* s_major != 0: idx holds position-major row ids s * batch + b */
int desta_scatter_rows_bf16(const void* in, const int32_t* idx, int rows, int hidden, void* out, void* stream);

/* Greedy decoding helper: out[r] = argmax over the first `cols` entries of bf16 row r (first maximum, two-stage
 * reduction; `workspace` holds desta_argmax_workspace_bytes(rows) bytes).
 * Replaces the argmax of `llm_model.generate(do_sample=False)` (modeling_desta25.py:1419). */
size_t desta_argmax_workspace_bytes(int rows);
/* logits[r, ids[i]] = -inf (bf16 rows of `cols` entries, row stride ld) for every row r: token suppression in front of the argmax
 * (Whisper ASR leg of generate(), modeling_desta25.py:1580-1590 -> TF:generation/logits_process.py SuppressTokensLogitsProcessor). */
int desta_mask_tokens_bf16(void* logits, int64_t ld, int rows, int cols, const int32_t* ids, int n, void* stream);
int desta_argmax_bf16(const void* x, int64_t ld, int rows, int cols, int64_t* out, void* workspace, void* stream);

/* `do_sample=True` step of `llm_model.generate` (modeling_desta25.py:1419-1427: temperature, top_p):
 * TemperatureLogitsWarper -> TopPLogitsWarper -> one multinomial draw per row
 * (TF:generation/logits_process.py, TF:generation/utils.py `_sample`).  Token i is kept iff the softmax mass of all
 * tokens not more probable than i exceeds 1 - top_p; tokens tying with the boundary logit are all kept.  The draw uses
 * the library's counter-based RNG (seed, step, row) — same distribution as torch.multinomial, not the same stream.
 * keep_mask (optional, uint8 [rows, cols]) exports the kept set (tests). */
int desta_sample_top_p_bf16(const void* logits, int64_t ld, int rows, int cols, float temperature, float top_p,
                            uint64_t seed, uint32_t step, int64_t* out, uint8_t* keep_mask, void* stream);

/* desta_rope (forward) on a fused q|k|v projection [rows, ld] that ALSO appends the rotated K heads and the V heads of
 * row (b, s) to a KV cache slab: kv_cache + b*kv_batch_stride + (slot0 + s)*kv_row_stride, K heads then V heads.
 * This is what `DynamicCache.update` does inside `llm_model.generate` (modeling_desta25.py:1419) for the prompt
 * (seq = prompt length, slot0 = 0) and for every decode step (seq = 1, slot0 = cache length). */
int desta_rope_kv_append(void* qkv, int64_t ld, int rows, int seq, int n_q_heads, int n_kv_heads, int head_dim,
                         const float* cos_sin, const float* q_norm_w, const float* k_norm_w, float eps,
                         const int32_t* pos_shift, void* kv_cache, int64_t kv_batch_stride, int64_t kv_row_stride,
                         int slot0, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange: the MEAN over ranks of the flat fp32 gradient arena as one
 * RCCL all-reduce (ncclAvg) on `stream`, in place — what DDP's bucketed reducer does for the reference (accelerate
 * `accelerator.py:1892`; loss semantics: each rank's loss is its own token mean, gradients are averaged over ranks, hazard H8).
 * For hosts without torch.distributed (the Python host layer issues the same collective through torch's RCCL binding):
 *   rank 0: desta_comm_get_unique_id(&id); hand the 128 bytes to every rank (any channel);
 *   every rank (its device current): desta_comm_create(&comm, world, rank, &id);
 *   every step: desta_allreduce_grads(comm, arena_grads, n, stream) between backward and desta_clip_adafactor_step;
 *   desta_comm_destroy(comm).
 * RCCL is loaded on first use (dlopen): the other entry points need no RCCL on the box.  One process per GPU. */
typedef struct { char internal[128]; } desta_comm_unique_id;        /* = ncclUniqueId */
typedef void* desta_comm;                                            /* = ncclComm_t */
int desta_comm_get_unique_id(desta_comm_unique_id* id);
int desta_comm_create(desta_comm* comm, int world_size, int rank, const desta_comm_unique_id* id);
int desta_allreduce_grads(desta_comm comm, float* grads, int64_t n, void* stream);
int desta_comm_destroy(desta_comm comm);

#ifdef __cplusplus
}
#endif
#endif /* DESTA_HIP_H */
