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

#define DESTA_ABI_VERSION 1

int desta_abi_version(void);
const char* desta_last_error(void);

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
    int act;                           /* 0 = none, 1 = GELU(erf) */
    int out_f32;                       /* C dtype: 0 = bf16, 1 = fp32 */
    void* preact;                      /* optional bf16 [M,N] */
    int64_t ldp, stride_p;
    float alpha;
} desta_gemm_desc;
int desta_gemm_bf16_nt(const desta_gemm_desc* d, void* stream);

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
 *   units    int32 [n_units][4]   : tensor index, batch index, first row, number of rows (<= 64)
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
} desta_opt_plan;
size_t desta_adafactor_workspace_floats(int n_units, int n_vec, int64_t sum_rows, int64_t sum_cols,
                                        int64_t colpart_floats);
int desta_clip_adafactor_step(const desta_opt_plan* plan, float* params, const float* grads, float* state,
                              float* workspace, float lr, float beta2t, float eps1, float clip_threshold,
                              float max_grad_norm, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whisper log-mel front end: wave [batch, n_samples] f32 (row stride wave_stride) ->
 * out [batch, n_mels, 3000] f32.  Replaces WhisperFeatureExtractor.__call__ at
 * desta/trainer/data/simple_dataset.py:239-243 and modeling_desta25.py:1570
 * (TF:models/whisper/feature_extraction_whisper.py:135-168, TF:audio_utils.py:638-729).
 * `tables` = device copy of the buffer desta_logmel_fill_tables() writes on the host
 * (hann window | cos | sin | slaney filter bank [201][n_mels]); workspace: batch*94 floats. */
size_t desta_logmel_table_floats(int n_mels);
size_t desta_logmel_workspace_floats(int batch);
int desta_logmel_fill_tables(int n_mels, float* host_out);
int desta_logmel_f32(const float* wave, int batch, int n_samples, int64_t wave_stride, const float* tables,
                     int n_mels, float* out, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DESTA_HIP_H */
