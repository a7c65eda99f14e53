// Embedding gather + audio splice, fused shifted cross-entropy (forward + dlogits), and the
// connector's tap mix — the glue ops of DeSTA25AudioModel around the three transformer stacks (gfx950).
//   embed/splice ... modeling_desta25.py:1009-1041 (`embed_tokens(input_ids)` then the slice-assign of
//                    cat(audio_features, transcription_embeddings) at each start position)
//   CE ............. TF:loss/loss_utils.py:49-71 (`ForCausalLMLoss`: fp32 upcast of the bf16 logits,
//                    labels shifted left by one, mean over labels != -100) and its autograd backward
//   tap mix ........ modeling_desta25.py:600-604 (softmax(layer_weights) weighted sum over the 4 taps)
#include "common.h"
#include "desta_hip.h"

namespace {

// out[r] = src[r] >= 0 ? table[src[r]] : audio[-(src[r]+1)]     (rows of h bf16, h % 8 == 0)
__global__ __launch_bounds__(256) void embed_gather_k(const bf16_t* __restrict__ table, const bf16_t* __restrict__ audio,
                                                      const int* __restrict__ src, int rows, int h,
                                                      bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int s = src[row];
    const bf16_t* in = s >= 0 ? table + (long)s * h : audio + (long)(-(s + 1)) * h;
    for (int c = lane * 8; c < h; c += 512) *(u16x8*)(out + (long)row * h + c) = *(const u16x8*)(in + c);
}

__global__ __launch_bounds__(256) void gather_rows_k(const bf16_t* __restrict__ in, const int* __restrict__ idx, int rows,
                                                     int h, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* src = in + (long)idx[row] * h;
    for (int c = lane * 8; c < h; c += 512) *(u16x8*)(out + (long)row * h + c) = *(const u16x8*)(src + c);
}

// out[idx[i]] = in[i]: rows of a compact [rows, h] matrix back into their places (the inverse of gather_rows)
__global__ __launch_bounds__(256) void scatter_rows_k(const bf16_t* __restrict__ in, const int* __restrict__ idx, int rows,
                                                      int h, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* dst = out + (long)idx[row] * h;
    for (int c = lane * 8; c < h; c += 512) *(u16x8*)(dst + c) = *(const u16x8*)(in + (long)row * h + c);
}

// Rows of the [B*S] token grid whose SHIFTED label is a real target (ForCausalLMLoss: row (b,s) predicts labels[b,s+1]):
// idx[0..n) = those rows in order, lab[0] = -100, lab[1 + i] = target of compact row i, lab[1 + n] = -100, count = n.
// One block; fixed-order scan (deterministic).  lab is laid out so that `desta_causal_lm_loss(batch = 1, seq = n + 1)`
// on a compact [n + 1, V] logits buffer reproduces the full-grid loss and gradients of exactly these rows.
__global__ __launch_bounds__(1024) void target_rows_k(const long* __restrict__ labels, int B, int S, int* __restrict__ idx,
                                                      long* __restrict__ lab, int* __restrict__ count, int s_major) {
    __shared__ int part[1024];
    const int M = B * S, tid = threadIdx.x;
    const int per = (M + 1023) / 1024, m0 = tid * per, m1 = min(M, m0 + per);
    __shared__ int first[1024];
    int n = 0, f = S;                                           // f: first POSITION (m % S) among this thread's target rows
    for (int m = m0; m < m1; ++m)
        if ((m % S) + 1 < S && labels[m + 1] != -100) { ++n; f = min(f, m % S); }
    part[tid] = n;
    first[tid] = f;
    __syncthreads();
    if (tid == 0) {
        int run = 0, fmin = S;
        for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; fmin = min(fmin, first[i]); }
        count[0] = run;
        count[1] = fmin;                                        // rows of positions < fmin carry no target in any sequence
        lab[0] = -100;
        lab[1 + run] = -100;
    }
    __syncthreads();
    int pos = part[tid];
    for (int m = m0; m < m1; ++m)
        if ((m % S) + 1 < S && labels[m + 1] != -100) {
            idx[pos] = s_major ? (m % S) * B + m / S : m;          // row id in the position-major / batch-major token grid
            lab[1 + pos] = labels[m + 1];
            ++pos;
        }
}

// scal[0] = number of valid (shifted) targets, scal[1] = 1/scal[0]
__global__ __launch_bounds__(256) void ce_count_k(const long* __restrict__ labels, int B, int S, float* __restrict__ scal) {
    __shared__ float red[4];
    float n = 0.f;
    for (long i = threadIdx.x; i < (long)B * S; i += 256) {
        const int s = (int)(i % S);
        if (s + 1 < S && labels[i + 1] != -100) n += 1.f;
    }
    n = block_sum<256>(n, red);
    if (threadIdx.x == 0) { scal[0] = n; scal[1] = n > 0.f ? 1.0f / n : 0.f; }
}

// one block per row m = (b, s); logits row is overwritten with dlogits (bf16)
__global__ __launch_bounds__(256) void ce_row_k(bf16_t* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                int S, int V, const float* __restrict__ scal,
                                                float* __restrict__ row_loss, int write_grad) {
    __shared__ float red[4];
    const int m = blockIdx.x, s = m % S;
    const long tgt = (s + 1 < S) ? labels[m + 1] : -100;
    bf16_t* row = logits + (long)m * ld;
    const int nv8 = V / 8;
    if (tgt == -100) {
        if (threadIdx.x == 0) row_loss[m] = 0.f;
        if (write_grad) {
            const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = threadIdx.x; i < nv8; i += 256) *(u16x8*)(row + i * 8) = z;
            for (int i = nv8 * 8 + threadIdx.x; i < V; i += 256) row[i] = 0;
        }
        return;
    }
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < nv8; i += 256) {
        const u16x8 v = *(const u16x8*)(row + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, bf2f(v[e]));
    }
    for (int i = nv8 * 8 + threadIdx.x; i < V; i += 256) mx = fmaxf(mx, bf2f(row[i]));
    mx = block_max<256>(mx, red);
    float sum = 0.f;
    for (int i = threadIdx.x; i < nv8; i += 256) {
        const u16x8 v = *(const u16x8*)(row + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf(bf2f(v[e]) - mx);
    }
    for (int i = nv8 * 8 + threadIdx.x; i < V; i += 256) sum += __expf(bf2f(row[i]) - mx);
    sum = block_sum<256>(sum, red);
    const float lse = mx + __logf(sum);
    if (threadIdx.x == 0) row_loss[m] = lse - bf2f(row[tgt]);
    if (!write_grad) return;
    __syncthreads();                                   // row[tgt] has been read before anyone overwrites it
    const float inv_n = scal[1];
    for (int i = threadIdx.x; i < nv8; i += 256) {
        const u16x8 v = *(const u16x8*)(row + i * 8);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g = __expf(bf2f(v[e]) - lse);
            if ((long)(i * 8 + e) == tgt) g -= 1.0f;
            o[e] = f2bf(g * inv_n);
        }
        *(u16x8*)(row + i * 8) = o;
    }
    for (int i = nv8 * 8 + threadIdx.x; i < V; i += 256) {
        float g = __expf(bf2f(row[i]) - lse);
        if ((long)i == tgt) g -= 1.0f;
        row[i] = f2bf(g * inv_n);
    }
}

// Same arithmetic with the WHOLE row held in registers (1024 threads x NV 16-byte chunks): the logits are read
// once and the gradient written once (2.6 GB per step at V = 128256) instead of three reads + one write — the
// 256-KiB rows of ~2000 concurrently resident blocks do not stay in L2 / the Infinity Cache between passes.
template <int NV>
__global__ __launch_bounds__(1024) void ce_row_reg_k(bf16_t* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                     int S, int V, const float* __restrict__ scal,
                                                     float* __restrict__ row_loss, int write_grad) {
    __shared__ float red[16];
    const int m = blockIdx.x, s = m % S, tid = threadIdx.x;
    const long tgt = (s + 1 < S) ? labels[m + 1] : -100;
    bf16_t* row = logits + (long)m * ld;
    const int nv8 = V / 8;
    if (tgt == -100) {
        if (tid == 0) row_loss[m] = 0.f;
        if (write_grad) {
            const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = tid; i < nv8; i += 1024) *(u16x8*)(row + i * 8) = z;
        }
        return;
    }
    u16x8 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid + 1024 * j;
        if (i < nv8) v[j] = *(const u16x8*)(row + i * 8);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (tid + 1024 * j < nv8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, bf2f(v[j][e]));
        }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (tid + 1024 * j < nv8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += __expf(bf2f(v[j][e]) - mx);
        }
    sum = wave_sum(sum);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) sum += red[w];
    const float lse = mx + __logf(sum);
    if (tid == 0) row_loss[m] = lse - bf2f(row[tgt]);
    if (!write_grad) return;
    __syncthreads();                                   // row[tgt] has been read before anyone overwrites it
    const float inv_n = scal[1];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid + 1024 * j;
        if (i < nv8) {
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float g = __expf(bf2f(v[j][e]) - lse);
                if ((long)(i * 8 + e) == tgt) g -= 1.0f;
                o[e] = f2bf(g * inv_n);
            }
            *(u16x8*)(row + i * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void ce_finish_k(const float* __restrict__ row_loss, int M, const float* __restrict__ scal,
                                                   float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) s += row_loss[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) loss[0] = s * scal[1];
}

constexpr int MAXT = 32;           // tapped encoder layers (4 in the shipped configs; 32 = every layer of whisper-large with orca_use_all_layers)

// out[n][c] = sum_j softmax(lw[k])[j] * x[j][n][c],  n = b*K + k;  x: [taps][N][d] f32
__global__ __launch_bounds__(256) void mix_fwd_k(const float* __restrict__ x, const float* __restrict__ lw, int taps, int N,
                                                 int K, int d, float* __restrict__ out) {
    const int n = blockIdx.x, k = n % K;
    float sm[MAXT];
    float mx = -INFINITY, den = 0.f;
    for (int j = 0; j < taps; ++j) mx = fmaxf(mx, lw[k * taps + j]);
    for (int j = 0; j < taps; ++j) { sm[j] = __expf(lw[k * taps + j] - mx); den += sm[j]; }
    for (int c = threadIdx.x * 4; c < d; c += 1024) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < taps; ++j) {
            const float wj = sm[j] / den;
            const float4 v = *(const float4*)(x + ((long)j * N + n) * d + c);
            acc.x += wj * v.x; acc.y += wj * v.y; acc.z += wj * v.z; acc.w += wj * v.w;
        }
        *(float4*)(out + (long)n * d + c) = acc;
    }
}

// one block per query slot k: dx[j][b*K+k][:] = sm[k][j] * dout[b*K+k][:];  dlw[k][:] = softmax'(...)
__global__ __launch_bounds__(256) void mix_bwd_k(const float* __restrict__ x, const float* __restrict__ lw,
                                                 const float* __restrict__ dout, int taps, int Bn, int K, int d,
                                                 float* __restrict__ dx, float* __restrict__ dlw) {
    __shared__ float red[4];
    const int k = blockIdx.x, N = Bn * K;
    float sm[MAXT], ds[MAXT];
    float mx = -INFINITY, den = 0.f;
    for (int j = 0; j < taps; ++j) mx = fmaxf(mx, lw[k * taps + j]);
    for (int j = 0; j < taps; ++j) { sm[j] = __expf(lw[k * taps + j] - mx); den += sm[j]; }
    for (int j = 0; j < taps; ++j) { sm[j] /= den; ds[j] = 0.f; }
    for (int b = 0; b < Bn; ++b) {
        const long n = (long)b * K + k;
        for (int c = threadIdx.x * 4; c < d; c += 1024) {
            const float4 g = *(const float4*)(dout + n * d + c);
            for (int j = 0; j < taps; ++j) {
                const float4 v = *(const float4*)(x + ((long)j * N + n) * d + c);
                ds[j] += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
                *(float4*)(dx + ((long)j * N + n) * d + c) = make_float4(sm[j] * g.x, sm[j] * g.y, sm[j] * g.z, sm[j] * g.w);
            }
        }
    }
    float dot = 0.f;
    for (int j = 0; j < taps; ++j) { ds[j] = block_sum<256>(ds[j], red); dot += sm[j] * ds[j]; }
    if (threadIdx.x == 0)
        for (int j = 0; j < taps; ++j) dlw[k * taps + j] = sm[j] * (ds[j] - dot);
}

// x32/x16[(j*B + b)][:] = prompts[j][:]   (rows of n = K*d floats; layer_prompts[j].expand(B,-1,-1))
__global__ __launch_bounds__(256) void prompt_expand_k(const float* __restrict__ prompts, int taps, int Bn, long n,
                                                       float* __restrict__ x32, bf16_t* __restrict__ x16) {
    const long n4 = n / 4, total = (long)taps * Bn * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long jb = i / n4, c = (i % n4) * 4;
        const int j = (int)(jb / Bn);
        const float4 v = *(const float4*)(prompts + (long)j * n + c);
        *(float4*)(x32 + jb * n + c) = v;
        u16x4 o;
        o[0] = f2bf(v.x); o[1] = f2bf(v.y); o[2] = f2bf(v.z); o[3] = f2bf(v.w);
        *(u16x4*)(x16 + jb * n + c) = o;
    }
}
// dprompts[j][:] = sum_b dx[(j*B + b)][:]   (fixed order)
__global__ __launch_bounds__(256) void prompt_grad_k(const float* __restrict__ dx, int taps, int Bn, long n,
                                                     float* __restrict__ dprompts) {
    const long n4 = n / 4, total = (long)taps * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i / n4);
        const long c = (i % n4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < Bn; ++b) {
            const float4 v = *(const float4*)(dx + ((long)j * Bn + b) * n + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *(float4*)(dprompts + (long)j * n + c) = acc;
    }
}

}  // namespace

extern "C" int desta_prompt_expand(const float* prompts, int taps, int batch, int64_t n, float* x_f32, void* x_bf16, void* stream) {
    DESTA_CHECK_ARG(prompts && x_f32 && x_bf16 && taps > 0 && batch > 0 && n > 0 && n % 4 == 0, "prompt_expand: bad argument");
    long blocks = ((long)taps * batch * (n / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prompt_expand_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, prompts, taps, batch, (long)n, x_f32, (bf16_t*)x_bf16);
    DESTA_CHECK_LAUNCH("prompt_expand");
    return DESTA_OK;
}

extern "C" int desta_prompt_grad(const float* dx, int taps, int batch, int64_t n, float* dprompts, void* stream) {
    DESTA_CHECK_ARG(dx && dprompts && taps > 0 && batch > 0 && n > 0 && n % 4 == 0, "prompt_grad: bad argument");
    long blocks = ((long)taps * (n / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prompt_grad_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dx, taps, batch, (long)n, dprompts);
    DESTA_CHECK_LAUNCH("prompt_grad");
    return DESTA_OK;
}

extern "C" int desta_embed_gather(const void* table, const void* audio_rows, const int32_t* src_row, int rows, int hidden,
                                  void* out, void* stream) {
    DESTA_CHECK_ARG(table && src_row && out && rows > 0 && hidden % 8 == 0, "embed_gather: bad argument");
    hipLaunchKernelGGL(embed_gather_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table,
                       (const bf16_t*)audio_rows, src_row, rows, hidden, (bf16_t*)out);
    DESTA_CHECK_LAUNCH("embed_gather");
    return DESTA_OK;
}

extern "C" int desta_gather_rows_bf16(const void* in, const int32_t* idx, int rows, int hidden, void* out, void* stream) {
    DESTA_CHECK_ARG(in && idx && out && rows > 0 && hidden % 8 == 0, "gather_rows: bad argument");
    hipLaunchKernelGGL(gather_rows_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, idx, rows,
                       hidden, (bf16_t*)out);
    DESTA_CHECK_LAUNCH("gather_rows");
    return DESTA_OK;
}

extern "C" int desta_scatter_rows_bf16(const void* in, const int32_t* idx, int rows, int hidden, void* out, void* stream) {
    DESTA_CHECK_ARG(in && idx && out && rows > 0 && hidden % 8 == 0, "scatter_rows: bad argument");
    hipLaunchKernelGGL(scatter_rows_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, idx, rows,
                       hidden, (bf16_t*)out);
    DESTA_CHECK_LAUNCH("scatter_rows");
    return DESTA_OK;
}

extern "C" int desta_target_rows(const int64_t* labels, int batch, int seq, int32_t* idx, int64_t* compact_labels, int32_t* count,
                                 int s_major, void* stream) {
    DESTA_CHECK_ARG(labels && idx && compact_labels && count && batch > 0 && seq > 0, "target_rows: bad argument");
    hipLaunchKernelGGL(target_rows_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long*)labels, batch, seq, idx,
                       (long*)compact_labels, count, s_major);
    DESTA_CHECK_LAUNCH("target_rows");
    return DESTA_OK;
}

extern "C" size_t desta_ce_workspace_floats(int batch, int seq) { return (size_t)batch * seq + 8; }

extern "C" int desta_causal_lm_loss(void* logits, int64_t ld, const int64_t* labels, int batch, int seq, int vocab,
                                    float* loss, float* workspace, int write_grad, void* stream) {
    DESTA_CHECK_ARG(logits && labels && loss && workspace, "causal_lm_loss: null argument");
    DESTA_CHECK_ARG(batch > 0 && seq > 0 && vocab > 0 && ld >= vocab && ld % 8 == 0, "causal_lm_loss: bad shape");
    DESTA_CHECK_ARG(((uintptr_t)logits % 16) == 0, "causal_lm_loss: logits must be 16-byte aligned");
    float* scal = workspace;
    float* row_loss = workspace + 8;
    hipStream_t st = (hipStream_t)stream;
    const int M = batch * seq;
    hipLaunchKernelGGL(ce_count_k, dim3(1), dim3(256), 0, st, (const long*)labels, batch, seq, scal);
    const int nv8 = vocab / 8;
#define CE_ARGS dim3(M), dim3(1024), 0, st, (bf16_t*)logits, (long)ld, (const long*)labels, seq, vocab, (const float*)scal, row_loss, write_grad
    if (vocab % 8 == 0 && nv8 > 2048 && nv8 <= 1024 * 16) hipLaunchKernelGGL(ce_row_reg_k<16>, CE_ARGS);
    else if (vocab % 8 == 0 && nv8 > 2048 && nv8 <= 1024 * 20) hipLaunchKernelGGL(ce_row_reg_k<20>, CE_ARGS);
    else
        hipLaunchKernelGGL(ce_row_k, dim3(M), dim3(256), 0, st, (bf16_t*)logits, (long)ld, (const long*)labels, seq, vocab,
                           (const float*)scal, row_loss, write_grad);
#undef CE_ARGS
    hipLaunchKernelGGL(ce_finish_k, dim3(1), dim3(256), 0, st, (const float*)row_loss, M, (const float*)scal, loss);
    DESTA_CHECK_LAUNCH("causal_lm_loss");
    return DESTA_OK;
}

extern "C" int desta_tap_mix_fwd(const float* x, const float* layer_weights, int taps, int batch, int prompt, int d,
                                 float* out, void* stream) {
    DESTA_CHECK_ARG(x && layer_weights && out && taps > 0 && taps <= MAXT && d % 4 == 0, "tap_mix_fwd: bad argument");
    hipLaunchKernelGGL(mix_fwd_k, dim3(batch * prompt), dim3(256), 0, (hipStream_t)stream, x, layer_weights, taps,
                       batch * prompt, prompt, d, out);
    DESTA_CHECK_LAUNCH("tap_mix_fwd");
    return DESTA_OK;
}

extern "C" int desta_tap_mix_bwd(const float* x, const float* layer_weights, const float* dout, int taps, int batch,
                                 int prompt, int d, float* dx, float* dlayer_weights, void* stream) {
    DESTA_CHECK_ARG(x && layer_weights && dout && dx && dlayer_weights && taps > 0 && taps <= MAXT && d % 4 == 0,
                    "tap_mix_bwd: bad argument");
    hipLaunchKernelGGL(mix_bwd_k, dim3(prompt), dim3(256), 0, (hipStream_t)stream, x, layer_weights, dout, taps, batch,
                       prompt, d, dx, dlayer_weights);
    DESTA_CHECK_LAUNCH("tap_mix_bwd");
    return DESTA_OK;
}
