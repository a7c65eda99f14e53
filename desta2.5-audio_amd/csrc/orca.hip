// ORCA hybrid, first slice, forward only: the small row-wise kernels the ORCA branch needs beside the GEMM /
// attention / LayerNorm entry points it shares with the qformer_1 path.  Reference: /root/reference/desta/models/modeling_desta25.py
//   :336-352  local branch  — softmax(local_layer_weights)-weighted sum of the tapped encoder states      -> desta_orca_local_mix
//   :22-95    compute_rope_freqs / apply_rotary_pos_emb on the WHOLE hidden vector, positions t / scale     -> desta_orca_rope
//   :456-490  hidden + sigmoid(gate_proj(hidden)) * LayerNorm(cross_attn)                                   -> desta_orca_gate_residual
//   :1174-1198 diversity / orthogonality losses on L2-normalised tokens                                      -> desta_orca_sim_loss
//   :460-486  per-layer alignment loss: 1 - cos(mean_t audio, mean_span hidden)                             -> desta_orca_align
// All HBM-bound, one wave (or one block) per row; none of them is on the qformer_1 hot path.  fp32 statistics, bf16 streams.
#include "common.h"
#include "desta_hip.h"
#include <math.h>

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {        // 256 threads; sh[4]
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// out[r, :] = sum_l softmax(w)[l] * x[l, r, :]   (x bf16 [taps][rows][d], out bf16)
__global__ __launch_bounds__(256) void local_mix_k(const bf16_t* __restrict__ x, const float* __restrict__ w, int taps, long rows, int d,
                                                   bf16_t* __restrict__ out) {
    float wl[8], mx = -INFINITY, den = 0.f;
    for (int l = 0; l < taps; ++l) mx = fmaxf(mx, w[l]);
    for (int l = 0; l < taps; ++l) { wl[l] = __expf(w[l] - mx); den += wl[l]; }
    for (int l = 0; l < taps; ++l) wl[l] /= den;
    const long n8 = rows * (d / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < taps; ++l) {
            const u16x8 v = *(const u16x8*)(x + ((long)l * rows * d) + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += wl[l] * bf2f(v[e]);
        }
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
        *(u16x8*)(out + i * 8) = o;
    }
}

// y[b, t, i] = x1 cos - x2 sin, y[b, t, i + H/2] = x1 sin + x2 cos, angle = (t / scale) * theta^(-i / (H/2)); cos / sin rounded to bf16
// like the reference's `cos.to(x.dtype)` on a bf16 model
__global__ __launch_bounds__(256) void orca_rope_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int batch, int T, int H, float theta,
                                                   float scale, int round_cs) {
    const int half = H >> 1;
    const long n = (long)batch * T * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % half);
        const long row = i / half;
        const int t = (int)(row % T);
        const float inv = __powf(theta, -(float)c / (float)half);
        const float ang = ((float)t / scale) * inv;
        float cs = cosf(ang), sn = sinf(ang);
        if (round_cs) { cs = bf2f(f2bf(cs)); sn = bf2f(f2bf(sn)); }
        const float x1 = bf2f(x[row * H + c]), x2 = bf2f(x[row * H + half + c]);
        y[row * H + c] = f2bf(x1 * cs - x2 * sn);
        y[row * H + half + c] = f2bf(x1 * sn + x2 * cs);
    }
}

// hs[m, :] += sigmoid(dot(g1[m, :], w2) + b2) * c[m, :]      one wave per row; hs bf16 in place, c bf16, g1 bf16 [M, Hq], w2 fp32 [Hq]
__global__ __launch_bounds__(256) void gate_residual_k(bf16_t* __restrict__ hs, long ld_hs, const bf16_t* __restrict__ c, const bf16_t* __restrict__ g1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, long M, int H, int Hq, float* __restrict__ gate_out) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (m >= M) return;
    float dot = 0.f;
    for (int k = lane; k < Hq; k += 64) dot += bf2f(g1[m * Hq + k]) * w2[k];
    dot = wsum(dot) + b2[0];
    const float gate = 1.0f / (1.0f + __expf(-dot));
    if (gate_out && lane == 0) gate_out[m] = gate;
    const float gq = bf2f(f2bf(gate));                         // the reference's gate is a bf16 tensor on a bf16 model
    for (int k = lane * 8; k < H; k += 512) {
        const u16x8 hv = *(const u16x8*)(hs + m * ld_hs + k), cv = *(const u16x8*)(c + m * H + k);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(hv[e]) + bf2f(f2bf(gq * bf2f(cv[e]))));
        *(u16x8*)(hs + m * ld_hs + k) = o;
    }
}

// out[0] += sum_{j} (xhat_i . yhat_j - (identity && i == j))^2  for one (b, i) per block, divided by the element count on the host side
// x [B, Nx, H], y [B, Ny, H] bf16 (row stride H); rows of y are picked through idx[Ny] (uniform sampling of the local tokens) or 0..Ny-1
__global__ __launch_bounds__(256) void sim_loss_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const int* __restrict__ idx, int Nx,
                                                  int Ny, long y_rows, int H, int identity, float* __restrict__ part) {
    __shared__ float sh[4];
    const int b = blockIdx.x / Nx, i = blockIdx.x % Nx;
    const bf16_t* xr = x + ((long)b * Nx + i) * H;
    float q = 0.f;
    for (int k = threadIdx.x; k < H; k += 256) { const float v = bf2f(xr[k]); q += v * v; }
    const float xn = fmaxf(sqrtf(block_sum(q, sh)), 1e-12f);               // F.normalize: x / max(||x||, eps)
    float acc = 0.f;
    for (int j = 0; j < Ny; ++j) {
        const bf16_t* yr = y + ((long)b * y_rows + (idx ? idx[j] : j)) * H;
        float d = 0.f, n2 = 0.f;
        for (int k = threadIdx.x; k < H; k += 256) { const float u = bf2f(yr[k]); d += bf2f(xr[k]) * u; n2 += u * u; }
        const float dd = block_sum(d, sh), yn = fmaxf(sqrtf(block_sum(n2, sh)), 1e-12f);
        const float s = dd / (xn * yn) - ((identity && i == j) ? 1.0f : 0.0f);
        acc += s * s;
    }
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// out[b] = 1 - cos(mean_t a[b, t, :], mean_{s in [s0, s1)} hs[row(b, s), :]);   one block per entry; spans [n][3] = (text row, s0, s1)
__global__ __launch_bounds__(256) void align_k(const bf16_t* __restrict__ a, int T, const bf16_t* __restrict__ hs, long hs_rs, long hs_bs, int H,
                                               const int* __restrict__ spans, float* __restrict__ out) {
    __shared__ float sh[4];
    const int e = blockIdx.x;
    const int row = spans[3 * e], s0 = spans[3 * e + 1], s1 = spans[3 * e + 2];
    float dab = 0.f, na = 0.f, nb = 0.f;
    for (int k = threadIdx.x; k < H; k += 256) {
        float ma = 0.f, mh = 0.f;
        for (int t = 0; t < T; ++t) ma += bf2f(a[((long)e * T + t) * H + k]);
        for (int s = s0; s < s1; ++s) mh += bf2f(hs[(long)row * hs_bs + (long)s * hs_rs + k]);
        ma /= (float)T; mh /= (float)max(s1 - s0, 1);
        dab += ma * mh; na += ma * ma; nb += mh * mh;
    }
    const float d = block_sum(dab, sh), x = fmaxf(sqrtf(block_sum(na, sh)), 1e-12f), y = fmaxf(sqrtf(block_sum(nb, sh)), 1e-12f);
    if (threadIdx.x == 0) out[e] = 1.0f - d / (x * y);
}

inline unsigned nblk(long n) { long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b)); }

}  // namespace

extern "C" int desta_orca_local_mix(const void* x, const float* layer_weights, int taps, int64_t rows, int d, void* out, void* stream) {
    DESTA_CHECK_ARG(x && layer_weights && out && taps > 0 && taps <= 8 && rows > 0 && d % 8 == 0, "orca_local_mix: bad argument");
    hipLaunchKernelGGL(local_mix_k, dim3(nblk(rows * (d / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, layer_weights, taps, (long)rows, d, (bf16_t*)out);
    DESTA_CHECK_LAUNCH("orca_local_mix");
    return DESTA_OK;
}

extern "C" int desta_orca_rope(const void* x, void* y, int batch, int tokens, int hidden, float theta, float position_scale, int round_cos_sin,
                               void* stream) {
    DESTA_CHECK_ARG(x && y && batch > 0 && tokens > 0 && hidden % 2 == 0 && theta > 0.f && position_scale > 0.f, "orca_rope: bad argument");
    hipLaunchKernelGGL(orca_rope_k, dim3(nblk((long)batch * tokens * (hidden / 2))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, batch,
                       tokens, hidden, theta, position_scale, round_cos_sin);
    DESTA_CHECK_LAUNCH("orca_rope");
    return DESTA_OK;
}

extern "C" int desta_orca_gate_residual(void* hidden, int64_t ld_hidden, const void* cross, const void* gate_hidden, const float* gate_w2, const float* gate_b2,
                                        int64_t rows, int hidden_size, int gate_width, float* gate_out, void* stream) {
    DESTA_CHECK_ARG(hidden && cross && gate_hidden && gate_w2 && gate_b2 && rows > 0 && hidden_size % 8 == 0 && gate_width > 0 && ld_hidden % 8 == 0,
                    "orca_gate_residual: bad argument");
    hipLaunchKernelGGL(gate_residual_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)hidden, (long)ld_hidden, (const bf16_t*)cross,
                       (const bf16_t*)gate_hidden, gate_w2, gate_b2, (long)rows, hidden_size, gate_width, gate_out);
    DESTA_CHECK_LAUNCH("orca_gate_residual");
    return DESTA_OK;
}

extern "C" int desta_orca_sim_loss(const void* x, const void* y, const int32_t* y_index, int batch, int nx, int ny, int64_t y_rows, int hidden,
                                   int subtract_identity, float* partials, void* stream) {
    DESTA_CHECK_ARG(x && y && partials && batch > 0 && nx > 0 && ny > 0 && y_rows >= ny && hidden > 0 && (!subtract_identity || nx == ny),
                    "orca_sim_loss: bad argument");
    hipLaunchKernelGGL(sim_loss_k, dim3((unsigned)(batch * nx)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)y, y_index, nx, ny,
                       (long)y_rows, hidden, subtract_identity, partials);
    DESTA_CHECK_LAUNCH("orca_sim_loss");
    return DESTA_OK;
}

extern "C" int desta_orca_align(const void* audio, int tokens, const void* hidden, int64_t hidden_row_stride, int64_t hidden_batch_stride, int hidden_size,
                                const int32_t* spans, int n_spans, float* out, void* stream) {
    DESTA_CHECK_ARG(audio && hidden && spans && out && tokens > 0 && n_spans > 0 && hidden_size > 0, "orca_align: bad argument");
    hipLaunchKernelGGL(align_k, dim3((unsigned)n_spans), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)audio, tokens, (const bf16_t*)hidden,
                       (long)hidden_row_stride, (long)hidden_batch_stride, hidden_size, spans, out);
    DESTA_CHECK_LAUNCH("orca_align");
    return DESTA_OK;
}
