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

constexpr int MAX_TAPS = 32;       // tapped encoder layers: 4 in the shipped configs, every layer of whisper-large with orca_use_all_layers
// out[r, :] = sum_l softmax(w)[l] * x[l, r, :]   (x bf16 [taps][rows][d], out bf16)
__global__ __launch_bounds__(256) void local_mix_k(const bf16_t* __restrict__ x, const float* __restrict__ w, int taps, long rows, int d,
                                                   bf16_t* __restrict__ out) {
    float wl[MAX_TAPS], mx = -INFINITY, den = 0.f;
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

// ---------------------------------------------------------------------------------------------- backward pieces (round 4)
// x_out = x + bf16(gate) * c  ->  dc[m,:] = gate[m] * dxo[m,:],  dg2[m] = (dxo[m,:] . c[m,:]) * gate[m] * (1 - gate[m])   (pre-sigmoid gradient)
__global__ __launch_bounds__(256) void gate_residual_bwd_k(const bf16_t* __restrict__ dxo, long ld, const bf16_t* __restrict__ c, const float* __restrict__ gate,
                                                           long M, int H, bf16_t* __restrict__ dc, float* __restrict__ dg2) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (m >= M) return;
    const float g = gate[m];
    float dot = 0.f;
    for (int k = lane * 8; k < H; k += 512) {
        const u16x8 dv = *(const u16x8*)(dxo + m * ld + k), cv = *(const u16x8*)(c + m * H + k);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = bf2f(dv[e]); dot += d * bf2f(cv[e]); o[e] = f2bf(g * d); }
        *(u16x8*)(dc + m * H + k) = o;
    }
    dot = wsum(dot);
    if (lane == 0) dg2[m] = dot * g * (1.0f - g);
}

// g2 = gelu(pre) . w2 + b2:  dpre[m,k] = dg2[m] * w2[k] * gelu'(pre[m,k])
__global__ __launch_bounds__(256) void gate_mlp_bwd_k(const float* __restrict__ dg2, const bf16_t* __restrict__ pre, const float* __restrict__ w2, long M, int Hq,
                                                      bf16_t* __restrict__ dpre) {
    const long n = M * Hq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long m = i / Hq;
        const int k = (int)(i % Hq);
        dpre[i] = f2bf(dg2[m] * w2[k] * gelu_erf_grad(bf2f(pre[i])));
    }
}
// dw2[k] = sum_m dg2[m] * g1[m,k]  (g1 = the bf16 GELU output the forward fed to the dot),  db2 = sum_m dg2[m].  Two launches, fixed order:
// block (x, y) = 64 columns x 4 row lanes over row slice y of W2_SPLIT -> part[y][0 .. Hq) and part[y][Hq] (the bias term); then the
// slices are added in slice order.  [One launch of Hq / 64 blocks walking all rows took 0.5 ms per layer at 5120 x 1024.]
constexpr int W2_SPLIT = 64;
__global__ __launch_bounds__(256) void gate_w2_grad_k(const float* __restrict__ dg2, const bf16_t* __restrict__ g1, long M, int Hq, float* __restrict__ part) {
    __shared__ float sh[4][64];
    const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6, col = blockIdx.x * 64 + c, y = blockIdx.y;
    const long per = (M + W2_SPLIT - 1) / W2_SPLIT, m0 = y * per, m1 = m0 + per < M ? m0 + per : M;
    float acc = 0.f, accb = 0.f;
    for (long m = m0 + r4; m < m1; m += 4) {
        const float d = dg2[m];
        if (col < Hq) acc += d * bf2f(g1[m * Hq + col]);
        if (blockIdx.x == 0 && c == 0) accb += d;
    }
    sh[r4][c] = acc;
    __syncthreads();
    if (r4 == 0 && col < Hq) part[(long)y * (Hq + 1) + col] = sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c];
    __syncthreads();
    if (blockIdx.x == 0) {
        if (c == 0) sh[r4][0] = accb;
        __syncthreads();
        if (threadIdx.x == 0) part[(long)y * (Hq + 1) + Hq] = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
    }
}
__global__ __launch_bounds__(256) void gate_w2_fin_k(const float* __restrict__ part, int Hq, float* __restrict__ dw2, float* __restrict__ db2) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col > Hq) return;
    float s = 0.f;
    for (int y = 0; y < W2_SPLIT; ++y) s += part[(long)y * (Hq + 1) + col];
    if (col < Hq) dw2[col] = s; else db2[0] = s;
}

// per-layer alignment loss, gradient w.r.t. the hidden rows of the span: L = coef * sum_e (1 - cos(a_e, t_e)), t_e = mean_{s in span} hs[row, s, :]
// (a_e = mean_t audio[e, t, :] carries no gradient: the reference pools it under no_grad).  dhs[row, s, :] += coef * d(1 - cos)/dt / span
__global__ __launch_bounds__(256) void align_bwd_k(const bf16_t* __restrict__ a, int T, const bf16_t* __restrict__ hs, long hs_rs, long hs_bs, int H,
                                                   const int* __restrict__ spans, float coef, bf16_t* __restrict__ dhs, long d_rs, long d_bs) {
    __shared__ float sh[4];
    const int e = blockIdx.x;
    const int row = spans[3 * e], s0 = spans[3 * e + 1], s1 = spans[3 * e + 2], len = max(s1 - s0, 1);
    float dab = 0.f, na = 0.f, nb = 0.f;
    for (int k = threadIdx.x; k < H; k += 256) {
        float ma = 0.f, mh = 0.f;
        for (int t = 0; t < T; ++t) ma += bf2f(a[((long)e * T + t) * H + k]);
        for (int s = s0; s < s1; ++s) mh += bf2f(hs[(long)row * hs_bs + (long)s * hs_rs + k]);
        ma /= (float)T; mh /= (float)len;
        dab += ma * mh; na += ma * ma; nb += mh * mh;
    }
    const float d = block_sum(dab, sh), x = fmaxf(sqrtf(block_sum(na, sh)), 1e-12f), y = fmaxf(sqrtf(block_sum(nb, sh)), 1e-12f);
    const float cs = d / (x * y);
    for (int k = threadIdx.x; k < H; k += 256) {
        float ma = 0.f, mh = 0.f;
        for (int t = 0; t < T; ++t) ma += bf2f(a[((long)e * T + t) * H + k]);
        for (int s = s0; s < s1; ++s) mh += bf2f(hs[(long)row * hs_bs + (long)s * hs_rs + k]);
        ma /= (float)T; mh /= (float)len;
        const float g = -coef * (ma / x - cs * mh / y) / y / (float)len;       // d(1 - cos)/d t_k, spread over the span's rows
        for (int s = s0; s < s1; ++s) {
            bf16_t* q = dhs + (long)row * d_bs + (long)s * d_rs + k;
            *q = f2bf(bf2f(*q) + g);
        }
    }
}

// transpose of the whole-vector rotation (rotation by the NEGATIVE angle) of the accumulated fp32 gradient of the rotated audio tokens,
// split back into its sources: tokens [0, n_first) of a clip go to dst0 [batch, n_first, H] (the global tokens under orca_global_cross_attn),
// the rest to dst1 [batch, T - n_first, H] (the local tokens); ADDED to what the destinations hold
__global__ __launch_bounds__(256) void orca_rope_bwd_k(const float* __restrict__ dy, int batch, int T, int H, float theta, float scale, int round_cs,
                                                       int n_first, float* __restrict__ dst0, float* __restrict__ dst1) {
    const int half = H >> 1;
    const long n = (long)batch * T * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % half);
        const long row = i / half;
        const int t = (int)(row % T), b = (int)(row / T);
        const float ang = ((float)t / scale) * __powf(theta, -(float)c / (float)half);
        float cs = cosf(ang), sn = sinf(ang);
        if (round_cs) { cs = bf2f(f2bf(cs)); sn = bf2f(f2bf(sn)); }
        const float y1 = dy[row * H + c], y2 = dy[row * H + half + c];
        float* d = t < n_first ? dst0 + ((long)b * n_first + t) * H : dst1 + ((long)b * (T - n_first) + (t - n_first)) * H;
        d[c] += y1 * cs + y2 * sn;
        d[half + c] += -y1 * sn + y2 * cs;
    }
}

// col2im of the strided Conv1d: dx_pad[b, t, c] = sum_{j < k, (t - j) % stride == 0, 0 <= (t - j) / stride < Tout} dcol[b, (t - j) / stride, j * H + c]
__global__ __launch_bounds__(256) void col2im_add_k(const bf16_t* __restrict__ dcol, int batch, int Tout, int Tp, int H, int k, int stride, bf16_t* __restrict__ dx) {
    const long n = (long)batch * Tp * (H / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % (H / 8)) * 8;
        const long bt = i / (H / 8);
        const int t = (int)(bt % Tp), b = (int)(bt / Tp);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < k; ++j) {
            const int u = t - j;
            if (u < 0 || u % stride) continue;
            const int to = u / stride;
            if (to >= Tout) continue;
            const u16x8 v = *(const u16x8*)(dcol + ((long)b * Tout + to) * ((long)k * H) + (long)j * H + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
        *(u16x8*)(dx + bt * H + c) = o;
    }
}

// d(local_layer_weights): dots[l] = sum_{r,c} dfused[r,c] * x[l,r,c] (per-block partials, fixed-order finish), then the softmax Jacobian
__global__ __launch_bounds__(256) void local_mix_bwd_k(const bf16_t* __restrict__ dfused, const bf16_t* __restrict__ x, int taps, long rows, int d, float* __restrict__ part) {
    __shared__ float sh[4];
    float acc[MAX_TAPS];
#pragma unroll
    for (int l = 0; l < MAX_TAPS; ++l) acc[l] = 0.f;
    const long n8 = rows * (d / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const u16x8 g = *(const u16x8*)(dfused + i * 8);
#pragma unroll
        for (int l = 0; l < MAX_TAPS; ++l) {                                  // (static indices: the accumulators stay in registers)
            if (l < taps) {
                const u16x8 v = *(const u16x8*)(x + (long)l * rows * d + i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[l] += bf2f(g[e]) * bf2f(v[e]);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < MAX_TAPS; ++l) {
        if (l < taps) {
            const float s = block_sum(acc[l], sh);
            if (threadIdx.x == 0) part[(long)blockIdx.x * MAX_TAPS + l] = s;
        }
    }
}
__global__ void local_mix_bwd_fin_k(const float* __restrict__ part, int nblk, const float* __restrict__ w, int taps, float* __restrict__ dw) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float dots[MAX_TAPS], p[MAX_TAPS], mx = -INFINITY, den = 0.f, mean = 0.f;
    for (int l = 0; l < taps; ++l) { float s = 0.f; for (int b = 0; b < nblk; ++b) s += part[(long)b * MAX_TAPS + l]; dots[l] = s; mx = fmaxf(mx, w[l]); }
    for (int l = 0; l < taps; ++l) { p[l] = __expf(w[l] - mx); den += p[l]; }
    for (int l = 0; l < taps; ++l) { p[l] /= den; mean += p[l] * dots[l]; }
    for (int l = 0; l < taps; ++l) dw[l] = p[l] * (dots[l] - mean);
}

// gradient of coef * sum_{b,i,j} (xhat_i . yhat_j - [identity and i == j])^2 w.r.t. the UN-normalised rows of x:
//   v = mult * coef * sum_j 2 s_ij yhat_j   (mult = 2 for the symmetric x == y case, where row i also appears as a y),
//   dx_i += (v - (v . xhat_i) xhat_i) / ||x_i||.      One block per (b, i); rows of x / y / dx are picked through optional index lists.
__global__ __launch_bounds__(256) void sim_loss_bwd_k(const bf16_t* __restrict__ x, const int* __restrict__ xidx, long x_rows, const bf16_t* __restrict__ y,
                                                      const int* __restrict__ yidx, long y_rows, int Nx, int Ny, int H, int identity, float coef,
                                                      float* __restrict__ dx) {
    __shared__ float sh[4];
    __shared__ float sij[128], ynorm[128];
    const int b = blockIdx.x / Nx, i = blockIdx.x % Nx;
    const long xr_ = (long)b * x_rows + (xidx ? xidx[i] : i);
    const bf16_t* xr = x + xr_ * H;
    float q = 0.f;
    for (int k = threadIdx.x; k < H; k += 256) { const float v = bf2f(xr[k]); q += v * v; }
    const float xn = fmaxf(sqrtf(block_sum(q, sh)), 1e-12f);
    for (int j = 0; j < Ny; ++j) {
        const bf16_t* yr = y + ((long)b * y_rows + (yidx ? yidx[j] : j)) * H;
        float d = 0.f, n2 = 0.f;
        for (int k = threadIdx.x; k < H; k += 256) { const float u = bf2f(yr[k]); d += bf2f(xr[k]) * u; n2 += u * u; }
        const float dd = block_sum(d, sh), yn = fmaxf(sqrtf(block_sum(n2, sh)), 1e-12f);
        if (threadIdx.x == 0) { sij[j] = dd / (xn * yn) - ((identity && i == j) ? 1.0f : 0.0f); ynorm[j] = yn; }
    }
    __syncthreads();
    const float mult = (identity ? 2.0f : 1.0f) * 2.0f * coef;
    float vdot = 0.f;
    for (int k = threadIdx.x; k < H; k += 256) {
        float v = 0.f;
        for (int j = 0; j < Ny; ++j) v += sij[j] * bf2f(y[((long)b * y_rows + (yidx ? yidx[j] : j)) * H + k]) / ynorm[j];
        vdot += v * bf2f(xr[k]) / xn;
    }
    vdot = block_sum(vdot, sh);
    for (int k = threadIdx.x; k < H; k += 256) {
        float v = 0.f;
        for (int j = 0; j < Ny; ++j) v += sij[j] * bf2f(y[((long)b * y_rows + (yidx ? yidx[j] : j)) * H + k]) / ynorm[j];
        dx[xr_ * H + k] += mult * (v - vdot * bf2f(xr[k]) / xn) / xn;
    }
}

inline unsigned nblk(long n) { long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b)); }

}  // namespace

extern "C" int desta_orca_local_mix(const void* x, const float* layer_weights, int taps, int64_t rows, int d, void* out, void* stream) {
    DESTA_CHECK_ARG(x && layer_weights && out && taps > 0 && taps <= 32 && rows > 0 && d % 8 == 0, "orca_local_mix: bad argument");
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

extern "C" int desta_orca_gate_residual_bwd(const void* d_out, int64_t ld, const void* cross, const float* gate, int64_t rows, int hidden_size, void* d_cross,
                                            float* d_gate_pre, void* stream) {
    DESTA_CHECK_ARG(d_out && cross && gate && d_cross && d_gate_pre && rows > 0 && hidden_size % 8 == 0 && ld % 8 == 0, "orca_gate_residual_bwd: bad argument");
    hipLaunchKernelGGL(gate_residual_bwd_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_out, (long)ld, (const bf16_t*)cross,
                       gate, (long)rows, hidden_size, (bf16_t*)d_cross, d_gate_pre);
    DESTA_CHECK_LAUNCH("orca_gate_residual_bwd");
    return DESTA_OK;
}

extern "C" int desta_orca_gate_mlp_bwd(const float* d_gate_pre, const void* gate_preact, const void* gate_hidden, const float* gate_w2, int64_t rows, int gate_width,
                                       void* d_preact, float* d_w2, float* d_b2, float* workspace, void* stream) {
    DESTA_CHECK_ARG(d_gate_pre && gate_preact && gate_hidden && gate_w2 && d_preact && d_w2 && d_b2 && workspace && rows > 0 && gate_width > 0,
                    "orca_gate_mlp_bwd: bad argument");
    hipLaunchKernelGGL(gate_mlp_bwd_k, dim3(nblk(rows * gate_width)), dim3(256), 0, (hipStream_t)stream, d_gate_pre, (const bf16_t*)gate_preact, gate_w2, (long)rows,
                       gate_width, (bf16_t*)d_preact);
    hipLaunchKernelGGL(gate_w2_grad_k, dim3((unsigned)((gate_width + 63) / 64), W2_SPLIT), dim3(256), 0, (hipStream_t)stream, d_gate_pre, (const bf16_t*)gate_hidden,
                       (long)rows, gate_width, workspace);
    hipLaunchKernelGGL(gate_w2_fin_k, dim3((unsigned)((gate_width + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, gate_width, d_w2, d_b2);
    DESTA_CHECK_LAUNCH("orca_gate_mlp_bwd");
    return DESTA_OK;
}

extern "C" int desta_orca_align_bwd(const void* audio, int tokens, const void* hidden, int64_t hidden_row_stride, int64_t hidden_batch_stride, int hidden_size,
                                    const int32_t* spans, int n_spans, float coef, void* d_hidden, int64_t d_row_stride, int64_t d_batch_stride, void* stream) {
    DESTA_CHECK_ARG(audio && hidden && spans && d_hidden && tokens > 0 && n_spans > 0 && hidden_size > 0, "orca_align_bwd: bad argument");
    hipLaunchKernelGGL(align_bwd_k, dim3((unsigned)n_spans), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)audio, tokens, (const bf16_t*)hidden,
                       (long)hidden_row_stride, (long)hidden_batch_stride, hidden_size, spans, coef, (bf16_t*)d_hidden, (long)d_row_stride, (long)d_batch_stride);
    DESTA_CHECK_LAUNCH("orca_align_bwd");
    return DESTA_OK;
}

extern "C" int desta_orca_rope_bwd(const float* d_rotated, int batch, int tokens, int hidden, float theta, float position_scale, int round_cos_sin, int n_first,
                                   float* d_first, float* d_rest, void* stream) {
    DESTA_CHECK_ARG(d_rotated && batch > 0 && tokens > 0 && hidden % 2 == 0 && n_first >= 0 && n_first <= tokens && (n_first == 0 || d_first) &&
                    (n_first == tokens || d_rest), "orca_rope_bwd: bad argument");
    hipLaunchKernelGGL(orca_rope_bwd_k, dim3(nblk((long)batch * tokens * (hidden / 2))), dim3(256), 0, (hipStream_t)stream, d_rotated, batch, tokens, hidden, theta,
                       position_scale, round_cos_sin, n_first, d_first, d_rest);
    DESTA_CHECK_LAUNCH("orca_rope_bwd");
    return DESTA_OK;
}

extern "C" int desta_orca_col2im_add(const void* d_col, int batch, int tokens_out, int tokens_padded, int hidden, int kernel, int stride, void* d_padded, void* stream) {
    DESTA_CHECK_ARG(d_col && d_padded && batch > 0 && tokens_out > 0 && tokens_padded > 0 && hidden % 8 == 0 && kernel > 0 && stride > 0, "orca_col2im_add: bad argument");
    hipLaunchKernelGGL(col2im_add_k, dim3(nblk((long)batch * tokens_padded * (hidden / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_col, batch, tokens_out,
                       tokens_padded, hidden, kernel, stride, (bf16_t*)d_padded);
    DESTA_CHECK_LAUNCH("orca_col2im_add");
    return DESTA_OK;
}

extern "C" int desta_orca_local_mix_bwd(const void* d_out, const void* x, const float* layer_weights, int taps, int64_t rows, int d, float* d_layer_weights,
                                        float* workspace, void* stream) {
    DESTA_CHECK_ARG(d_out && x && layer_weights && d_layer_weights && workspace && taps > 0 && taps <= 32 && rows > 0 && d % 8 == 0, "orca_local_mix_bwd: bad argument");
    const int nb = 256;                                                      /* workspace: 256 * 32 floats */
    hipLaunchKernelGGL(local_mix_bwd_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_out, (const bf16_t*)x, taps, (long)rows, d, workspace);
    hipLaunchKernelGGL(local_mix_bwd_fin_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, nb, layer_weights, taps, d_layer_weights);
    DESTA_CHECK_LAUNCH("orca_local_mix_bwd");
    return DESTA_OK;
}

extern "C" int desta_orca_sim_loss_bwd(const void* x, const int32_t* x_index, int64_t x_rows, const void* y, const int32_t* y_index, int64_t y_rows, int batch, int nx,
                                       int ny, int hidden, int subtract_identity, float coef, float* d_x, void* stream) {
    DESTA_CHECK_ARG(x && y && d_x && batch > 0 && nx > 0 && ny > 0 && ny <= 128 && x_rows >= nx && y_rows >= ny && hidden > 0 && (!subtract_identity || nx == ny),
                    "orca_sim_loss_bwd: bad argument (ny <= 128)");
    hipLaunchKernelGGL(sim_loss_bwd_k, dim3((unsigned)(batch * nx)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_index, (long)x_rows, (const bf16_t*)y, y_index,
                       (long)y_rows, nx, ny, hidden, subtract_identity, coef, d_x);
    DESTA_CHECK_LAUNCH("orca_sim_loss_bwd");
    return DESTA_OK;
}
