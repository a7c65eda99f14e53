// Fused global-norm clip + Adafactor over a FLAT fp32 parameter / gradient arena (gfx950, HBM-bound).
//
// Replaces, per optimizer step, `torch.nn.utils.clip_grad_norm_(params, 1.0)` (TF:trainer.py:1780-1782)
// followed by `transformers.optimization.Adafactor.step` (TF:optimization.py:1203-1294) with the
// HF-Trainer kwargs scale_parameter=False, relative_step=False, beta1=None
// (TF:trainer_optimizer.py:197): ~8 elementwise torch passes x ~230 tensors become 4 launches:
//
//   K1 stats    : one pass over g: per-row sum g^2, per-(unit,col) partial sum g^2, per-unit sum g^2
//   K2 finalize : global norm -> clip coef c; row/col EMA update with c^2*mean(g^2)+eps1; r/c factors
//   K3 usq      : sum u^2 per unit, u = c*g*rfac[row]*cfac[col]        (g re-read)
//   K4 apply    : p = p*(1 - wd*lr) - lr * u / max(1, rms(u)/clip_thr)  (g re-read, p read+write)
//   (1-D tensors: V1 sum g^2, V2 everything else, one block per tensor.)
//
// All reductions are two-stage with a fixed order (no float atomics), so every data-parallel rank
// computes bit-identical updates from bit-identical all-reduced gradients.
// Algorithmic bytes: 12 N (g read, p read, p write); this implementation moves 20 N (g is read three
// times) — stated in DESIGN.md.
#include "common.h"
#include "desta_hip.h"

namespace {

constexpr int UNIT_ROWS = 64;      // rows per work unit (4 waves x 16 rows)
constexpr int MAXSEG = 16;         // 256-column segments per row handled in registers (cols <= 4096)

struct Tab {
    const long* ten;               // [T][8]: offset, batch, rows, cols, row_state_off, col_state_off, unit0, nunits
    const float* ten_wd;           // [T]
    int T;
    const int* unit;               // [U][4]: tensor, batch index, row0, nrows
    const long* unit_col_off;      // [U] offset of this unit's column partials in colpart
    int U;
    const long* vec;               // [V][3]: offset, n, sq_state_off
    const float* vec_wd;           // [V]
    int V;
};

// workspace carve (floats): scalars[8] | unit_sumsq[U] | unit_usq[U] | vec_sumsq[V] | rowsum[SR] | rfac[SR] | cfac[SC] | colpart[...]
struct Ws {
    float* scalars; float* unit_sumsq; float* unit_usq; float* vec_sumsq;
    float* rowsum; float* rfac; float* cfac; float* colpart;
};

__global__ __launch_bounds__(256) void k1_stats(Tab tb, Ws ws, const float* __restrict__ g) {
    __shared__ float colred[4][256 * 4];           // one 256-col segment at a time, 4 waves
    __shared__ float red[4];
    const int u = blockIdx.x;
    const int t = tb.unit[u * 4 + 0], b = tb.unit[u * 4 + 1], row0 = tb.unit[u * 4 + 2], nrows = tb.unit[u * 4 + 3];
    const long* tt = tb.ten + (long)t * 8;
    const long off = tt[0];
    const int R = (int)tt[2], Cn = (int)tt[3];
    const long rs_off = tt[4];
    const float* gt = g + off + (long)b * R * Cn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec4 = (Cn % 4 == 0);
    float4 cacc[MAXSEG];
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) cacc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float tot = 0.f;
    for (int rr = wave; rr < nrows; rr += 4) {
        const int r = row0 + rr;
        const float* gr = gt + (long)r * Cn;
        float rsum = 0.f;
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) {
            const int c0 = s * 256 + lane * 4;
            if (c0 < Cn) {
                float4 v;
                if (vec4) v = *(const float4*)(gr + c0);
                else {
                    v.x = gr[c0]; v.y = c0 + 1 < Cn ? gr[c0 + 1] : 0.f; v.z = c0 + 2 < Cn ? gr[c0 + 2] : 0.f; v.w = c0 + 3 < Cn ? gr[c0 + 3] : 0.f;
                }
                v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w;
                cacc[s].x += v.x; cacc[s].y += v.y; cacc[s].z += v.z; cacc[s].w += v.w;
                rsum += (v.x + v.y) + (v.z + v.w);
            }
        }
        rsum = wave_sum(rsum);
        if (lane == 0) ws.rowsum[rs_off + (long)b * R + r] = rsum;
        tot += rsum;
    }
    // combine the 4 waves' column partials, one 256-col segment at a time (fixed order)
    float* cp = ws.colpart + tb.unit_col_off[u];
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) {
        if (s * 256 < Cn) {
            __syncthreads();
            *(float4*)&colred[wave][lane * 4] = cacc[s];
            __syncthreads();
            const int c = s * 256 + threadIdx.x;
            if (threadIdx.x < 256 && c < Cn) {
                const int l4 = threadIdx.x;
                cp[c] = (colred[0][l4] + colred[1][l4]) + (colred[2][l4] + colred[3][l4]);
            }
        }
    }
    __syncthreads();
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) ws.unit_sumsq[u] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void v1_stats(Tab tb, Ws ws, const float* __restrict__ g) {
    __shared__ float red[4];
    const int v = blockIdx.x;
    const long off = tb.vec[v * 3 + 0];
    const int n = (int)tb.vec[v * 3 + 1];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float x = g[off + i]; s += x * x; }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) ws.vec_sumsq[v] = s;
}

// global L2 norm of all gradients + clip coefficient (clip_grad_norm_ semantics), one block, fixed order
__global__ __launch_bounds__(256) void k2_scalars(Tab tb, Ws ws, float max_norm) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < tb.U; i += 256) s += ws.unit_sumsq[i];
    for (int i = threadIdx.x; i < tb.V; i += 256) s += ws.vec_sumsq[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) {
        const float gn = sqrtf(s);
        ws.scalars[0] = gn;
        ws.scalars[1] = max_norm > 0.f ? fminf(max_norm / (gn + 1e-6f), 1.0f) : 1.0f;
    }
}

__global__ __launch_bounds__(256) void k2_finalize(Tab tb, Ws ws, float* __restrict__ state, float beta2t,
                                                   float eps1) {
    __shared__ float red[4];
    const int t = blockIdx.x, b = blockIdx.y;
    const long* tt = tb.ten + (long)t * 8;
    const int nb = (int)tt[1], R = (int)tt[2], Cn = (int)tt[3];
    if (b >= nb) return;
    const float c = ws.scalars[1];
    const float c2 = c * c, omb = 1.0f - beta2t;
    float* srow = state + tt[4] + (long)b * R;
    float* scol = state + tt[5] + (long)b * Cn;
    const float* rowsum = ws.rowsum + tt[4] + (long)b * R;
    float* rfac = ws.rfac + tt[4] + (long)b * R;
    float* cfac = ws.cfac + tt[5] + (long)b * Cn;
    // rows
    float rs = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
        const float nv = beta2t * srow[r] + omb * (c2 * rowsum[r] / (float)Cn + eps1);
        srow[r] = nv;
        rs += nv;
    }
    rs = block_sum<256>(rs, red);
    const float rmean = rs / (float)R;
    for (int r = threadIdx.x; r < R; r += 256) rfac[r] = rsqrtf(srow[r] / rmean);
    // columns: reduce this (tensor,batch)'s unit partials in unit order
    const int unit0 = (int)tt[6], nun = (int)tt[7];
    const int upb = nun / nb;                       // units per batch entry
    for (int cc = threadIdx.x; cc < Cn; cc += 256) {
        float s = 0.f;
        for (int k = 0; k < upb; ++k) s += ws.colpart[tb.unit_col_off[unit0 + b * upb + k] + cc];
        const float nv = beta2t * scol[cc] + omb * (c2 * s / (float)R + eps1);
        scol[cc] = nv;
        cfac[cc] = rsqrtf(nv);
    }
}

template <bool APPLY>
__global__ __launch_bounds__(256) void k34_update(Tab tb, Ws ws, const float* __restrict__ g, float* __restrict__ p,
                                                  float lr, float clip_thr) {
    __shared__ float red[4];
    const int u = blockIdx.x;
    const int t = tb.unit[u * 4 + 0], b = tb.unit[u * 4 + 1], row0 = tb.unit[u * 4 + 2], nrows = tb.unit[u * 4 + 3];
    const long* tt = tb.ten + (long)t * 8;
    const int nb = (int)tt[1], R = (int)tt[2], Cn = (int)tt[3];
    const long base = tt[0] + (long)b * R * Cn;
    const float* rfac = ws.rfac + tt[4] + (long)b * R;
    const float* cfac = ws.cfac + tt[5] + (long)b * Cn;
    const float c = ws.scalars[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec4 = (Cn % 4 == 0);
    float scale = 0.f, decay = 1.f;
    if (APPLY) {
        const int unit0 = (int)tt[6], nun = (int)tt[7];
        float s = 0.f;
        for (int k = 0; k < nun; ++k) s += ws.unit_usq[unit0 + k];       // fixed order, same in every block
        const float rms = sqrtf(s / ((float)nb * (float)R * (float)Cn));
        scale = lr / fmaxf(1.0f, rms / clip_thr);
        decay = 1.0f - tb.ten_wd[t] * lr;
    }
    float tot = 0.f;
    for (int rr = wave; rr < nrows; rr += 4) {
        const int r = row0 + rr;
        const float rf = rfac[r] * c;
        const float* gr = g + base + (long)r * Cn;
        float* pr = p + base + (long)r * Cn;
#pragma unroll 4
        for (int s = 0; s < MAXSEG; ++s) {
            const int c0 = s * 256 + lane * 4;
            if (c0 >= Cn) break;
            if (vec4) {
                const float4 gv = *(const float4*)(gr + c0);
                const float4 cf = *(const float4*)(cfac + c0);
                float4 uv = make_float4(gv.x * rf * cf.x, gv.y * rf * cf.y, gv.z * rf * cf.z, gv.w * rf * cf.w);
                if (APPLY) {
                    float4 pv = *(const float4*)(pr + c0);
                    pv.x = pv.x * decay - scale * uv.x; pv.y = pv.y * decay - scale * uv.y;
                    pv.z = pv.z * decay - scale * uv.z; pv.w = pv.w * decay - scale * uv.w;
                    *(float4*)(pr + c0) = pv;
                } else {
                    tot += (uv.x * uv.x + uv.y * uv.y) + (uv.z * uv.z + uv.w * uv.w);
                }
            } else {
                for (int e = 0; e < 4 && c0 + e < Cn; ++e) {
                    const float uv = gr[c0 + e] * rf * cfac[c0 + e];
                    if (APPLY) pr[c0 + e] = pr[c0 + e] * decay - scale * uv;
                    else tot += uv * uv;
                }
            }
        }
    }
    if (!APPLY) {
        tot = block_sum<256>(tot, red);
        if (threadIdx.x == 0) ws.unit_usq[u] = tot;
    }
}

__global__ __launch_bounds__(256) void v2_update(Tab tb, Ws ws, const float* __restrict__ g, float* __restrict__ p,
                                                 float* __restrict__ state, float beta2t, float eps1, float lr,
                                                 float clip_thr) {
    __shared__ float red[4];
    const int v = blockIdx.x;
    const long off = tb.vec[v * 3 + 0];
    const int n = (int)tb.vec[v * 3 + 1];
    float* sq = state + tb.vec[v * 3 + 2];
    const float c = ws.scalars[1];
    const float omb = 1.0f - beta2t;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gc = g[off + i] * c;
        const float nv = beta2t * sq[i] + omb * (gc * gc + eps1);
        sq[i] = nv;
        const float uv = gc * rsqrtf(nv);
        s += uv * uv;
    }
    s = block_sum<256>(s, red);
    const float rms = sqrtf(s / (float)n);
    const float scale = lr / fmaxf(1.0f, rms / clip_thr);
    const float decay = 1.0f - tb.vec_wd[v] * lr;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float uv = g[off + i] * c * rsqrtf(sq[i]);
        p[off + i] = p[off + i] * decay - scale * uv;
    }
}

}  // namespace

extern "C" size_t desta_adafactor_workspace_floats(int U, int V, int64_t sum_rows, int64_t sum_cols, int64_t colpart_floats) {
    return (size_t)(8 + 2 * (size_t)U + (size_t)V + 2 * (size_t)sum_rows + (size_t)sum_cols + (size_t)colpart_floats + 64);
}

extern "C" int desta_clip_adafactor_step(const desta_opt_plan* pl, float* params, const float* grads, float* state,
                                         float* workspace, float lr, float beta2t, float eps1, float clip_threshold,
                                         float max_grad_norm, void* stream) {
    DESTA_CHECK_ARG(pl && params && grads && state && workspace, "adafactor: null argument");
    DESTA_CHECK_ARG(pl->n_tensors >= 0 && pl->n_units >= 0 && pl->n_vec >= 0, "adafactor: bad plan");
    DESTA_CHECK_ARG(pl->max_cols <= MAXSEG * 256, "adafactor: factored tensor with %d columns > %d unsupported",
                    pl->max_cols, MAXSEG * 256);
    Tab tb;
    tb.ten = (const long*)pl->tensors; tb.ten_wd = pl->tensor_wd; tb.T = pl->n_tensors;
    tb.unit = pl->units; tb.unit_col_off = (const long*)pl->unit_col_off; tb.U = pl->n_units;
    tb.vec = (const long*)pl->vecs; tb.vec_wd = pl->vec_wd; tb.V = pl->n_vec;
    Ws ws;
    float* w = workspace;
    ws.scalars = w; w += 8;
    ws.unit_sumsq = w; w += tb.U;
    ws.unit_usq = w; w += tb.U;
    ws.vec_sumsq = w; w += tb.V;
    w += (4 - ((w - workspace) & 3)) & 3;
    ws.rowsum = w; w += pl->sum_rows;
    ws.rfac = w; w += pl->sum_rows;
    w += (4 - ((w - workspace) & 3)) & 3;
    ws.cfac = w; w += pl->sum_cols;
    w += (4 - ((w - workspace) & 3)) & 3;
    ws.colpart = w;
    hipStream_t st = (hipStream_t)stream;
    if (tb.U > 0) hipLaunchKernelGGL(k1_stats, dim3(tb.U), dim3(256), 0, st, tb, ws, grads);
    if (tb.V > 0) hipLaunchKernelGGL(v1_stats, dim3(tb.V), dim3(256), 0, st, tb, ws, grads);
    hipLaunchKernelGGL(k2_scalars, dim3(1), dim3(256), 0, st, tb, ws, max_grad_norm);
    if (tb.T > 0) {
        hipLaunchKernelGGL(k2_finalize, dim3(tb.T, pl->max_batch), dim3(256), 0, st, tb, ws, state, beta2t, eps1);
        hipLaunchKernelGGL(k34_update<false>, dim3(tb.U), dim3(256), 0, st, tb, ws, grads, params, lr, clip_threshold);
        hipLaunchKernelGGL(k34_update<true>, dim3(tb.U), dim3(256), 0, st, tb, ws, grads, params, lr, clip_threshold);
    }
    if (tb.V > 0) hipLaunchKernelGGL(v2_update, dim3(tb.V), dim3(256), 0, st, tb, ws, grads, params, state, beta2t, eps1, lr, clip_threshold);
    DESTA_CHECK_LAUNCH("clip_adafactor_step");
    return DESTA_OK;
}
