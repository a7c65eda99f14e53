// Fused global-norm clip + Adafactor over a FLAT fp32 parameter / gradient arena (gfx950, HBM-bound).
//
// Replaces, per optimizer step, `torch.nn.utils.clip_grad_norm_(params, 1.0)` (TF:trainer.py:1780-1782)
// followed by `transformers.optimization.Adafactor.step` (TF:optimization.py:1203-1294) with the
// HF-Trainer kwargs scale_parameter=False, relative_step=False, beta1=None
// (TF:trainer_optimizer.py:197): ~8 elementwise torch passes x ~230 tensors become THREE launches:
//
//   A  af_stats    : one pass over g: per-row sum g^2, per-(unit, col) partial sum g^2, per-unit sum g^2
//                    (1-D tensors: sum g^2 per tensor, extra blocks of the same grid)
//   B  af_finalize : every block re-derives the global norm / clip coefficient c from the unit sums (fixed order);
//                    row / column EMA update with c^2 * mean(g^2) + eps1; r / c factors
//   C  af_usq      : per 16384-element CHUNK of one tensor: u = c g r[row] c[col], sum u^2 of the chunk      (g read)
//   D  af_apply    : rms(u) of the WHOLE tensor from its chunk sums in chunk order, then
//                    p = p (1 - wd lr) - lr u / max(1, rms(u) / clip_thr)                     (g re-read, p read + write)
//      C and D run per GROUP of whole tensors (<= 64 MB of g): D's re-read of the group's g is served by the 256 MB
//      Infinity Cache, not HBM; D of group i and C of group i+1 share one launch.  Every thread issues all 16 float4 loads of its chunk before the first use
//      (64 KB in flight per block).  1-D tensors ride in the last D launch, one work item per vector.
//
// HBM traffic: g twice (A, C) + p read + p write = 16 N bytes (N = trainable fp32 values); the algorithm's floor
// with nothing cached is 12 N (g once) — the global gradient norm has to be known before ANY update can start
// (the EMA is not homogeneous in c), so g cannot be consumed in one pass unless the producers of the gradients
// emit the norm.  C / D walk the tensors in the REVERSE order of A, so the tail of A's stream is still in the
// Infinity Cache when C starts.  (Round 1 moved 20 N: every pass read g from HBM.)
//
// A fused C+D (persistent grid, chunk kept in registers, per-tensor arrival counter with memory-side atomics) was built
// and measured SLOWER (0.71-1.3 ms vs 0.50 for C + D): the blocks of one tensor move in lockstep through load / wait /
// apply, so the memory system idles at every hand-off; removed again.
//
// Every reduction has a fixed order (no float atomics), so all data-parallel ranks compute bit-identical updates from
// bit-identical all-reduced gradients.
#include "common.h"
#include "desta_hip.h"

namespace {

constexpr int UNIT_ROWS = 64;      // rows per stats unit (4 waves x 16 rows)
constexpr int MAXSEG = 16;         // 256-column segments per row handled in registers (cols <= 4096)
constexpr int CHUNK_V4 = 16;       // float4 per thread and chunk in the update kernel
constexpr int CHUNK = 256 * 4 * CHUNK_V4;
constexpr int NARROW = 16;         // ragged tensors with <= 16 columns: one thread per row in the unit kernels

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
    const int* chunk;              // [NC][4]: tensor, batch index, first element inside the [R, C] matrix, count
    const int* ten_chunk;          // [T][2]: first chunk (queue position), number of chunks
    int NC;
};

// workspace carve (floats): scalars[8] | unit_sumsq[U] | vec_sumsq[V] | chunk_usq[NC] | rowsum[SR] | rfac[SR] | cfac[SC] | colpart[...]
// scalars: [0] global grad norm  [1] clip coefficient  [2..7] unused
struct Ws {
    float* scalars; float* unit_sumsq; float* vec_sumsq; float* chunk_usq;
    float* rowsum; float* rfac; float* cfac; float* colpart;
};

// ------------------------------------------------------------------------------------------------ A: statistics
// ROWS rows per wave in flight (NSEG float4 loads each, <= 24 loads = 96 registers): the one-row-at-a-time loop of round 1
// ran at 2.3 TB/s
template <int NSEG, int ROWS>
__device__ __forceinline__ void stats_rows(const float* __restrict__ gt, int Cn, int row0, int nrows, int wave, int lane,
                                           float4 (&cacc)[MAXSEG], float* __restrict__ rowsum_out, float& tot) {
    for (int rr = wave * ROWS; rr < nrows; rr += 4 * ROWS) {
        float4 v[ROWS][NSEG];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const int r = min(rr + k, nrows - 1);                       // clamped rows are loaded but not counted
            const float* gr = gt + (long)(row0 + r) * Cn;
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                const int c0 = s * 256 + lane * 4;
                v[k][s] = (c0 < Cn) ? *(const float4*)(gr + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const bool live = rr + k < nrows;
            float rsum = 0.f;
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                float4 x = v[k][s];
                x.x *= x.x; x.y *= x.y; x.z *= x.z; x.w *= x.w;
                if (live) { cacc[s].x += x.x; cacc[s].y += x.y; cacc[s].z += x.z; cacc[s].w += x.w; }
                rsum += (x.x + x.y) + (x.z + x.w);
            }
            rsum = wave_sum(rsum);
            if (live) {
                if (lane == 0) rowsum_out[row0 + rr + k] = rsum;
                tot += rsum;
            }
        }
    }
}

__global__ __launch_bounds__(256) void af_stats(Tab tb, Ws ws, const float* __restrict__ g) {
    __shared__ float colred[4][256 * 4];           // one 256-col segment at a time, 4 waves
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= tb.U) {                 // 1-D tensor: sum g^2
        const int v = blockIdx.x - tb.U;
        const long off = tb.vec[v * 3 + 0];
        const int n = (int)tb.vec[v * 3 + 1];
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) { const float x = g[off + i]; s += x * x; }
        s = block_sum<256>(s, red);
        if (threadIdx.x == 0) ws.vec_sumsq[v] = s;
        return;
    }
    const int u = blockIdx.x;
    const int t = tb.unit[u * 4 + 0], b = tb.unit[u * 4 + 1], row0 = tb.unit[u * 4 + 2], nrows = tb.unit[u * 4 + 3];
    const long* tt = tb.ten + (long)t * 8;
    const int R = (int)tt[2], Cn = (int)tt[3];
    const float* gt = g + tt[0] + (long)b * R * Cn;
    float* rowsum_out = ws.rowsum + tt[4] + (long)b * R;
    float4 cacc[MAXSEG];
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) cacc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    float tot = 0.f;
    if (Cn % 4 == 0) {
        const int nseg = (Cn + 255) / 256;
        if (nseg <= 5) stats_rows<5, 4>(gt, Cn, row0, nrows, wave, lane, cacc, rowsum_out, tot);
        else if (nseg <= 12) stats_rows<12, 2>(gt, Cn, row0, nrows, wave, lane, cacc, rowsum_out, tot);
        else stats_rows<MAXSEG, 1>(gt, Cn, row0, nrows, wave, lane, cacc, rowsum_out, tot);
    } else if (Cn <= NARROW) {
        // a few columns per row (Conv1d weight [out, in, k]: k columns): one THREAD per row — a wave reads 64 consecutive rows, and the
        // column sums are k block reductions at the end (fixed order)
        float ca[NARROW];
#pragma unroll
        for (int c = 0; c < NARROW; ++c) ca[c] = 0.f;
        for (int rr = threadIdx.x; rr < nrows; rr += 256) {
            const float* gr = gt + (long)(row0 + rr) * Cn;
            float rsum = 0.f;
#pragma unroll
            for (int c = 0; c < NARROW; ++c) {
                if (c < Cn) { float x = gr[c]; x *= x; ca[c] += x; rsum += x; }
            }
            rowsum_out[row0 + rr] = rsum;
            tot += rsum;
        }
        float* cpn = ws.colpart + tb.unit_col_off[u];
#pragma unroll
        for (int c = 0; c < NARROW; ++c) {
            if (c < Cn) {
                const float s = block_sum<256>(ca[c], red);
                if (threadIdx.x == 0) cpn[c] = s;
                __syncthreads();
            }
        }
        tot = block_sum<256>(tot, red);
        if (threadIdx.x == 0) ws.unit_sumsq[u] = tot;
        return;
    } else {                                        // ragged rows: scalar loads
        for (int rr = wave; rr < nrows; rr += 4) {
            const float* gr = gt + (long)(row0 + rr) * Cn;
            float rsum = 0.f;
#pragma unroll
            for (int s = 0; s < MAXSEG; ++s) {
                if (s * 256 >= Cn) continue;        // (a Conv1d weight has 3..7 columns: one segment)
                const int c0 = s * 256 + lane * 4;
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = c0 + e < Cn ? gr[c0 + e] : 0.f; x[e] *= x[e]; }
                cacc[s].x += x[0]; cacc[s].y += x[1]; cacc[s].z += x[2]; cacc[s].w += x[3];
                rsum += (x[0] + x[1]) + (x[2] + x[3]);
            }
            rsum = wave_sum(rsum);
            if (lane == 0) rowsum_out[row0 + rr] = rsum;
            tot += rsum;
        }
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
            if (c < Cn) {
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

// global L2 norm of all gradients + clip coefficient (clip_grad_norm_ semantics); the same fixed-order sum in every block
__device__ __forceinline__ float clip_coef(const Tab& tb, const Ws& ws, float max_norm, float* red, float* gn_out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < tb.U; i += 256) s += ws.unit_sumsq[i];
    for (int i = threadIdx.x; i < tb.V; i += 256) s += ws.vec_sumsq[i];
    s = block_sum<256>(s, red);
    const float gn = sqrtf(s);
    if (gn_out) *gn_out = gn;
    return max_norm > 0.f ? fminf(max_norm / (gn + 1e-6f), 1.0f) : 1.0f;
}

// ------------------------------------------------------------------------------------------------ B: factors
// grid = sum over (tensor, batch) of (1 row block + ceil(C / 256) column blocks), resolved through `fin` [n][3]:
// tensor, batch, part (0 = rows, k >= 1 = columns [256 (k-1), 256 k))
__global__ __launch_bounds__(256) void af_finalize(Tab tb, Ws ws, const int* __restrict__ fin, float* __restrict__ state,
                                                   float beta2t, float eps1, float max_norm) {
    __shared__ float red[4];
    float gn;
    const float c = clip_coef(tb, ws, max_norm, red, &gn);
    if (blockIdx.x == 0 && threadIdx.x == 0) { ws.scalars[0] = gn; ws.scalars[1] = c; }
    const int t = fin[blockIdx.x * 3 + 0], b = fin[blockIdx.x * 3 + 1], part = fin[blockIdx.x * 3 + 2];
    const long* tt = tb.ten + (long)t * 8;
    const int nb = (int)tt[1], R = (int)tt[2], Cn = (int)tt[3];
    const float c2 = c * c, omb = 1.0f - beta2t;
    if (part == 0) {
        float* srow = state + tt[4] + (long)b * R;
        const float* rowsum = ws.rowsum + tt[4] + (long)b * R;
        float* rfac = ws.rfac + tt[4] + (long)b * R;
        float rs = 0.f;
        for (int r = threadIdx.x; r < R; r += 256) {
            const float nv = beta2t * srow[r] + omb * (c2 * rowsum[r] / (float)Cn + eps1);
            srow[r] = nv;
            rs += nv;
        }
        __syncthreads();
        rs = block_sum<256>(rs, red);
        const float rmean = rs / (float)R;
        for (int r = threadIdx.x; r < R; r += 256) rfac[r] = rsqrtf(srow[r] / rmean);
    } else {
        // columns: reduce this (tensor, batch)'s unit partials in unit order
        float* scol = state + tt[5] + (long)b * Cn;
        float* cfac = ws.cfac + tt[5] + (long)b * Cn;
        const int unit0 = (int)tt[6], upb = (int)tt[7] / nb;
        const int cc = (part - 1) * 256 + threadIdx.x;
        if (cc < Cn) {
            float s = 0.f;
            for (int k = 0; k < upb; ++k) s += ws.colpart[tb.unit_col_off[unit0 + b * upb + k] + cc];
            const float nv = beta2t * scol[cc] + omb * (c2 * s / (float)R + eps1);
            scol[cc] = nv;
            cfac[cc] = rsqrtf(nv);
        }
    }
}

// ------------------------------------------------------------------------------------------------ C: update
__device__ __forceinline__ void vec_item(const Tab& tb, const Ws& ws, int v, const float* __restrict__ g, float* __restrict__ p,
                                         float* __restrict__ state, float c, float beta2t, float eps1, float lr, float clip_thr,
                                         float* red) {
    const long off = tb.vec[v * 3 + 0];
    const int n = (int)tb.vec[v * 3 + 1];
    float* sq = state + tb.vec[v * 3 + 2];
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
    for (int i = threadIdx.x; i < n; i += 256) {    // each thread re-reads exactly the sq[i] it wrote
        const float uv = g[off + i] * c * rsqrtf(sq[i]);
        p[off + i] = p[off + i] * decay - scale * uv;
    }
}

// u of one chunk into registers: all loads first, then the factors (L2-resident)
__device__ __forceinline__ float chunk_u(const Tab& tb, const Ws& ws, const float* __restrict__ g, int item, float c,
                                         float4 (&u)[CHUNK_V4], int& t, long& pbase, int& cnt) {
    t = tb.chunk[item * 4 + 0];
    const int b = tb.chunk[item * 4 + 1], e0 = tb.chunk[item * 4 + 2];
    cnt = tb.chunk[item * 4 + 3];
    const long* tt = tb.ten + (long)t * 8;
    const int R = (int)tt[2], Cn = (int)tt[3];
    const long base = tt[0] + (long)b * R * Cn;
    const float* rfac = ws.rfac + tt[4] + (long)b * R;
    const float* cfac = ws.cfac + tt[5] + (long)b * Cn;
    pbase = base + e0;
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < CHUNK_V4; ++i) {
        const int e = (i * 256 + threadIdx.x) * 4;
        u[i] = e < cnt ? *(const float4*)(g + base + e0 + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < CHUNK_V4; ++i) {
        if ((i & 3) == 0) __builtin_amdgcn_sched_barrier(0);        // at most 4 factor loads in flight (register budget)
        const int e = (i * 256 + threadIdx.x) * 4;
        if (e < cnt) {
            const int idx = e0 + e, r = idx / Cn, cc = idx - r * Cn;
            const float rf = rfac[r] * c;
            const float4 cf = *(const float4*)(cfac + cc);
            u[i].x *= rf * cf.x; u[i].y *= rf * cf.y; u[i].z *= rf * cf.z; u[i].w *= rf * cf.w;
            tot += (u[i].x * u[i].x + u[i].y * u[i].y) + (u[i].z * u[i].z + u[i].w * u[i].w);
        }
    }
    return tot;
}

// C: sum u^2 per chunk (rows are multiples of 4 floats: the host routes plans with ragged rows to the unit-based kernels)
__global__ __launch_bounds__(256, 3) void af_usq(Tab tb, Ws ws, const float* __restrict__ g, int chunk0) {
    __shared__ float red[4];
    const int item = chunk0 + blockIdx.x;
    float4 u[CHUNK_V4];
    int t, cnt; long pbase;
    float tot = chunk_u(tb, ws, g, item, ws.scalars[1], u, t, pbase, cnt);
    tot = block_sum<256>(tot, red);
    if (threadIdx.x == 0) ws.chunk_usq[item] = tot;
}

// D: apply of one group; the blocks past its chunks run C (sum u^2) of the NEXT group in the same launch (independent work: one
// launch boundary less per group and twice the blocks to fill the chip), and after those the 1-D tensors (last group only)
__global__ __launch_bounds__(256, 3) void af_apply(Tab tb, Ws ws, const float* __restrict__ g, float* __restrict__ p,
                                                   float* __restrict__ state, float beta2t, float eps1, float lr, float clip_thr,
                                                   int chunk0, int nchunks, int usq_chunk0, int usq_n) {
    __shared__ float red[4];
    const float c = ws.scalars[1];
    if ((int)blockIdx.x >= nchunks) {
        const int j = blockIdx.x - nchunks;
        if (j < usq_n) {
            float4 u[CHUNK_V4];
            int t, cnt; long pbase;
            float tot = chunk_u(tb, ws, g, usq_chunk0 + j, c, u, t, pbase, cnt);
            tot = block_sum<256>(tot, red);
            if (threadIdx.x == 0) ws.chunk_usq[usq_chunk0 + j] = tot;
            return;
        }
        vec_item(tb, ws, j - usq_n, g, p, state, c, beta2t, eps1, lr, clip_thr, red);
        return;
    }
    const int item = chunk0 + blockIdx.x;
    float4 u[CHUNK_V4];
    int t, cnt; long pbase;
    (void)chunk_u(tb, ws, g, item, c, u, t, pbase, cnt);
    const long* tt = tb.ten + (long)t * 8;
    const int c0 = tb.ten_chunk[t * 2 + 0], nch = tb.ten_chunk[t * 2 + 1];
    float s = 0.f;
    for (int k = threadIdx.x; k < nch; k += 256) s += ws.chunk_usq[c0 + k];           // same partition + order in every block
    s = block_sum<256>(s, red);
    const float rms = sqrtf(s / ((float)tt[1] * (float)tt[2] * (float)tt[3]));
    const float scale = lr / fmaxf(1.0f, rms / clip_thr);
    const float decay = 1.0f - tb.ten_wd[t] * lr;
    float* pt = p + pbase;
    // p in two halves of 8 float4 per thread (32 KB in flight per block; u stays in registers: 168-register budget)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 pv[CHUNK_V4 / 2];
#pragma unroll
        for (int j = 0; j < CHUNK_V4 / 2; ++j) {
            const int e = ((h * (CHUNK_V4 / 2) + j) * 256 + threadIdx.x) * 4;
            if (e < cnt) pv[j] = *(const float4*)(pt + e);
        }
#pragma unroll
        for (int j = 0; j < CHUNK_V4 / 2; ++j) {
            const int i = h * (CHUNK_V4 / 2) + j;
            const int e = (i * 256 + threadIdx.x) * 4;
            if (e < cnt) {
                pv[j].x = pv[j].x * decay - scale * u[i].x; pv[j].y = pv[j].y * decay - scale * u[i].y;
                pv[j].z = pv[j].z * decay - scale * u[i].z; pv[j].w = pv[j].w * decay - scale * u[i].w;
                *(float4*)(pt + e) = pv[j];
            }
        }
    }
}

// sum u^2 of a whole tensor from its unit sums, in place: unit_usq[first unit] <- total (fixed order: thread-strided partials, then
// the block tree).  One block per tensor (`t0 + blockIdx.x`).  [Rounds 1-3 summed the units serially in EVERY block of the apply
// kernel: quadratic in the unit count, 2.3 s for a [4096, 4096, 5] Conv1d weight = 262 144 units.]
__global__ __launch_bounds__(256) void k34_totals(Tab tb, int t0, float* __restrict__ unit_usq) {
    __shared__ float red[4];
    const long* tt = tb.ten + (long)(t0 + blockIdx.x) * 8;
    const int unit0 = (int)tt[6], nun = (int)tt[7];
    float s = 0.f;
    for (int k = threadIdx.x; k < nun; k += 256) s += unit_usq[unit0 + k];
    s = block_sum<256>(s, red);
    __syncthreads();
    if (threadIdx.x == 0) unit_usq[unit0] = s;
}

__global__ __launch_bounds__(256) void k34_totals_range(int unit0, int nun, float* __restrict__ unit_usq) {
    __shared__ float red[4];
    float s = 0.f;
    for (int k = threadIdx.x; k < nun; k += 256) s += unit_usq[unit0 + k];
    s = block_sum<256>(s, red);
    __syncthreads();
    if (threadIdx.x == 0) unit_usq[unit0] = s;
}

// unit-based update (round-1 structure): tensors with ragged rows (cols % 4 != 0); unit = unit_base + blockIdx.x
template <bool APPLY>
__global__ __launch_bounds__(256) void k34_update(Tab tb, Ws ws, const float* __restrict__ g, float* __restrict__ p,
                                                  float lr, float clip_thr, float* __restrict__ unit_usq, int unit_base) {
    __shared__ float red[4];
    const int u = unit_base + blockIdx.x;
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
        const float s = unit_usq[(int)tt[6]];                             // the tensor's total (k34_totals)
        const float rms = sqrtf(s / ((float)nb * (float)R * (float)Cn));
        scale = lr / fmaxf(1.0f, rms / clip_thr);
        decay = 1.0f - tb.ten_wd[t] * lr;
    }
    float tot = 0.f;
    if (!vec4 && Cn <= NARROW) {                    // one thread per row (see af_stats)
        float cf[NARROW];
#pragma unroll
        for (int cc = 0; cc < NARROW; ++cc) cf[cc] = cc < Cn ? cfac[cc] : 0.f;
        for (int rr = threadIdx.x; rr < nrows; rr += 256) {
            const int r = row0 + rr;
            const float rf = rfac[r] * c;
            const float* gr = g + base + (long)r * Cn;
            float* pr = p + base + (long)r * Cn;
#pragma unroll
            for (int cc = 0; cc < NARROW; ++cc) {
                if (cc < Cn) {
                    const float uv = gr[cc] * rf * cf[cc];
                    if (APPLY) pr[cc] = pr[cc] * decay - scale * uv;
                    else tot += uv * uv;
                }
            }
        }
        if (!APPLY) {
            tot = block_sum<256>(tot, red);
            if (threadIdx.x == 0) unit_usq[u] = tot;
        }
        return;
    }
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
        if (threadIdx.x == 0) unit_usq[u] = tot;
    }
}

__global__ __launch_bounds__(256) void v2_update(Tab tb, Ws ws, const float* __restrict__ g, float* __restrict__ p,
                                                 float* __restrict__ state, float beta2t, float eps1, float lr, float clip_thr) {
    __shared__ float red[4];
    vec_item(tb, ws, blockIdx.x, g, p, state, ws.scalars[1], beta2t, eps1, lr, clip_thr, red);
}

}  // namespace

// floats: scalars | unit_sumsq[U] | vec_sumsq[V] | chunk_usq[NC] (>= U: the two-launch path keeps unit sums there) | rowsum | rfac | cfac |
// colpart
extern "C" size_t desta_adafactor_workspace_floats(int U, int V, int64_t sum_rows, int64_t sum_cols, int64_t colpart_floats) {
    return (size_t)(8 + 2 * (size_t)U + (size_t)V + 2 * (size_t)sum_rows + (size_t)sum_cols + (size_t)colpart_floats + 64);
}
extern "C" size_t desta_adafactor_workspace_floats_v3(const desta_opt_plan* pl, int64_t colpart_floats) {
    const size_t nc = (size_t)(pl->n_chunks > pl->n_units ? pl->n_chunks : pl->n_units);
    return 8 + (size_t)pl->n_units + (size_t)pl->n_vec + nc + 2 * (size_t)pl->sum_rows + (size_t)pl->sum_cols +
           (size_t)colpart_floats + 64;
}

extern "C" int desta_clip_adafactor_step(const desta_opt_plan* pl, float* params, const float* grads, float* state,
                                         float* workspace, float lr, float beta2t, float eps1, float clip_threshold,
                                         float max_grad_norm, void* stream) {
    DESTA_CHECK_ARG(pl && params && grads && state && workspace, "adafactor: null argument");
    DESTA_CHECK_ARG(pl->n_tensors >= 0 && pl->n_units >= 0 && pl->n_vec >= 0, "adafactor: bad plan");
    DESTA_CHECK_ARG(pl->max_cols <= MAXSEG * 256, "adafactor: factored tensor with %d columns > %d unsupported",
                    pl->max_cols, MAXSEG * 256);
    DESTA_CHECK_ARG(pl->n_tensors == 0 || (pl->chunks && pl->ten_chunks && pl->fin && pl->n_fin > 0 && (pl->n_chunks > 0 || pl->n_ragged > 0)),
                    "adafactor: plan without chunk / finalize tables (ABI 3)");
    DESTA_CHECK_ARG(pl->n_ragged >= 0 && (pl->n_ragged == 0 || pl->ragged_units), "adafactor: bad ragged-tensor table");
    Tab tb;
    tb.ten = (const long*)pl->tensors; tb.ten_wd = pl->tensor_wd; tb.T = pl->n_tensors;
    tb.unit = pl->units; tb.unit_col_off = (const long*)pl->unit_col_off; tb.U = pl->n_units;
    tb.vec = (const long*)pl->vecs; tb.vec_wd = pl->vec_wd; tb.V = pl->n_vec;
    tb.chunk = pl->chunks; tb.ten_chunk = pl->ten_chunks; tb.NC = pl->n_chunks;
    const size_t nc = (size_t)(tb.NC > tb.U ? tb.NC : tb.U);
    Ws ws;
    float* w = workspace;
    ws.scalars = w; w += 8;
    ws.unit_sumsq = w; w += tb.U;
    ws.vec_sumsq = w; w += tb.V;
    ws.chunk_usq = w; w += nc;
    w += (4 - ((w - workspace) & 3)) & 3;
    ws.rowsum = w; w += pl->sum_rows;
    ws.rfac = w; w += pl->sum_rows;
    w += (4 - ((w - workspace) & 3)) & 3;
    ws.cfac = w; w += pl->sum_cols;
    w += (4 - ((w - workspace) & 3)) & 3;
    ws.colpart = w;
    hipStream_t st = (hipStream_t)stream;
    DESTA_CHECK_ARG(tb.T > 0, "adafactor: a plan without any factored (>= 2-D) tensor is not supported");
    hipLaunchKernelGGL(af_stats, dim3(tb.U + tb.V), dim3(256), 0, st, tb, ws, grads);
    hipLaunchKernelGGL(af_finalize, dim3(pl->n_fin), dim3(256), 0, st, tb, ws, pl->fin, state, beta2t, eps1, max_grad_norm);
    // tensors with ragged rows (ABI 7): unit kernels over their own unit ranges; their unit sums live in chunk_usq[first unit ...]
    // while the chunk sums of the other tensors live in chunk_usq[chunk index]: disjoint in TIME (this loop ends before af_usq starts)
    for (int r = 0; r < pl->n_ragged; ++r) {
        const int u0 = pl->ragged_units[2 * r], n = pl->ragged_units[2 * r + 1];
        DESTA_CHECK_ARG(u0 >= 0 && n > 0 && u0 + n <= tb.U, "adafactor: ragged unit range [%d, %d) outside [0, %d)", u0, u0 + n, tb.U);
        hipLaunchKernelGGL(k34_update<false>, dim3(n), dim3(256), 0, st, tb, ws, grads, params, lr, clip_threshold, ws.chunk_usq, u0);
        hipLaunchKernelGGL(k34_totals_range, dim3(1), dim3(256), 0, st, u0, n, ws.chunk_usq);
        hipLaunchKernelGGL(k34_update<true>, dim3(n), dim3(256), 0, st, tb, ws, grads, params, lr, clip_threshold, ws.chunk_usq, u0);
    }
    if (pl->n_ragged > 0 && tb.NC == 0) {
        if (tb.V > 0) hipLaunchKernelGGL(v2_update, dim3(tb.V), dim3(256), 0, st, tb, ws, grads, params, state, beta2t, eps1, lr, clip_threshold);
    } else if (pl->cols_multiple_of_4) {
        DESTA_CHECK_ARG(pl->group_bounds && pl->n_groups > 0 && pl->group_bounds[0] == 0 && pl->group_bounds[pl->n_groups] == tb.NC,
                        "adafactor: bad group table");
        for (int gi = 0; gi < pl->n_groups; ++gi) {
            const int c0 = pl->group_bounds[gi], n = pl->group_bounds[gi + 1] - c0;
            const bool last = gi == pl->n_groups - 1;
            const int nv = last ? tb.V : 0;
            const int c1 = last ? 0 : pl->group_bounds[gi + 1], n1 = last ? 0 : pl->group_bounds[gi + 2] - c1;
            DESTA_CHECK_ARG(n > 0, "adafactor: empty group");
            if (gi == 0) hipLaunchKernelGGL(af_usq, dim3(n), dim3(256), 0, st, tb, ws, grads, c0);
            // apply(group gi) + sum u^2 of group gi+1 (whole tensors per group: a tensor's chunk sums are complete before its apply)
            hipLaunchKernelGGL(af_apply, dim3(n + n1 + nv), dim3(256), 0, st, tb, ws, grads, params, state, beta2t, eps1, lr, clip_threshold,
                               c0, n, c1, n1);
        }
    } else {
        DESTA_CHECK_ARG(pl->n_ragged == 0, "adafactor: a ragged-tensor table needs cols_multiple_of_4 = 1 for the tensors with chunks");
        hipLaunchKernelGGL(k34_update<false>, dim3(tb.U), dim3(256), 0, st, tb, ws, grads, params, lr, clip_threshold, ws.chunk_usq, 0);
        hipLaunchKernelGGL(k34_totals, dim3(tb.T), dim3(256), 0, st, tb, 0, ws.chunk_usq);
        hipLaunchKernelGGL(k34_update<true>, dim3(tb.U), dim3(256), 0, st, tb, ws, grads, params, lr, clip_threshold, ws.chunk_usq, 0);
        if (tb.V > 0) hipLaunchKernelGGL(v2_update, dim3(tb.V), dim3(256), 0, st, tb, ws, grads, params, state, beta2t, eps1, lr, clip_threshold);
    }
    DESTA_CHECK_LAUNCH("clip_adafactor_step");
    return DESTA_OK;
}
