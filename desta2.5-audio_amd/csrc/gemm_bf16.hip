// bf16 "NT" GEMM on MFMA for gfx950:  C[M,N] = epilogue( A[M,K] · B[N,K]^T ),  fp32 accumulate.
//
// Every dense contraction of the DeSTA2.5 step is routed to this one kernel family: nn.Linear
// forward (B = weight [out,in]), dX backward (B = pre-transposed weight copy, kept resident in HBM),
// dW backward (A = dY^T, B = X^T), and the Whisper conv stem as a zero-copy im2col (overlapping rows:
// lda < K).  Replaces the torch/rocBLAS calls behind `nn.Linear` / `nn.Conv1d` at
// modeling_desta25.py:563-606, TF:models/whisper/modeling_whisper.py:279-330,
// TF:models/bert/modeling_bert.py:354-416, TF:models/llama/modeling_llama.py:163-281,480.
//
// Structure: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// A/B tiles go HBM -> LDS with `global_load_lds` (16 B/lane, no VGPR round trip), double buffered.
// LDS image rows are 128 B; the 16-B chunk index is XOR-swizzled with (row & 7) so the
// ds_read_b128 fragment reads are bank-conflict free; because the LDS-DMA destination is
// lane-linear the swizzle is applied to the per-lane SOURCE address and again on the read.
// MFMA operands are swapped (mfma(Bfrag, Afrag)) so a lane ends with 4 consecutive N outputs of one
// row and stores them as one 8-B (bf16) / 16-B (f32) access.
#include "common.h"
#include "desta_hip.h"
#include <math.h>

#ifndef GEMM_GROUP_M
#define GEMM_GROUP_M 8        // tile-group height of the 256x256 kernel (8 vs 4: gate_up +2-3 %, others equal; tools/gemm_bench.py)
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;                 // 16 KiB per operand tile

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    long sA, sB, sC;                                    // batch strides (elements)
    const float* bias;                                  // [N] or null
    const void* res; long ldr; long sR; int res_f32;    // residual added AFTER the activation
    int act;                                            // 0 none, 1 gelu(erf)
    int out_f32;
    bf16_t* preact; long ldp; long sP;                  // optional copy of (acc+bias) before act
    float alpha;                                        // scales the accumulator before bias
    bf16_t* aux; long lda_x;                            // act 2: SwiGLU output [M,N/2]; act 3: saved gate|up input [M,2N]
    unsigned drop_thresh, seed_lo, seed_hi; float drop_scale;   // dropout on (acc+bias) after act, before residual (0 = off)
    int tilesM, tilesN;
    int full_tiles, split;                              // 256-kernel: tiles [0,full) whole-K; the rest in `split` K-slices
    float* ws;                                          // fp32 partial slabs [(tile-full)*split + slice][256][256]
    int* tickets;                                       // per split tile: arrivals of its K-slices (in-kernel reduction) or null
    const float* rms_w; float rms_eps;                  // skinny kernel: RMSNorm(A rows; weight rms_w) applied on the fly
    const float* rope_cs; const int* rope_pos; int rope_cols, rope_hd;   // rotary embedding of output columns [0, rope_cols) in adjacent pairs
    int tail_skip;                                      // 256-kernel, 2-phase schedule: past the last K-tile the half-tile stream STOPS (counted waits shrink) instead of re-loading dead slots (option 10)
};

// rotary embedding of 4 consecutive outputs (two adjacent pairs) of row m, columns n0 .. n0 + 3 (see desta_gemm_desc.rope_*)
__device__ __forceinline__ void rope4(const GemmArgs& p, int m, int n0, float (&v)[4]) {
    if (p.rope_cs && n0 < p.rope_cols) {
        const int pos = p.rope_pos[m];
        const float4 cs = *(const float4*)(p.rope_cs + ((long)pos * (p.rope_hd >> 1) + ((n0 & (p.rope_hd - 1)) >> 1)) * 2);
        const float a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
        v[0] = a0 * cs.x - b0 * cs.y; v[1] = b0 * cs.x + a0 * cs.y;
        v[2] = a1 * cs.z - b1 * cs.w; v[3] = b1 * cs.z + a1 * cs.w;
    }
}

// bias + GELU of 4 adjacent outputs destined for a bf16 store: packed polynomial form (common.h), or the A&S form (option 9 = 0)
__device__ __forceinline__ void gelu4_bf16(float (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_fast(v[e]);
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// epilogue for 4 consecutive N outputs of row m: alpha, bias, pre-activation copy, GELU, residual, store
// ROPE: compiled in only where a rotary-epilogue GEMM can arrive without taking the wide-store path (128x128 kernels, split-K
// fix-up / in-kernel reduce); inside the 256x256 kernels' generic loops the extra body made hipcc give up the full unroll and
// keep the 128 accumulators in SCRATCH (528 B per thread: WRITE_SIZE 106 -> 472 MB per launch) — rotary GEMMs satisfy the
// wide-store conditions there (checked on the host).
template <bool ROPE = false>
__device__ __forceinline__ void epilogue4(const GemmArgs& p, int z, int m, int n0, const f32x4& a) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = a[e] * p.alpha;
    if (p.bias) {
        const float4 b = *(const float4*)(p.bias + n0);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if constexpr (ROPE) rope4(p, m, n0, v);
    if (p.preact) {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *(u16x4*)(p.preact + (long)z * p.sP + (long)m * p.ldp + n0) = o;
    }
    if (p.act == 1) {
        if (p.out_f32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else {                                           // bf16 store: the approximation error sits far below the rounding step
            gelu4_bf16(v);
        }
    }
    if (p.drop_thresh) {
        const unsigned long base = ((unsigned long)z * p.M + m) * p.N + n0;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = desta_keep(p.seed_lo, p.seed_hi, base + e, p.drop_thresh) ? v[e] * p.drop_scale : 0.f;
    }
    if (p.res) {
        if (p.res_f32) {
            const float4 r = *(const float4*)((const float*)p.res + (long)z * p.sR + (long)m * p.ldr + n0);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        } else {
            const u16x4 r = *(const u16x4*)((const bf16_t*)p.res + (long)z * p.sR + (long)m * p.ldr + n0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(r[e]);
        }
    }
    if (p.out_f32) {
        *(float4*)((float*)p.C + (long)z * p.sC + (long)m * p.ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *(u16x4*)((bf16_t*)p.C + (long)z * p.sC + (long)m * p.ldc + n0) = o;
    }
}

// out[m, n0 .. n0+4) (fp32) = alpha * acc + bias + residual (fp32): the Whisper encoder's out-proj / fc2 (fp32 residual stream), with
// no run-time feature tests (the caller decides once per tile)
__device__ __forceinline__ void epilogue4_f32_stream(const GemmArgs& p, int z, int m, int n0, const f32x4& a) {
    const float4 b = *(const float4*)(p.bias + n0);
    const float4 r = *(const float4*)((const float*)p.res + (long)z * p.sR + (long)m * p.ldr + n0);
    *(float4*)((float*)p.C + (long)z * p.sC + (long)m * p.ldc + n0) =
        make_float4((a[0] * p.alpha + b.x) + r.x, (a[1] * p.alpha + b.y) + r.y, (a[2] * p.alpha + b.z) + r.z, (a[3] * p.alpha + b.w) + r.w);
}

// Fast bf16 epilogue for two horizontally adjacent 16x16 tiles (bias / GELU / bf16 or fp32 residual):
// v_permlane16_swap exchanges the odd 16-lane rows of tile j with the even rows of tile j+1, after which
// every lane owns 8 CONTIGUOUS outputs of one row -> one 16-B store instead of two 8-B stores
// (half the store instructions of the epilogue, which is issue-bound with one block per CU).
// MODE: what the tile needs, decided ONCE per tile by the caller (bits: 1 bias, 2 GELU, 4 rotary, 8 residual); MODE < 0 keeps every
// test at run time.  [Until round 4 every one of the 16 calls per wave tested bias / act / rope / residual itself: the 256x256
// kernel's epilogue is ~30 000 instructions of mostly skipped code, and each feature added to it (rotary in round 3, a second GELU
// form in round 4) made the PLAIN path slower — 4.7 -> 8 -> 14.5 us per tile, i.e. 104 -> 130 us on the Whisper q|k|v shape.]
template <int MODE = -1>
__device__ __forceinline__ void epilogue_pair_bf16(const GemmArgs& p, int z, int m, bool row_ok, int ncol0, int fq,
                                                   const f32x4& a0, const f32x4& a1) {
    constexpr bool DYN = MODE < 0;
    const bool has_bias = DYN ? p.bias != nullptr : (MODE & 1) != 0;
    const bool has_gelu = DYN ? p.act == 1 : (MODE & 2) != 0;
    const bool has_rope = DYN ? true : (MODE & 4) != 0;                    // (rope4 tests the pointer itself)
    const bool has_res = DYN ? p.res != nullptr : (MODE & 8) != 0;
    const bool has_pre = DYN ? p.preact != nullptr : (MODE & 16) != 0;     // copy of the output in front of the (absent) activation: Qwen3's q|k|v before the q/k-norm
    if (has_res && !p.res_f32 && (p.ldr & 7) == 0 && (p.sR & 7) == 0) {
        // bf16 residual: swap the fp32 values first, so the lane's 8 contiguous outputs take ONE 16-byte residual load
        // (instead of two 8-byte loads before the swap); same arithmetic: add in fp32, round once.
        float v[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4& a = t ? a1 : a0;
            const int n0 = ncol0 + t * 16 + fq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][e] = a[e] * p.alpha;
            if (has_bias) {
                const float4 b = *(const float4*)(p.bias + n0);
                v[t][0] += b.x; v[t][1] += b.y; v[t][2] += b.z; v[t][3] += b.w;
            }
            if (has_gelu) gelu4_bf16(v[t]);
        }
        const int col = ncol0 + (fq & 1) * 16 + (fq >> 1) * 8;
        const u16x8 r = *(const u16x8*)((const bf16_t*)p.res + (long)z * p.sR + (long)m * p.ldr + col);
        float w[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0][e]), __float_as_uint(v[1][e]), false, false);
            // after the swap: first result = columns col + e (e < 4: from tile 0/1 half), second = col + 4 + e
            w[e] = __uint_as_float(sw[0]);
            w[4 + e] = __uint_as_float(sw[1]);
        }
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(w[e] + bf2f(r[e]));
        if (row_ok) *(u16x8*)((bf16_t*)p.C + (long)z * p.sC + (long)m * p.ldc + col) = o;
        return;
    }
    unsigned pk[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4& a = t ? a1 : a0;
        const int n0 = ncol0 + t * 16 + fq * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e] * p.alpha;
        if (has_bias) {
            const float4 b = *(const float4*)(p.bias + n0);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (has_rope) rope4(p, min(m, p.M - 1), n0, v);
        if (has_gelu) gelu4_bf16(v);
        if (has_res) {
            if (p.res_f32) {
                const float4 r = *(const float4*)((const float*)p.res + (long)z * p.sR + (long)m * p.ldr + n0);
                v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
            } else {
                const u16x4 r = *(const u16x4*)((const bf16_t*)p.res + (long)z * p.sR + (long)m * p.ldr + n0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bf2f(r[e]);
            }
        }
        pk[t][0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        pk[t][1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    }
    // rows (= lane >> 4) after the swap: X' = [X0, Y0, X2, Y2], Y' = [X1, Y1, X3, Y3]
    const auto lo = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
    const int col = ncol0 + (fq & 1) * 16 + (fq >> 1) * 8;
    if (row_ok) {
        const uint4 o = make_uint4(lo[0], hi[0], lo[1], hi[1]);
        *(uint4*)((bf16_t*)p.C + (long)z * p.sC + (long)m * p.ldc + col) = o;
        if (has_pre) *(uint4*)(p.preact + (long)z * p.sP + (long)m * p.ldp + col) = o;
    }
}

// ---- fused SwiGLU epilogues over the BLOCKED gate|up layout (act 2 / act 3; round 4) ------------------------------------------
// Column n of the gate|up projection: 64-column block b = n / 64 holds gate_{32b .. 32b+31} in its first 32 columns and
// up_{32b .. 32b+31} in its last 32 (the frozen weight rows are permuted once at load).  A wave of every tile kernel here owns
// whole 64-column blocks — the four adjacent 16x16 accumulators a0..a3 of one row group are (gate, gate, up, up) of the SAME 32
// activations — so silu(gate) * up is formed in registers, lane-locally, and every store stays a 16-byte piece after the
// permlane16 exchange (the first version interleaved (gate_i, up_i) column pairs: 4-byte activation stores and no wide-store
// path for the projection itself, and measured no faster than the separate kernels).
// Exchange: odd 16-lane rows of X swap with the even rows of Y; afterwards the lane owns 8 CONTIGUOUS columns of the 32-column
// pair (X = columns [0,16), Y = [16,32)), starting at col8 = (fq & 1) * 16 + (fq >> 1) * 8.
__device__ __forceinline__ uint4 pair_swap_pk(unsigned x0, unsigned x1, unsigned y0, unsigned y1) {
    const auto lo = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    return make_uint4(lo[0], hi[0], lo[1], hi[1]);
}
__device__ __forceinline__ unsigned pk2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// act 2 — forward.  C[M, N] (bf16) keeps the projection (saved for the backward), aux[M, N/2] = bf16(bf16(silu(g)) * u) with the
// rounding points of the unfused path (projection rounded to bf16, then swiglu_fwd_k).  ncol0 = first column of the 64-block.
__device__ __forceinline__ void epilogue_swiglu_fwd(const GemmArgs& p, int z, int m, bool row_ok, int ncol0, int fq,
                                                    const f32x4& a0, const f32x4& a1, const f32x4& a2, const f32x4& a3) {
    float g[2][4], u[2][4], o[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        g[0][e] = bf2f(f2bf(a0[e] * p.alpha)); g[1][e] = bf2f(f2bf(a1[e] * p.alpha));
        u[0][e] = bf2f(f2bf(a2[e] * p.alpha)); u[1][e] = bf2f(f2bf(a3[e] * p.alpha));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = g[t][e] / (1.0f + __expf(-g[t][e]));
            o[t][e] = bf2f(f2bf(sg)) * u[t][e];                  // HF rounds silu(gate) to bf16 before the product
        }
    const int col8 = (fq & 1) * 16 + (fq >> 1) * 8;
    const uint4 vg = pair_swap_pk(pk2(g[0][0], g[0][1]), pk2(g[0][2], g[0][3]), pk2(g[1][0], g[1][1]), pk2(g[1][2], g[1][3]));
    const uint4 vu = pair_swap_pk(pk2(u[0][0], u[0][1]), pk2(u[0][2], u[0][3]), pk2(u[1][0], u[1][1]), pk2(u[1][2], u[1][3]));
    const uint4 vo = pair_swap_pk(pk2(o[0][0], o[0][1]), pk2(o[0][2], o[0][3]), pk2(o[1][0], o[1][1]), pk2(o[1][2], o[1][3]));
    if (row_ok) {
        bf16_t* c = (bf16_t*)p.C + (long)z * p.sC + (long)m * p.ldc + ncol0 + col8;
        *(uint4*)c = vg;
        *(uint4*)(c + 32) = vu;
        *(uint4*)(p.aux + (long)m * p.lda_x + (ncol0 >> 1) + col8) = vo;
    }
}
// act 3 — backward.  The GEMM computes d(act)[M, N]; ncol0 = first of 64 activation columns = TWO 32-blocks: (a0, a1) belong to
// gate|up block ncol0 / 32, (a2, a3) to the next.  aux[M, 2N] (in) = the saved blocked gate|up, C[M, 2N] (out, ldc = its row
// length) = d(gate|up) in the same blocked layout; d(act) is rounded to bf16 first (the unfused path stores it), then the
// arithmetic of swiglu_bwd_k.
__device__ __forceinline__ void epilogue_swiglu_bwd(const GemmArgs& p, int z, int m, bool row_ok, int ncol0, int fq,
                                                    const f32x4& a0, const f32x4& a1, const f32x4& a2, const f32x4& a3) {
    const int col8 = (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4& x = h ? a2 : a0;
        const f32x4& y = h ? a3 : a1;
        float d[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(bf2f(f2bf(x[e] * p.alpha))), __float_as_uint(bf2f(f2bf(y[e] * p.alpha))), false, false);
            d[e] = __uint_as_float(sw[0]);
            d[4 + e] = __uint_as_float(sw[1]);
        }
        const long off = (long)m * p.lda_x + 2 * (ncol0 + 32 * h) + col8;          // gate piece; the up piece sits 32 columns on
        const u16x8 gv = *(const u16x8*)(p.aux + off);
        const u16x8 uv = *(const u16x8*)(p.aux + off + 32);
        u16x8 og, ou;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = bf2f(gv[e]), uu = bf2f(uv[e]);
            const float sig = 1.0f / (1.0f + __expf(-gg));
            const float sg = gg * sig;
            ou[e] = f2bf(d[e] * sg);
            og[e] = f2bf(d[e] * uu * (sig * (1.0f + gg * (1.0f - sig))));
        }
        if (row_ok) {
            bf16_t* c = (bf16_t*)p.C + (long)z * p.sC + (long)m * p.ldc + 2 * (ncol0 + 32 * h) + col8;
            *(u16x8*)c = og;
            *(u16x8*)(c + 32) = ou;
        }
    }
}

// TA / TB: the operand is stored TRANSPOSED, i.e. as [K, M] (resp. [K, N]) row-major with the reduction index as
// the slow dimension — what autograd's dW = dY^T X and dX = dY W need without materialising a transpose.  Its LDS
// tile image is [64 k][128 out] (256-B rows); the 16-B chunk c of row k sits at slot c ^ f(k), f(k) = (k & 3) |
// ((k >> 3) & 3) << 2, so that the 16 (k-group, k) rows a `ds_read_b64_tr_b16` fragment read touches fall on 16
// distinct chunk slots (2 wave-cycles for 512 B: conflict-free).  A fragment = two transpose reads: each gives a
// lane 4 consecutive k of ONE output index, exactly the 16x16x32 operand layout (lane = out index, k = 8g..8g+7).
__device__ __forceinline__ int tr_swz(int k) { return (k & 3) | (((k >> 3) & 3) << 2); }
__device__ __forceinline__ bf16x8 frag_trans(const char* img, int out0, int kk, int fr, int fq) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
    const int col = out0 + 4 * (fr & 3);                                 // this lane's 4-column piece of the 16-column block
    bf16x8 o;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int k = kk * 32 + 8 * fq + 4 * s2 + (fr >> 2);
        const int off = k * 256 + (((col >> 3) ^ tr_swz(k)) << 4) + ((col & 4) << 1);
        const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(img + off));
        o[4 * s2 + 0] = v[0]; o[4 * s2 + 1] = v[1]; o[4 * s2 + 2] = v[2]; o[4 * s2 + 3] = v[3];
    }
    return o;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE_BYTES];   // [buf][A|B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware, grouped tile order
    const int ntile = p.tilesM * p.tilesN;
    const int L = xcd_remap(blockIdx.x, ntile);
    constexpr int GROUP_M = 8;
    const int gspan = GROUP_M * p.tilesN;
    const int first_m = (L / gspan) * GROUP_M;
    const int gsz = min(p.tilesM - first_m, GROUP_M);
    const int tm = first_m + (L % gspan) % gsz;
    const int tn = (L % gspan) / gsz;
    const int brow = tm * BM, bcol = tn * BN;
    const int z = blockIdx.y;

    const bf16_t* A = p.A + (long)z * p.sA;
    const bf16_t* B = p.B + (long)z * p.sB;

    // per-thread staging addresses: 4 x 16-B chunks for A and for B per K-tile
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + tid;                  // chunk index in the tile image
        if (TA) {                                       // image [64 k][16 chunks of 8 outputs]
            const int k = idx >> 4, c = (idx & 15) ^ tr_swz(k);
            srcA[i] = A + (long)k * p.lda + min(brow + c * 8, p.M - 8);      // clamped chunks feed rows >= M only
        } else {
            const int r = idx >> 3, c = (idx & 7) ^ (r & 7);                 // logical chunk stored at physical slot idx & 7
            srcA[i] = A + (long)min(brow + r, p.M - 1) * p.lda + c * 8;
        }
        if (TB) {
            const int k = idx >> 4, c = (idx & 15) ^ tr_swz(k);
            srcB[i] = B + (long)k * p.ldb + min(bcol + c * 8, p.N - 8);
        } else {
            const int r = idx >> 3, c = (idx & 7) ^ (r & 7);
            srcB[i] = B + (long)min(bcol + r, p.N - 1) * p.ldb + c * 8;
        }
    }
    const long stepA = TA ? (long)BK * p.lda : BK, stepB = TB ? (long)BK * p.ldb : BK;
    auto stage = [&](int buf, int kt) {
        char* la = lds + buf * 2 * TILE_BYTES;
        char* lb = la + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int wbase = (i * 256 + wave * 64) * 16;           // wave-uniform LDS base
            glds16(srcA[i] + kt * stepA, la + wbase);
            glds16(srcB[i] + kt * stepB, lb + wbase);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside a tile image
    const int fr = lane & 15, fq = lane >> 4;
    int offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + fr;
        const int rb = wc * 64 + i * 16 + fr;
        offA[i] = ra * 128 + ((fq ^ (ra & 7)) << 4);
        offB[i] = rb * 128 + ((fq ^ (rb & 7)) << 4);
    }

    const int nk = p.K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* la = lds + cur * 2 * TILE_BYTES;
        const char* lb = la + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // logical chunk = kk*4 + fq  ->  physical = (kk*4 + fq) ^ (row&7) = offX ^ (kk<<6)
                if (TA) af[i] = frag_trans(la, wr * 64 + i * 16, kk, fr, fq);
                else af[i] = *(const bf16x8*)(la + (offA[i] ^ (kk << 6)));
                if (TB) bfr[i] = frag_trans(lb, wc * 64 + i * 16, kk, fr, fq);
                else bfr[i] = *(const bf16x8*)(lb + (offB[i] ^ (kk << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n0..n0+3], m = ..+fr, n0 = ..+fq*4
    if constexpr (!TA && !TB) {
        if (p.act == 2 || p.act == 3) {                                  // fused SwiGLU over the blocked gate|up layout (block-uniform)
            const int ncol0 = bcol + wc * 64;
            if (ncol0 < p.N) {                                           // wave-uniform (N is a multiple of 64: host check)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = brow + wr * 64 + i * 16 + fr;          // all lanes take part in the exchange; rows >= M only skip the stores
                    if (p.act == 2) epilogue_swiglu_fwd(p, z, min(m, p.M - 1), m < p.M, ncol0, fq, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                    else epilogue_swiglu_bwd(p, z, min(m, p.M - 1), m < p.M, ncol0, fq, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = brow + wr * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = bcol + wc * 64 + j * 16 + fq * 4;
            if (n0 >= p.N) continue;
            epilogue4<true>(p, z, m, n0, acc[i][j]);
        }
    }
}


// frag_trans for the ring kernel below.  The compiler orders the `ds_read_tr` BUILTIN behind every LDS-DMA in flight (it puts
// an `s_waitcnt vmcnt(0)` in front: no alias information on the intrinsic), which would drain the ring on every K-tile; as
// inline asm the read is invisible to its counters, so the consumer waits itself (`landed()`: explicit lgkmcnt(0)).
__device__ __forceinline__ bf16x8 frag_trans_raw(const char* img, int out0, int kk, int fr, int fq) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
    const int col = out0 + 4 * (fr & 3);
    bf16x8 o;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int k = kk * 32 + 8 * fq + 4 * s2 + (fr >> 2);
        const int off = k * 256 + (((col >> 3) ^ tr_swz(k)) << 4) + ((col & 4) << 1);
        bf16x4_t v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(unsigned long)(img + off)) : "memory");   // low 32 bits of a flat LDS address = the LDS offset
        o[4 * s2 + 0] = v[0]; o[4 * s2 + 1] = v[1]; o[4 * s2 + 2] = v[2]; o[4 * s2 + 3] = v[3];
    }
    return o;
}

// The same 128x128x64 tile and 4 waves, software-pipelined for grids that leave a block alone on its CU (the Q-Former's
// 2048-row GEMMs: 100-640 tiles of 20-80 K-tiles).  With one wave per SIMD the double buffer above serialises, per K-tile,
// the fragment reads (64 KiB of ds_read per block), the 32 MFMAs of a wave and the wait for the next tile's DMA: 0.72 us per
// K-tile measured (tools/qformer_gemm_bench.py) against 0.21 us of MFMA time.  Here:
//   * FOUR-slot ring (128 KiB, one block per CU), K-tiles t+1..t+3 in flight behind a counted `s_waitcnt vmcnt(16)`
//     (8 LDS-DMA ops per thread and K-tile), ONE raw barrier per K-tile; past the last K-tile the stream re-loads the last tile
//     into dead slots so the count stays constant;
//   * fragments double-buffered in registers: the ds_reads of half K-tile h+1 are issued before the 16 MFMAs of half h, so the
//     LDS pipe and the matrix pipe run side by side inside one wave.
// Same accumulation order per output element as the double-buffered kernel (K-tiles, then halves, in order): bit-identical.
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_ring_kernel(GemmArgs p) {
    constexpr int NS = 4;
    __shared__ __attribute__((aligned(16))) char lds[NS * 2 * TILE_BYTES];   // [slot][A|B]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int ntile = p.tilesM * p.tilesN;
    const int L = xcd_remap(blockIdx.x, ntile);
    constexpr int GROUP_M = 8;
    const int gspan = GROUP_M * p.tilesN;
    const int first_m = (L / gspan) * GROUP_M;
    const int gsz = min(p.tilesM - first_m, GROUP_M);
    const int tm = first_m + (L % gspan) % gsz;
    const int tn = (L % gspan) / gsz;
    const int brow = tm * BM, bcol = tn * BN;
    const int z = blockIdx.y;

    const bf16_t* A = p.A + (long)z * p.sA;
    const bf16_t* B = p.B + (long)z * p.sB;

    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + tid;
        if (TA) {
            const int k = idx >> 4, c = (idx & 15) ^ tr_swz(k);
            srcA[i] = A + (long)k * p.lda + min(brow + c * 8, p.M - 8);
        } else {
            const int r = idx >> 3, c = (idx & 7) ^ (r & 7);
            srcA[i] = A + (long)min(brow + r, p.M - 1) * p.lda + c * 8;
        }
        if (TB) {
            const int k = idx >> 4, c = (idx & 15) ^ tr_swz(k);
            srcB[i] = B + (long)k * p.ldb + min(bcol + c * 8, p.N - 8);
        } else {
            const int r = idx >> 3, c = (idx & 7) ^ (r & 7);
            srcB[i] = B + (long)min(bcol + r, p.N - 1) * p.ldb + c * 8;
        }
    }
    const long stepA = TA ? (long)BK * p.lda : BK, stepB = TB ? (long)BK * p.ldb : BK;
    const int nk = p.K / BK;
    auto stage = [&](int kt) {                                       // K-tile kt -> slot kt & 3
        const int kc = min(kt, nk - 1);
        char* la = lds + (kt & (NS - 1)) * 2 * TILE_BYTES;
        char* lb = la + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int wbase = (i * 256 + wave * 64) * 16;
            glds16(srcA[i] + kc * stepA, la + wbase);
            glds16(srcB[i] + kc * stepB, lb + wbase);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    int offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + fr;
        const int rb = wc * 64 + i * 16 + fr;
        offA[i] = ra * 128 + ((fq ^ (ra & 7)) << 4);
        offB[i] = rb * 128 + ((fq ^ (rb & 7)) << 4);
    }
    auto rd_a = [&](int kt, int kk, int i) __attribute__((always_inline)) -> bf16x8 {
        const char* la = lds + (kt & (NS - 1)) * 2 * TILE_BYTES;
        if (TA) return frag_trans_raw(la, wr * 64 + i * 16, kk, fr, fq);
        return *(const bf16x8*)(la + (offA[i] ^ (kk << 6)));
    };
    auto rd_b = [&](int kt, int kk, int i) __attribute__((always_inline)) -> bf16x8 {
        const char* lb = lds + (kt & (NS - 1)) * 2 * TILE_BYTES + TILE_BYTES;
        if (TB) return frag_trans_raw(lb, wc * 64 + i * 16, kk, fr, fq);
        return *(const bf16x8*)(lb + (offB[i] ^ (kk << 6)));
    };
    // transposed-storage variants: the inline-asm reads are invisible to sched_group_barrier, so the (MFMA, memory op) pairs are
    // fenced one by one instead
    auto fence = [&]() __attribute__((always_inline)) { if (TA || TB) __builtin_amdgcn_sched_barrier(0); };
    auto landed = [&](bf16x8 (&af)[4], bf16x8 (&bfr)[4]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3]) :: "memory");
    };

    stage(0); stage(1); stage(2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                // K-tile 0 of this wave's share has landed ...
    __builtin_amdgcn_s_barrier();                                    // ... and everybody else's
    bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a0[i] = rd_a(0, 0, i); b0[i] = rd_b(0, 0, i); }
    landed(a0, b0);                                                  // (both ways into the loop header agree: no pending reads)
    for (int kt = 0; kt < nk; ++kt) {
        // first half: the 16 MFMAs of fragments (kt, 0) with one memory instruction behind each, inside the issue slack an MFMA
        // leaves (8 of its 16 cycles): alternately a DMA op of K-tile kt+3 (into the slot of K-tile kt-1, which every wave left
        // before the last barrier) and a read of fragments (kt, 1).  (Reads first / DMA last measured the same on the row-major
        // form and 15 % slower on the transposed-storage form.)
        {
            const int kc = min(kt + 3, nk - 1);
            char* la = lds + ((kt + 3) & (NS - 1)) * 2 * TILE_BYTES;
            char* lb = la + TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int wbase = (i * 256 + wave * 64) * 16;
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[0], a0[i], acc[i][0], 0, 0, 0);
                glds16(srcA[i] + kc * stepA, la + wbase);
                fence();
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[1], a0[i], acc[i][1], 0, 0, 0);
                a1[i] = rd_a(kt, 1, i);
                fence();
                acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[2], a0[i], acc[i][2], 0, 0, 0);
                glds16(srcB[i] + kc * stepB, lb + wbase);
                fence();
                acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[3], a0[i], acc[i][3], 0, 0, 0);
                b1[i] = rd_b(kt, 1, i);
                fence();
            }
            if (!TA && !TB) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read (the LDS-DMA)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // `landed` puts the wait for a fragment set HERE, after the MFMAs that cover its reads were issued.  K-tile kt+1 has
        // landed when all but the 16 youngest DMA ops (tiles kt+2, kt+3) are done; this wave's reads of slot kt are complete,
        // so after the barrier the slot may be re-staged
        landed(a1, b1);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // second half: the MFMAs of (kt, 1) with the reads of (kt+1, 0) between them (past the end: a landed, unused tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[0], a1[i], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[1], a1[i], acc[i][1], 0, 0, 0);
            a0[i] = rd_a(kt + 1, 0, i);
            fence();
            acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[2], a1[i], acc[i][2], 0, 0, 0);
            acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[3], a1[i], acc[i][3], 0, 0, 0);
            b0[i] = rd_b(kt + 1, 0, i);
            fence();
        }
        if (!TA && !TB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        landed(a0, b0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // no LDS-DMA may outlive the block

    if constexpr (!TA && !TB) {
        if (p.act == 2 || p.act == 3) {                                  // fused SwiGLU over the blocked gate|up layout (block-uniform)
            const int ncol0 = bcol + wc * 64;
            if (ncol0 < p.N) {                                           // wave-uniform (N is a multiple of 64: host check)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = brow + wr * 64 + i * 16 + fr;          // all lanes take part in the exchange; rows >= M only skip the stores
                    if (p.act == 2) epilogue_swiglu_fwd(p, z, min(m, p.M - 1), m < p.M, ncol0, fq, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                    else epilogue_swiglu_bwd(p, z, min(m, p.M - 1), m < p.M, ncol0, fq, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = brow + wr * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = bcol + wc * 64 + j * 16 + fq * 4;
            if (n0 >= p.N) continue;
            epilogue4<true>(p, z, m, n0, acc[i][j]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves (2M x 4N, 128x64 per wave), 4 phases per K-tile, one 16-KiB half-tile staged
// per phase by LDS-DMA, five half-tiles (80 KiB) in flight behind a COUNTED s_waitcnt vmcnt(10) and raw
// s_barriers (never vmcnt(0) in the loop).  LDS = 2 buffers x {A0, A1, B0, B1} x 16 KiB = 128 KiB.
//   A-half s = rows {wm*128 + s*64 + r}, B-half s = cols {wn*64 + s*32 + c}: a half is what ONE phase
//   reads, so its slot is dead (re-stageable) one phase later:
//     P1 reads A0,B0 -> Q00 | P2 reads B1 -> Q01 | P3 reads A1 -> Q11 | P4 (B0 kept in VGPRs) -> Q10
//   stream order of half-tiles S = (A0,B0,B1,A1)(t), t = 0,1,2..; global phase g issues S[g+7] and, before
//   its closing barrier, waits until all but the 5 youngest half-tiles (2 LDS-DMA each per wave) landed:
//   exactly what phase g+1 reads.  Past the last K-tile the stream re-loads the last tile into dead
//   slots so the count stays constant.
constexpr int HT = 128 * BK * 2;                         // half-tile bytes (128 rows x 128 B)

// PHASES = 4: the schedule described above.  PHASES = 2: the same half-tile stream and slots, but a K-tile is multiplied
// in TWO phases of 32 MFMAs per wave (P1 reads A0,B0,B1 -> Q00,Q01; P2 reads A1 -> Q11,Q10 with B in registers): half the
// workgroup barriers per K-tile (rocprofv3 PMC: the waves of the 4-phase kernel are parked at s_waitcnt / s_barrier 37 %
// of their cycles).  P1 stages A1(t+1) into the slot P2(t-1) vacated, P2 stages A0,B0,B1(t+2) into the slots P1
// vacated; every barrier is preceded by vmcnt(8): all but the 4 youngest half-tiles have landed.
// EPI = 1: the fused SwiGLU epilogues (act 2 / act 3) as their OWN instantiation, so that the default kernel's code and register
// allocation stay exactly what they were (an extra body in its unrolled epilogue once cost a 528-byte scratch frame, see epilogue4).
template <bool STAGGER, int PHASES, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_256_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 4 * HT];            // [buf][A0,A1,B0,B1]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // work item -> (tile, K-slice): the first `full_tiles` items are whole tiles; the remainder of the
    // tile grid (the tail round that would leave CUs idle) is cut into `split` K-slices per tile
    int L, slice = 0;
    if ((int)blockIdx.x < p.full_tiles) {
        L = xcd_remap(blockIdx.x, p.full_tiles);
    } else {
        const int j = blockIdx.x - p.full_tiles;
        L = p.full_tiles + j / p.split;
        slice = j % p.split;
    }
    constexpr int GROUP_M = GEMM_GROUP_M;
    const int gspan = GROUP_M * p.tilesN;
    const int first_m = (L / gspan) * GROUP_M;
    const int gsz = min(p.tilesM - first_m, GROUP_M);
    const int tm = first_m + (L % gspan) % gsz;
    const int tn = (L % gspan) / gsz;
    const int brow = tm * 256, bcol = tn * 256;
    const int z = blockIdx.y;
    const bf16_t* A = p.A + (long)z * p.sA;
    const bf16_t* B = p.B + (long)z * p.sB;

    // staging sources: this thread's two 16-B chunks of each of the four half-tile kinds
    const bf16_t* src[4][2];                              // [A0, A1, B0, B1][i]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = i * 512 + tid;
        const int r = idx >> 3, pc = idx & 7;
        const int c = pc ^ (r & 7);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int arow = (r >> 6) * 128 + s2 * 64 + (r & 63);
            const int bcolr = (r >> 5) * 64 + s2 * 32 + (r & 31);
            src[s2][i] = A + (long)min(brow + arow, p.M - 1) * p.lda + c * 8;
            src[2 + s2][i] = B + (long)min(bcol + bcolr, p.N - 1) * p.ldb + c * 8;
        }
    }
    const int nk_all = p.K / BK;
    const bool partial = (int)blockIdx.x >= p.full_tiles && p.split > 1;
    const int kt0 = partial ? (int)((long)nk_all * slice / p.split) : 0;
    const int kt1 = partial ? (int)((long)nk_all * (slice + 1) / p.split) : nk_all;
    const int nk = kt1 - kt0;
    // issue half-tile number j of the stream
    // tail_skip (2-phase schedule): half-tiles past the end of the stream are NOT issued — without it the stream re-loads the last K-tile
    // into dead slots so that the counted vmcnt stays constant: 7 x 16 KiB of L2 -> LDS traffic per tile that nothing reads (2.7 % of
    // a K = 4096 tile's operand traffic, 8.8 % at K = 1280) and a drain behind the last MFMA
    const bool tskip = PHASES == 2 && p.tail_skip != 0;
    const int n_ht = 4 * nk;
    auto stage = [&](int j) {
        if (tskip && j >= n_ht) return;                   // (wave-uniform)
        const int t = j >> 2, q = j & 3;                  // q: 0 A0, 1 B0, 2 B1, 3 A1
        const int kind = (q == 0) ? 0 : (q == 1) ? 2 : (q == 2) ? 3 : 1;
        const long koff = (long)(kt0 + min(t, nk - 1)) * BK;
        char* dst = lds + ((t & 1) * 4 + kind) * HT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            glds16(src[kind][i] + koff, dst + (i * 512 + wave * 64) * 16);
    };
    // counted wait in front of a barrier: all but the 4 youngest half-tiles of the UNCAPPED stream must have landed; `issued` = stream
    // position after this slot's stage() calls.  Steady state: 4 half-tiles x 2 ops = vmcnt(8); in the tail the stream is shorter by
    // the overshoot, so fewer ops may stay outstanding (wave-uniform branch on a scalar)
    auto stage_steady = [&](int j) __attribute__((always_inline)) {       // j < 4 nk: no end-of-stream test, no clamp
        const int t = j >> 2, q = j & 3;
        const int kind = (q == 0) ? 0 : (q == 1) ? 2 : (q == 2) ? 3 : 1;
        const long koff = (long)(kt0 + t) * BK;
        char* dst = lds + ((t & 1) * 4 + kind) * HT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            glds16(src[kind][i] + koff, dst + (i * 512 + wave * 64) * 16);
    };
    auto wait_steady = [&](int) __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); };
    auto tail_wait = [&](int issued) __attribute__((always_inline)) {
        const int ov = tskip ? issued - n_ht : 0;
        if (ov <= 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ov == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (ov == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ov == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    int offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        offA[i] = r * 128 + ((fq ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wn * 32 + i * 16 + fr;
        offB[i] = r * 128 + ((fq ^ (r & 7)) << 4);
    }

#pragma unroll
    for (int j = 0; j < 7; ++j) stage(j);
    if (PHASES == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else tail_wait(7);
    __builtin_amdgcn_s_barrier();

    bf16x8 a[4][2], b0[2][2], b1[2][2];
    // STAGGER: waves 4-7 (the SIMD partners of waves 0-3) run half a phase behind, so on every SIMD one wave
    // issues its 16 MFMAs while the other issues its LDS reads / LDS-DMA: each phase is split into a LOAD
    // slot and an MFMA slot with a barrier after each; group B enters the loop one barrier late and group A
    // leaves it one barrier late.  Every slot ends with lgkmcnt(0) (its ds_reads have returned before any
    // other wave may re-stage the slot) and the counted vmcnt (the shares this wave issued have landed).
#define SLOT_END()                                                                                                  \
    if (PHASES == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();
#define MFMA_SLOT(ACC_I0, ACC_J0, BF)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
                acc[ACC_I0 + i][ACC_J0 + j] =                                                                \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][kk], a[i][kk], acc[ACC_I0 + i][ACC_J0 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);
    // One K-tile of the two-phase schedule.  STAGE / WAIT are the half-tile issue and the counted wait in front of a barrier: the
    // STEADY forms (no end-of-stream test, constant vmcnt(8)) for every K-tile whose slots stay inside the stream, the checked forms
    // for the last two.  [Round 4 first shipped the checked forms for EVERY K-tile: the scalar compare / branch chains in front of the
    // barriers cost the LLM shapes 4-5 % (gate_up 903 -> 955 us) — invisible to the same-binary A/B of the run-time switch.]
#define KTILE_2PHASE(STAGE, WAIT)                                                                              \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                 \
                b0[i][kk] = *(const bf16x8*)(B0 + (offB[i] ^ (kk << 6)));                                      \
                b1[i][kk] = *(const bf16x8*)(B1 + (offB[i] ^ (kk << 6)));                                      \
            }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A0 + (offA[i] ^ (kk << 6))); \
        STAGE(g + 7);                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        if (STAGGER) { WAIT(g + 8); __builtin_amdgcn_s_barrier(); }                                            \
        MFMA_SLOT(0, 0, b0)                                                                                    \
        MFMA_SLOT(0, 2, b1)                                                                                    \
        WAIT(g + 8); __builtin_amdgcn_s_barrier();                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A1 + (offA[i] ^ (kk << 6))); \
        STAGE(g + 8);                                                                                          \
        STAGE(g + 9);                                                                                          \
        STAGE(g + 10);                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        if (STAGGER) { WAIT(g + 11); __builtin_amdgcn_s_barrier(); }                                           \
        MFMA_SLOT(4, 2, b1)                                                                                    \
        MFMA_SLOT(4, 0, b0)                                                                                    \
        WAIT(g + 11); __builtin_amdgcn_s_barrier();                                                            \
    }
    if (STAGGER && wave >= 4) __builtin_amdgcn_s_barrier();
    // two-phase schedule: K-tiles [0, nk - 2) in the steady form (every slot they stage, <= 4 t + 10, lies inside the stream), the last
    // two in the checked form — two loops, one body each
    const int nk_steady = PHASES == 2 ? max(nk - 2, 0) : 0;
    if constexpr (PHASES == 2) {
        for (int t = 0; t < nk_steady; ++t) {
            const char* base = lds + (t & 1) * 4 * HT;
            const char* A0 = base, *A1 = base + HT, *B0 = base + 2 * HT, *B1 = base + 3 * HT;
            const int g = 4 * t;
            KTILE_2PHASE(stage_steady, wait_steady)
        }
        for (int t = nk_steady; t < nk; ++t) {
            const char* base = lds + (t & 1) * 4 * HT;
            const char* A0 = base, *A1 = base + HT, *B0 = base + 2 * HT, *B1 = base + 3 * HT;
            const int g = 4 * t;
            KTILE_2PHASE(stage, tail_wait)
        }
    }
    for (int t = (PHASES == 2 ? nk : 0); t < nk; ++t) {
        const char* base = lds + (t & 1) * 4 * HT;
        const char* A0 = base, *A1 = base + HT, *B0 = base + 2 * HT, *B1 = base + 3 * HT;
        const int g = 4 * t;
        // ---------------- P1: A0, B0 -> Q00
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b0[i][kk] = *(const bf16x8*)(B0 + (offB[i] ^ (kk << 6)));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A0 + (offA[i] ^ (kk << 6)));
        stage(g + 7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (STAGGER) { SLOT_END() }
        MFMA_SLOT(0, 0, b0)
        SLOT_END()
        // ---------------- P2: B1 -> Q01
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b1[i][kk] = *(const bf16x8*)(B1 + (offB[i] ^ (kk << 6)));
        stage(g + 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (STAGGER) { SLOT_END() }
        MFMA_SLOT(0, 2, b1)
        SLOT_END()
        // ---------------- P3: A1 -> Q11
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A1 + (offA[i] ^ (kk << 6)));
        stage(g + 9);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (STAGGER) { SLOT_END() }
        MFMA_SLOT(4, 2, b1)
        SLOT_END()
        // ---------------- P4: (B0 in registers) -> Q10
        stage(g + 10);
        if (STAGGER) { SLOT_END() }
        MFMA_SLOT(4, 0, b0)
        SLOT_END()
    }
    if (STAGGER && wave < 4) __builtin_amdgcn_s_barrier();
#undef KTILE_2PHASE
#undef MFMA_SLOT
#undef SLOT_END
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the dummy tail loads before LDS is released

    if (partial) {
        // raw fp32 accumulators -> this item's slab
        float* slab = p.ws + ((long)z * (p.tilesM * p.tilesN - p.full_tiles) * p.split +
                              (long)(blockIdx.x - p.full_tiles)) * (256 * 256);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4*)(slab + (wm * 128 + i * 16 + fr) * 256 + wn * 64 + j * 16 + fq * 4) = acc[i][j];
        if (!p.tickets) return;                            // a fix-up launch sums the slices in order
        // In-kernel reduction, scattered: once ALL `split` slices of the tile have arrived, slice s sums rows
        // [256 s / split, 256 (s+1) / split) of the tile over the slabs IN SLICE ORDER (deterministic) and runs the epilogue on
        // them.  (A single last-arriving block summing the whole tile reads `split` x 256 KB at one block's ~70 GB/s: measured
        // 6 ms per step SLOWER than the fix-up launch.)  Hand-off per cdna_hip_programming.md Guideline 16: every storing wave
        // drains its stores, workgroup barrier, lane 0 agent-scope release + arrival add, bounded poll, agent-scope acquire,
        // barrier, plain loads.  All slices of a tail tile are among the last <= 256 work items of a one-block-per-CU grid, so
        // they are resident together (or become so as earlier tiles retire: those never wait).  tickets[2 t] counts arrivals,
        // tickets[2 t + 1] departures; the last to depart zeroes both for the next launch.
        __shared__ int s_go;
        int* tk = p.tickets + 2 * ((long)z * (p.tilesM * p.tilesN - p.full_tiles) + (L - p.full_tiles));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0, ok = 1;
            while (__hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.split) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 21)) { ok = 0; break; }                   // bounded: never hang (the tile is then left unwritten)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_go = ok;
        }
        __syncthreads();
        if (s_go) {
            const float* s0 = p.ws + ((long)z * (p.tilesM * p.tilesN - p.full_tiles) + (L - p.full_tiles)) * p.split * (256 * 256);
            const int r0 = 256 * slice / p.split, r1 = 256 * (slice + 1) / p.split;
            for (int idx = tid; idx < (r1 - r0) * 64; idx += 512) {
                const int ml = r0 + (idx >> 6), nl = (idx & 63) * 4;
                const float* q = s0 + ml * 256 + nl;
                f32x4 v = *(const f32x4*)q;
                for (int s2 = 1; s2 < p.split; ++s2) {
                    const f32x4 w = *(const f32x4*)(q + (long)s2 * (256 * 256));
                    v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
                }
                const int m = brow + ml, n0 = bcol + nl;
                if (m < p.M && n0 < p.N) epilogue4<true>(p, z, m, n0, v);
            }
        }
        __syncthreads();                                                       // every thread is done with the slabs
        if (tid == 0) {
            const int dpt = __hip_atomic_fetch_add(tk + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dpt == p.split - 1) {
                __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(tk + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if constexpr (EPI == 1) {                                  // act 2 / act 3 (never split: host)
        const int ncol0 = bcol + wn * 64;
        if (ncol0 < p.N) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = brow + wm * 128 + i * 16 + fr;
                if (p.act == 2) epilogue_swiglu_fwd(p, z, min(m, p.M - 1), m < p.M, ncol0, fq, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                else epilogue_swiglu_bwd(p, z, min(m, p.M - 1), m < p.M, ncol0, fq, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            }
        }
        return;
    }
    // whole 32-column pairs inside N, bf16 output, no side outputs: wide-store epilogue (block-uniform choice;
    // the permlane swap needs all 64 lanes, so row guards only predicate the store)
    const bool pre_ok = !p.preact || (p.act == 0 && !p.res && (p.ldp & 7) == 0 && (p.sP & 7) == 0);   // pre-activation copy == output: a second wide store
    const bool wide = !p.out_f32 && pre_ok && p.act <= 1 && !p.drop_thresh && (p.N % 32 == 0) && (p.ldc % 8 == 0);
    if (wide) {
        // the tile's feature set, once (block-uniform): the combinations the training step uses get straight-line code
#define DESTA_WIDE_TILE(MODE)                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
            const int m = brow + wm * 128 + i * 16 + fr;                                                               \
            const int mc = min(m, p.M - 1);                                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; j += 2) {                                                         \
                const int ncol0 = bcol + wn * 64 + j * 16;                                                             \
                if (ncol0 >= p.N) continue;                                   /* wave-uniform */                       \
                epilogue_pair_bf16<MODE>(p, z, mc, m < p.M, ncol0, fq, acc[i][j], acc[i][j + 1]);   /* all lanes swap; rows >= M only skip the store */ \
            }                                                                                                          \
        }
        const int mode = (p.bias ? 1 : 0) | (p.act == 1 ? 2 : 0) | (p.rope_cs ? 4 : 0) | (p.res ? 8 : 0) | (p.preact ? 16 : 0);
        if (mode == 0) { DESTA_WIDE_TILE(0) }                                 // dX GEMMs, lm_head, q|k|v without the fused rotary
        else if (mode == 8) { DESTA_WIDE_TILE(8) }                            // o_proj / down_proj (+ residual)
        else if (mode == 4) { DESTA_WIDE_TILE(4) }                            // q|k|v with the rotary epilogue
        else if (mode == 1) { DESTA_WIDE_TILE(1) }                            // Whisper q|k|v, Q-Former K|V (bias)
        else if (mode == 3) { DESTA_WIDE_TILE(3) }                            // Whisper fc1 (bias + GELU)
        else if (mode == 16) { DESTA_WIDE_TILE(16) }                          // Qwen3 q|k|v (+ the copy its q/k-norm backward reads)
        else { DESTA_WIDE_TILE(-1) }
#undef DESTA_WIDE_TILE
        return;
    }
    if (p.out_f32 && p.bias && p.res && p.res_f32 && !p.preact && p.act == 0 && !p.drop_thresh) {      // Whisper out-proj / fc2: fp32 residual stream
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = brow + wm * 128 + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n0 = bcol + wn * 64 + j * 16 + fq * 4;
                if (n0 >= p.N) continue;
                epilogue4_f32_stream(p, z, m, n0, acc[i][j]);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = brow + wm * 128 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = bcol + wn * 64 + j * 16 + fq * 4;
            if (n0 >= p.N) continue;
            epilogue4(p, z, m, n0, acc[i][j]);               // (no rotary epilogue here: see epilogue4)
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// PERSISTENT form of the 256x256 kernel: gridDim.x = min(items, CUs) blocks, block b walks items b, b+G, ...
// and the half-tile stream simply CONTINUES across item boundaries: while the last phases of item s run, the
// LDS-DMA stream is already loading the first K-tiles of item s+1 (no per-tile prologue), and the epilogue
// stores of item s drain behind the main loop of item s+1 (with one block per CU nothing else would hide
// them).  With the staggered schedule the two wave groups run their epilogues half a phase apart, each
// beside the other group's MFMA slot.  Same phase structure, LDS slot rule and counted vmcnt(10) as above.
template <bool STAGGER, int PHASES>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_256p_kernel(GemmArgs p, int n_items) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 4 * HT];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int G = gridDim.x, z = blockIdx.y;
    const int n_my = (n_items - (int)blockIdx.x + G - 1) / G;
    const bf16_t* A = p.A + (long)z * p.sA;
    const bf16_t* B = p.B + (long)z * p.sB;
    const int nk_all = p.K / BK;
    constexpr int GROUP_M = GEMM_GROUP_M;              // the fix-up / persistent kernels must map L -> tile like the main kernel

    // item id -> tile origin + K range
    auto item_geom = [&](int s, int& brow, int& bcol, int& kt0, int& nk, bool& partial, int& slab) {
        const int id = blockIdx.x + s * G;
        int L, slice = 0;
        partial = false;
        if (id < p.full_tiles) {
            L = xcd_remap(id, p.full_tiles);
        } else {
            const int j = id - p.full_tiles;
            L = p.full_tiles + j / p.split;
            slice = j % p.split;
            partial = p.split > 1;
        }
        const int gspan = GROUP_M * p.tilesN;
        const int first_m = (L / gspan) * GROUP_M;
        const int gsz = min(p.tilesM - first_m, GROUP_M);
        brow = (first_m + (L % gspan) % gsz) * 256;
        bcol = ((L % gspan) / gsz) * 256;
        kt0 = partial ? (int)((long)nk_all * slice / p.split) : 0;
        const int kt1 = partial ? (int)((long)nk_all * (slice + 1) / p.split) : nk_all;
        nk = kt1 - kt0;
        slab = id - p.full_tiles;
    };

    // ---- issue side: stream state
    const bf16_t* src[4][2];
    int iss_s = 0, iss_t = 0, iss_nk = 1, iss_kt0 = 0, issT = 0;     // item, local K-tile, its count / first tile, global K-tile count
    auto setup_issue = [&](int s) {
        int brow, bcol, slab; bool partial;
        item_geom(s, brow, bcol, iss_kt0, iss_nk, partial, slab);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 512 + tid;
            const int r = idx >> 3, pc = idx & 7;
            const int c = pc ^ (r & 7);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int arow = (r >> 6) * 128 + s2 * 64 + (r & 63);
                const int bcolr = (r >> 5) * 64 + s2 * 32 + (r & 31);
                src[s2][i] = A + (long)min(brow + arow, p.M - 1) * p.lda + c * 8;
                src[2 + s2][i] = B + (long)min(bcol + bcolr, p.N - 1) * p.ldb + c * 8;
            }
        }
    };
    // issue the next half-tile of the stream; Q = position within the (A0,B0,B1,A1) K-tile group (static per call site)
#define STAGE_NEXT(Q)                                                                                   \
    {                                                                                                   \
        constexpr int kind_ = ((Q) == 0) ? 0 : ((Q) == 1) ? 2 : ((Q) == 2) ? 3 : 1;                      \
        const long koff_ = (long)(iss_kt0 + min(iss_t, iss_nk - 1)) * BK;                               \
        char* dst_ = lds + ((issT & 1) * 4 + kind_) * HT;                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                \
            glds16(src[kind_][i_] + koff_, dst_ + (i_ * 512 + wave * 64) * 16);                         \
        if ((Q) == 3) {                                                                                 \
            ++issT; ++iss_t;                                                                            \
            if (iss_t == iss_nk && iss_s + 1 < n_my) { ++iss_s; iss_t = 0; setup_issue(iss_s); }        \
        }                                                                                               \
    }

    f32x4 acc[8][4];
    const int fr = lane & 15, fq = lane >> 4;
    int offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        offA[i] = r * 128 + ((fq ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wn * 32 + i * 16 + fr;
        offB[i] = r * 128 + ((fq ^ (r & 7)) << 4);
    }

    setup_issue(0);
    STAGE_NEXT(0) STAGE_NEXT(1) STAGE_NEXT(2) STAGE_NEXT(3) STAGE_NEXT(0) STAGE_NEXT(1) STAGE_NEXT(2)
    if (PHASES == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8 a[4][2], b0[2][2], b1[2][2];
#define SLOT_END()                                                                                                  \
    if (PHASES == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();
#define MFMA_SLOT(ACC_I0, ACC_J0, BF)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
                acc[ACC_I0 + i][ACC_J0 + j] =                                                                \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][kk], a[i][kk], acc[ACC_I0 + i][ACC_J0 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);
    if (STAGGER && wave >= 4) __builtin_amdgcn_s_barrier();
    int Tc = 0;                                                       // global K-tile counter of the consumer
    for (int s = 0; s < n_my; ++s) {
        int brow, bcol, kt0, nk, slab; bool partial;
        item_geom(s, brow, bcol, kt0, nk, partial, slab);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nk; ++t, ++Tc) {
            const char* base = lds + (Tc & 1) * 4 * HT;
            const char* A0 = base, *A1 = base + HT, *B0 = base + 2 * HT, *B1 = base + 3 * HT;
            if constexpr (PHASES == 2) {
                // P1: A0, B0, B1 -> Q00, Q01
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        b0[i][kk] = *(const bf16x8*)(B0 + (offB[i] ^ (kk << 6)));
                        b1[i][kk] = *(const bf16x8*)(B1 + (offB[i] ^ (kk << 6)));
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A0 + (offA[i] ^ (kk << 6)));
                STAGE_NEXT(3)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (STAGGER) { SLOT_END() }
                MFMA_SLOT(0, 0, b0)
                MFMA_SLOT(0, 2, b1)
                SLOT_END()
                // P2: A1 (B in registers) -> Q11, Q10
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A1 + (offA[i] ^ (kk << 6)));
                STAGE_NEXT(0)
                STAGE_NEXT(1)
                STAGE_NEXT(2)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (STAGGER) { SLOT_END() }
                MFMA_SLOT(4, 2, b1)
                MFMA_SLOT(4, 0, b0)
                SLOT_END()
                continue;
            }
            // P1: A0, B0 -> Q00            (stream position g+7 : A1 of the K-tile after next)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) b0[i][kk] = *(const bf16x8*)(B0 + (offB[i] ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A0 + (offA[i] ^ (kk << 6)));
            STAGE_NEXT(3)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (STAGGER) { SLOT_END() }
            MFMA_SLOT(0, 0, b0)
            SLOT_END()
            // P2: B1 -> Q01
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) b1[i][kk] = *(const bf16x8*)(B1 + (offB[i] ^ (kk << 6)));
            STAGE_NEXT(0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (STAGGER) { SLOT_END() }
            MFMA_SLOT(0, 2, b1)
            SLOT_END()
            // P3: A1 -> Q11
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) a[i][kk] = *(const bf16x8*)(A1 + (offA[i] ^ (kk << 6)));
            STAGE_NEXT(1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (STAGGER) { SLOT_END() }
            MFMA_SLOT(4, 2, b1)
            SLOT_END()
            // P4: (B0 in registers) -> Q10
            STAGE_NEXT(2)
            if (STAGGER) { SLOT_END() }
            MFMA_SLOT(4, 0, b0)
            SLOT_END()
        }
        // ---- epilogue of item s (no barrier inside: the other wave group keeps streaming)
        if (partial) {
            float* slabp = p.ws + ((long)z * (p.tilesM * p.tilesN - p.full_tiles) * p.split + (long)slab) * (256 * 256);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(f32x4*)(slabp + (wm * 128 + i * 16 + fr) * 256 + wn * 64 + j * 16 + fq * 4) = acc[i][j];
            continue;
        }
        const bool wide = !p.out_f32 && !p.preact && p.act <= 1 && !p.drop_thresh && (p.N % 32 == 0) && (p.ldc % 8 == 0);
        if (wide) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = brow + wm * 128 + i * 16 + fr;
                const int mc = min(m, p.M - 1);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int ncol0 = bcol + wn * 64 + j * 16;
                    if (ncol0 >= p.N) continue;
                    epilogue_pair_bf16(p, z, mc, m < p.M, ncol0, fq, acc[i][j], acc[i][j + 1]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = brow + wm * 128 + i * 16 + fr;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n0 = bcol + wn * 64 + j * 16 + fq * 4;
                    if (n0 >= p.N) continue;
                    epilogue4(p, z, m, n0, acc[i][j]);
                }
            }
        }
    }
    if (STAGGER && wave < 4) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the tail of the stream before LDS is released
#undef KTILE_2PHASE
#undef MFMA_SLOT
#undef SLOT_END
#undef STAGE_NEXT
}

// sum the K-slice slabs of the split tiles (fixed order) and run the normal epilogue
__global__ __launch_bounds__(256) void gemm_splitk_fixup_kernel(GemmArgs p) {
    const int rem = p.tilesM * p.tilesN - p.full_tiles;
    const int ti = blockIdx.x / 64, part = blockIdx.x % 64;          // 64 blocks per tile: 4 rows x 256 cols each
    const int z = blockIdx.y;
    const int L = p.full_tiles + ti;
    constexpr int GROUP_M = GEMM_GROUP_M;              // the fix-up / persistent kernels must map L -> tile like the main kernel
    const int gspan = GROUP_M * p.tilesN;
    const int first_m = (L / gspan) * GROUP_M;
    const int gsz = min(p.tilesM - first_m, GROUP_M);
    const int tm = first_m + (L % gspan) % gsz;
    const int tn = (L % gspan) / gsz;
    const int ml = part * 4 + (threadIdx.x >> 6), nl = (threadIdx.x & 63) * 4;
    const float* slab = p.ws + ((long)z * rem + ti) * p.split * (256 * 256) + ml * 256 + nl;
    f32x4 acc = *(const f32x4*)slab;
    for (int s2 = 1; s2 < p.split; ++s2) {
        const f32x4 v = *(const f32x4*)(slab + (long)s2 * (256 * 256));
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    const int m = tm * 256 + ml, n0 = tn * 256 + nl;
    if (m < p.M && n0 < p.N) epilogue4<true>(p, z, m, n0, acc);
}

// ---------------------------------------------------------------------------------------------------------
// Skinny GEMM, M <= 16 (the projections of a KV-cached decode step: M = batch).  Pure weight streaming, bound
// by HBM: every weight byte is read once, 2*M flops per weight element.  One block owns COLS output columns;
// its 8 waves walk K in interleaved 64-element chunks, so at any moment the block reads COLS rows x 1 KiB of
// contiguous weight bytes (each lane 2 x 16 B of one row; 4 lanes = 128 B per row per wave).  The WEIGHT
// tile is the MFMA A operand (rows = n) and the activations the B operand (cols = m): a lane then owns 4
// consecutive n of one row m, which is exactly what epilogue4 stores.  k inside a chunk is permuted
// (lane group g holds k = 16g..16g+15, first 8 in MFMA 0, last 8 in MFMA 1) identically for both operands.
// COLS = 32/64 reuse one activation fragment for 2/4 weight tiles (fewer L2 loads per HBM byte; used when N
// is large enough to still fill the chip).  The K loop is software pipelined over two register stages of U
// chunks: the loads of group i+1 are in flight while group i feeds the MFMAs, so a block streams continuously
// instead of paying one memory latency per group.  Partial sums of the 8 waves are reduced through LDS in
// fixed order (deterministic).
template <int COLS, int U>
struct SkinnyStage {
    bf16x8 w0[U][COLS / 16], w1[U][COLS / 16], x0[U], x1[U];
};

// SWIGLU: B holds gate rows [0,N) and up rows [N,2N) (the concatenated gate|up projection); a tile is 8 output
// columns = the 8 gate + 8 matching up rows, the epilogue stores bf16(silu(gate)) * up to C[M,N] with the
// rounding points of the unfused path (projection rounded to bf16, then swiglu_fwd).
// RMS: A holds the UN-normalised residual stream; the block first normalises the M rows itself (RMSNorm with
// weight rms_w, the rounding points of rmsnorm_fwd_k: rms_w * bf16(x * rstd) -> bf16) into LDS while its first
// weight loads are in flight, and the MFMA activation fragments are then read from LDS instead of global memory.
// This removes the separate norm launch of a decode layer (~8 us for 8 rows, launch-latency bound).  The 16-byte
// slots of LDS row m are XOR-swizzled with m so the 16 row-strided ds_read_b128 of a fragment hit distinct banks
// without padding (8 rows x 4096 + the reduction buffers = exactly 80 KiB: two blocks per CU).
// The grid is PERSISTENT (<= 2 blocks per CU): a block walks column tiles blockIdx.x, +gridDim.x, ... and the
// two-stage load pipeline runs across tile boundaries, so the weight stream never drains while a tile is being
// reduced and the RMS prologue is paid once per block, not once per tile.
constexpr int SKINNY_XS_BYTES = 8 * 4096 * 2;                           // M * 2K must fit: 8 rows x 4096
struct SkinnyCursor { int tile, grp; };

template <int COLS, int U, bool SWIGLU, bool RMS>
__global__ __launch_bounds__(512, (COLS == 16 && U == 2) ? 2 : 1) void gemm_bf16_nt_skinny_kernel(GemmArgs p, int ntiles) {
    static_assert(!SWIGLU || COLS == 16, "SwiGLU pairing uses one 16-row weight tile");
    constexpr int T = COLS / 16;
    constexpr int RED_BYTES = 8 * T * 64 * 16;                           // one buffer of per-wave partial tiles
    __shared__ __attribute__((aligned(16))) char smem[2 * RED_BYTES + (RMS ? SKINNY_XS_BYTES : 0)];
    const char* xs = smem + 2 * RED_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, z = blockIdx.y;
    const int r = lane & 15, g = lane >> 4;
    const bool mok = r < p.M;
    const bf16_t* Bz = p.B + (long)z * p.sB + g * 16;
    const bf16_t* ap = p.A + (long)z * p.sA + (long)(mok ? r : 0) * p.lda + g * 16;
    const int nchunks = p.K >> 6;
    constexpr int STEP = 8 * U;
    const int gpt = (nchunks + STEP - 1) / STEP;                         // groups per tile, the same for every wave
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int G = my_tiles * gpt;
    const int rowb = p.K * 2, rr = mok ? r : 0;
    const char* xrow = xs + rr * rowb;                                   // this lane's fragment row in LDS (RMS only)
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    SkinnyStage<COLS, U> sa, sb;
    SkinnyCursor lc = {(int)blockIdx.x, 0}, mc = {(int)blockIdx.x, 0};
    int parity = 0;

    auto advance = [&](SkinnyCursor& c) { if (++c.grp == gpt) { c.grp = 0; c.tile += (int)gridDim.x; } };
#define SK_LOAD(S)                                                                          \
    {                                                                                       \
        const int n0 = lc.tile * (SWIGLU ? 8 : COLS), c0 = wave + lc.grp * STEP;            \
        const bf16_t* bp[T];                                                                \
        _Pragma("unroll") for (int t = 0; t < T; ++t) {      /* clamped rows: computed, never stored */ \
            const int wrow = SWIGLU ? (r < 8 ? min(n0 + r, p.N - 1) : p.N + min(n0 + r - 8, p.N - 1)) \
                                    : min(n0 + t * 16 + r, p.N - 1);                        \
            bp[t] = Bz + (long)wrow * p.ldb;                                                \
        }                                                                                   \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                     \
            const int cc = c0 + 8 * u;                                                      \
            if (cc < nchunks) {                                                             \
                _Pragma("unroll") for (int t = 0; t < T; ++t) {                             \
                    S.w0[u][t] = *(const bf16x8*)(bp[t] + (long)cc * 64);                   \
                    S.w1[u][t] = *(const bf16x8*)(bp[t] + (long)cc * 64 + 8);               \
                }                                                                           \
                if (!RMS) {                                                                 \
                    S.x0[u] = *(const bf16x8*)(ap + (long)cc * 64);                         \
                    S.x1[u] = *(const bf16x8*)(ap + (long)cc * 64 + 8);                     \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        advance(lc);                                                                        \
    }
#define SK_MMA(S)                                                                           \
    {                                                                                       \
        const int c0 = wave + mc.grp * STEP;                                                \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                     \
            if (c0 + 8 * u < nchunks) {                                                     \
                const bf16x8 zero = {};                                                     \
                if (RMS) {                                                                  \
                    const int slot = (c0 + 8 * u) * 8 + g * 2;                              \
                    S.x0[u] = *(const bf16x8*)(xrow + ((slot ^ rr) << 4));                  \
                    S.x1[u] = *(const bf16x8*)(xrow + (((slot + 1) ^ rr) << 4));            \
                }                                                                           \
                const bf16x8 a0 = mok ? S.x0[u] : zero, a1 = mok ? S.x1[u] : zero;          \
                _Pragma("unroll") for (int t = 0; t < T; ++t) {                             \
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S.w0[u][t], a0, acc[t], 0, 0, 0); \
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S.w1[u][t], a1, acc[t], 0, 0, 0); \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        if (mc.grp == gpt - 1) finish_tile(mc.tile);                                        \
        advance(mc);                                                                        \
    }
    // reduce the 8 waves' partial tiles in fixed order, epilogue, reset; one barrier per tile (red is double buffered:
    // a wave can run at most one tile ahead of the wave that still reads the other buffer)
    auto finish_tile = [&](int tile) {
        f32x4 (*red)[T][64] = (f32x4 (*)[T][64])(smem + parity * RED_BYTES);
        parity ^= 1;
#pragma unroll
        for (int t = 0; t < T; ++t) { red[wave][t][lane] = acc[t]; acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        __syncthreads();
        if (wave < T) {                                                  // wave t finishes column tile t
            f32x4 sum = red[0][wave][lane];
#pragma unroll
            for (int w = 1; w < 8; ++w) sum += red[w][wave][lane];
            const int n0 = tile * (SWIGLU ? 8 : COLS);
            if (SWIGLU) {
                // lane groups 0,1 hold gate columns n0+4g.., groups 2,3 the matching up columns: fetch up from lane+32
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float up = __shfl(sum[e], (lane + 32) & 63, 64);
                    const float gt = bf2f(f2bf(sum[e] * p.alpha)), uu = bf2f(f2bf(up * p.alpha));
                    o[e] = f2bf(bf2f(f2bf(gt / (1.0f + __expf(-gt)))) * uu);
                }
                const int n = n0 + g * 4;
                if (mok && g < 2 && n < p.N) *(u16x4*)((bf16_t*)p.C + (long)z * p.sC + (long)r * p.ldc + n) = o;
            } else {
                const int n = n0 + wave * 16 + g * 4;
                if (mok && n < p.N) epilogue4(p, z, r, n, sum);
            }
        }
    };

    if (G > 0) SK_LOAD(sa)
    if (RMS) {
        for (int m = wave; m < p.M; m += 8) {                            // one wave per row, same summation order as rmsnorm_fwd_k
            const bf16_t* xr = p.A + (long)z * p.sA + (long)m * p.lda;
            float q = 0.f;
#pragma unroll 4
            for (int c = lane * 8; c < p.K; c += 512) {
                const u16x8 v = *(const u16x8*)(xr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = bf2f(v[e]); q += f * f; }
            }
            const float rstd = rsqrtf(wave_sum(q) / (float)p.K + p.rms_eps);
#pragma unroll 2
            for (int c = lane * 8; c < p.K; c += 512) {
                const u16x8 v = *(const u16x8*)(xr + c);
                const float4 w0 = *(const float4*)(p.rms_w + c), w1 = *(const float4*)(p.rms_w + c + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(wv[e] * bf2f(f2bf(bf2f(v[e]) * rstd)));
                *(u16x8*)(smem + 2 * RED_BYTES + m * rowb + (((c >> 3) ^ m) << 4)) = o;
            }
        }
        __syncthreads();
    }
    for (int gi = 0; gi < G; gi += 2) {
        if (gi + 1 < G) SK_LOAD(sb)
        SK_MMA(sa)
        if (gi + 2 < G) SK_LOAD(sa)
        if (gi + 1 < G) SK_MMA(sb)
    }
#undef SK_LOAD
#undef SK_MMA
}

}  // namespace

static int g_last_kernel = 0;      // kernel family of the most recent launch: 1 = 128x128, 2 = 256x256 (+ split-K fix-up), 3 = skinny
extern "C" int desta_gemm_last_kernel(void) { return g_last_kernel; }
static int g_force_variant = 0;   // 0 auto, 1 = 128x128, 2 = 256x256 lockstep, 3 = staggered, 4 = staggered persistent (tuning / tests)
extern "C" int desta_gemm_force_variant(int v) { g_force_variant = v; return DESTA_OK; }
static int g_persistent = 0;      // automatic choice may use the persistent kernel (in-situ A/B: no gain)
static int g_stagger = 1;         // automatic choice uses the staggered schedule
static int g_inkernel_splitk = 0; // option 5: K-slices of tail tiles reduced inside the GEMM launch instead of by the fix-up launch (measured: no gain)
static int g_small_ring = 1;      // option 6: 0 never, 1 the four-slot ring form of the 128x128 kernel when its grid leaves one block per CU, 2 always (A/B runs)
static int g_phases2 = 1;         // automatic choice uses the 2-phase (32 MFMAs per phase) staggered schedule (+5-16 % on every shape)
extern "C" int desta_gemm_set_persistent(int on) { g_persistent = on; return DESTA_OK; }
static int g_skinny = 0;          // 0 auto, else COLS*10 + U of the skinny (M <= 16) kernel (tuning)
static int g_skinny_blocks = 512;  // persistent grid of the skinny kernel (2 blocks per CU)
static int g_tail_skip = 1;        // option 10: the 2-phase 256x256 kernel stops its half-tile stream at the last K-tile (1, default) or re-loads dead slots (0: rounds 1-3)
extern "C" int desta_gemm_set_option(int option, int value) {
    if (option == 0) g_persistent = value;
    else if (option == 1) g_stagger = value;
    else if (option == 2) {
        if (value != 0 && value != 162 && value != 164 && value != 322 && value != 641) {
            desta_set_error("gemm_set_option: skinny variant %d unknown (COLS*10+U: 162 164 322 641)", value);
            return DESTA_EINVAL;
        }
        g_skinny = value;
    }
    else if (option == 4) g_phases2 = value;
    else if (option == 5) g_inkernel_splitk = value;
    else if (option == 6) g_small_ring = value;
    else if (option == 7 || option == 8) { /* (round 3's de-synchronised start: measured no gain, removed) */ }
    else if (option == 9) { /* (round 4's packed-polynomial GELU: measured equal in the step, and its code in every epilogue fragment cost the PLAIN path 17 % at K = 1280: removed) */ }
    else if (option == 10) g_tail_skip = value;
    else if (option == 3) {
        if (value < 1 || value > 65535) { desta_set_error("gemm_set_option: skinny grid %d out of range", value); return DESTA_EINVAL; }
        g_skinny_blocks = value;
    }
    else { desta_set_error("gemm_set_option: unknown option %d", option); return DESTA_EINVAL; }
    return DESTA_OK;
}

extern "C" int desta_gemm_bf16_nt(const desta_gemm_desc* d, void* stream) {
    DESTA_CHECK_ARG(d && d->A && d->B && d->C, "gemm: null operand");
    DESTA_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    DESTA_CHECK_ARG(d->K % BK == 0, "gemm: K=%d must be a multiple of %d", d->K, BK);
    DESTA_CHECK_ARG(d->N % 4 == 0, "gemm: N=%d must be a multiple of 4", d->N);
    DESTA_CHECK_ARG(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "gemm: lda/ldb must be multiples of 8, ldc of 4");
    DESTA_CHECK_ARG(((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0) && ((uintptr_t)d->C % 16 == 0),
                    "gemm: operands must be 16-byte aligned");
    DESTA_CHECK_ARG(d->batch >= 1 && d->batch <= 65535, "gemm: bad batch %d", d->batch);
    DESTA_CHECK_ARG(!d->residual || d->ldr % 4 == 0, "gemm: ldr must be a multiple of 4");
    DESTA_CHECK_ARG(!d->preact || d->ldp % 4 == 0, "gemm: ldp must be a multiple of 4");
    DESTA_CHECK_ARG(!d->trans_a || (d->M % 8 == 0 && d->M >= 8), "gemm: trans_a needs M to be a multiple of 8");
    DESTA_CHECK_ARG(!d->trans_b || (d->N % 8 == 0 && d->N >= 8), "gemm: trans_b needs N to be a multiple of 8");
    DESTA_CHECK_ARG(!(d->trans_a || d->trans_b) || (d->act != 4 && !d->a_rms_weight), "gemm: trans_a / trans_b not available on the decode path");
    GemmArgs a;
    a.A = (const bf16_t*)d->A; a.B = (const bf16_t*)d->B; a.C = d->C;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.sA = d->stride_a; a.sB = d->stride_b; a.sC = d->stride_c;
    a.bias = d->bias;
    a.res = d->residual; a.ldr = d->ldr; a.sR = d->stride_r; a.res_f32 = d->residual_f32;
    a.act = d->act; a.out_f32 = d->out_f32; a.tail_skip = g_tail_skip;
    a.preact = (bf16_t*)d->preact; a.ldp = d->ldp; a.sP = d->stride_p;
    a.alpha = d->alpha;
    a.aux = (bf16_t*)d->aux; a.lda_x = d->ld_aux;
    DESTA_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "gemm: dropout_p must be in [0,1)");
    a.drop_thresh = d->dropout_p > 0.f ? desta_drop_thresh(d->dropout_p) : 0u;
    a.drop_scale = 1.0f / (1.0f - d->dropout_p);
    a.seed_lo = (unsigned)d->dropout_seed; a.seed_hi = (unsigned)(d->dropout_seed >> 32);
    DESTA_CHECK_ARG(d->act >= 0 && d->act <= 4, "gemm: unknown act %d", d->act);
    DESTA_CHECK_ARG(d->act < 2 || d->act == 4 || (d->aux && !d->out_f32 && !d->bias && !d->residual && !d->preact && d->ld_aux % 8 == 0),
                    "gemm: SwiGLU epilogues need aux, bf16 output and no bias/residual/preact");
    DESTA_CHECK_ARG(d->act != 3 || d->ldc % 8 == 0, "gemm: act 3 needs ldc (2N-wide rows) to be a multiple of 8");
    DESTA_CHECK_ARG((d->act != 2 && d->act != 3) || (d->N % 64 == 0 && d->ldc % 8 == 0 && d->batch == 1 && !d->trans_a && !d->trans_b && d->dropout_p == 0.f),
                    "gemm: the SwiGLU epilogues work on whole 64-column blocks (N %% 64 == 0), row-major operands, batch 1");
    // Tile choice.  The 256x256 8-phase kernel runs ONE block per CU, so the tile grid executes in rounds
    // of 256; a partial last round leaves CUs idle (M=5120 x N=4096: 320 tiles = 1.25 rounds).  Those tail
    // tiles are cut into `split` K-slices (<= 256 items, each 1/split long: "1 + 1/split" rounds instead
    // of 2), partial sums go to the caller's workspace and a fix-up launch reduces them in fixed order.
    // The 128x128 kernel (2 blocks/CU) covers small shapes and calls without a workspace.
    constexpr int NCU = 256;
    const int tM = (d->M + 255) / 256, tN = (d->N + 255) / 256, nk = d->K / BK;
    const long T = (long)tM * tN;
    int split = 1, full = (int)T;
    if (d->batch == 1 && d->workspace && nk >= 8 && d->act != 2 && d->act != 3) {      // (the SwiGLU epilogues need the whole accumulator tile in registers: no K-slices)
        const int rem = (int)(T % NCU);
        if (rem > 0 && rem <= NCU / 2) {
            int sp = NCU / rem;
            if (sp > 8) sp = 8;
            if (sp > nk / 16) sp = nk / 16;              // >= 16 K-tiles per slice, or the 7-half-tile prologue dominates
            if (sp >= 2 && (size_t)rem * sp * 256 * 256 * sizeof(float) + 4096 <= d->workspace_bytes) { split = sp; full = (int)(T - rem); }
        }
    }
    a.rope_cs = d->rope_cos_sin; a.rope_pos = d->rope_pos; a.rope_cols = d->rope_cols; a.rope_hd = d->rope_head_dim;
    if (d->rope_cos_sin)
        DESTA_CHECK_ARG(d->rope_pos && (d->rope_head_dim == 64 || d->rope_head_dim == 128) && d->rope_cols > 0 && d->rope_cols <= d->N &&
                        d->rope_cols % d->rope_head_dim == 0 && d->act == 0 && !d->bias && !d->residual && !d->preact && !d->out_f32 &&
                        d->dropout_p == 0.f && d->batch == 1 && d->M > 16 && !d->trans_a && !d->trans_b && d->N % 32 == 0 && d->ldc % 8 == 0,
                        "gemm: the rotary epilogue needs rope_pos, head_dim 64 / 128, rope_cols a multiple of it, a plain bf16 output (no bias / residual / act) and M > 16");
    a.rms_w = d->a_rms_weight; a.rms_eps = d->a_rms_eps;
    if (d->a_rms_weight)
        DESTA_CHECK_ARG(d->M <= 16 && (size_t)d->M * 2 * (size_t)d->K <= (size_t)SKINNY_XS_BYTES && d->K % 512 == 0 &&
                        (d->act == 0 || d->act == 4) && g_force_variant == 0,
                        "gemm: a_rms_weight (fused RMSNorm of A) needs M <= 16, M*2K <= %d bytes of LDS, K %% 512 == 0, act 0 or 4", SKINNY_XS_BYTES);
    if (d->act == 4 || (d->M <= 16 && g_force_variant == 0 && !d->trans_a && !d->trans_b && d->act != 2 && d->act != 3)) {          // decode-time projections: weight streaming
        DESTA_CHECK_ARG(d->act != 4 || (d->M <= 16 && !d->out_f32 && !d->bias && !d->residual && !d->preact && d->dropout_p == 0.f),
                        "gemm: act 4 (SwiGLU over concatenated gate|up rows) is the decode path: M <= 16, bf16 out, no other epilogue");
        // 16 columns x 8 K-slices per tile measured fastest on every decode shape (tools/skinny_bench.py; wider
        // column tiles, deeper stages and non-temporal weight loads all measured slower); option 2 = COLS*10 + U
        int cols = 16, u = 2;
        if (g_skinny && d->act != 4) { cols = g_skinny / 10; u = g_skinny % 10; }
        const int ntiles = d->act == 4 ? (d->N + 7) / 8 : (d->N + cols - 1) / cols;
        a.tilesM = 1; a.tilesN = ntiles; a.full_tiles = ntiles; a.split = 1; a.ws = nullptr; a.tickets = nullptr;
        const dim3 grid(ntiles < g_skinny_blocks ? ntiles : g_skinny_blocks, d->batch);
        hipStream_t st = (hipStream_t)stream;
        const bool rms = a.rms_w != nullptr;
#define SK_LAUNCH(C_, U_, SW_) \
        do { if (rms) hipLaunchKernelGGL((gemm_bf16_nt_skinny_kernel<C_, U_, SW_, true>), grid, dim3(512), 0, st, a, ntiles); \
             else hipLaunchKernelGGL((gemm_bf16_nt_skinny_kernel<C_, U_, SW_, false>), grid, dim3(512), 0, st, a, ntiles); } while (0)
        if (d->act == 4) SK_LAUNCH(16, 2, true);
        else if (cols == 64) SK_LAUNCH(64, 1, false);
        else if (cols == 32) SK_LAUNCH(32, 2, false);
        else if (u == 4) SK_LAUNCH(16, 4, false);
        else SK_LAUNCH(16, 2, false);
#undef SK_LAUNCH
        DESTA_CHECK_LAUNCH("gemm_bf16_nt_skinny");
        g_last_kernel = 3;
        return DESTA_OK;
    }
    bool big = false;
    if (d->M >= 128 && d->N >= 128 && d->K >= 512) {
        const double rounds = (double)full * d->batch / NCU + (split > 1 ? 1.0 / split : 0.0);
        const double ideal = (double)T * d->batch / NCU;
        const double eff = ideal / (split > 1 ? rounds : ceil(rounds));      // fraction of CU-time doing work
        big = eff >= 0.78;
    }
    if (g_force_variant == 1) big = false;
    if (g_force_variant >= 2) big = true;
    if (d->trans_a || d->trans_b) big = false;                          // transposed-storage operands: 128x128 kernel only
    if (big) {
        a.tilesM = tM; a.tilesN = tN;
        a.full_tiles = full; a.split = split; a.ws = (float*)d->workspace;
        const int items = full + (int)(T - full) * split;
        // tickets: the last 4 KiB of the workspace (<= 128 split tiles); zero at first use (the caller hands over a zeroed
        // workspace once) and self-resetting afterwards.  Only the plain (non-persistent) kernels carry the in-kernel reduce.
        const bool persistent_ = g_force_variant == 4 || g_force_variant == 8 || (g_force_variant == 0 && g_persistent && items > NCU);
        const bool inkernel = split > 1 && g_inkernel_splitk && !persistent_;
        a.tickets = inkernel ? (int*)((char*)d->workspace + d->workspace_bytes - 4096) : nullptr;
        // variants: 2 = lockstep, 3 = staggered (+4-7 %), 4 = staggered + persistent cross-tile streaming
        //           (default when a block gets more than one item; +3-7 % on the LLM shapes)
        const bool persistent = g_force_variant == 4 || (g_force_variant == 0 && g_persistent && items > NCU);
        if (d->act == 2 || d->act == 3) hipLaunchKernelGGL((gemm_bf16_nt_256_kernel<true, 2, 1>), dim3(items, d->batch), dim3(512), 0, (hipStream_t)stream, a);
        else if (g_force_variant == 2 || (g_force_variant == 0 && !g_stagger)) hipLaunchKernelGGL((gemm_bf16_nt_256_kernel<false, 4>), dim3(items, d->batch), dim3(512), 0, (hipStream_t)stream, a);
        else if (g_force_variant == 8) hipLaunchKernelGGL((gemm_bf16_nt_256p_kernel<true, 2>), dim3(items < NCU ? items : NCU, d->batch), dim3(512), 0, (hipStream_t)stream, a, items);
        else if (persistent) hipLaunchKernelGGL((gemm_bf16_nt_256p_kernel<true, 4>), dim3(items < NCU ? items : NCU, d->batch), dim3(512), 0, (hipStream_t)stream, a, items);
        else if (g_force_variant == 6 || (g_force_variant == 0 && g_phases2)) hipLaunchKernelGGL((gemm_bf16_nt_256_kernel<true, 2>), dim3(items, d->batch), dim3(512), 0, (hipStream_t)stream, a);
        else if (g_force_variant == 7) hipLaunchKernelGGL((gemm_bf16_nt_256_kernel<false, 2>), dim3(items, d->batch), dim3(512), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gemm_bf16_nt_256_kernel<true, 4>), dim3(items, d->batch), dim3(512), 0, (hipStream_t)stream, a);
        if (split > 1 && !inkernel)
            hipLaunchKernelGGL(gemm_splitk_fixup_kernel, dim3((unsigned)(T - full) * 64, d->batch), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        a.tilesM = (d->M + BM - 1) / BM; a.tilesN = (d->N + BN - 1) / BN;
        a.full_tiles = a.tilesM * a.tilesN; a.split = 1; a.ws = nullptr; a.tickets = nullptr;
        dim3 grid(a.tilesM * a.tilesN, d->batch);
        const long nblk = (long)a.tilesM * a.tilesN * d->batch;
        const bool ring = d->K / BK >= 4 && (g_small_ring == 2 || (g_small_ring == 1 && nblk <= NCU));
        if (ring) {
            if (d->trans_a && d->trans_b) hipLaunchKernelGGL((gemm_bf16_nt_ring_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
            else if (d->trans_b) hipLaunchKernelGGL((gemm_bf16_nt_ring_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
            else if (d->trans_a) hipLaunchKernelGGL((gemm_bf16_nt_ring_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((gemm_bf16_nt_ring_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
        } else if (d->trans_a && d->trans_b) hipLaunchKernelGGL((gemm_bf16_nt_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else if (d->trans_b) hipLaunchKernelGGL((gemm_bf16_nt_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else if (d->trans_a) hipLaunchKernelGGL((gemm_bf16_nt_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gemm_bf16_nt_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    DESTA_CHECK_LAUNCH("gemm_bf16_nt");
    g_last_kernel = big ? 2 : 1;
    return DESTA_OK;
}
