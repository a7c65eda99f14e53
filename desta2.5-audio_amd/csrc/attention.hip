// Flash-style attention for the three attention shapes of the DeSTA2.5 step on gfx950 (MFMA 32x32x16,
// fp32 online softmax, no S x S matrix in HBM), forward + backward:
//   Whisper encoder self-attention (non-causal, D=64)   TF:models/whisper/modeling_whisper.py:241-357
//   Q-Former self-/cross-attention (non-causal, D=64)   TF:models/bert/modeling_bert.py:100-293 (eager)
//   Llama / Qwen3 GQA causal attention with left-pad key mask (D=128 or 64)
//                                                       TF:models/llama/modeling_llama.py:179-281
// and their autograd backward (dQ, dK, dV).
//
// Orientation ("key on the register rows, query on the lane"): S^T = K·Q^T is computed with K as the
// MFMA A operand (ds_read_b128 rows of an XOR-swizzled LDS image) and Q as the B operand (registers),
// so every lane owns ONE query column: row max / row sum / rescale are lane-local scalars.  The fp32
// S^T accumulator converts pairwise to bf16 and is used directly as the B operand of O^T += V^T·P^T
// (no LDS round trip); V^T fragments come from the row-major V image with ds_read_b64_tr_b16.
// K/V tiles are register-staged (global load of tile t+1 issued before the MFMAs of tile t).
// Backward = two kernels without atomics (deterministic): dq (same skeleton as forward) and dkdv
// (one wave owns 32 keys, sweeps the query heads of its GQA group and 32-row query slices).
//
// Fully masked query rows (left-pad positions) produce O = 0 and lse = +inf (their gradients are 0).
#include "common.h"
#include <mutex>
#include "desta_hip.h"

#ifndef ATTN_ABL
#define ATTN_ABL 0     /* forward-kernel timing ablations for tools/attn_bench.py (results are garbage): 1 no LDS stores, 2 no global
                          loads in the loop, 3 no exp, 4 one S MFMA step, 5 one P.V MFMA column block */
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    const bf16_t* dO; bf16_t* dQ; bf16_t* dK; bf16_t* dV;
    float* lse; const float* delta;
    float* O32;                                                      // optional fp32 copy of O (same strides as O): delta = rowsum(dO * O) from the UNROUNDED output
    const float* rope_cs;                                            // bwd: [seq][D/2][2] (cos, sin): dQ / dK are rotated back (adjacent-pair layout) before the store
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;           // batch / row strides (elements)
    long do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    int B, Hq, Hkv, Sq, Sk;
    int causal;
    const int* kv_start;                                             // [B] first valid key (left padding) or null
    float scale_log2;                                                // softmax scale * log2(e)
    float scale;
    unsigned drop_thresh, seed_lo, seed_hi; float drop_scale;       // attention-probability dropout (DROP kernels)
    long dkv_t_ld;                                                   // one-query-tile backward: dK / dV stored transposed ([head * 64 + d][dkv_t_ld], column = batch * Sk + key); 0 = off
};

template <int D>
__device__ __forceinline__ int img_off(int row, int ch) {
    const int sw = (D == 128) ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
    return row * (D * 2) + ((ch ^ sw) << 4);
}

// Register stage of a ROWS x D bf16 tile: N = ROWS*D/8/256 16-B chunks per thread held in ONE first-class
// vector value (ext_vector_type) — an array here ends up in scratch once two stages are alive
// (hipcc keeps allocas that are conditionally re-loaded inside the unrolled-by-two loop in memory).
template <int N> struct stage_vec;
template <> struct stage_vec<1> { typedef __attribute__((ext_vector_type(4))) unsigned type; };
template <> struct stage_vec<2> { typedef __attribute__((ext_vector_type(8))) unsigned type; };
template <> struct stage_vec<4> { typedef __attribute__((ext_vector_type(16))) unsigned type; };
template <int D, int ROWS, int NT = 256> using stage_t = typename stage_vec<ROWS * (D / 8) / NT>::type;

// global -> registers (16-B chunks), rows clamped to [0, max_row]; NT = threads of the block
template <int D, int ROWS, int NT = 256>
__device__ __forceinline__ stage_t<D, ROWS, NT> tile_load(const bf16_t* base, long rs, int row0, int max_row) {
    constexpr int CH = D / 8, N = ROWS * CH / NT;
    stage_t<D, ROWS, NT> out;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int id = i * NT + threadIdx.x;
        const int r = id / CH, c = id % CH;
        const int gr = min(row0 + r, max_row);
        const uint4 v = *(const uint4*)(base + (long)gr * rs + c * 8);
        out[4 * i + 0] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
    }
    return out;
}
template <int D, int ROWS, int NT = 256>
__device__ __forceinline__ void tile_store(char* img, const stage_t<D, ROWS, NT>& regs) {
    constexpr int CH = D / 8, N = ROWS * CH / NT;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int id = i * NT + threadIdx.x;
        const int r = id / CH, c = id % CH;
        *(uint4*)(img + img_off<D>(r, c)) = make_uint4(regs[4 * i + 0], regs[4 * i + 1], regs[4 * i + 2], regs[4 * i + 3]);
    }
}

// A operand (32x32x16) read by rows: lane (row = row0 + (lane&31), k = 16*ks + 8*(lane>>5) + j)
template <int D>
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int row0, int ks, int lane) {
    const int r = row0 + (lane & 31);
    return *(const bf16x8*)(img + img_off<D>(r, 2 * ks + (lane >> 5)));
}

// Transposed operand from a row-major [row][col] image: returns, for lane (c = col0 + (lane&31), h = lane>>5),
// the 8 elements img[row0 + 8*(j>>2) + 4*h + (j&3)][c], j = 0..7  (the k order of an accumulator-fed MFMA).
template <int D>
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int row0, int col0, int lane) {
    const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int q = i >> 2, pp = i & 3;
    const int c = col0 + 16 * g + 4 * pp;                  // first column of this lane's 4-element piece
    const int r_lo = row0 + 4 * h + q;
    const int a0 = img_off<D>(r_lo, c >> 3) + ((c & 4) << 1);
    const int a1 = img_off<D>(r_lo + 8, c >> 3) + ((c & 4) << 1);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + a1));
    bf16x8 out;
    out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
    out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
    return out;
}

// accumulator registers 8s..8s+7 -> bf16 fragment of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)x[8 * s + j];
    return o;
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// XCD-aware work order for the query-block kernels.  Blocks are dealt round-robin over the 8 XCDs (block id % 8), each with
// its own L2: with (q-block, head, batch) = blockIdx the q-blocks of one (batch, head) landed on 8 different XCDs and every
// XCD fetched that head's K / V for itself (rocprofv3 FETCH_SIZE of the Whisper shape: 528 MB per launch against 92 MB of
// Q, K, V).  Here XCD x walks the (batch, kv head) groups g = x, x + 8, ... and, inside a group, the query heads of the GQA
// group and their q-blocks back to back, so the K / V tiles of a group are re-read from that XCD's L2.  Speed only: any
// placement is correct.  Causal: q-blocks in descending order (most keys first).
__device__ __forceinline__ void attn_work_item(const AttnArgs& p, int nq, int& qblk, int& h, int& b) {
    const int id = blockIdx.x, per = nq * (p.Hq / p.Hkv), ngrp = p.B * p.Hkv;       // per = blocks of one (batch, kv head) group
    int g, r;
    if ((ngrp & 7) == 0) {
        const int xcd = id & 7, slot = id >> 3;
        g = (slot / per) * 8 + xcd;
        r = slot % per;
    } else {
        g = id / per;
        r = id % per;
    }
    b = g / p.Hkv;
    h = (g % p.Hkv) * (p.Hq / p.Hkv) + r / nq;
    qblk = r % nq;
    if (p.causal) qblk = nq - 1 - qblk;
}

// ------------------------------------------------------------------------------------------ forward
// Occupancy beats prefetch depth here (measured, tools/attn_bench.py): with K/V register-staged ONE tile ahead the
// kernel fits 256 VGPRs at D=128 (160 at D=64) -> 2 (3) blocks per CU, whose MFMA / softmax / staging phases overlap
// each other: LLM shape 104 -> 69 us, Whisper shape 190 -> 157 us vs two tiles ahead at one block per CU.
#ifndef ATTN_FWD_PF
#define ATTN_FWD_PF 1
#endif
// NW = waves per block (32 query rows each): 4, or 2 for Sq <= 64 (the Q-Former's 64 prompt queries: with 4 waves half of
// every block computed clamped duplicate rows, and 640 four-wave blocks left most of the chip's wave slots empty)
#ifndef ATTN_FWD64_WAVES
#define ATTN_FWD64_WAVES 3
#endif
template <int D, bool DROP, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? (D == 64 ? ATTN_FWD64_WAVES : 2) : 2) void attn_fwd_k(AttnArgs p) {
    constexpr int NT = 64 * NW, QB = 32 * NW;
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * D * 2];
    char* kimg = lds;
    char* vimg = lds + 64 * D * 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h2 = lane >> 5;    // wave index in an SGPR: everything derived from it (q0, mask tests) is scalar -> real branches
    int qblk, h, b;
    attn_work_item(p, (p.Sq + QB - 1) / QB, qblk, h, b);
    const int qb0 = qblk * QB, q0 = qb0 + wave * 32;
    const int hk = h / (p.Hq / p.Hkv);
    const int qcol = q0 + (lane & 31);

    const bf16_t* qptr = p.Q + (long)b * p.q_bs + (long)min(qcol, p.Sq - 1) * p.q_rs + (long)h * D;
    bf16x8 qf[D / 16];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) qf[ds] = *(const bf16x8*)(qptr + 16 * ds + 8 * h2);

    const int coff = p.Sk - p.Sq;
    const int kv_lo = p.kv_start ? max(0, min(p.kv_start[b], p.Sk)) : 0;
    int kv_hi = p.Sk;
    if (p.causal) kv_hi = min(p.Sk, min(qb0 + QB - 1, p.Sq - 1) + coff + 1);
    const int t_lo = kv_lo / 64, t_hi = (kv_hi + 63) / 64;
    const int q_abs = qcol + coff;
    const int wave_kmax = p.causal ? min(q0 + 31, p.Sq - 1) + coff : p.Sk - 1;   // last key any row of this wave may see

    const bf16_t* kbase = p.K + (long)b * p.k_bs + (long)hk * D;
    const bf16_t* vbase = p.V + (long)b * p.v_bs + (long)hk * D;

    f32x16 oacc[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // K/V tiles are register-staged PF tiles ahead (PF = 2: two register sets, loop unrolled by two)
    constexpr int PF = ATTN_FWD_PF;                        // prefetch distance in tiles (1 or 2)
    stage_t<D, 64, NT> kr0 = {}, vr0 = {}, kr1 = {}, vr1 = {};
    if (t_lo < t_hi) {
        kr0 = tile_load<D, 64, NT>(kbase, p.k_rs, t_lo * 64, p.Sk - 1);
        vr0 = tile_load<D, 64, NT>(vbase, p.v_rs, t_lo * 64, p.Sk - 1);
    }
    if (PF == 2 && t_lo + 1 < t_hi) {
        kr1 = tile_load<D, 64, NT>(kbase, p.k_rs, (t_lo + 1) * 64, p.Sk - 1);
        vr1 = tile_load<D, 64, NT>(vbase, p.v_rs, (t_lo + 1) * 64, p.Sk - 1);
    }
    auto compute = [&](const int kt) __attribute__((always_inline)) {
        if (kt * 64 > wave_kmax) return;                   // wave-uniform: nothing visible in this tile
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
        {
            // K fragments of the first 32 keys are all read before the first MFMA; those of the second 32 keys are read while
            // the first chain runs (hipcc otherwise emits read -> wait -> MFMA pairs: one exposed LDS latency per MFMA)
            constexpr int NS = (ATTN_ABL == 4 ? 1 : D / 16);
            bf16x8 ka[NS], kbf[NS];
#pragma unroll
            for (int ds = 0; ds < NS; ++ds) ka[ds] = frag_rows<D>(kimg, 0, ds, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ds = 0; ds < NS; ++ds) {
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ds], qf[ds], st[0], 0, 0, 0);
                kbf[ds] = frag_rows<D>(kimg, 32, ds, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ds = 0; ds < NS; ++ds) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kbf[ds], qf[ds], st[1], 0, 0, 0);
        }
        // masks only on boundary tiles (wave-uniform): ragged Sk, left padding, causal diagonal
        const bool need_mask = (kt * 64 + 63 >= p.Sk) || (kt * 64 < kv_lo) || (p.causal && kt * 64 + 63 > q0 + coff);
        if (need_mask) {
            asm volatile("; boundary tile" ::: "memory");          // keeps this a BRANCH: hipcc otherwise if-converts the 128 compares
#pragma unroll                                                      // + selects into every tile's straight-line code
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + acc_row(r, lane);
                    const bool ok = key < p.Sk && key >= kv_lo && (!p.causal || key <= q_abs);
                    st[kb][r] = ok ? st[kb][r] : -INFINITY;
                }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[kb][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * p.scale_log2;         // scale > 0: max commutes
        const float mnew = fmaxf(m, tmax);
        const float muse = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = __builtin_amdgcn_exp2f(m - muse);
        // score -> exponent argument and the row sum on register PAIRS (v_pk_fma_f32 / v_pk_add_f32: half the VALU instructions
        // of the two passes; the kernel is bound by instruction issue)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 rs2 = {0.f, 0.f};
        const f32x2 sc2 = {p.scale_log2, p.scale_log2}, nm2 = {-muse, -muse};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 sv = {st[kb][r], st[kb][r + 1]};
                const f32x2 av = sv * sc2 + nm2;
                f32x2 pv;
                pv.x = ATTN_ABL == 3 ? sv.x : __builtin_amdgcn_exp2f(av.x);
                pv.y = ATTN_ABL == 3 ? sv.y : __builtin_amdgcn_exp2f(av.y);
                st[kb][r] = pv.x;
                st[kb][r + 1] = pv.y;
                rs2 += pv;
            }
        float rs = rs2.x + rs2.y;
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        if constexpr (DROP) {
            // inverted dropout of the probabilities that enter P·V (the normaliser l keeps the full sum).  Accumulator
            // registers 2j, 2j+1 are ADJACENT keys = one element pair = one hash (common.h; Sk even is checked on the host)
            const unsigned long rowbase = (((unsigned long)b * p.Hq + h) * p.Sq + min(qcol, p.Sq - 1)) * p.Sk;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const unsigned long i0 = rowbase + (kt * 64 + kb * 32 + acc_row(r, lane));
                    const unsigned hsh = desta_rng32(p.seed_lo, p.seed_hi, i0 >> 1);     // Sk is even (host check): i0 is even
                    st[kb][r] = (hsh & 0xffffu) >= p.drop_thresh ? st[kb][r] * p.drop_scale : 0.f;
                    st[kb][r + 1] = (hsh >> 16) >= p.drop_thresh ? st[kb][r + 1] * p.drop_scale : 0.f;
                }
        }
        if (__any(mnew != m)) {                            // rescale only when some row's max moved
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        m = mnew;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pb = acc_frag(st[kb], s);
#pragma unroll
                for (int i = 0; i < (ATTN_ABL == 5 ? 1 : D / 32); ++i)
                    oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(vimg, kb * 32 + 16 * s, i * 32, lane), pb, oacc[i], 0, 0, 0);
            }
    };
#define DESTA_KV_STAGE(KT, KR, VR)                                                   \
    __syncthreads(); /* previous tile's LDS reads are done */                        \
    asm volatile("; stage " #KR ::: "memory"); /* distinct text: keeps the two halves from being tail-merged */ \
    if (ATTN_ABL != 1 || (KT) == t_lo) {                                             \
    tile_store<D, 64, NT>(kimg, KR);                                                     \
    tile_store<D, 64, NT>(vimg, VR);                                                     \
    }                                                                                \
    __syncthreads();                                                                 \
    if (ATTN_ABL != 2 && (KT) + PF < t_hi) {                                         \
        KR = tile_load<D, 64, NT>(kbase, p.k_rs, ((KT) + PF) * 64, p.Sk - 1);            \
        VR = tile_load<D, 64, NT>(vbase, p.v_rs, ((KT) + PF) * 64, p.Sk - 1);            \
    }
    for (int kt = t_lo; kt < t_hi; kt += 2) {
        DESTA_KV_STAGE(kt, kr0, vr0)
        compute(kt);
        if (kt + 1 < t_hi) {
            if constexpr (PF == 2) {
                DESTA_KV_STAGE(kt + 1, kr1, vr1)
            } else {
                DESTA_KV_STAGE(kt + 1, kr0, vr0)
            }
            compute(kt + 1);
        }
    }
#undef DESTA_KV_STAGE

    if (qcol < p.Sq) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        bf16_t* optr = p.O + (long)b * p.o_bs + (long)qcol * p.o_rs + (long)h * D;
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(oacc[i][4 * rq + e] * inv);
                *(u16x4*)(optr + i * 32 + 8 * rq + 4 * h2) = o;
            }
        if (p.O32) {
            float* o32 = p.O32 + (long)b * p.o_bs + (long)qcol * p.o_rs + (long)h * D;
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *(f32x4*)(o32 + i * 32 + 8 * rq + 4 * h2) = f32x4{oacc[i][4 * rq] * inv, oacc[i][4 * rq + 1] * inv, oacc[i][4 * rq + 2] * inv, oacc[i][4 * rq + 3] * inv};
        }
        if (p.lse && h2 == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qcol] = l > 0.f ? m + log2f(l) : INFINITY;
    }
}

// ------------------------------------------------------------------------------------------ forward, 8 waves per block
// The forward for long query ranges (LLM prefill / training: D = 128 causal GQA; Whisper: D = 64, 1500 x 1500).  Same
// orientation as above (key on the accumulator rows, query on the lane, P^T straight from the S^T accumulator into the P.V
// MFMAs), restructured around what the four-wave kernel spends its time on (rocprofv3: mfma-busy 0.18-0.29; per 64-key tile and
// wave 32 MFMAs beside ~330 VALU instructions, and the two waves of a SIMD doing the same phase at the same time):
//   * EIGHT waves share one K / V tile: a block is 256 query rows of one head or, under GQA, HPB heads x 256 / HPB rows of one
//     kv head (the heads of a group read the same K / V): half / a quarter of the staging work per unit of MFMA work, and
//     under the causal mask 64- or 128-row blocks waste less of the diagonal tiles;
//   * LDS double buffer, ONE raw s_barrier per tile (no vmcnt drain): tile t+1 is written to the other buffer in the middle of
//     tile t from registers loaded one tile earlier, and the global loads of tile t+2 are issued right behind it: every load
//     has a whole tile of compute to land;
//   * (STAGGER = true, off by default: measured) waves 4-7 — the SIMD partners of waves 0-3 — half a tile behind, on a
//     three-slot ring: between two barriers waves 0-3 do [S, softmax, P.V] of tile t, waves 4-7 [P.V of t-1, S, softmax of t].
//     Equal on the LLM shape (55.9 vs 55.3 us), SLOWER on the Whisper shape (154 vs 120 us): the partners de-synchronise by
//     themselves once one of them is ahead in its MFMA chain, and the forced pairing is worse than the one that forms;
//   * deferred rescale (guide T13): the running reference m only moves when some row's tile maximum exceeds it by more than
//     2^ATTN_DEFER_LOG2: on all other tiles there is no alpha, no O rescale, no l rescale (P <= 2^ATTN_DEFER_LOG2: relative bf16
//     rounding is unchanged, sums stay far inside fp32 range);
//   * softmax VALU diet: row maximum by v_max3 written as asm (fmaxf on MFMA results makes hipcc canonicalise every operand
//     with an extra v_max: 51 instead of 16 instructions), one v_permlane32_swap to cross the half-waves, the row sum kept as a
//     per-lane partial (the halves are combined once, in the epilogue); LDS read addresses and the per-thread staging offsets
//     are computed once (one add of the slot base per read; global loads use a scalar tile base + 32-bit lane offset);
//   * O is stored as 16-byte pieces after a permlane32 exchange of the packed halves (guide T21).
// Work order: XCD x walks the (batch, kv head) groups x, x + 8, ...; causal launches the heaviest q-blocks of all its groups
// first (the K / V of an XCD's ten groups fit its L2 together), non-causal keeps the q-blocks of a group back to back.
#ifndef ATTN_DEFER_LOG2
#define ATTN_DEFER_LOG2 6.0f
#endif
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float swap_halves_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = [lo, lo], r[1] = [hi, hi]
    return vmax2(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float swap_halves_sum(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ bf16x8 tr_pair(const char* lo, const char* hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)lo);
    const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)hi);
    return __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);            // one register tuple: no element-wise moves
}

// MINW = 4 (D = 64 only): two 8-wave blocks per CU (<= 128 registers)
template <int D, bool CAUSAL, int HPB, int MINW = 2, bool STAGGER = false>
__global__ __launch_bounds__(512, MINW) void attn_fwd8_k(AttnArgs p) {
    constexpr int NT = 512, RB = 256 / HPB, WPH = RB / 32;         // query rows of a block (per head), waves per head
    constexpr int TILE = 64 * D * 2;                               // bytes of one 64-key K (or V) image
    constexpr int SLOT = 2 * TILE, NSLOT = STAGGER ? 3 : 2;        // lockstep: double buffer; staggered: tile t-1's V is still read while t+1 is written
    constexpr int NCH = 64 * (D / 8) / NT;                         // 16-byte chunks per thread, tile and tensor (2 / 1)
    __shared__ __attribute__((aligned(16))) char lds[NSLOT * SLOT];   // ring of [K image, V image]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h2 = lane >> 5;
    const int G = p.Hq / p.Hkv, nq = (p.Sq + RB - 1) / RB, hgroups = G / HPB, ngrp = p.B * p.Hkv;
    int g, qblk, hg;
    {
        const int id = blockIdx.x, per = nq * hgroups;
        if ((ngrp & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3, gpx = ngrp >> 3;
            if (CAUSAL) {
                const int r = slot % (hgroups * gpx);
                qblk = nq - 1 - slot / (hgroups * gpx); hg = r / gpx; g = (r % gpx) * 8 + xcd;
            } else {
                const int r = slot % per;
                g = (slot / per) * 8 + xcd; hg = r / nq; qblk = r % nq;
            }
        } else if (CAUSAL) {
            const int r = id % (hgroups * ngrp);
            qblk = nq - 1 - id / (hgroups * ngrp); hg = r / ngrp; g = r % ngrp;
        } else {
            const int r = id % per;
            g = id / per; hg = r / nq; qblk = r % nq;
        }
    }
    const int b = g / p.Hkv, hk = g % p.Hkv;
    const int h = hk * G + hg * HPB + wave / WPH;
    const int qb0 = qblk * RB, q0 = qb0 + (wave % WPH) * 32;
    const int qcol = q0 + (lane & 31);
    const bool wave_on = q0 < p.Sq;                                 // scalar: a wave past the ragged end only helps staging

    const int coff = p.Sk - p.Sq;
    const int kv_lo = p.kv_start ? max(0, min(p.kv_start[b], p.Sk)) : 0;
    int kv_hi = p.Sk;
    if (CAUSAL) kv_hi = min(p.Sk, min(qb0 + RB - 1, p.Sq - 1) + coff + 1);
    const int t_lo = kv_lo / 64, t_hi = (kv_hi + 63) / 64;
    const int q_abs = qcol + coff;
    const int wave_kmax = CAUSAL ? min(q0 + 31, p.Sq - 1) + coff : p.Sk - 1;

    const char* kbase = (const char*)(p.K + (long)b * p.k_bs + (long)hk * D);
    const char* vbase = (const char*)(p.V + (long)b * p.v_bs + (long)hk * D);
    // staging: this thread's NCH chunks of a tile: (row r, chunk c) -> byte offset from the tile's first row (32-bit; the tile
    // base is wave-uniform: scalar base + vector offset loads) and its place in the swizzled LDS image
    int srow[NCH], sdst[NCH];
    unsigned ksrc[NCH], vsrc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = i * NT + threadIdx.x, r = id / (D / 8), c = id % (D / 8);
        srow[i] = r;
        sdst[i] = img_off<D>(r, c);
        ksrc[i] = (unsigned)(r * (int)p.k_rs + c * 8) * 2u;
        vsrc[i] = (unsigned)(r * (int)p.v_rs + c * 8) * 2u;
    }
    // ONE first-class vector value per tensor (an array that is conditionally re-loaded inside the loop is kept in memory by
    // hipcc — promoted to LDS here: every load was waited for and parked in LDS right behind its issue)
    typedef typename stage_vec<NCH>::type stage8_t;
    stage8_t kr = {}, vr = {};
    auto load_tile = [&](const int kt) __attribute__((always_inline)) {
        const char* kb_ = kbase + (long)kt * 64 * p.k_rs * 2;
        const char* vb_ = vbase + (long)kt * 64 * p.v_rs * 2;
        const bool ragged = kt * 64 + 64 > p.Sk;                    // last tile (or a block without tiles): clamp the rows
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            unsigned ko = ksrc[i], vo = vsrc[i];
            if (ragged) {
                const int rc = max(min(srow[i], p.Sk - 1 - kt * 64), -kt * 64), c = (i * NT + (int)threadIdx.x) % (D / 8);
                ko = (unsigned)(rc * (int)p.k_rs + c * 8) * 2u;     // (rc < 0 only when the block has no tile: row 0 of the tensor)
                vo = (unsigned)(rc * (int)p.v_rs + c * 8) * 2u;
            }
            const uint4 a = *(const uint4*)(kb_ + (long)(int)ko), c4 = *(const uint4*)(vb_ + (long)(int)vo);
            kr[4 * i + 0] = a.x; kr[4 * i + 1] = a.y; kr[4 * i + 2] = a.z; kr[4 * i + 3] = a.w;
            vr[4 * i + 0] = c4.x; vr[4 * i + 1] = c4.y; vr[4 * i + 2] = c4.z; vr[4 * i + 3] = c4.w;
        }
    };
    auto store_tile = [&](char* slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *(uint4*)(slot + sdst[i]) = make_uint4(kr[4 * i + 0], kr[4 * i + 1], kr[4 * i + 2], kr[4 * i + 3]);
            *(uint4*)(slot + TILE + sdst[i]) = make_uint4(vr[4 * i + 0], vr[4 * i + 1], vr[4 * i + 2], vr[4 * i + 3]);
        }
    };

    // prologue, straight-line on purpose: first K / V tile, then the Q fragments; loads return in order, so once the Q
    // fragments are in (the empty asm makes the compiler wait for them HERE) nothing of the prologue is still in flight and
    // the loop's vmcnt bookkeeping covers only the K / V stream (with the Q loads pending at the loop head hipcc makes every
    // tile's S MFMAs wait for vmcnt(3..0), i.e. for the loads of tile t+2 issued half a tile earlier).  Rows are clamped:
    // the loads are legal even when the block has no tile at all.
    load_tile(t_lo);
    const bf16_t* qptr = p.Q + (long)b * p.q_bs + (long)min(qcol, p.Sq - 1) * p.q_rs + (long)h * D;
    bf16x8 qf[D / 16];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) qf[ds] = *(const bf16x8*)(qptr + 16 * ds + 8 * h2);
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) asm volatile("" :: "v"(qf[ds]));
    store_tile(lds);
    load_tile(t_lo + 1);

    // LDS read addresses inside a slot (lane constants; "+v": keep them in registers instead of re-deriving them per tile)
    int kofs[D / 16], vofs[D / 32][2];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) {
        kofs[ds] = img_off<D>(lane & 31, 2 * ds + h2);             // K rows 0..31 (rows 32..63: + 32 rows, same swizzle)
        asm volatile("" : "+v"(kofs[ds]));
    }
    {
        const int i16 = lane & 15, gq = (lane >> 4) & 1, qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
        for (int i = 0; i < D / 32; ++i) {
            const int c = i * 32 + 16 * gq + 4 * pp;
            vofs[i][0] = TILE + img_off<D>(4 * h2 + qq, c >> 3) + ((c & 4) << 1);          // V rows r0 + 4 h2 + q (r0 = multiples of 16: same swizzle)
            vofs[i][1] = TILE + img_off<D>(4 * h2 + qq + 8, c >> 3) + ((c & 4) << 1);
            asm volatile("" : "+v"(vofs[i][0]), "+v"(vofs[i][1]));
        }
    }

    f32x16 oacc[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;                                   // m: reference in scaled log2 units (may lag the true maximum by <= 2^DEFER); l: this lane's partial sum
    f32x16 st[2];                                                   // S^T of the current tile, then its probabilities (waves 4-7 carry them across the barrier)
#define ATTN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   /* LDS traffic of this wave is done; global loads stay in flight */
    ATTN_BARRIER();

    // S^T = K Q^T of tile kt (slot base `sb`), boundary masks, deferred-rescale online softmax: st := probabilities
    auto scores = [&](const int kt, const int sb) __attribute__((always_inline)) {
        if (!(wave_on && kt * 64 <= wave_kmax)) return;             // scalar
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
        bf16x8 ka[D / 16], kbf[D / 16];
        int kad[D / 16];                                            // ONE add of the slot base per address register; the rest are immediates
#pragma unroll
        for (int ds = 0; ds < D / 16; ++ds) { kad[ds] = kofs[ds] + sb; asm volatile("" : "+v"(kad[ds])); }
#pragma unroll
        for (int ds = 0; ds < D / 16; ++ds) ka[ds] = *(const bf16x8*)(lds + kad[ds]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ds = 0; ds < D / 16; ++ds) {
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ds], qf[ds], st[0], 0, 0, 0);
            kbf[ds] = *(const bf16x8*)(lds + kad[ds] + 32 * D * 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ds = 0; ds < D / 16; ++ds) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kbf[ds], qf[ds], st[1], 0, 0, 0);
        const bool need_mask = (kt * 64 + 63 >= p.Sk) || (kt * 64 < kv_lo) || (CAUSAL && kt * 64 + 63 > q0 + coff);
        if (need_mask) {
            asm volatile("; boundary tile" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + acc_row(r, lane);
                    const bool ok = key < p.Sk && key >= kv_lo && (!CAUSAL || key <= q_abs);
                    st[kb][r] = ok ? st[kb][r] : -INFINITY;
                }
        }
        float tmax = vmax3(st[0][0], st[1][0], st[0][1]);
        tmax = vmax3(tmax, st[1][1], st[0][2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) tmax = vmax3(tmax, st[1][r], st[0][r + 1]);
        tmax = vmax2(tmax, st[1][15]);
        tmax = swap_halves_max(tmax) * p.scale_log2;                                          // scale > 0: max commutes
        if (!__all(!(tmax > m + ATTN_DEFER_LOG2))) {                                          // some row outgrew its reference: move every row's
            const float mnew = vmax2(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - (mnew == -INFINITY ? 0.f : mnew)); // m = -inf: 0 (O = l = 0 anyway)
            l *= alpha;
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            m = mnew;
        }
        const float nmu = (m == -INFINITY) ? 0.f : -m;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[kb][r] = __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.scale_log2, nmu));
                l += st[kb][r];
            }
    };
    // O^T += V^T P^T of tile kt
    auto pv = [&](const int kt, const int sb) __attribute__((always_inline)) {
        if (!(wave_on && kt * 64 <= wave_kmax)) return;
        int vad[D / 32][2];
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { vad[i][j] = vofs[i][j] + sb; asm volatile("" : "+v"(vad[i][j])); }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pb = acc_frag(st[kb], s);
#pragma unroll
                for (int i = 0; i < D / 32; ++i)
                    oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_pair(lds + vad[i][0] + (kb * 32 + 16 * s) * D * 2, lds + vad[i][1] + (kb * 32 + 16 * s) * D * 2), pb, oacc[i], 0, 0, 0);
            }
    };
    // tile kt+1: registers -> its ring slot (last read two barriers ago); then the global loads of tile kt+2
    auto stage = [&](const int kt, const int sb_next) __attribute__((always_inline)) {
        if (kt + 1 < t_hi) {
            store_tile(lds + sb_next);
            if (kt + 2 < t_hi) load_tile(kt + 2);
        }
    };
    auto next_slot = [](int sb) { return sb == (NSLOT - 1) * SLOT ? 0 : sb + SLOT; };
    int sb = 0;                                                     // slot of tile kt (scalar)
    if (!STAGGER || wave < 4) {
        for (int kt = t_lo; kt < t_hi; ++kt) {
            const int sn = next_slot(sb);
            scores(kt, sb);
            stage(kt, sn);
            pv(kt, sb);
            ATTN_BARRIER();
            sb = sn;
        }
    } else if (t_lo < t_hi) {
        // half a tile behind: the same number of barriers as waves 0-3 (one per tile), P.V of tile kt after the barrier that ends it
        scores(t_lo, sb);
        stage(t_lo, next_slot(sb));
        ATTN_BARRIER();
        for (int kt = t_lo; kt + 1 < t_hi; ++kt) {
            const int sn = next_slot(sb);
            pv(kt, sb);
            stage(kt + 1, next_slot(sn));
            scores(kt + 1, sn);
            ATTN_BARRIER();
            sb = sn;
        }
        pv(t_hi - 1, sb);
    }
#undef ATTN_BARRIER

    if (wave_on && qcol < p.Sq) {
        const float lt = swap_halves_sum(l);
        const float inv = lt > 0.f ? 1.0f / lt : 0.f;
        bf16_t* optr = p.O + (long)b * p.o_bs + (long)qcol * p.o_rs + (long)h * D;
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // registers 8j..8j+3: d = 32i + 16j + 4*h2 + e; 8j+4..8j+7: d = 32i + 16j + 8 + 4*h2 + e.  After the exchange a lane
                // of the lower half holds d = 32i + 16j + 0..7, its partner in the upper half 32i + 16j + 8..15: one 16-byte store each
                unsigned a[2], c[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    a[e] = (unsigned)f2bf(oacc[i][8 * j + 2 * e] * inv) | ((unsigned)f2bf(oacc[i][8 * j + 2 * e + 1] * inv) << 16);
                    c[e] = (unsigned)f2bf(oacc[i][8 * j + 4 + 2 * e] * inv) | ((unsigned)f2bf(oacc[i][8 * j + 4 + 2 * e + 1] * inv) << 16);
                }
                const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                *(uint4*)(optr + i * 32 + 16 * j + 8 * h2) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
        if (p.O32) {
            float* o32 = p.O32 + (long)b * p.o_bs + (long)qcol * p.o_rs + (long)h * D;
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *(f32x4*)(o32 + i * 32 + 8 * rq + 4 * h2) = f32x4{oacc[i][4 * rq] * inv, oacc[i][4 * rq + 1] * inv, oacc[i][4 * rq + 2] * inv, oacc[i][4 * rq + 3] * inv};
        }
        if (p.lse && h2 == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qcol] = lt > 0.f ? m + log2f(lt) : INFINITY;
    }
}

// delta[b,h,q] = sum_d dO[q,d] * O[q,d]
template <int D>
__global__ __launch_bounds__(256) void attn_delta_k(AttnArgs p, float* __restrict__ delta) {
    constexpr int G = D / 8;                               // lanes per row
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long rowid = gid / G;
    const int j = (int)(gid % G);
    const long total = (long)p.B * p.Hq * p.Sq;
    const bool active = rowid < total;
    const long rr = active ? rowid : 0;
    const int q = (int)(rr % p.Sq), h = (int)((rr / p.Sq) % p.Hq), b = (int)(rr / ((long)p.Sq * p.Hq));
    const u16x8 g = *(const u16x8*)(p.dO + (long)b * p.do_bs + (long)q * p.do_rs + (long)h * D + 8 * j);
    float s = 0.f;
    if (p.O32) {                                           // the unrounded forward output (see desta_attn_desc.O_f32)
        const float* o32 = p.O32 + (long)b * p.o_bs + (long)q * p.o_rs + (long)h * D + 8 * j;
        const f32x4 o0 = *(const f32x4*)o32, o1 = *(const f32x4*)(o32 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += o0[e] * bf2f(g[e]) + o1[e] * bf2f(g[4 + e]);
    } else {
        const u16x8 o = *(const u16x8*)(p.O + (long)b * p.o_bs + (long)q * p.o_rs + (long)h * D + 8 * j);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += bf2f(o[e]) * bf2f(g[e]);
    }
#pragma unroll
    for (int o2 = 1; o2 < G; o2 <<= 1) s += __shfl_xor(s, o2, 64);
    if (active && j == 0) delta[rowid] = s;
}

// ------------------------------------------------------------------------------------------ backward: dQ
// same trade as the forward: one tile ahead + two blocks per CU (131 -> 103 us on the LLM shape; two tiles ahead
// squeezed into 256 registers spills and is slower than either)
#ifndef ATTN_DQ_PF
#define ATTN_DQ_PF 1
#endif
#ifndef ATTN_DQ_BLOCKS
#define ATTN_DQ_BLOCKS 2
#endif
template <int D, bool DROP, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? ATTN_DQ_BLOCKS : 2) void attn_bwd_dq_k(AttnArgs p) {
    constexpr int NT = 64 * NW, QB = 32 * NW;
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * D * 2];
    char* kimg = lds;
    char* vimg = lds + 64 * D * 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h2 = lane >> 5;    // wave index in an SGPR: everything derived from it (q0, mask tests) is scalar -> real branches
    int qblk, h, b;
    attn_work_item(p, (p.Sq + QB - 1) / QB, qblk, h, b);
    const int qb0 = qblk * QB, q0 = qb0 + wave * 32;
    const int hk = h / (p.Hq / p.Hkv);
    const int qcol = q0 + (lane & 31), qc = min(qcol, p.Sq - 1);

    const bf16_t* qptr = p.Q + (long)b * p.q_bs + (long)qc * p.q_rs + (long)h * D;
    const bf16_t* gptr = p.dO + (long)b * p.do_bs + (long)qc * p.do_rs + (long)h * D;
    bf16x8 qf[D / 16], gf[D / 16];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) {
        qf[ds] = *(const bf16x8*)(qptr + 16 * ds + 8 * h2);
        gf[ds] = *(const bf16x8*)(gptr + 16 * ds + 8 * h2);
    }
    const long stat = ((long)b * p.Hq + h) * p.Sq + qc;
    const float lse = p.lse[stat], dlt = p.delta[stat];

    const int coff = p.Sk - p.Sq;
    const int kv_lo = p.kv_start ? max(0, min(p.kv_start[b], p.Sk)) : 0;
    int kv_hi = p.Sk;
    if (p.causal) kv_hi = min(p.Sk, min(qb0 + QB - 1, p.Sq - 1) + coff + 1);
    const int t_lo = kv_lo / 64, t_hi = (kv_hi + 63) / 64;
    const int q_abs = qcol + coff;
    const int wave_kmax = p.causal ? min(q0 + 31, p.Sq - 1) + coff : p.Sk - 1;

    const bf16_t* kbase = p.K + (long)b * p.k_bs + (long)hk * D;
    const bf16_t* vbase = p.V + (long)b * p.v_bs + (long)hk * D;

    f32x16 dq[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    // K/V tiles are register-staged PF tiles ahead (PF = 2: two register sets, loop unrolled by two)
    constexpr int PF = ATTN_DQ_PF;                         // prefetch distance in tiles (1 or 2)
    stage_t<D, 64, NT> kr0 = {}, vr0 = {}, kr1 = {}, vr1 = {};
    if (t_lo < t_hi) {
        kr0 = tile_load<D, 64, NT>(kbase, p.k_rs, t_lo * 64, p.Sk - 1);
        vr0 = tile_load<D, 64, NT>(vbase, p.v_rs, t_lo * 64, p.Sk - 1);
    }
    if (PF == 2 && t_lo + 1 < t_hi) {
        kr1 = tile_load<D, 64, NT>(kbase, p.k_rs, (t_lo + 1) * 64, p.Sk - 1);
        vr1 = tile_load<D, 64, NT>(vbase, p.v_rs, (t_lo + 1) * 64, p.Sk - 1);
    }
    auto compute = [&](const int kt) __attribute__((always_inline)) {
        if (kt * 64 > wave_kmax) return;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < D / 16; ++ds) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(kimg, kb * 32, ds, lane), qf[ds], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(vimg, kb * 32, ds, lane), gf[ds], dp, 0, 0, 0);
            }
            const bool need_mask = (kt * 64 + kb * 32 + 31 >= p.Sk) || (kt * 64 + kb * 32 < kv_lo) ||
                                   (p.causal && kt * 64 + kb * 32 + 31 > q0 + coff);
            unsigned keep = 0xffffu;                                          // bit r: probability r of this lane survived dropout
            if constexpr (DROP) {
                // registers 2j, 2j+1 hold adjacent keys = one element pair = one hash (common.h; Sk even: host check)
                const unsigned long rowbase = (((unsigned long)b * p.Hq + h) * p.Sq + qc) * p.Sk;
                keep = 0;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const unsigned long i0 = rowbase + (kt * 64 + kb * 32 + acc_row(r, lane));
                    const unsigned hsh = desta_rng32(p.seed_lo, p.seed_hi, i0 >> 1);
                    keep |= ((hsh & 0xffffu) >= p.drop_thresh ? 1u : 0u) << r;
                    keep |= ((hsh >> 16) >= p.drop_thresh ? 1u : 0u) << (r + 1);
                }
            }
            if (need_mask) {                                           // boundary tiles only (scalar branch, outside the element loop)
                asm volatile("; boundary tile" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + acc_row(r, lane);
                    const bool ok = key < p.Sk && key >= kv_lo && (!p.causal || key <= q_abs);
                    st[r] = ok ? st[r] : -INFINITY;                   // exp2(-inf) = 0
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[r], p.scale_log2, -lse));
                float dpr = dp[r];
                if constexpr (DROP) dpr = ((keep >> r) & 1u) ? dpr * p.drop_scale : 0.f;
                st[r] = pv * (dpr - dlt) * p.scale;        // dS^T
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 dsf = acc_frag(st, s);
#pragma unroll
                for (int i = 0; i < D / 32; ++i)
                    dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(kimg, kb * 32 + 16 * s, i * 32, lane), dsf, dq[i], 0, 0, 0);
            }
        }
    };
#define DESTA_KV_STAGE(KT, KR, VR)                                                   \
    __syncthreads(); /* previous tile's LDS reads are done */                        \
    asm volatile("; stage " #KR ::: "memory"); /* distinct text: keeps the two halves from being tail-merged */ \
    tile_store<D, 64, NT>(kimg, KR);                                                     \
    tile_store<D, 64, NT>(vimg, VR);                                                     \
    __syncthreads();                                                                 \
    if ((KT) + PF < t_hi) {                                                          \
        KR = tile_load<D, 64, NT>(kbase, p.k_rs, ((KT) + PF) * 64, p.Sk - 1);            \
        VR = tile_load<D, 64, NT>(vbase, p.v_rs, ((KT) + PF) * 64, p.Sk - 1);            \
    }
    for (int kt = t_lo; kt < t_hi; kt += 2) {
        DESTA_KV_STAGE(kt, kr0, vr0)
        compute(kt);
        if (kt + 1 < t_hi) {
            if constexpr (PF == 2) {
                DESTA_KV_STAGE(kt + 1, kr1, vr1)
            } else {
                DESTA_KV_STAGE(kt + 1, kr0, vr0)
            }
            compute(kt + 1);
        }
    }
#undef DESTA_KV_STAGE
    if (qcol < p.Sq) {
        bf16_t* optr = p.dQ + (long)b * p.dq_bs + (long)qcol * p.dq_rs + (long)h * D;
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(dq[i][4 * rq + e]);
                *(u16x4*)(optr + i * 32 + 8 * rq + 4 * h2) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ, 8 waves per block
// attn_fwd8_k's geometry for the dQ pass of long query ranges (seq_q >= 128, no dropout): 8 waves share each K / V tile (HPB
// heads of a GQA group x 256 / HPB query rows), LDS double buffer with ONE raw barrier per tile, loads two tiles ahead in
// registers, heaviest q-blocks first per XCD.  Per 32-key half of a tile and wave: S^T = K Q^T and dP^T = V dO^T (16 MFMAs,
// K / V rows by ds_read_b128), dS^T = P (dP - delta) scale in registers, dQ^T += K^T dS^T (8 MFMAs, K^T by ds_read_tr from
// the same image).  delta = rowsum(dO * O) is computed HERE, in the prologue, from the fragments the wave loads anyway (each
// query row belongs to exactly one wave), and written to `delta_out` for the dK / dV kernel that follows on the stream: the
// separate delta launch (0.53 ms per step) is gone.
template <int D, bool CAUSAL, int HPB>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq8_k(AttnArgs p, float* __restrict__ delta_out) {
    constexpr int NT = 512, RB = 256 / HPB, WPH = RB / 32;
    constexpr int TILE = 64 * D * 2, SLOT = 2 * TILE;
    constexpr int NCH = 64 * (D / 8) / NT;
    __shared__ __attribute__((aligned(16))) char lds[2 * SLOT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h2 = lane >> 5;
    const int G = p.Hq / p.Hkv, nq = (p.Sq + RB - 1) / RB, hgroups = G / HPB, ngrp = p.B * p.Hkv;
    int g, qblk, hg;
    {
        const int id = blockIdx.x, per = nq * hgroups;
        if ((ngrp & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3, gpx = ngrp >> 3;
            if (CAUSAL) {
                const int r = slot % (hgroups * gpx);
                qblk = nq - 1 - slot / (hgroups * gpx); hg = r / gpx; g = (r % gpx) * 8 + xcd;
            } else {
                const int r = slot % per;
                g = (slot / per) * 8 + xcd; hg = r / nq; qblk = r % nq;
            }
        } else if (CAUSAL) {
            const int r = id % (hgroups * ngrp);
            qblk = nq - 1 - id / (hgroups * ngrp); hg = r / ngrp; g = r % ngrp;
        } else {
            const int r = id % per;
            g = id / per; hg = r / nq; qblk = r % nq;
        }
    }
    const int b = g / p.Hkv, hk = g % p.Hkv;
    const int h = hk * G + hg * HPB + wave / WPH;
    const int qb0 = qblk * RB, q0 = qb0 + (wave % WPH) * 32;
    const int qcol = q0 + (lane & 31), qc = min(qcol, p.Sq - 1);
    const bool wave_on = q0 < p.Sq;

    const int coff = p.Sk - p.Sq;
    const int kv_lo = p.kv_start ? max(0, min(p.kv_start[b], p.Sk)) : 0;
    int kv_hi = p.Sk;
    if (CAUSAL) kv_hi = min(p.Sk, min(qb0 + RB - 1, p.Sq - 1) + coff + 1);
    const int t_lo = kv_lo / 64, t_hi = (kv_hi + 63) / 64;
    const int q_abs = qcol + coff;
    const int wave_kmax = CAUSAL ? min(q0 + 31, p.Sq - 1) + coff : p.Sk - 1;

    const char* kbase = (const char*)(p.K + (long)b * p.k_bs + (long)hk * D);
    const char* vbase = (const char*)(p.V + (long)b * p.v_bs + (long)hk * D);
    int srow[NCH], sdst[NCH];
    unsigned ksrc[NCH], vsrc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = i * NT + threadIdx.x, r = id / (D / 8), c = id % (D / 8);
        srow[i] = r;
        sdst[i] = img_off<D>(r, c);
        ksrc[i] = (unsigned)(r * (int)p.k_rs + c * 8) * 2u;
        vsrc[i] = (unsigned)(r * (int)p.v_rs + c * 8) * 2u;
    }
    typedef typename stage_vec<NCH>::type stage8_t;
    stage8_t kr = {}, vr = {};
    auto load_tile = [&](const int kt) __attribute__((always_inline)) {
        const char* kb_ = kbase + (long)kt * 64 * p.k_rs * 2;
        const char* vb_ = vbase + (long)kt * 64 * p.v_rs * 2;
        const bool ragged = kt * 64 + 64 > p.Sk;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            unsigned ko = ksrc[i], vo = vsrc[i];
            if (ragged) {
                const int rc = max(min(srow[i], p.Sk - 1 - kt * 64), -kt * 64), c = (i * NT + (int)threadIdx.x) % (D / 8);
                ko = (unsigned)(rc * (int)p.k_rs + c * 8) * 2u;
                vo = (unsigned)(rc * (int)p.v_rs + c * 8) * 2u;
            }
            const uint4 a = *(const uint4*)(kb_ + (long)(int)ko), c4 = *(const uint4*)(vb_ + (long)(int)vo);
            kr[4 * i + 0] = a.x; kr[4 * i + 1] = a.y; kr[4 * i + 2] = a.z; kr[4 * i + 3] = a.w;
            vr[4 * i + 0] = c4.x; vr[4 * i + 1] = c4.y; vr[4 * i + 2] = c4.z; vr[4 * i + 3] = c4.w;
        }
    };
    auto store_tile = [&](char* slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *(uint4*)(slot + sdst[i]) = make_uint4(kr[4 * i + 0], kr[4 * i + 1], kr[4 * i + 2], kr[4 * i + 3]);
            *(uint4*)(slot + TILE + sdst[i]) = make_uint4(vr[4 * i + 0], vr[4 * i + 1], vr[4 * i + 2], vr[4 * i + 3]);
        }
    };

    load_tile(t_lo);
    const bf16_t* qptr = p.Q + (long)b * p.q_bs + (long)qc * p.q_rs + (long)h * D;
    const bf16_t* gptr = p.dO + (long)b * p.do_bs + (long)qc * p.do_rs + (long)h * D;
    const bf16_t* optr = p.O + (long)b * p.o_bs + (long)qc * p.o_rs + (long)h * D;
    bf16x8 qf[D / 16], gf[D / 16];
    float dl = 0.f;
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) {
        qf[ds] = *(const bf16x8*)(qptr + 16 * ds + 8 * h2);
        gf[ds] = *(const bf16x8*)(gptr + 16 * ds + 8 * h2);
        const bf16x8 of = *(const bf16x8*)(optr + 16 * ds + 8 * h2);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)of[j] * (float)gf[ds][j];
    }
    const long stat = ((long)b * p.Hq + h) * p.Sq + qc;
    const float nlse = -p.lse[stat];
    const float dlt = swap_halves_sum(dl);                          // this lane's half of the head dim + its partner's
    if (wave_on && qcol < p.Sq && h2 == 0) delta_out[stat] = dlt;
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) asm volatile("" :: "v"(qf[ds]), "v"(gf[ds]));
    asm volatile("" :: "v"(nlse));
    store_tile(lds);
    load_tile(t_lo + 1);

    int kofs[D / 16], tofs[D / 32][2];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) {
        kofs[ds] = img_off<D>(lane & 31, 2 * ds + h2);
        asm volatile("" : "+v"(kofs[ds]));
    }
    {
        const int i16 = lane & 15, gq = (lane >> 4) & 1, qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
        for (int i = 0; i < D / 32; ++i) {
            const int c = i * 32 + 16 * gq + 4 * pp;
            tofs[i][0] = img_off<D>(4 * h2 + qq, c >> 3) + ((c & 4) << 1);           // K^T operand: rows r0 + 4 h2 + q of the K image
            tofs[i][1] = img_off<D>(4 * h2 + qq + 8, c >> 3) + ((c & 4) << 1);
            asm volatile("" : "+v"(tofs[i][0]), "+v"(tofs[i][1]));
        }
    }
    f32x16 dq[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
#define ATTN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    ATTN_BARRIER();

    int sb = 0;
    for (int kt = t_lo; kt < t_hi; ++kt) {
        const int sn = SLOT - sb;
        const bool act = wave_on && kt * 64 <= wave_kmax;
        if (act) {
            int kad[D / 16];
#pragma unroll
            for (int ds = 0; ds < D / 16; ++ds) { kad[ds] = kofs[ds] + sb; asm volatile("" : "+v"(kad[ds])); }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (CAUSAL && kt * 64 + kb * 32 > wave_kmax) break;                  // upper half of a diagonal tile: masked for every row of the wave
                f32x16 st, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ds = 0; ds < D / 16; ++ds) {
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lds + kad[ds] + kb * 32 * D * 2), qf[ds], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lds + kad[ds] + TILE + kb * 32 * D * 2), gf[ds], dp, 0, 0, 0);
                }
                const bool need_mask = (kt * 64 + kb * 32 + 31 >= p.Sk) || (kt * 64 + kb * 32 < kv_lo) ||
                                       (CAUSAL && kt * 64 + kb * 32 + 31 > q0 + coff);
                if (need_mask) {
                    asm volatile("; boundary tile" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 64 + kb * 32 + acc_row(r, lane);
                        const bool ok = key < p.Sk && key >= kv_lo && (!CAUSAL || key <= q_abs);
                        st[r] = ok ? st[r] : -INFINITY;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[r], p.scale_log2, nlse));
                    st[r] = pv * ((dp[r] - dlt) * p.scale);                          // dS^T
                }
                int tad[D / 32][2];
#pragma unroll
                for (int i = 0; i < D / 32; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) { tad[i][j] = tofs[i][j] + sb + kb * 32 * D * 2; asm volatile("" : "+v"(tad[i][j])); }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 dsf = acc_frag(st, s);
#pragma unroll
                    for (int i = 0; i < D / 32; ++i)
                        dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(lds + tad[i][0] + 16 * s * D * 2, lds + tad[i][1] + 16 * s * D * 2), dsf, dq[i], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < t_hi) {
            store_tile(lds + sn);
            if (kt + 2 < t_hi) load_tile(kt + 2);
        }
        ATTN_BARRIER();
        sb = sn;
    }
#undef ATTN_BARRIER

    if (wave_on && qcol < p.Sq) {
        bf16_t* dqp = p.dQ + (long)b * p.dq_bs + (long)qcol * p.dq_rs + (long)h * D;
        if (p.rope_cs) {
            // Q is the rotary-embedded projection (adjacent-pair layout): gradient of the projection output = R(pos)^T dQ
            const float* cs = p.rope_cs + (long)qcol * D;           // (D / 2) pairs x 2 floats per position
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {                    // registers 4 jj .. 4 jj + 3: d = 32 i + 8 jj + 4 h2 + e
                    const float4 c4 = *(const float4*)(cs + (32 * i + 8 * jj + 4 * h2));
                    const float g0 = dq[i][4 * jj], g1 = dq[i][4 * jj + 1], g2 = dq[i][4 * jj + 2], g3 = dq[i][4 * jj + 3];
                    dq[i][4 * jj] = g0 * c4.x + g1 * c4.y;     dq[i][4 * jj + 1] = g1 * c4.x - g0 * c4.y;
                    dq[i][4 * jj + 2] = g2 * c4.z + g3 * c4.w; dq[i][4 * jj + 3] = g3 * c4.z - g2 * c4.w;
                }
        }
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned a[2], c[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    a[e] = (unsigned)f2bf(dq[i][8 * j + 2 * e]) | ((unsigned)f2bf(dq[i][8 * j + 2 * e + 1]) << 16);
                    c[e] = (unsigned)f2bf(dq[i][8 * j + 4 + 2 * e]) | ((unsigned)f2bf(dq[i][8 * j + 4 + 2 * e + 1]) << 16);
                }
                const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                *(uint4*)(dqp + i * 32 + 16 * j + 8 * h2) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// block = 128 keys of one (batch, kv head); wave w owns keys [k0 + 32w, k0 + 32w + 32).  QR query rows (32 or 64) are
// staged per iteration: QR = 64 halves the barriers / staging round trips per unit of MFMA work.
#ifndef ATTN_DKDV_QR
#define ATTN_DKDV_QR 32
#endif

// D = 64: two blocks per CU (<= 256 VGPR + AGPR; the dropout variant took 284 and ran ONE block per CU: 250 us per cross-attention
// launch); D = 128 keeps its 384 registers and one block per CU
// NW = waves per block (32 keys each).  NW = 2 (64-key blocks, D = 128 causal): twice as many, half as long work items — the
// 128-key form's 320 blocks of length 5..1 on 256 CUs run at 0.75 of the balanced makespan, 640 blocks of length 10..1 at 0.93 —
// for twice the Q / dO staging per wave.
template <int D, bool DROP, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 4 && D == 64) ? 2 : 1) void attn_bwd_dkdv_k(AttnArgs p) {   // (NW = 2: 384 registers = one wave per SIMD, so two 2-wave blocks share a CU anyway)
    constexpr int QR = (D == 128) ? ATTN_DKDV_QR : 32;
    constexpr int NT = 64 * NW, KB = 32 * NW;
    constexpr int LDS_LOOP = 2 * QR * D * 2 + 2 * QR * 4, LDS_EPI = NW * 32 * 64 * 4;     // epilogue: a 32 x 64 fp32 transpose tile per wave
    __shared__ __attribute__((aligned(16))) char lds[LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI];
    char* qimg = lds;
    char* gimg = lds + QR * D * 2;
    float* lse_s = (float*)(lds + 2 * QR * D * 2);
    float* dlt_s = lse_s + QR;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h2 = lane >> 5;    // wave index in an SGPR: everything derived from it (q0, mask tests) is scalar -> real branches
    // 1-D grid with the key block as the SLOWEST index: under the causal mask low key blocks sweep the
    // most query tiles, so the heaviest work items are dispatched first (longest-processing-time order)
    const int nhb = p.Hkv * p.B;
    const int kb0 = (blockIdx.x / nhb) * KB, k0 = kb0 + wave * 32;
    const int hk = (blockIdx.x % nhb) % p.Hkv, b = (blockIdx.x % nhb) / p.Hkv, group = p.Hq / p.Hkv;
    const int kcol = k0 + (lane & 31), kc = min(kcol, p.Sk - 1);
    const int kv_lo = p.kv_start ? max(0, min(p.kv_start[b], p.Sk)) : 0;
    const int coff = p.Sk - p.Sq;

    const bf16_t* kptr = p.K + (long)b * p.k_bs + (long)kc * p.k_rs + (long)hk * D;
    const bf16_t* vptr = p.V + (long)b * p.v_bs + (long)kc * p.v_rs + (long)hk * D;
    bf16x8 kf[D / 16], vf[D / 16];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) {
        kf[ds] = *(const bf16x8*)(kptr + 16 * ds + 8 * h2);
        vf[ds] = *(const bf16x8*)(vptr + 16 * ds + 8 * h2);
    }
    const bool key_ok = kcol < p.Sk && kcol >= kv_lo;

    f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

    // first query row that can see any key of this block
    const int q_first = p.causal ? max(0, kb0 - coff) : 0;
    const int qt_lo = q_first / QR, qt_hi = (p.Sq + QR - 1) / QR;
    // (query head of the GQA group, QR-row query slice) flattened into one iteration space; the Q / dO
    // tiles and their lse / delta rows are register-staged TWO iterations ahead.
    const int nq = qt_hi - qt_lo, n_it = group * nq;
    stage_t<D, QR, NT> qr0 = {}, gr0 = {}, qr1 = {}, gr1 = {};
    float ls0 = 0.f, dl0 = 0.f, ls1 = 0.f, dl1 = 0.f;
#define DESTA_Q_FETCH(IT, QR_, GR, LS, DL)                                                                      \
    {                                                                                                          \
        const int h_ = hk * group + (IT) / nq, qt_ = qt_lo + (IT) % nq;                                        \
        QR_ = tile_load<D, QR, NT>(p.Q + (long)b * p.q_bs + (long)h_ * D, p.q_rs, qt_ * QR, p.Sq - 1);        \
        GR = tile_load<D, QR, NT>(p.dO + (long)b * p.do_bs + (long)h_ * D, p.do_rs, qt_ * QR, p.Sq - 1);      \
        if (threadIdx.x < QR) {                                                                                \
            const long st_ = ((long)b * p.Hq + h_) * p.Sq + min(qt_ * QR + (int)threadIdx.x, p.Sq - 1);       \
            LS = p.lse[st_];                                                                                   \
            DL = p.delta[st_];                                                                                 \
        }                                                                                                      \
    }
#define DESTA_Q_STAGE(IT, QR_, GR, LS, DL)                                                                      \
    __syncthreads();                                                                                           \
    asm volatile("; stage " #QR_ ::: "memory");                                                                \
    tile_store<D, QR, NT>(qimg, QR_);                                                                          \
    tile_store<D, QR, NT>(gimg, GR);                                                                           \
    if (threadIdx.x < QR) { lse_s[threadIdx.x] = LS; dlt_s[threadIdx.x] = DL; }                                \
    __syncthreads();                                                                                           \
    if ((IT) + 2 < n_it) DESTA_Q_FETCH((IT) + 2, QR_, GR, LS, DL)
    if (n_it > 0) DESTA_Q_FETCH(0, qr0, gr0, ls0, dl0)
    if (n_it > 1) DESTA_Q_FETCH(1, qr1, gr1, ls1, dl1)
    auto compute = [&](const int it) __attribute__((always_inline)) {
        const int qt = qt_lo + it % nq;
        constexpr int NS = QR / 32;
        f32x16 stv[NS], dpv[NS];
        bool act[NS];
        // phase A for every 32-row slice first: with QR = 64 the S / dP MFMAs of slice 1 are in flight while the
        // VALU softmax of slice 0 runs, and slice 0's dV / dK MFMAs overlap the softmax of slice 1
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            const int qb = qt * QR + sub * 32;                                // first query row of this 32-row slice
            // wave-uniform skips: slice past the end, or (causal) all of this wave's keys in the slice's future
            act[sub] = qb < p.Sq && !(p.causal && k0 > qb + 31 + coff);
            if (!act[sub]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) { stv[sub][r] = 0.f; dpv[sub][r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < D / 16; ++ds) {
                stv[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(qimg, sub * 32, ds, lane), kf[ds], stv[sub], 0, 0, 0);
                dpv[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(gimg, sub * 32, ds, lane), vf[ds], dpv[sub], 0, 0, 0);
            }
        }
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            if (!act[sub]) continue;
            const int qb = qt * QR + sub * 32;
            f32x16& st = stv[sub];
            f32x16& dp = dpv[sub];
            // st[r]: S[q = qb + acc_row(r)][key = kcol]
            // masks only on boundary tiles (wave-uniform): ragged Sq/Sk, left padding, causal diagonal
            const bool need_mask = (qb + 31 >= p.Sq) || (k0 + 31 >= p.Sk) || (k0 < kv_lo) ||
                                   (p.causal && k0 + 31 > qb + coff);
            // this lane's 16 accumulator rows are 4 runs of 4 consecutive query rows (acc_row): fetch lse / delta as
            // 4 + 4 ds_read_b128 instead of 32 scalar LDS reads inside the dependent exp chain
            float lsv[16], dlv[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 a4 = *(const float4*)(lse_s + sub * 32 + 8 * q4 + 4 * h2);
                const float4 b4 = *(const float4*)(dlt_s + sub * 32 + 8 * q4 + 4 * h2);
                lsv[4 * q4 + 0] = a4.x; lsv[4 * q4 + 1] = a4.y; lsv[4 * q4 + 2] = a4.z; lsv[4 * q4 + 3] = a4.w;
                dlv[4 * q4 + 0] = b4.x; dlv[4 * q4 + 1] = b4.y; dlv[4 * q4 + 2] = b4.z; dlv[4 * q4 + 3] = b4.w;
            }
            unsigned keep = 0xffffu;                                          // bit r: probability (query row r, this key) survived
            if constexpr (DROP) {
                const unsigned long hb = ((unsigned long)b * p.Hq + (hk * group + it / nq)) * p.Sq;
                keep = 0;
                // lanes 2j, 2j+1 hold ADJACENT keys = one element pair = one hash (common.h; Sk even: host check): the even lane
                // hashes query rows 0..7 of the slice, the odd lane rows 8..15, and they swap through DPP (half the integer multiplies)
                const int half = lane & 1, kpair = k0 + (lane & 30);
                unsigned hh[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ql = (j & 3) + 8 * (2 * half + (j >> 2)) + 4 * h2;               // acc_row(8 * half + j, lane)
                    hh[j] = desta_rng32(p.seed_lo, p.seed_hi, ((hb + min(qb + ql, p.Sq - 1)) * p.Sk + kpair) >> 1);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned oth = (unsigned)__builtin_amdgcn_mov_dpp((int)hh[j], 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
                    const unsigned mine = half ? (hh[j] >> 16) : (hh[j] & 0xffffu), theirs = half ? (oth >> 16) : (oth & 0xffffu);
                    keep |= (mine >= p.drop_thresh ? 1u : 0u) << (8 * half + j);
                    keep |= (theirs >= p.drop_thresh ? 1u : 0u) << (8 * (1 - half) + j);
                }
            }
            if (need_mask) {                                           // boundary slices only (scalar branch, outside the element loop)
                asm volatile("; boundary slice" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qb + acc_row(r, lane);
                    const bool ok = key_ok && q < p.Sq && (!p.causal || kcol <= q + coff);
                    st[r] = ok ? st[r] : -INFINITY;                   // exp2(-inf) = 0
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[r], p.scale_log2, -lsv[r]));
                float ms = 1.0f;
                if constexpr (DROP) ms = ((keep >> r) & 1u) ? p.drop_scale : 0.f;
                st[r] = pv * ms;                                             // (dropped) P for dV
                dp[r] = pv * (ms * dp[r] - dlv[r]) * p.scale;                // dS
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = acc_frag(st, s), dsf = acc_frag(dp, s);
#pragma unroll
                for (int i = 0; i < D / 32; ++i) {
                    dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_tr<D>(gimg, sub * 32 + 16 * s, i * 32, lane), dv[i], 0, 0, 0);
                    dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, frag_tr<D>(qimg, sub * 32 + 16 * s, i * 32, lane), dk[i], 0, 0, 0);
                }
            }
        }
    };
    for (int it = 0; it < n_it; it += 2) {
        DESTA_Q_STAGE(it, qr0, gr0, ls0, dl0)
        compute(it);
        if (it + 1 < n_it) {
            DESTA_Q_STAGE(it + 1, qr1, gr1, ls1, dl1)
            compute(it + 1);
        }
    }
#undef DESTA_Q_STAGE
#undef DESTA_Q_FETCH
    // dk[i][r]: dK[key = k0 + acc_row(r)][d = i*32 + (lane&31)]
    bf16_t* dkp = p.dK + (long)b * p.dk_bs + (long)hk * D;
    bf16_t* dvp = p.dV + (long)b * p.dv_bs + (long)hk * D;
    const bool wide = ((p.dk_rs | p.dv_rs | p.dk_bs | p.dv_bs) & 7) == 0 && (((size_t)p.dK | (size_t)p.dV) & 15) == 0;
    if (wide) {
        // A lane owns ONE column of 16 keys: stored directly that is 64 two-byte store instructions per tensor and wave (the
        // cross-attention dK / dV are 245 MB per layer).  Transposed through a wave-private 32 x 64 fp32 LDS tile a lane owns 8
        // CONSECUTIVE columns of one key: 4 x 16-byte stores per 64 columns, 128-byte row segments.
        __syncthreads();                                                     // every wave is done with the Q / dO images
        float* tb = (float*)lds + wave * (32 * 64);
        auto emit = [&](const f32x16 (&acc)[D / 32], bf16_t* base, long rs, const bool rope) __attribute__((always_inline)) {
#pragma unroll
            for (int hf = 0; hf < D / 64; ++hf) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tb[acc_row(r, lane) * 64 + ii * 32 + (lane & 31)] = acc[hf * 2 + ii][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // same wave: LDS ops complete in order, keep the compiler in order too
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (lane >> 3) + 8 * j, key = k0 + row;
                    float4 x0 = *(const float4*)(tb + row * 64 + (lane & 7) * 8), x1 = *(const float4*)(tb + row * 64 + (lane & 7) * 8 + 4);
                    if (rope) {                                              // K is the rotary-embedded projection: R(position = key)^T dK
                        const float* cs = p.rope_cs + (long)min(key, p.Sk - 1) * D + hf * 64 + (lane & 7) * 8;
                        const float4 c0 = *(const float4*)cs, c1 = *(const float4*)(cs + 4);
                        const float4 y0 = make_float4(x0.x * c0.x + x0.y * c0.y, x0.y * c0.x - x0.x * c0.y, x0.z * c0.z + x0.w * c0.w, x0.w * c0.z - x0.z * c0.w);
                        const float4 y1 = make_float4(x1.x * c1.x + x1.y * c1.y, x1.y * c1.x - x1.x * c1.y, x1.z * c1.z + x1.w * c1.w, x1.w * c1.z - x1.z * c1.w);
                        x0 = y0; x1 = y1;
                    }
                    u16x8 o;
                    o[0] = f2bf(x0.x); o[1] = f2bf(x0.y); o[2] = f2bf(x0.z); o[3] = f2bf(x0.w);
                    o[4] = f2bf(x1.x); o[5] = f2bf(x1.y); o[6] = f2bf(x1.z); o[7] = f2bf(x1.w);
                    if (key < p.Sk) *(u16x8*)(base + (long)key * rs + hf * 64 + (lane & 7) * 8) = o;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // reads done before the next half overwrites the tile
            }
        };
        emit(dk, dkp, p.dk_rs, p.rope_cs != nullptr);
        emit(dv, dvp, p.dv_rs, false);
        return;
    }
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + acc_row(r, lane);
            if (key < p.Sk) {
                dkp[(long)key * p.dk_rs + i * 32 + (lane & 31)] = f2bf(dk[i][r]);
                dvp[(long)key * p.dv_rs + i * 32 + (lane & 31)] = f2bf(dv[i][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------ backward, ONE query tile (seq_q <= 64, D = 64): dQ, dK, dV in one pass
// The Q-Former's cross-attention (64 queries over 1500 encoder frames, 640 (batch, head) pairs per launch) is HBM-bound: K / V
// are 246 MB, dK / dV another 246 MB.  The two-kernel backward reads K / V twice (the dQ kernel re-computes S and dP) and
// re-fetches Q / dO per 128-key block: 930 MB of traffic, 260 us.  With a single query tile a block can own the WHOLE query
// range: it walks a chunk of key blocks, stages each K / V tile once (coalesced, through LDS), forms S and dP, emits dK / dV
// of that block and keeps dQ = sum_keys dS K in registers across the walk.  The keys of a (batch, head) are cut into `nch`
// chunks (work items = nch x batch x heads, equal length) whose dQ partials are summed in fixed order by attn_dq_sum_k:
// K / V read once, dK / dV written once, + 2 x nch x 16 KB per (batch, head) of partials.  No atomics: bit-identical run to run.
// block = 4 waves, wave w owns keys [128 kb + 32 w, +32) of every key block kb of its chunk.
// TR: dK / dV are written TRANSPOSED (the operand layout of the K / V projection's weight-gradient GEMM: no transpose pass over
// the 246 MB) and their sums over keys (the projection's bias gradients) leave the kernel as one partial row per (chunk, batch).
template <bool DROP, bool TR>
__global__ __launch_bounds__(256, 2) void attn_bwd_q64_k(AttnArgs p, float* __restrict__ dq_part, float* __restrict__ bias_part, int blocks_per_chunk) {
    constexpr int D = 64, WREG = 12288;
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * D * 2 + 4 * WREG + 2 * 64 * 4];
    char* qimg = lds;                                                       // [64 q][64 d]
    char* gimg = lds + 64 * D * 2;                                          // dO
    char* wreg = lds + 2 * 64 * D * 2;                                      // per wave: its 32 K rows | its 32 V rows | dS [32 keys][64 q]  (= 12 KB, re-used as the 32 x 64 fp32 epilogue tile)
    float* lse_s = (float*)(wreg + 4 * WREG);
    float* dlt_s = lse_s + 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h2 = lane >> 5;
    const int nbh = p.B * p.Hq;
    const int chunk = blockIdx.x / nbh, bh = blockIdx.x % nbh, b = bh / p.Hq, h = bh % p.Hq;
    const int nkb = (p.Sk + 127) / 128;
    const int kb_lo = chunk * blocks_per_chunk, kb_hi = min(kb_lo + blocks_per_chunk, nkb);
    const int kv_lo = p.kv_start ? max(0, min(p.kv_start[b], p.Sk)) : 0;
    char* kimg_w = wreg + wave * WREG;
    char* vimg_w = kimg_w + 4096;
    char* dsimg_w = kimg_w + 8192;
    const bf16_t* kbase = p.K + (long)b * p.k_bs + (long)h * D;
    const bf16_t* vbase = p.V + (long)b * p.v_bs + (long)h * D;

    // the query tile, once
    {
        const stage_t<D, 64> qr = tile_load<D, 64>(p.Q + (long)b * p.q_bs + (long)h * D, p.q_rs, 0, p.Sq - 1);
        const stage_t<D, 64> gr = tile_load<D, 64>(p.dO + (long)b * p.do_bs + (long)h * D, p.do_rs, 0, p.Sq - 1);
        tile_store<D, 64>(qimg, qr);
        tile_store<D, 64>(gimg, gr);
        if (threadIdx.x < 64) {
            const long st_ = ((long)b * p.Hq + h) * p.Sq + min((int)threadIdx.x, p.Sq - 1);
            lse_s[threadIdx.x] = p.lse[st_];
            dlt_s[threadIdx.x] = p.delta[st_];
        }
    }
    // dQ: wave w owns the 32 x 32 tile (query slice w >> 1, d half w & 1) over ALL 128 keys of a block (every wave's dS image and K rows
    // are in LDS anyway): 16 accumulator registers instead of 64 for per-wave partials, and no cross-wave sum at the end
    const int qa = wave >> 1, di = wave & 1;
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;

    stage_t<D, 128> kst = {}, vst = {};
    if (kb_lo < kb_hi) {
        kst = tile_load<D, 128>(kbase, p.k_rs, kb_lo * 128, p.Sk - 1);
        vst = tile_load<D, 128>(vbase, p.v_rs, kb_lo * 128, p.Sk - 1);
    }
    const unsigned long hb = ((unsigned long)b * p.Hq + h) * p.Sq;
    bf16_t* dkp = TR ? p.dK + (long)h * D * p.dkv_t_ld + (long)b * p.Sk : p.dK + (long)b * p.dk_bs + (long)h * D;
    bf16_t* dvp = TR ? p.dV + (long)h * D * p.dkv_t_ld + (long)b * p.Sk : p.dV + (long)b * p.dv_bs + (long)h * D;
    float bsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                            // TR: [K | V][d half]: sums over this lane's keys of every block

    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        __syncthreads();                                                    // every wave is done with its region (epilogue tile of the previous block) / the query images are written
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                       // 128 x 64 tile: row r of the tile lives in the region of wave r >> 5
            const int id = i * 256 + threadIdx.x, r = id >> 3, c = id & 7;
            char* dst = wreg + (r >> 5) * WREG + img_off<D>(r & 31, c);
            *(uint4*)dst = make_uint4(kst[4 * i + 0], kst[4 * i + 1], kst[4 * i + 2], kst[4 * i + 3]);
            *(uint4*)(dst + 4096) = make_uint4(vst[4 * i + 0], vst[4 * i + 1], vst[4 * i + 2], vst[4 * i + 3]);
        }
        __syncthreads();
        if (kb + 1 < kb_hi) {                                               // next tile: in flight during this block's arithmetic and epilogue
            kst = tile_load<D, 128>(kbase, p.k_rs, (kb + 1) * 128, p.Sk - 1);
            vst = tile_load<D, 128>(vbase, p.v_rs, (kb + 1) * 128, p.Sk - 1);
        }
        const int k0 = kb * 128 + wave * 32, kcol = k0 + (lane & 31);
        const bool key_ok = kcol < p.Sk && kcol >= kv_lo;
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub) {                                 // (not unrolled: both slices in flight at once spill)
            const int qb = sub * 32;
            if (qb >= p.Sq) {                                                // (wave-uniform) slice past the end: its dS is zero
#pragma unroll
                for (int j = 0; j < 4; ++j) *(uint2*)(dsimg_w + img_off<D>(lane & 31, sub * 4 + j) + 8 * h2) = make_uint2(0u, 0u);
                continue;
            }
            unsigned keep = 0xffffu;                                         // (hashes first: nothing else is live yet)
            if constexpr (DROP) {                                            // same element -> hash map as attn_bwd_dkdv_k (adjacent keys share one hash)
                keep = 0;
                const int half = lane & 1, kpair = k0 + (lane & 30);
                unsigned hh[8];
                const unsigned long e0 = (hb + qb) * p.Sk + kpair;            // element index of (query qb, key pair); rows >= Sq are masked below, their bits are unused
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ql = (j & 3) + 8 * (2 * half + (j >> 2)) + 4 * h2;
                    hh[j] = desta_rng32(p.seed_lo, p.seed_hi, (e0 + (unsigned)(ql * p.Sk)) >> 1);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned oth = (unsigned)__builtin_amdgcn_mov_dpp((int)hh[j], 0xB1, 0xf, 0xf, true);
                    const unsigned mine = half ? (hh[j] >> 16) : (hh[j] & 0xffffu), theirs = half ? (oth >> 16) : (oth & 0xffffu);
                    keep |= (mine >= p.drop_thresh ? 1u : 0u) << (8 * half + j);
                    keep |= (theirs >= p.drop_thresh ? 1u : 0u) << (8 * (1 - half) + j);
                }
            }
            if constexpr (DROP) asm volatile("" : "+v"(keep));               // keep the hash arithmetic in front of the accumulators' live range
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < D / 16; ++ds) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(qimg, qb, ds, lane), frag_rows<D>(kimg_w, 0, ds, lane), st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(gimg, qb, ds, lane), frag_rows<D>(vimg_w, 0, ds, lane), dp, 0, 0, 0);
            }
            // st[r]: S[q = qb + acc_row(r)][key = kcol]
            const bool need_mask = (qb + 31 >= p.Sq) || (k0 + 31 >= p.Sk) || (k0 < kv_lo);
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = key_ok && (qb + acc_row(r, lane)) < p.Sq;
                    st[r] = ok ? st[r] : -INFINITY;
                }
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {                                 // this lane's rows: 4 runs of 4 consecutive queries (acc_row)
                const float4 a4 = *(const float4*)(lse_s + qb + 8 * q4 + 4 * h2);
                const float4 b4 = *(const float4*)(dlt_s + qb + 8 * q4 + 4 * h2);
                const float ls4[4] = {a4.x, a4.y, a4.z, a4.w}, dl4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q4 + e;
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[r], p.scale_log2, -ls4[e]));
                    float ms = 1.0f;
                    if constexpr (DROP) ms = ((keep >> r) & 1u) ? p.drop_scale : 0.f;
                    st[r] = pv * ms;                                         // (dropped) P for dV
                    dp[r] = pv * (ms * dp[r] - dl4[e]) * p.scale;            // dS
                }
            }
            // dS of this slice -> the wave's [key][q] image (row = this lane's key): 4 runs of 4 consecutive query rows
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned lo = (unsigned)f2bf(dp[4 * j + 0]) | ((unsigned)f2bf(dp[4 * j + 1]) << 16);
                const unsigned hi = (unsigned)f2bf(dp[4 * j + 2]) | ((unsigned)f2bf(dp[4 * j + 3]) << 16);
                *(uint2*)(dsimg_w + img_off<D>(lane & 31, sub * 4 + j) + 8 * h2) = make_uint2(lo, hi);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = acc_frag(st, s2), dsf = acc_frag(dp, s2);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_tr<D>(gimg, qb + 16 * s2, i * 32, lane), dv[i], 0, 0, 0);
                    dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, frag_tr<D>(qimg, qb + 16 * s2, i * 32, lane), dk[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                                    // every wave's dS image is complete
        // dQ[q][d] += sum over the block's 128 keys of dS[q][key] K[key][d]: both operands are transposed reads with the key as the row
#pragma unroll
        for (int wv = 0; wv < 4; ++wv)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(wreg + wv * WREG + 8192, 16 * s2, qa * 32, lane),
                                                             frag_tr<D>(wreg + wv * WREG, 16 * s2, di * 32, lane), dq, 0, 0, 0);
        __syncthreads();                                                    // the regions are free: each becomes its wave's 32 x 64 fp32 transpose tile
        if constexpr (TR) {
            // transposed store: [d][key] through a wave-private [64 d][36] fp32 tile (a lane holds 4 runs of 4 consecutive keys of
            // ONE d: float4 writes; then a lane reads 4 consecutive keys of one d: 8-byte bf16 stores, 64-byte row segments)
            float* tb = (float*)kimg_w;
            auto emit_t = [&](const f32x16 (&acc)[2], bf16_t* base, float (&bs)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        *(float4*)(tb + (ii * 32 + (lane & 31)) * 36 + 8 * q4 + 4 * h2) =
                            make_float4(acc[ii][4 * q4], acc[ii][4 * q4 + 1], acc[ii][4 * q4 + 2], acc[ii][4 * q4 + 3]);
                        bs[ii] += (acc[ii][4 * q4] + acc[ii][4 * q4 + 1]) + (acc[ii][4 * q4 + 2] + acc[ii][4 * q4 + 3]);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int dr = 8 * j + (lane >> 3), key = k0 + 4 * (lane & 7);
                    const float4 x = *(const float4*)(tb + dr * 36 + 4 * (lane & 7));
                    u16x4 o;
                    o[0] = f2bf(x.x); o[1] = f2bf(x.y); o[2] = f2bf(x.z); o[3] = f2bf(x.w);
                    if (key < p.Sk) *(u16x4*)(base + (long)dr * p.dkv_t_ld + key) = o;        // Sk % 4 == 0 (host check): a group is whole or absent
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            emit_t(dk, dkp, bsum[0]);
            emit_t(dv, dvp, bsum[1]);
        } else {
            float* tb = (float*)kimg_w;
            auto emit = [&](const f32x16 (&acc)[2], bf16_t* base, long rs) __attribute__((always_inline)) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tb[acc_row(r, lane) * 64 + ii * 32 + (lane & 31)] = acc[ii][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (lane >> 3) + 8 * j, key = k0 + row;
                    const float4 x0 = *(const float4*)(tb + row * 64 + (lane & 7) * 8), x1 = *(const float4*)(tb + row * 64 + (lane & 7) * 8 + 4);
                    u16x8 o;
                    o[0] = f2bf(x0.x); o[1] = f2bf(x0.y); o[2] = f2bf(x0.z); o[3] = f2bf(x0.w);
                    o[4] = f2bf(x1.x); o[5] = f2bf(x1.y); o[6] = f2bf(x1.z); o[7] = f2bf(x1.w);
                    if (key < p.Sk) *(u16x8*)(base + (long)key * rs + (lane & 7) * 8) = o;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            emit(dk, dkp, p.dk_rs);
            emit(dv, dvp, p.dv_rs);
        }
    }
    // the four 32 x 32 tiles -> one [64 q][64 d] fp32 image -> one coalesced partial per work item
    float* acc_s = (float*)wreg;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_s[(qa * 32 + acc_row(r, lane)) * 64 + di * 32 + (lane & 31)] = dq[r];
    __syncthreads();
    float* out = dq_part + ((long)chunk * nbh + bh) * (64 * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (i * 256 + threadIdx.x) * 4;
        *(float4*)(out + e) = *(const float4*)(acc_s + e);
    }
    if constexpr (TR) {
        if (bias_part) {
            // bias gradients: sum the 8 (wave, key half) rows in fixed order -> this (chunk, batch)'s partial row [K | V][head * 64 + d]
            __syncthreads();
            float* bs_s = (float*)wreg;                                     // [8][128]
#pragma unroll
            for (int kv = 0; kv < 2; ++kv)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) bs_s[(wave * 2 + h2) * 128 + kv * 64 + ii * 32 + (lane & 31)] = bsum[kv][ii];
            __syncthreads();
            if (threadIdx.x < 128) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) t += bs_s[r * 128 + threadIdx.x];
                const int kv = threadIdx.x >> 6, dd = threadIdx.x & 63, W = 2 * p.Hq * 64;
                bias_part[((long)chunk * p.B + b) * W + kv * (p.Hq * 64) + h * 64 + dd] = t;
            }
        }
    }
}

// bias[c] = sum over the (chunk, batch) partial rows: 64 columns per block, the rows dealt to the block's 4 wave groups (4 independent
// accumulators each: 128 rows = 8 dependent rounds instead of 32), combined in fixed order
__global__ __launch_bounds__(256) void attn_bias_sum_k(const float* __restrict__ part, int nrows, int width, float* __restrict__ out) {
    __shared__ float comb[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < width) {
        int r = grp;
        for (; r + 12 < nrows; r += 16) {
            a0 += part[(long)r * width + c];
            a1 += part[(long)(r + 4) * width + c];
            a2 += part[(long)(r + 8) * width + c];
            a3 += part[(long)(r + 12) * width + c];
        }
        for (; r < nrows; r += 4) a0 += part[(long)r * width + c];
    }
    comb[grp][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && c < width) out[c] = (comb[0][lane] + comb[1][lane]) + (comb[2][lane] + comb[3][lane]);
}

// dQ[b][q][h][:] = bf16(sum over the key chunks, in chunk order, of the fp32 partials of attn_bwd_q64_k)
__global__ __launch_bounds__(256) void attn_dq_sum_k(AttnArgs p, const float* __restrict__ dq_part, int nch) {
    const int nbh = p.B * p.Hq, bh = blockIdx.x, b = bh / p.Hq, h = bh % p.Hq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (i * 256 + threadIdx.x) * 4, q = e >> 6, d0 = e & 63;
        float4 s = *(const float4*)(dq_part + (long)bh * 4096 + e);
        for (int c = 1; c < nch; ++c) {
            const float4 v = *(const float4*)(dq_part + ((long)c * nbh + bh) * 4096 + e);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (q < p.Sq) {
            u16x4 o;
            o[0] = f2bf(s.x); o[1] = f2bf(s.y); o[2] = f2bf(s.z); o[3] = f2bf(s.w);
            *(u16x4*)(p.dQ + (long)b * p.dq_bs + (long)q * p.dq_rs + (long)h * 64 + d0) = o;
        }
    }
}

int fill_args(const desta_attn_desc* d, AttnArgs& a) {
    DESTA_CHECK_ARG(d && d->Q && d->K && d->V, "attention: null operand");
    DESTA_CHECK_ARG(d->head_dim == 64 || d->head_dim == 128, "attention: head_dim %d unsupported (64 or 128)", d->head_dim);
    DESTA_CHECK_ARG(d->batch > 0 && d->n_q_heads > 0 && d->n_kv_heads > 0 && d->n_q_heads % d->n_kv_heads == 0,
                    "attention: bad head counts");
    DESTA_CHECK_ARG(d->seq_q > 0 && d->seq_k > 0, "attention: bad sequence lengths");
    DESTA_CHECK_ARG(!d->causal || d->seq_k >= d->seq_q, "attention: causal needs seq_k >= seq_q");
    DESTA_CHECK_ARG(d->q_row_stride % 8 == 0 && d->k_row_stride % 8 == 0 && d->v_row_stride % 8 == 0,
                    "attention: row strides must be multiples of 8 elements");
    DESTA_CHECK_ARG(d->batch <= 65535 && d->n_q_heads <= 65535, "attention: grid too large");
    a.Q = (const bf16_t*)d->Q; a.K = (const bf16_t*)d->K; a.V = (const bf16_t*)d->V; a.O = (bf16_t*)d->O;
    a.dO = (const bf16_t*)d->dO; a.dQ = (bf16_t*)d->dQ; a.dK = (bf16_t*)d->dK; a.dV = (bf16_t*)d->dV;
    a.lse = d->lse; a.delta = nullptr; a.O32 = d->O_f32; a.rope_cs = d->rope_cos_sin;
    a.q_bs = d->q_batch_stride; a.q_rs = d->q_row_stride; a.k_bs = d->k_batch_stride; a.k_rs = d->k_row_stride;
    a.v_bs = d->v_batch_stride; a.v_rs = d->v_row_stride; a.o_bs = d->o_batch_stride; a.o_rs = d->o_row_stride;
    a.do_bs = d->do_batch_stride; a.do_rs = d->do_row_stride; a.dq_bs = d->dq_batch_stride; a.dq_rs = d->dq_row_stride;
    a.dk_bs = d->dk_batch_stride; a.dk_rs = d->dk_row_stride; a.dv_bs = d->dv_batch_stride; a.dv_rs = d->dv_row_stride;
    a.B = d->batch; a.Hq = d->n_q_heads; a.Hkv = d->n_kv_heads; a.Sq = d->seq_q; a.Sk = d->seq_k;
    a.causal = d->causal; a.kv_start = d->kv_start;
    a.scale = d->scale; a.scale_log2 = d->scale * 1.44269504088896340736f;
    DESTA_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "attention: dropout_p must be in [0,1)");
    DESTA_CHECK_ARG(d->dropout_p == 0.f || d->head_dim == 64, "attention: dropout is built for head_dim 64 (the Q-Former) only");
    DESTA_CHECK_ARG(d->dropout_p == 0.f || d->seq_k % 2 == 0, "attention: dropout needs an even seq_k (adjacent keys share one hash)");
    a.drop_thresh = d->dropout_p > 0.f ? desta_drop_thresh(d->dropout_p) : 0u;
    a.drop_scale = 1.0f / (1.0f - d->dropout_p);
    a.seed_lo = (unsigned)d->dropout_seed; a.seed_hi = (unsigned)(d->dropout_seed >> 32);
    a.dkv_t_ld = d->dkv_transposed ? d->dkv_t_ld : 0;
    return DESTA_OK;
}

}  // namespace

namespace { int g_attn_opt[8] = {1, 0, 0, 0, 1, 0, 0, 0}; }      // [4]: one-pass backward for a single query tile (seq_q <= 64, D = 64)
extern "C" int desta_attention_set_option(int which, int value) {
    DESTA_CHECK_ARG(which >= 0 && which < 8, "attention_set_option: unknown option %d", which);
    g_attn_opt[which] = value;
    return DESTA_OK;
}

extern "C" int desta_attention_fwd(const desta_attn_desc* d, void* stream) {
    AttnArgs a;
    if (int rc = fill_args(d, a)) return rc;
    DESTA_CHECK_ARG(d->O, "attention_fwd: null output");
    DESTA_CHECK_ARG(d->o_row_stride % 8 == 0 && d->o_batch_stride % 8 == 0 && ((size_t)d->O & 15) == 0,
                    "attention_fwd: O must be 16-byte aligned with row / batch strides that are multiples of 8 elements");
    if (g_attn_opt[0] && !a.drop_thresh && a.Sq >= 128) {
        // 8-wave blocks: HPB heads of one GQA group x 256 / HPB query rows (the heads share the K / V tiles)
        const int G = a.Hq / a.Hkv;
        const int hpb = !a.causal ? 1 : (d->head_dim == 128 ? (G % 4 == 0 ? 4 : (G % 2 == 0 ? 2 : 1)) : (G % 2 == 0 ? 2 : 1));
        const int rb = 256 / hpb;
        dim3 g8((unsigned)((a.Sq + rb - 1) / rb) * (unsigned)(G / hpb) * (unsigned)(a.B * a.Hkv));
        hipStream_t st = (hipStream_t)stream;
        if (d->head_dim == 128) {
            if (!a.causal) hipLaunchKernelGGL((attn_fwd8_k<128, false, 1>), g8, dim3(512), 0, st, a);
            else if (hpb == 4 && g_attn_opt[2]) hipLaunchKernelGGL((attn_fwd8_k<128, true, 4, 2, true>), g8, dim3(512), 0, st, a);
            else if (hpb == 4) hipLaunchKernelGGL((attn_fwd8_k<128, true, 4>), g8, dim3(512), 0, st, a);
            else if (hpb == 2) hipLaunchKernelGGL((attn_fwd8_k<128, true, 2>), g8, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((attn_fwd8_k<128, true, 1>), g8, dim3(512), 0, st, a);
        } else {
            if (!a.causal && g_attn_opt[1]) hipLaunchKernelGGL((attn_fwd8_k<64, false, 1, 4, false>), g8, dim3(512), 0, st, a);
            else if (!a.causal && g_attn_opt[2]) hipLaunchKernelGGL((attn_fwd8_k<64, false, 1, 2, true>), g8, dim3(512), 0, st, a);
            else if (!a.causal) hipLaunchKernelGGL((attn_fwd8_k<64, false, 1>), g8, dim3(512), 0, st, a);
            else if (hpb == 2) hipLaunchKernelGGL((attn_fwd8_k<64, true, 2>), g8, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((attn_fwd8_k<64, true, 1>), g8, dim3(512), 0, st, a);
        }
        DESTA_CHECK_LAUNCH("attention_fwd");
        return DESTA_OK;
    }
    const bool two = d->head_dim == 64 && a.Sq <= 64;
    dim3 grid((unsigned)((a.Sq + (two ? 63 : 127)) / (two ? 64 : 128)) * a.Hq * a.B);       // 1-D: attn_work_item() orders it per XCD                  // 64-row blocks of two waves (Q-Former queries)
    if (d->head_dim == 128) hipLaunchKernelGGL((attn_fwd_k<128, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (two && a.drop_thresh) hipLaunchKernelGGL((attn_fwd_k<64, true, 2>), grid, dim3(128), 0, (hipStream_t)stream, a);
    else if (two) hipLaunchKernelGGL((attn_fwd_k<64, false, 2>), grid, dim3(128), 0, (hipStream_t)stream, a);
    else if (a.drop_thresh) hipLaunchKernelGGL((attn_fwd_k<64, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_k<64, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    DESTA_CHECK_LAUNCH("attention_fwd");
    return DESTA_OK;
}

// delta [batch][heads][seq_q], then (seq_q <= 64) up to ATTN_Q64_MAX_CHUNKS dQ partials of 64 x 64 floats per (batch, head)
#define ATTN_Q64_MAX_CHUNKS 4
extern "C" size_t desta_attention_bwd_workspace_floats(int batch, int n_q_heads, int seq_q) {
    const size_t delta = ((size_t)batch * n_q_heads * seq_q + 3) / 4 * 4;
    return delta + (seq_q <= 64 ? (size_t)ATTN_Q64_MAX_CHUNKS * batch * n_q_heads * (4096 + 128) : 0);   // + the K | V bias-gradient partial rows
}

// dQ and dK/dV are independent given delta.  The dK/dV grid is makespan-bound under the causal mask (320 blocks of very
// different length on 256 CUs, one block per CU: ~25 % of the CU-time idles behind the heaviest blocks), so the dQ kernel
// is launched on a side stream (fork after delta, join before returning to the caller's stream) and its blocks fill
// the CUs that dK/dV leaves idle.  Results are unchanged (no shared outputs, no atomics).
namespace {
struct BwdFork { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; int handles = 0; };
BwdFork g_bwd_fork[16];
std::mutex g_bwd_fork_mu;                    // creation / release only; the data path reads a fully built entry
int g_bwd_concurrent = 1;
BwdFork* bwd_fork_locked(int dev) {
    BwdFork& f = g_bwd_fork[dev];
    if (!f.side) {
        hipStream_t s = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            (void)hipStreamDestroy(s);
            return nullptr;
        }
        f.fork = e0; f.join = e1; f.side = s;      // published last: a half-built entry is never visible
    }
    return &f;
}
BwdFork* bwd_fork() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (g_bwd_fork[dev].side) return &g_bwd_fork[dev];
    std::lock_guard<std::mutex> lk(g_bwd_fork_mu);
    return bwd_fork_locked(dev);
}
}  // namespace
extern "C" int desta_attention_set_concurrent_bwd(int on) { g_bwd_concurrent = on; return DESTA_OK; }
// desta_create / desta_destroy (api.hip): the current device's internal fork stream and events are shared by every stateless
// desta_attention_bwd caller of the process, so handles only COUNT references: the stream is made with the first handle (or
// lazily by the first backward call) and destroyed when the LAST handle of the device goes, never under another live handle.
// A process that never creates a handle keeps the lazily made stream for its lifetime.  desta_destroy must still not race a
// data-path call of the same device that is inside desta_attention_bwd (documented in include/desta_hip.h).
int desta_internal_reserve(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return DESTA_EINVAL;
    std::lock_guard<std::mutex> lk(g_bwd_fork_mu);
    BwdFork* f = bwd_fork_locked(dev);
    if (!f) return DESTA_ELAUNCH;
    ++f->handles;
    return DESTA_OK;
}
int desta_internal_release(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return DESTA_EINVAL;
    std::lock_guard<std::mutex> lk(g_bwd_fork_mu);
    BwdFork& f = g_bwd_fork[dev];
    if (f.handles > 0 && --f.handles > 0) return DESTA_OK;      // another handle of this device is alive
    if (f.side) { (void)hipStreamSynchronize(f.side); (void)hipStreamDestroy(f.side); }
    if (f.fork) (void)hipEventDestroy(f.fork);
    if (f.join) (void)hipEventDestroy(f.join);
    f = BwdFork();
    return DESTA_OK;
}

extern "C" int desta_attention_bwd(const desta_attn_desc* d, float* workspace, void* stream) {
    AttnArgs a;
    if (int rc = fill_args(d, a)) return rc;
    DESTA_CHECK_ARG(d->O && d->dO && d->lse && workspace && d->dQ, "attention_bwd: null argument");
    DESTA_CHECK_ARG((d->dK == nullptr) == (d->dV == nullptr), "attention_bwd: dK and dV go together");
    DESTA_CHECK_ARG(d->dq_row_stride % 4 == 0 && d->do_row_stride % 8 == 0 && d->o_row_stride % 8 == 0,
                    "attention_bwd: bad strides");
    a.delta = workspace;
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)a.B * a.Hq * a.Sq;
    const int G = d->head_dim / 8;
    dim3 gd((unsigned)((rows * G + 255) / 256));
    dim3 gk_((unsigned)((a.Sk + 127) / 128) * a.Hkv * a.B);
    if (a.rope_cs)
        DESTA_CHECK_ARG(g_attn_opt[0] && !g_attn_opt[3] && !a.drop_thresh && !a.O32 && a.Sq >= 128 && a.Sq == a.Sk && d->dq_row_stride % 8 == 0 &&
                        d->dq_batch_stride % 8 == 0 && ((size_t)d->dQ & 15) == 0 &&
                        (!d->dK || ((((d->dk_row_stride | d->dv_row_stride | d->dk_batch_stride | d->dv_batch_stride) & 7) == 0) &&
                                    (((size_t)d->dK | (size_t)d->dV) & 15) == 0)),
                        "attention_bwd: rope_cos_sin needs seq_q == seq_k >= 128, no dropout and 16-byte aligned dQ / dK / dV");
    if (g_attn_opt[0] && !g_attn_opt[3] && !a.drop_thresh && !a.O32 && a.Sq >= 128 && d->dq_row_stride % 8 == 0 && d->dq_batch_stride % 8 == 0 &&
        ((size_t)d->dQ & 15) == 0) {
        // 8-wave dQ kernel (computes delta itself and leaves it in the workspace), then dK / dV on the same stream
        const int G = a.Hq / a.Hkv;
        const int hpb = !a.causal ? 1 : (d->head_dim == 128 ? (G % 4 == 0 ? 4 : (G % 2 == 0 ? 2 : 1)) : (G % 2 == 0 ? 2 : 1));
        const int rb = 256 / hpb;
        dim3 g8((unsigned)((a.Sq + rb - 1) / rb) * (unsigned)(G / hpb) * (unsigned)(a.B * a.Hkv));
        // (the 8-wave dQ kernel on a side stream beside dK / dV, with delta by its own launch again, measured equal in the step:
        // 168.41 / 169.16 vs 168.54 / 169.16 ms in alternating runs on one box; removed)
        if (d->head_dim == 128) {
            if (!a.causal) hipLaunchKernelGGL((attn_bwd_dq8_k<128, false, 1>), g8, dim3(512), 0, st, a, workspace);
            else if (hpb == 4) hipLaunchKernelGGL((attn_bwd_dq8_k<128, true, 4>), g8, dim3(512), 0, st, a, workspace);
            else if (hpb == 2) hipLaunchKernelGGL((attn_bwd_dq8_k<128, true, 2>), g8, dim3(512), 0, st, a, workspace);
            else hipLaunchKernelGGL((attn_bwd_dq8_k<128, true, 1>), g8, dim3(512), 0, st, a, workspace);
            if (d->dK) {
                if (g_attn_opt[5] && a.causal) {                            // 64-key blocks: better balanced under the causal mask
                    dim3 gk2((unsigned)((a.Sk + 63) / 64) * a.Hkv * a.B);
                    hipLaunchKernelGGL((attn_bwd_dkdv_k<128, false, 2>), gk2, dim3(128), 0, st, a);
                } else hipLaunchKernelGGL((attn_bwd_dkdv_k<128, false>), gk_, dim3(256), 0, st, a);
            }
        } else {
            if (!a.causal) hipLaunchKernelGGL((attn_bwd_dq8_k<64, false, 1>), g8, dim3(512), 0, st, a, workspace);
            else if (hpb == 2) hipLaunchKernelGGL((attn_bwd_dq8_k<64, true, 2>), g8, dim3(512), 0, st, a, workspace);
            else hipLaunchKernelGGL((attn_bwd_dq8_k<64, true, 1>), g8, dim3(512), 0, st, a, workspace);
            if (d->dK) hipLaunchKernelGGL((attn_bwd_dkdv_k<64, false>), gk_, dim3(256), 0, st, a);
        }
        DESTA_CHECK_LAUNCH("attention_bwd");
        return DESTA_OK;
    }
    const bool q64 = d->head_dim == 64 && a.Sq <= 64 && a.Sk >= 256 && !a.causal && a.Hq == a.Hkv && d->dK && !a.rope_cs &&
                     ((d->q_batch_stride | d->k_batch_stride | d->v_batch_stride | d->do_batch_stride) & 7) == 0 && ((size_t)d->dO & 15) == 0 &&
                     (d->dq_batch_stride & 3) == 0 && ((size_t)d->dQ & 7) == 0;
    if (d->dkv_transposed)
        DESTA_CHECK_ARG(q64 && a.Sk % 4 == 0 && d->dkv_t_ld % 4 == 0 && d->dkv_t_ld >= (int64_t)a.B * a.Sk && (((size_t)d->dK | (size_t)d->dV) & 7) == 0,
                        "attention_bwd: dkv_transposed needs the one-query-tile path (head_dim 64, seq_q <= 64, seq_k >= 256 and a multiple of 4, no GQA) and 8-byte aligned dK / dV");
    if ((g_attn_opt[4] || d->dkv_transposed) && q64 &&
        (d->dkv_transposed || (((d->dk_row_stride | d->dv_row_stride | d->dk_batch_stride | d->dv_batch_stride) & 7) == 0 && (((size_t)d->dK | (size_t)d->dV) & 15) == 0))) {
        // one query tile (the Q-Former's cross-attention): delta, then dQ / dK / dV in ONE pass over K / V, then the chunk sum
        hipLaunchKernelGGL(attn_delta_k<64>, gd, dim3(256), 0, st, a, workspace);
        const int nkb = (a.Sk + 127) / 128;
        const int nch = nkb >= 8 ? ATTN_Q64_MAX_CHUNKS : (nkb >= 4 ? 2 : 1), bpc = (nkb + nch - 1) / nch;
        float* part = workspace + (rows + 3) / 4 * 4;
        float* bpart = d->dkv_bias_grad ? part + (size_t)ATTN_Q64_MAX_CHUNKS * a.B * a.Hq * 4096 : nullptr;
        dim3 gf((unsigned)(nch * a.B * a.Hq));
        if (d->dkv_transposed) {
            if (a.drop_thresh) hipLaunchKernelGGL((attn_bwd_q64_k<true, true>), gf, dim3(256), 0, st, a, part, bpart, bpc);
            else hipLaunchKernelGGL((attn_bwd_q64_k<false, true>), gf, dim3(256), 0, st, a, part, bpart, bpc);
            if (bpart) hipLaunchKernelGGL(attn_bias_sum_k, dim3((unsigned)(2 * a.Hq)), dim3(256), 0, st, (const float*)bpart,
                                          nch * a.B, 2 * a.Hq * 64, d->dkv_bias_grad);
        } else if (a.drop_thresh) hipLaunchKernelGGL((attn_bwd_q64_k<true, false>), gf, dim3(256), 0, st, a, part, bpart, bpc);
        else hipLaunchKernelGGL((attn_bwd_q64_k<false, false>), gf, dim3(256), 0, st, a, part, bpart, bpc);
        hipLaunchKernelGGL(attn_dq_sum_k, dim3((unsigned)(a.B * a.Hq)), dim3(256), 0, st, a, (const float*)part, nch);
        DESTA_CHECK_LAUNCH("attention_bwd");
        return DESTA_OK;
    }
    const bool two_q = d->head_dim == 64 && a.Sq <= 64;
    dim3 gq((unsigned)((a.Sq + (two_q ? 63 : 127)) / (two_q ? 64 : 128)) * a.Hq * a.B);
    dim3 gk((unsigned)((a.Sk + 127) / 128) * a.Hkv * a.B);
    if (d->head_dim == 128) {
        hipLaunchKernelGGL(attn_delta_k<128>, gd, dim3(256), 0, st, a, workspace);
        BwdFork* f = (g_bwd_concurrent && d->dK) ? bwd_fork() : nullptr;
        if (f && hipEventRecord(f->fork, st) == hipSuccess && hipStreamWaitEvent(f->side, f->fork, 0) == hipSuccess) {
            hipLaunchKernelGGL((attn_bwd_dkdv_k<128, false>), gk, dim3(256), 0, st, a);          // the long one first
            hipLaunchKernelGGL((attn_bwd_dq_k<128, false>), gq, dim3(256), 0, f->side, a);
            DESTA_CHECK_ARG(hipEventRecord(f->join, f->side) == hipSuccess && hipStreamWaitEvent(st, f->join, 0) == hipSuccess,
                            "attention_bwd: could not join the side stream");
        } else {
            hipLaunchKernelGGL((attn_bwd_dq_k<128, false>), gq, dim3(256), 0, st, a);
            if (d->dK) hipLaunchKernelGGL((attn_bwd_dkdv_k<128, false>), gk, dim3(256), 0, st, a);
        }
    } else {
        hipLaunchKernelGGL(attn_delta_k<64>, gd, dim3(256), 0, st, a, workspace);
        const bool two = two_q;
        if (a.drop_thresh) {
            if (two) hipLaunchKernelGGL((attn_bwd_dq_k<64, true, 2>), gq, dim3(128), 0, st, a);
            else hipLaunchKernelGGL((attn_bwd_dq_k<64, true>), gq, dim3(256), 0, st, a);
            if (d->dK) hipLaunchKernelGGL((attn_bwd_dkdv_k<64, true>), gk, dim3(256), 0, st, a);
        } else {
            if (two) hipLaunchKernelGGL((attn_bwd_dq_k<64, false, 2>), gq, dim3(128), 0, st, a);
            else hipLaunchKernelGGL((attn_bwd_dq_k<64, false>), gq, dim3(256), 0, st, a);
            if (d->dK) hipLaunchKernelGGL((attn_bwd_dkdv_k<64, false>), gk, dim3(256), 0, st, a);
        }
    }
    DESTA_CHECK_LAUNCH("attention_bwd");
    return DESTA_OK;
}
