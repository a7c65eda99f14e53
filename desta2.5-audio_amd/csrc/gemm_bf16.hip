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
    int tilesM, tilesN;
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

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
        const int r = idx >> 3, pc = idx & 7;
        const int c = pc ^ (r & 7);                     // logical chunk stored at physical slot pc
        const int ra = min(brow + r, p.M - 1), rb = min(bcol + r, p.N - 1);
        srcA[i] = A + (long)ra * p.lda + c * 8;
        srcB[i] = B + (long)rb * p.ldb + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* la = lds + buf * 2 * TILE_BYTES;
        char* lb = la + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int wbase = (i * 256 + wave * 64) * 16;           // wave-uniform LDS base
            glds16(srcA[i] + (long)kt * BK, la + wbase);
            glds16(srcB[i] + (long)kt * BK, lb + wbase);
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
                af[i] = *(const bf16x8*)(la + (offA[i] ^ (kk << 6)));
                bfr[i] = *(const bf16x8*)(lb + (offB[i] ^ (kk << 6)));
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
    const long zc = (long)z * p.sC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = brow + wr * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = bcol + wc * 64 + j * 16 + fq * 4;
            if (n0 >= p.N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha;
            if (p.bias) {
                const float4 b = *(const float4*)(p.bias + n0);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (p.preact) {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
                *(u16x4*)(p.preact + (long)z * p.sP + (long)m * p.ldp + n0) = o;
            }
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
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
                *(float4*)((float*)p.C + zc + (long)m * p.ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
                *(u16x4*)((bf16_t*)p.C + zc + (long)m * p.ldc + n0) = o;
            }
        }
    }
}

}  // namespace

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
    GemmArgs a;
    a.A = (const bf16_t*)d->A; a.B = (const bf16_t*)d->B; a.C = d->C;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.sA = d->stride_a; a.sB = d->stride_b; a.sC = d->stride_c;
    a.bias = d->bias;
    a.res = d->residual; a.ldr = d->ldr; a.sR = d->stride_r; a.res_f32 = d->residual_f32;
    a.act = d->act; a.out_f32 = d->out_f32;
    a.preact = (bf16_t*)d->preact; a.ldp = d->ldp; a.sP = d->stride_p;
    a.alpha = d->alpha;
    a.tilesM = (d->M + BM - 1) / BM; a.tilesN = (d->N + BN - 1) / BN;
    dim3 grid(a.tilesM * a.tilesN, d->batch);
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    DESTA_CHECK_LAUNCH("gemm_bf16_nt");
    return DESTA_OK;
}
