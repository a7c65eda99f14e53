// Whisper log-mel front end on the GPU (gfx950): [B, n] f32 waveform -> [B, n_mels, 3000] f32.
//
// Replaces `WhisperFeatureExtractor.__call__` as called by the collate function at
// desta/trainer/data/simple_dataset.py:239-243 (math: TF:models/whisper/feature_extraction_whisper.py
// :135-168; filters TF:audio_utils.py:638-729), which today runs on CPU DataLoader workers:
//   zero-pad / truncate to 480000 samples; centred STFT (n_fft 400, hop 160, periodic hann, reflect
//   pad) -> 3001 frames, last one dropped; |.|^2; slaney mel filter bank; log10(max(., 1e-10));
//   clamp to (per-clip global max - 8); (x + 4) / 4.
//
// K1: one block = 32 frames of one clip.  The 32 windowed frames are staged in LDS as x[n][frame]
//     (coalesced HBM read of the 5.3 K-sample span), folded in place into their even / odd parts, then every
//     thread owns the bin PAIR (k, 200-k) x 8 frames and runs the real-input DFT as broadcast LDS reads + FMAs
//     with a 400-entry twiddle table (index (k*n) mod 400 kept incrementally): a quarter of the multiplies of the
//     plain 400-point sum.  Power -> LDS -> dense mel product -> log10 -> raw store + per-block max.
// K2: per-clip max from the block maxima, clamp + affine in place.
#include "common.h"
#include "desta_hip.h"
#include <math.h>
#include <vector>

namespace {

constexpr int N_FFT = 400, HOP = 160, N_BINS = 201, N_FRAMES = 3000, N_SAMPLES = 480000;
constexpr int FT = 32;                       // frames per block
constexpr int NT = 448;                      // 7 waves: 2 x 201 (bin, frame-half) workers
constexpr int XS = 36;                       // LDS row stride (floats) of x[n][frame]; 16-B aligned rows
constexpr int NBLK = (N_FRAMES + FT - 1) / FT;

__global__ __launch_bounds__(NT) void logmel_k1(const float* __restrict__ wave, long wave_stride, int n_samples,
                                                const float* __restrict__ tables, int n_mels,
                                                float* __restrict__ out, float* __restrict__ blockmax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;                          // [400][XS]
    float* tw = xs + N_FFT * XS;                       // cos[400] | sin[400]
    float* red = tw + 2 * N_FFT;                       // [8]
    float* pw = xs;                                    // [FT][201], aliases xs once the DFT is done

    const int b = blockIdx.y, blk = blockIdx.x, t0 = blk * FT;
    const float* w = wave + (long)b * wave_stride;
    const float* win = tables;
    const float* fb = tables + 3 * N_FFT;              // [201][n_mels]
    const float* krange = fb + N_BINS * n_mels;        // [n_mels][2]: non-zero DFT-bin range of every mel filter

    for (int i = threadIdx.x; i < 2 * N_FFT; i += NT) tw[i] = tables[N_FFT + i];
    // stage windowed frames: idx -> (f, n), n fastest: coalesced global reads
    for (int idx = threadIdx.x; idx < FT * N_FFT; idx += NT) {
        const int f = idx / N_FFT, n = idx - f * N_FFT;
        int s = (t0 + f) * HOP - N_FFT / 2 + n;
        if (s < 0) s = -s;                                         // reflect (centre=True)
        if (s >= N_SAMPLES) s = 2 * (N_SAMPLES - 1) - s;
        const float v = (s < n_samples) ? w[s] : 0.f;             // zero padding up to 30 s
        xs[n * XS + f] = v * win[n];
    }
    __syncthreads();

    // Real-input symmetries cut the 400-point DFT to a quarter of its multiplies:
    //   * fold: with e[n] = x[n] + x[400-n], o[n] = x[n] - x[400-n] (n = 1..199; e[0] = x[0], e[200] = x[200]),
    //       Re X[k] = sum_{n=0..200} e[n] cos(2 pi k n / 400),   Im X[k] = -sum_{n=1..199} o[n] sin(2 pi k n / 400);
    //   * bins k and 200-k share every twiddle up to (-1)^n: accumulating even and odd n apart gives both,
    //       Re X[k] = Ee + Eo, Re X[200-k] = Ee - Eo;  |Im X[k]| = |Oe + Oo|, |Im X[200-k]| = |Oe - Oo|.
    // The fold is done in place in LDS (row n <- e, row 400-n <- o).
    for (int idx = threadIdx.x; idx < 199 * FT; idx += NT) {
        const int n = 1 + (idx >> 5), f = idx & (FT - 1);
        const float u = xs[n * XS + f], v = xs[(N_FFT - n) * XS + f];
        xs[n * XS + f] = u + v;
        xs[(N_FFT - n) * XS + f] = u - v;
    }
    __syncthreads();
    // worker (k in 0..100, quarter): bins k and 200-k for frames quarter*8 .. +7; 404 of the 448 threads
    constexpr int NK = 101, FQ = 8;
    const int k = threadIdx.x % NK, quarter = threadIdx.x / NK;
    const bool worker = quarter < FT / FQ;
    float ee[FQ], eo[FQ], oe[FQ], oo[FQ];
#pragma unroll
    for (int f = 0; f < FQ; ++f) { ee[f] = 0.f; eo[f] = 0.f; oe[f] = 0.f; oo[f] = 0.f; }
    if (worker) {
        const float* xq = xs + quarter * FQ;
        auto ld8 = [&](int row, float (&x)[FQ]) {
            const float4 a = *(const float4*)(xq + row * XS), c = *(const float4*)(xq + row * XS + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = c.x; x[5] = c.y; x[6] = c.z; x[7] = c.w;
        };
        auto step = [&](int n, int ti, float (&re)[FQ], float (&im)[FQ]) {
            const float c = tw[ti], sn = tw[N_FFT + ti];
            float e[FQ], o[FQ];
            ld8(n, e);
            ld8(N_FFT - n, o);
#pragma unroll
            for (int f = 0; f < FQ; ++f) { re[f] = fmaf(e[f], c, re[f]); im[f] = fmaf(o[f], sn, im[f]); }
        };
        {                                                                  // n = 0 (cos = 1) and n = 200 (cos = (-1)^k): even n, no sine part
            float x0[FQ], x200[FQ];
            ld8(0, x0);
            ld8(200, x200);
            const float sg = (k & 1) ? -1.f : 1.f;
#pragma unroll
            for (int f = 0; f < FQ; ++f) ee[f] = fmaf(x200[f], sg, x0[f]);
        }
        int ti = 0;
        for (int m = 0; m < 99; ++m) {                                     // n = 2m+1 (odd), 2m+2 (even)
            ti += k; if (ti >= N_FFT) ti -= N_FFT;
            step(2 * m + 1, ti, eo, oo);
            ti += k; if (ti >= N_FFT) ti -= N_FFT;
            step(2 * m + 2, ti, ee, oe);
        }
        ti += k; if (ti >= N_FFT) ti -= N_FFT;
        step(199, ti, eo, oo);
    }
    __syncthreads();                                   // everyone is done reading xs (pw aliases it)
    if (worker) {
#pragma unroll
        for (int f = 0; f < FQ; ++f) {
            const float r1 = ee[f] + eo[f], i1 = oe[f] + oo[f], r2 = ee[f] - eo[f], i2 = oe[f] - oo[f];
            pw[(quarter * FQ + f) * N_BINS + k] = r1 * r1 + i1 * i1;
            pw[(quarter * FQ + f) * N_BINS + (N_BINS - 1 - k)] = r2 * r2 + i2 * i2;      // k = 100 writes bin 100 twice with equal values
        }
    }
    __syncthreads();

    // mel product + log10; item -> (m, f) with f fastest (coalesced 128-B output rows)
    float mx = -INFINITY;
    for (int item = threadIdx.x; item < FT * n_mels; item += NT) {
        const int f = item & (FT - 1), m = item >> 5;
        const float* pf = pw + f * N_BINS;
        float acc = 0.f;
        const int klo = (int)krange[2 * m], khi = (int)krange[2 * m + 1];
        for (int kk = klo; kk < khi; ++kk) acc = fmaf(fb[kk * n_mels + m], pf[kk], acc);
        const float lv = log10f(fmaxf(acc, 1e-10f));
        const int t = t0 + f;
        if (t < N_FRAMES) {
            out[((long)b * n_mels + m) * N_FRAMES + t] = lv;
            mx = fmaxf(mx, lv);
        }
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m2 = red[0];
        for (int i = 1; i < NT / 64; ++i) m2 = fmaxf(m2, red[i]);
        blockmax[b * NBLK + blk] = m2;
    }
}

__global__ __launch_bounds__(256) void logmel_k2(float* __restrict__ out, const float* __restrict__ blockmax, long per_clip) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < NBLK; i += 256) mx = fmaxf(mx, blockmax[b * NBLK + i]);
    mx = block_max<256>(mx, red);
    const float floor_v = mx - 8.0f;
    float4* o = (float4*)(out + (long)b * per_clip);
    const long n4 = per_clip / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = o[i];
        v.x = (fmaxf(v.x, floor_v) + 4.0f) * 0.25f; v.y = (fmaxf(v.y, floor_v) + 4.0f) * 0.25f;
        v.z = (fmaxf(v.z, floor_v) + 4.0f) * 0.25f; v.w = (fmaxf(v.w, floor_v) + 4.0f) * 0.25f;
        o[i] = v;
    }
}

double hz_to_mel(double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; }
double mel_to_hz(double m) { return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; }

}  // namespace

extern "C" size_t desta_logmel_table_floats(int n_mels) { return (size_t)(3 * N_FFT + N_BINS * n_mels + 2 * n_mels); }
extern "C" size_t desta_logmel_workspace_floats(int batch) { return (size_t)batch * NBLK; }

// Host helper: window[400] | cos[400] | sin[400] | slaney filter bank [201][n_mels] (computed in double) | per mel bin the
// [first, last + 1) range of DFT bins with a non-zero weight (the triangles cover 2-30 of the 201 bins; the kernel sums only those
// — the same terms in the same order as the dense product, minus exact zeros).
extern "C" int desta_logmel_fill_tables(int n_mels, float* host_out) {
    DESTA_CHECK_ARG(host_out && n_mels > 0 && n_mels <= 256, "logmel tables: bad n_mels %d", n_mels);
    const double PI = 3.14159265358979323846;
    for (int n = 0; n < N_FFT; ++n) {
        host_out[n] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / N_FFT));          // periodic hann
        host_out[N_FFT + n] = (float)cos(2.0 * PI * n / N_FFT);
        host_out[2 * N_FFT + n] = (float)sin(2.0 * PI * n / N_FFT);
    }
    std::vector<double> ff(n_mels + 2);
    const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(8000.0);
    for (int i = 0; i < n_mels + 2; ++i) ff[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
    float* fb = host_out + 3 * N_FFT;
    for (int k = 0; k < N_BINS; ++k) {
        const double f = 8000.0 * k / (N_BINS - 1);
        for (int m = 0; m < n_mels; ++m) {
            const double down = (f - ff[m]) / (ff[m + 1] - ff[m]);
            const double up = (ff[m + 2] - f) / (ff[m + 2] - ff[m + 1]);
            double v = fmin(down, up);
            if (v < 0.0) v = 0.0;
            fb[k * n_mels + m] = (float)(v * 2.0 / (ff[m + 2] - ff[m]));
        }
    }
    float* rng = fb + N_BINS * n_mels;
    for (int m = 0; m < n_mels; ++m) {
        int lo = N_BINS, hi = 0;
        for (int k = 0; k < N_BINS; ++k)
            if (fb[k * n_mels + m] != 0.0f) { if (k < lo) lo = k; hi = k + 1; }
        if (lo > hi) lo = hi = 0;                                       // an empty filter: sum of nothing = 0, as in the dense product
        rng[2 * m] = (float)lo;
        rng[2 * m + 1] = (float)hi;
    }
    return DESTA_OK;
}

extern "C" int desta_logmel_f32(const float* wave, int batch, int n_samples, int64_t wave_stride,
                                const float* tables, int n_mels, float* out, float* workspace, void* stream) {
    DESTA_CHECK_ARG(wave && tables && out && workspace, "logmel: null argument");
    DESTA_CHECK_ARG(batch > 0 && n_samples > 0 && n_mels > 0 && n_mels <= 256, "logmel: bad shape");
    DESTA_CHECK_ARG(((uintptr_t)out % 16) == 0, "logmel: out must be 16-byte aligned");
    const int ns = n_samples < N_SAMPLES ? n_samples : N_SAMPLES;       // truncate to 30 s
    const size_t lds = (size_t)(N_FFT * XS + 2 * N_FFT + 8) * sizeof(float);   // 60.8 KB
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(logmel_k1, dim3(NBLK, batch), dim3(NT), lds, st, wave, (long)wave_stride, ns, tables, n_mels, out, workspace);
    const long per_clip = (long)n_mels * N_FRAMES;
    hipLaunchKernelGGL(logmel_k2, dim3(64, batch), dim3(256), 0, st, out, (const float*)workspace, per_clip);
    DESTA_CHECK_LAUNCH("logmel");
    return DESTA_OK;
}
