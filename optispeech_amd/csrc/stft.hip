// STFT magnitude forward / backward with the FFT in LDS (reference rows A15, A15b):
//   disc/loss.py:123-142   stft(): |torch.stft(center=True, reflect, hann zero-padded to n_fft)| with clamp 1e-7
//   _discriminators.py:196-216  DiscriminatorR.spectrogram: rectangular window, plain abs
//   disc/loss.py:94-118    torchaudio MelSpectrogram's inner spectrogram (power=1)
//
// One workgroup of N/4 threads per frame.  The frame is formed directly from the waveform (reflect padding and
// the window are applied on load -- the padded / framed signal is never materialised in HBM), transformed by a
// Stockham autosort FFT (radix-4 stages, one radix-2 stage when log2 N is odd) ping-ponging between two LDS
// buffers, and only the N/2+1 magnitudes are stored.  Algorithmic HBM bytes per frame: hop*4 read (each sample is
// used by n_fft/hop frames but fetched from L2 after the first touch) + (N/2+1)*4 written.
// Backward recomputes the spectrum (cheaper than saving 2x the magnitudes), forms G = dmag * X/|X| on the half
// spectrum, evaluates Re(sum_k G_k e^{+i 2 pi k n / N}) as Re(FFT(conj G)) with the same routine, applies the window
// and overlap-adds into dx with the reflect index map (f32 atomics).
#include "osp_common.h"
#include <stdlib.h>

__global__ void fft_twiddles_kernel(float2* tw, int N) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < N) {
        double s, c;
        sincos(-2.0 * 3.14159265358979323846 * (double)m / (double)N, &s, &c);
        tw[m] = make_float2((float)c, (float)s);
    }
}
// tw[m] = exp(-2 pi i m / N), m in [0, N)
extern "C" int osp_fft_twiddles(float* tw, int64_t N, hipStream_t stream) {
    OSP_CHECK_ARG(tw && N >= 8 && (N & (N - 1)) == 0, "N must be a power of two >= 8");
    hipLaunchKernelGGL(fft_twiddles_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, stream, (float2*)tw, (int)N);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// In-LDS Stockham FFT of N complex points held in buf0 (natural order); returns the buffer holding the
// result (natural order).  blockDim.x == N/4.  Forward transform (e^{-i...}).
// ``tws``: stride into the twiddle table (a transform of size N out of a table made for tws * N points).
__device__ __forceinline__ float2* lds_fft(float2* buf0, float2* buf1, const float2* __restrict__ tw, int N, int tws = 1) {
    const int j = threadIdx.x, Q = N >> 2;
    float2 *in = buf0, *out = buf1;
    int Ns = 1;
    for (; Ns * 4 <= N; Ns *= 4) {
        __syncthreads();
        const int k = j & (Ns - 1);
        const int tstep = N / (Ns * 4);                       // twiddle index of angle -2 pi k / (4 Ns)
        float2 v0 = in[j], v1 = in[j + Q], v2 = in[j + 2 * Q], v3 = in[j + 3 * Q];
        if (Ns > 1) {
            v1 = cmul(v1, tw[k * tstep * tws]);
            v2 = cmul(v2, tw[2 * k * tstep * tws]);
            v3 = cmul(v3, tw[3 * k * tstep * tws]);
        }
        const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), d = csub(v1, v3);
        const float2 a3 = make_float2(d.y, -d.x);              // -i * (v1 - v3)
        const int base = ((j - k) << 2) + k;                   // (j / Ns) * 4 Ns + k
        out[base] = cadd(a0, a2);
        out[base + Ns] = cadd(a1, a3);
        out[base + 2 * Ns] = csub(a0, a2);
        out[base + 3 * Ns] = csub(a1, a3);
        float2* t = in; in = out; out = t;
    }
    if (Ns < N) {                                              // one radix-2 stage: N/2 butterflies, 2 per thread
        __syncthreads();
        const int H = N >> 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int jj = j + r * Q, k = jj & (Ns - 1);
            const float2 a = in[jj], b = cmul(in[jj + H], tw[k * (N / (Ns * 2)) * tws]);
            const int base = ((jj - k) << 1) + k;
            out[base] = cadd(a, b);
            out[base + Ns] = csub(a, b);
        }
        float2* t = in; in = out; out = t;
    }
    __syncthreads();
    return in;
}

__device__ __forceinline__ int reflect_index(int p, int T) {   // index into x of padded position p - N/2 (reflect, no edge repeat)
    if (p < 0) p = -p;
    if (p >= T) p = 2 * (T - 1) - p;
    return p;
}

// N is a template parameter (static LDS, stage loop unrolled at compile time).  (Tried as a remedy for the occasional wrong frames
// seen when two PROCESSES share one GPU -- rocFFT behind torch.stft shows them too -- and it is not one: DESIGN.md section 7.)
template <int N>
__global__ __launch_bounds__(N / 4 < 64 ? 64 : N / 4) void stft_mag_fwd_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                    const float2* __restrict__ tw, float clamp_min, float* __restrict__ mag, int T,
                                    int hop, int frames) {
    __shared__ __attribute__((aligned(16))) float2 lds[2 * N];
    float2 *b0 = lds, *b1 = lds + N;
    const int b = blockIdx.y, f = blockIdx.x, j = threadIdx.x, Q = N >> 2;
    const float* xb = x + (int64_t)b * T;
    const int start = f * hop - (N >> 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = j + r * Q;
        float v = xb[reflect_index(start + n, T)];
        if (window) v *= window[n];
        b0[n] = make_float2(v, 0.f);
    }
    const float2* X = lds_fft(b0, b1, tw, N);
    const int bins = (N >> 1) + 1;
    float* out = mag + ((int64_t)b * frames + f) * bins;
    for (int k = j; k < bins; k += Q) {
        const float2 c = X[k];
        const float p = c.x * c.x + c.y * c.y;
        out[k] = sqrtf(clamp_min >= 0.f ? fmaxf(p, clamp_min) : p);
    }
}

template <int N>
__global__ __launch_bounds__(N / 4 < 64 ? 64 : N / 4) void stft_mag_bwd_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                    const float2* __restrict__ tw, float clamp_min, const float* __restrict__ dmag,
                                    float* __restrict__ dx, int T, int hop, int frames) {
    __shared__ __attribute__((aligned(16))) float2 lds[2 * N];
    float2 *b0 = lds, *b1 = lds + N;
    const int b = blockIdx.y, f = blockIdx.x, j = threadIdx.x, Q = N >> 2;
    const float* xb = x + (int64_t)b * T;
    const int start = f * hop - (N >> 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = j + r * Q;
        float v = xb[reflect_index(start + n, T)];
        if (window) v *= window[n];
        b0[n] = make_float2(v, 0.f);
    }
    float2* X = lds_fft(b0, b1, tw, N);
    float2* other = (X == b0) ? b1 : b0;
    const int bins = (N >> 1) + 1;
    const float* g = dmag + ((int64_t)b * frames + f) * bins;
    // conj(G) on the half spectrum, zero above Nyquist
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = j + r * Q;
        float2 v = make_float2(0.f, 0.f);
        if (k < bins) {
            const float2 c = X[k];
            const float p = c.x * c.x + c.y * c.y;
            const bool live = clamp_min >= 0.f ? (p >= clamp_min) : (p > 0.f);
            if (live) {
                const float s = g[k] * rsqrtf(p);
                v = make_float2(s * c.x, -s * c.y);
            }
        }
        other[k] = v;
    }
    const float2* Y = lds_fft(other, X, tw, N);
    float* dxb = dx + (int64_t)b * T;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = j + r * Q;
        float v = Y[n].x;
        if (window) v *= window[n];
        if (v != 0.f) atomicAdd(dxb + reflect_index(start + n, T), v);
    }
}

// ------------------------------------------------------------------------------------------------ real-input transform (round 3)
// A frame is REAL: the kernels above transform it as N complex points (imaginary parts zero), i.e. twice the butterflies, LDS and
// threads a real-input transform needs.  Here the frame is packed as M = N / 2 complex points z[m] = x[2m] + i x[2m+1], transformed
// by the same in-LDS Stockham routine at size M (N / 8 threads), and unpacked with one post-twiddle pass:
//     X[k] = (Z[k] + conj Z[M-k]) / 2  -  i e^{-2 pi i k / N} (Z[k] - conj Z[M-k]) / 2,      k = 0 .. M,  Z[M] = Z[0].
// Backward: with H = conj(dmag * X / |X|) on k = 0 .. M the gradient is dx[n] = Re sum_k H_k e^{-2 pi i k n / N}.  Extending H to the
// Hermitian sequence A (A_0 = 2 Re H_0, A_M = 2 Re H_M, A_k = H_k) gives dx[2m] + i dx[2m+1] = FFT_M(P)[m] / 2 with
//     P[k] = (A_k + conj A_{M-k}) + i e^{-2 pi i k / N} (A_k - conj A_{M-k}),                  k = 0 .. M-1
// -- again one size-M transform.  Same twiddle table as the complex kernels (stride 2 for the size-M stages).
template <int N>
__device__ __forceinline__ float2 rfft_unpack(const float2* __restrict__ Z, const float2* __restrict__ tw, int k) {
    constexpr int M = N / 2;
    const float2 a = Z[k == M ? 0 : k], b0 = Z[k == 0 ? 0 : M - k];
    const float2 b = make_float2(b0.x, -b0.y);                                  // conj Z[M - k]
    const float2 s = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y)), d = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
    const float2 w = k == M ? make_float2(-1.f, 0.f) : tw[k];                   // e^{-2 pi i k / N}
    const float2 wd = cmul(w, d);
    return make_float2(s.x + wd.y, s.y - wd.x);                                 // s - i w d
}

template <int N>
__global__ __launch_bounds__(N / 8 < 64 ? 64 : N / 8) void stft_mag_fwd_real_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                    const float2* __restrict__ tw, float clamp_min, float* __restrict__ mag, int T,
                                    int hop, int frames) {
    constexpr int M = N / 2, Q = M / 4;
    __shared__ __attribute__((aligned(16))) float2 lds[2 * M];
    float2 *b0 = lds, *b1 = lds + M;
    const int b = blockIdx.y, f = blockIdx.x, j = threadIdx.x;
    const float* xb = x + (int64_t)b * T;
    const int start = f * hop - (N >> 1);
    if (j < Q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = j + r * Q;
            float v0 = xb[reflect_index(start + 2 * m, T)], v1 = xb[reflect_index(start + 2 * m + 1, T)];
            if (window) { v0 *= window[2 * m]; v1 *= window[2 * m + 1]; }
            b0[m] = make_float2(v0, v1);
        }
    }
    const float2* Z = lds_fft(b0, b1, tw, M, 2);
    const int bins = M + 1;
    float* out = mag + ((int64_t)b * frames + f) * bins;
    for (int k = j; k < bins; k += blockDim.x) {
        const float2 c = rfft_unpack<N>(Z, tw, k);
        const float p = c.x * c.x + c.y * c.y;
        out[k] = sqrtf(clamp_min >= 0.f ? fmaxf(p, clamp_min) : p);
    }
}

template <int N>
__global__ __launch_bounds__(N / 8 < 64 ? 64 : N / 8) void stft_mag_bwd_real_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                    const float2* __restrict__ tw, float clamp_min, const float* __restrict__ dmag,
                                    float* __restrict__ dx, int T, int hop, int frames) {
    constexpr int M = N / 2, Q = M / 4;
    __shared__ __attribute__((aligned(16))) float2 lds[2 * M];
    __shared__ __attribute__((aligned(16))) float2 hs[M + 1];                     // A_k, k = 0 .. M
    float2 *b0 = lds, *b1 = lds + M;
    const int b = blockIdx.y, f = blockIdx.x, j = threadIdx.x;
    const float* xb = x + (int64_t)b * T;
    const int start = f * hop - (N >> 1);
    if (j < Q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = j + r * Q;
            float v0 = xb[reflect_index(start + 2 * m, T)], v1 = xb[reflect_index(start + 2 * m + 1, T)];
            if (window) { v0 *= window[2 * m]; v1 *= window[2 * m + 1]; }
            b0[m] = make_float2(v0, v1);
        }
    }
    float2* Z = lds_fft(b0, b1, tw, M, 2);
    float2* other = (Z == b0) ? b1 : b0;
    const float* g = dmag + ((int64_t)b * frames + f) * (M + 1);
    for (int k = j; k <= M; k += blockDim.x) {                                    // H_k = conj(dmag_k X_k / |X_k|), Hermitian end points doubled
        const float2 c = rfft_unpack<N>(Z, tw, k);
        const float p = c.x * c.x + c.y * c.y;
        const bool live = clamp_min >= 0.f ? (p >= clamp_min) : (p > 0.f);
        float2 h = make_float2(0.f, 0.f);
        if (live) { const float sc = g[k] * rsqrtf(p); h = make_float2(sc * c.x, -sc * c.y); }
        if (k == 0 || k == M) h = make_float2(2.f * h.x, 0.f);
        hs[k] = h;
    }
    __syncthreads();
    for (int k = j; k < M; k += blockDim.x) {                                     // P_k
        const float2 a = hs[k], c0 = hs[M - k];
        const float2 cj = make_float2(c0.x, -c0.y);
        const float2 s = cadd(a, cj), d = csub(a, cj);
        const float2 wd = cmul(tw[k], d);
        other[k] = make_float2(s.x - wd.y, s.y + wd.x);                           // s + i w d
    }
    const float2* U = lds_fft(other, Z, tw, M, 2);
    float* dxb = dx + (int64_t)b * T;
    for (int m = j; m < M; m += blockDim.x) {
        const float2 u = U[m];
        float v0 = 0.5f * u.x, v1 = 0.5f * u.y;
        if (window) { v0 *= window[2 * m]; v1 *= window[2 * m + 1]; }
        if (v0 != 0.f) atomicAdd(dxb + reflect_index(start + 2 * m, T), v0);
        if (v1 != 0.f) atomicAdd(dxb + reflect_index(start + 2 * m + 1, T), v1);
    }
}

static int stft_check(const void* x, const void* tw, int64_t B, int64_t T, int64_t N, int64_t hop) {
    if (!x || !tw || B <= 0 || hop <= 0) return 0;
    if (N < 16 || N > 4096 || (N & (N - 1))) return 0;
    if (T <= N / 2) return 0;                                  // reflect padding needs T > n_fft/2 (as torch.stft)
    return 1;
}

// mag: (B, frames, N/2+1), frames = 1 + T / hop (center=True).  window: N floats (already centred / zero padded)
// or null for the rectangular window; clamp_min < 0 selects plain |X|.
extern "C" int osp_stft_mag_fwd(const float* x, const float* window, const float* tw, float clamp_min, float* mag,
                                int64_t B, int64_t T, int64_t N, int64_t hop, hipStream_t stream) {
    OSP_CHECK_ARG(stft_check(x, tw, B, T, N, hop) && mag, "bad STFT arguments");
    const int frames = (int)(1 + T / hop);
    static int use_real = -1;
    if (use_real < 0) { const char* e = getenv("OSP_STFT_REAL"); use_real = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_real && N >= 64) {
#define OSP_STFT_FWDR(NN) case NN: hipLaunchKernelGGL((stft_mag_fwd_real_kernel<NN>), dim3((unsigned)frames, (unsigned)B), dim3(NN / 8), 0, stream, \
                       x, window, (const float2*)tw, clamp_min, mag, (int)T, (int)hop, frames); break
        switch ((int)N) {
            OSP_STFT_FWDR(64); OSP_STFT_FWDR(128); OSP_STFT_FWDR(256); OSP_STFT_FWDR(512); OSP_STFT_FWDR(1024); OSP_STFT_FWDR(2048); OSP_STFT_FWDR(4096);
            default: OSP_CHECK_ARG(false, "n_fft must be a power of two in [16, 4096]");
        }
#undef OSP_STFT_FWDR
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
#define OSP_STFT_FWD(NN) case NN: hipLaunchKernelGGL((stft_mag_fwd_kernel<NN>), dim3((unsigned)frames, (unsigned)B), dim3(NN / 4), 0, stream, \
                       x, window, (const float2*)tw, clamp_min, mag, (int)T, (int)hop, frames); break
    switch ((int)N) {
        OSP_STFT_FWD(16); OSP_STFT_FWD(32); OSP_STFT_FWD(64); OSP_STFT_FWD(128); OSP_STFT_FWD(256); OSP_STFT_FWD(512);
        OSP_STFT_FWD(1024); OSP_STFT_FWD(2048); OSP_STFT_FWD(4096);
        default: OSP_CHECK_ARG(false, "n_fft must be a power of two in [16, 4096]");
    }
#undef OSP_STFT_FWD
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// dx (B,T) must be zero-initialised by the caller (or hold a gradient to accumulate into).
extern "C" int osp_stft_mag_bwd(const float* x, const float* window, const float* tw, float clamp_min, const float* dmag,
                                float* dx, int64_t B, int64_t T, int64_t N, int64_t hop, hipStream_t stream) {
    OSP_CHECK_ARG(stft_check(x, tw, B, T, N, hop) && dmag && dx, "bad STFT arguments");
    const int frames = (int)(1 + T / hop);
    static int use_real = -1;
    if (use_real < 0) { const char* e = getenv("OSP_STFT_REAL"); use_real = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_real && N >= 64) {
#define OSP_STFT_BWDR(NN) case NN: hipLaunchKernelGGL((stft_mag_bwd_real_kernel<NN>), dim3((unsigned)frames, (unsigned)B), dim3(NN / 8), 0, stream, \
                       x, window, (const float2*)tw, clamp_min, dmag, dx, (int)T, (int)hop, frames); break
        switch ((int)N) {
            OSP_STFT_BWDR(64); OSP_STFT_BWDR(128); OSP_STFT_BWDR(256); OSP_STFT_BWDR(512); OSP_STFT_BWDR(1024); OSP_STFT_BWDR(2048); OSP_STFT_BWDR(4096);
            default: OSP_CHECK_ARG(false, "n_fft must be a power of two in [16, 4096]");
        }
#undef OSP_STFT_BWDR
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
#define OSP_STFT_BWD(NN) case NN: hipLaunchKernelGGL((stft_mag_bwd_kernel<NN>), dim3((unsigned)frames, (unsigned)B), dim3(NN / 4), 0, stream, \
                       x, window, (const float2*)tw, clamp_min, dmag, dx, (int)T, (int)hop, frames); break
    switch ((int)N) {
        OSP_STFT_BWD(16); OSP_STFT_BWD(32); OSP_STFT_BWD(64); OSP_STFT_BWD(128); OSP_STFT_BWD(256); OSP_STFT_BWD(512);
        OSP_STFT_BWD(1024); OSP_STFT_BWD(2048); OSP_STFT_BWD(4096);
        default: OSP_CHECK_ARG(false, "n_fft must be a power of two in [16, 4096]");
    }
#undef OSP_STFT_BWD
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ log-mel + energy
// Offline feature extraction of the reference (dataset/feature_extractors/__init__.py:114-147,176-200): per frame
//   m'[k] = sqrt(|X[k]|^2 + eps)      mel[j] = log(max(sum_k fb[j][k] m'[k], clip))      energy = sqrt(sum_k m'[k]^2)
// mag: (F, bins) magnitudes from osp_stft_mag_fwd (clamp_min < 0); fbT: (bins, n_mels) transposed mel basis (coalesced over
// the mel index); mel_out: (n_mels, F) -- the reference's (n_feats, T) layout; energy_out: (F).  One workgroup per frame.
// HBM-bound: (bins + n_mels + 1) * 4 bytes per frame, the basis (bins * n_mels * 4 B) stays in L2.
__global__ __launch_bounds__(128) void logmel_energy_kernel(const float* __restrict__ mag, const float* __restrict__ fbT,
                                                            float* __restrict__ mel_out, float* __restrict__ energy_out, int F,
                                                            int bins, int n_mels, float eps, float clip) {
    extern __shared__ float lm_m[];                      // bins magnitudes + 2 partial sums
    const int f = blockIdx.x, tid = threadIdx.x;
    float part = 0.f;
    for (int k = tid; k < bins; k += 128) {
        const float m = mag[(int64_t)f * bins + k], m2 = m * m + eps;
        lm_m[k] = sqrtf(m2);
        part += m2;
    }
    part = wave_sum(part);
    if ((tid & 63) == 0) lm_m[bins + (tid >> 6)] = part;
    __syncthreads();
    if (tid == 0 && energy_out) energy_out[f] = sqrtf(lm_m[bins] + lm_m[bins + 1]);
    for (int j = tid; j < n_mels; j += 128) {
        float acc = 0.f;
        for (int k = 0; k < bins; ++k) acc = fmaf(fbT[(int64_t)k * n_mels + j], lm_m[k], acc);
        mel_out[(int64_t)j * F + f] = logf(fmaxf(acc, clip));
    }
}

extern "C" int osp_logmel_energy(const float* mag, const float* fbT, float* mel_out, float* energy_out, int64_t F, int64_t bins,
                                 int64_t n_mels, float eps, float clip, hipStream_t stream) {
    OSP_CHECK_ARG(mag && fbT && mel_out && F > 0 && bins > 0 && n_mels > 0 && bins <= 8192, "bad log-mel arguments");
    hipLaunchKernelGGL(logmel_energy_kernel, dim3((unsigned)F), dim3(128), (size_t)(bins + 2) * 4, stream, mag, fbT, mel_out,
                       energy_out, (int)F, (int)bins, (int)n_mels, eps, clip);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
