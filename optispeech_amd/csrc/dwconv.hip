// Depthwise Conv1d of any odd width on channels-last frames + element-wise dropout: the separable-convolution backbones
// (reference rows 8(f)-4: LightSpeech EncSepConvLayer / ConvSeparable, modules/layers.py:455-506; LeanSpeech ConvGLU,
// modules/leanspeech.py:66-97; the Conformer convolution module, _conformer/convolution.py).  The ConvNeXt block keeps its own
// fused k = 7 kernels (convnext.hip).
//
//   y[b, t, c]   = (bias[c] + sum_{j<K} w[j, c] * x[b, t + j - K/2, c]) * rowmask[b*T + t]        (zero padding, K odd <= 63)
//   dx[b, t, c]  = sum_j w[j, c] * dy[b, t - j + K/2, c] * rowmask[...]                            (same kernel, flipped taps)
//   dw[j, c]    += sum_{b,t} dy[b, t, c] * rowmask * x[b, t + j - K/2, c];   db[c] += sum dy * rowmask
//
// Layout as convnext.hip: (B, T, C) f32, C % 4 == 0; a wave owns a run of frames of one utterance, lane l holds channels
// {256k + 4l ..}: every access is a coalesced 1 KiB row segment.  HBM-bound: one read and one write of the activations; the K
// re-reads of a row by the neighbouring output frames of the same wave come from L1 / L2 (taps in tap-major (K, C) order, also
// cached).  Algorithmic bytes per frame: 2 * C * 4 forward / input gradient, 2 * C * 4 (reads) for the weight gradient.
#include "osp_common.h"

#define DW_FR 16          // frames per wave
#define DW_MAXK 63

__device__ __forceinline__ float4 dwf4fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// flip = 0: forward (tap j reads frame t + j - pad); flip = 1: input gradient (tap j reads frame t - j + pad); the row mask
// multiplies the OUTPUT in the forward pass and the INPUT rows (dy) in the gradient pass.
// KT > 0: the width is a compile-time constant -- taps in registers, the row loop fully unrolled (every (row, output) pair is a
// static FMA): the widths of the reference configs (configs/model/generator/{encoder,decoder}/*.yaml).  KT = 0: any odd width,
// taps re-read from L1.
template <int KT>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     const float* __restrict__ rowmask, float* __restrict__ y, int B, int T, int C, int Krt,
                                                     int flip) {
    const int K = KT > 0 ? KT : Krt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int runs_per_utt = (T + DW_FR - 1) / DW_FR;
    const int run = blockIdx.x * 4 + wave;
    if (run >= B * runs_per_utt) return;
    const int b = run / runs_per_utt, t0 = (run - b * runs_per_utt) * DW_FR, pad = K / 2;
    const float* xb = x + (int64_t)b * T * C;
    const float* mb = rowmask ? rowmask + (int64_t)b * T : nullptr;
    for (int ch = lane * 4; ch < C; ch += 256) {
        const float4 bv = (bias && !flip) ? *reinterpret_cast<const float4*>(bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acc[DW_FR];
#pragma unroll
        for (int f = 0; f < DW_FR; ++f) acc[f] = bv;
        if constexpr (KT > 0) {
            float4 wr[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) wr[j] = *reinterpret_cast<const float4*>(w + (int64_t)(flip ? KT - 1 - j : j) * C + ch);
#pragma unroll
            for (int r = 0; r < DW_FR + KT - 1; ++r) {
                const int t = t0 + r - pad;
                if (t >= 0 && t < T) {                            // wave-uniform
                    float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)t * C + ch);
                    if (flip && mb) { const float m = mb[t]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
#pragma unroll
                    for (int f = 0; f < DW_FR; ++f)
                        if (r - f >= 0 && r - f < KT) acc[f] = dwf4fma(wr[r - f], v, acc[f]);      // compile-time condition
                }
            }
        } else {
            // input rows t0 - pad .. t0 + DW_FR - 1 + pad, each loaded once per channel chunk and scattered to the outputs it feeds
            for (int r = 0; r < DW_FR + K - 1; ++r) {
                const int t = t0 + r - pad;
                if (t < 0 || t >= T) continue;                    // wave-uniform
                float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)t * C + ch);
                if (flip && mb) { const float m = mb[t]; v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
                // row r feeds output f through tap j = r - f (forward) or j = K - 1 - (r - f) (gradient), 0 <= r - f < K
#pragma unroll
                for (int f = 0; f < DW_FR; ++f) {
                    const int d = r - f;
                    if (d >= 0 && d < K) {                        // wave-uniform
                        const float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)(flip ? K - 1 - d : d) * C + ch);
                        acc[f] = dwf4fma(wv, v, acc[f]);
                    }
                }
            }
        }
#pragma unroll
        for (int f = 0; f < DW_FR; ++f) {
            const int t = t0 + f;
            if (t < T) {
                float4 o = acc[f];
                if (!flip && mb) { const float m = mb[t]; o.x *= m; o.y *= m; o.z *= m; o.w *= m; }
                *reinterpret_cast<float4*>(y + ((int64_t)b * T + t) * C + ch) = o;
            }
        }
    }
}

// weight / bias gradient: a wave owns a run of WG_FR frames and one chunk of 4 channels per lane; tap partials in registers
// are not possible for a run-time K, so the taps are the outer loop: dy rows of the run stay in registers, x rows stream from
// L1 / L2; per-wave partials -> LDS -> one atomic per (tap, channel) and workgroup.
#define WG_FR 16
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ rowmask, float* __restrict__ dw, float* __restrict__ db,
                                                           int B, int T, int C, int K, int runs_per_wave) {
    __shared__ float red[4][256 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int runs_per_utt = (T + WG_FR - 1) / WG_FR, nruns = B * runs_per_utt, pad = K / 2;
    const int first = (blockIdx.x * 4 + wave) * runs_per_wave;
    for (int c0 = 0; c0 < C; c0 += 1024) {                          // 1024 channels per pass (4 chunks of 256 per lane)
        for (int j = -1; j < K; ++j) {                              // j = -1: the bias gradient
            float4 part[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < runs_per_wave; ++q) {
                const int run = first + q;
                if (run >= nruns) break;
                const int b = run / runs_per_utt, t0 = (run - b * runs_per_utt) * WG_FR;
                for (int f = 0; f < WG_FR; ++f) {
                    const int t = t0 + f, tx = t + j - pad;
                    if (t >= T) break;
                    if (j >= 0 && (tx < 0 || tx >= T)) continue;
                    const float m = rowmask ? rowmask[(int64_t)b * T + t] : 1.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ch = c0 + k * 256 + lane * 4;
                        if (ch < C) {
                            float4 g = *reinterpret_cast<const float4*>(dy + ((int64_t)b * T + t) * C + ch);
                            g.x *= m; g.y *= m; g.z *= m; g.w *= m;
                            if (j < 0) { part[k].x += g.x; part[k].y += g.y; part[k].z += g.z; part[k].w += g.w; }
                            else part[k] = dwf4fma(g, *reinterpret_cast<const float4*>(x + ((int64_t)b * T + tx) * C + ch), part[k]);
                        }
                    }
                }
            }
            float* dst = j < 0 ? db : dw + (int64_t)j * C;
            if (!dst) continue;                                     // kernel-uniform (no bias)
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(&red[wave][k * 256 + lane * 4]) = part[k];
            __syncthreads();
            for (int i = threadIdx.x; i < 1024; i += 256)
                if (c0 + i < C) atomicAdd(dst + c0 + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
        }
    }
}

extern "C" int osp_dwconv_fwd(const float* x, const float* w, const float* bias, const float* rowmask, float* y, int64_t B, int64_t T,
                              int64_t C, int64_t K, int64_t flip, hipStream_t stream) {
    OSP_CHECK_ARG(x && w && y, "null operand");
    OSP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0, "C must be a multiple of 4");
    OSP_CHECK_ARG(K > 0 && K % 2 == 1 && K <= DW_MAXK, "kernel width must be odd, <= 63");
#define L(KT_) hipLaunchKernelGGL((dwconv_kernel<KT_>), dim3((unsigned)cdiv(B * cdiv(T, DW_FR), 4)), dim3(256), 0, stream, x, w, bias, rowmask, y, (int)B, (int)T, (int)C, (int)K, (int)(flip != 0))
    switch (K) {
        case 5: L(5); break; case 7: L(7); break; case 9: L(9); break; case 13: L(13); break; case 17: L(17); break;
        case 21: L(21); break; case 25: L(25); break; case 31: L(31); break; default: L(0);
    }
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_dwconv_wgrad(const float* dy, const float* x, const float* rowmask, float* dw, float* db, int64_t B, int64_t T, int64_t C,
                                int64_t K, hipStream_t stream) {
    OSP_CHECK_ARG(dy && x && dw, "null operand");
    OSP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0, "C must be a multiple of 4");
    OSP_CHECK_ARG(K > 0 && K % 2 == 1 && K <= DW_MAXK, "kernel width must be odd, <= 63");
    // ~512 workgroups: enough to fill the chip, few enough that the (K + 1) * C atomics per workgroup stay cheap
    const int64_t nruns = B * cdiv(T, WG_FR);
    int64_t rpw = cdiv(nruns, 512 * 4);
    if (rpw < 1) rpw = 1;
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3((unsigned)cdiv(nruns, 4 * rpw)), dim3(256), 0, stream, dy, x, rowmask, dw, db, (int)B, (int)T,
                       (int)C, (int)K, (int)rpw);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ dropout (+ residual add)
// y[i] = res[i] + x[i] * keep(i)     keep(i) = 0 with probability p, else 1 / (1 - p): Philox key (seed, stream), counter i.
// The backward pass is the same kernel on the gradient (no res).  F.dropout call sites of layers.py:497-503 and the
// positional-dropout / residual sites of the separable-conv and Conformer layers.
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                                                          int64_t n4, float p, uint64_t seed, const int64_t* __restrict__ seed_dev,
                                                          uint32_t stream_id) {
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const float keep = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        if (p > 0.f) {
            const uint4 r = philox4(seed, (uint64_t)i, stream_id);
            v.x *= u32_to_unit(r.x) < p ? 0.f : keep;
            v.y *= u32_to_unit(r.y) < p ? 0.f : keep;
            v.z *= u32_to_unit(r.z) < p ? 0.f : keep;
            v.w *= u32_to_unit(r.w) < p ? 0.f : keep;
        }
        if (res) { const float4 q = reinterpret_cast<const float4*>(res)[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

extern "C" int osp_dropout_add(const float* x, const float* res, float* y, int64_t n, float p, int64_t seed, const int64_t* seed_dev,
                               int64_t stream_id, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "n must be a positive multiple of 4");
    OSP_CHECK_ARG(p >= 0.f && p < 1.f, "drop probability outside [0, 1)");
    const int64_t blocks = cdiv(n / 4, 256 * 4);
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, x, res, y, n / 4, p,
                       (uint64_t)seed, seed_dev, (uint32_t)stream_id);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
