// Masked softmax (+ dropout) of attention scores and its backward: the element-wise half of MultiHeadedAttention
// (generator/modules/_transformer/attention.py:80-98, :120-125).  The two batched GEMMs on either side (Q K^T, P V and
// their gradients) run on the conv-GEMM / wgrad kernels with a batch stride (optispeech_amd/ops.py: AttentionFn).
//   forward : P[z, i, j] = softmax_j(scale * S[z, i, j]) over the valid keys j < klen[z / H], 0 elsewhere   (in place)
//             Pd = dropout(P)  (Philox, regenerated in backward)                                             (optional)
//   backward: dS[z, i, j] = scale * P * (dP - sum_j P dP),  dP = dPd * keep / (1 - p)                        (in place in dPd)
// One wavefront per score row, 16 columns per lane (T2 <= 1024).  HBM-bound: 8 (+4) bytes per score forward, 12 backward.
#include "osp_common.h"

#define ATT_MAXC 16

__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(float* __restrict__ S, float* __restrict__ Pd,
                                                               const int64_t* __restrict__ klen, int H, int T1, int T2, int64_t rows,
                                                               float scale, float drop_p, uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id) {
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int z = (int)(row / T1), b = z / H;
    const int kl = (int)min((int64_t)T2, klen[b]);
    float* s = S + row * T2;
    float v[ATT_MAXC];
    float mx = -3.0e38f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        v[q] = (j < kl) ? s[j] * scale : -3.0e38f;
        mx = fmaxf(mx, v[q]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        v[q] = (j < kl) ? __expf(v[q] - mx) : 0.f;
        sum += v[q];
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;                   // no valid key: all-zero row (masked_fill(mask, 0))
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        if (j < T2) {
            const float pr = v[q] * inv;
            s[j] = pr;
            if (Pd) Pd[row * T2 + j] = drop_p > 0.f ? pr * dropout_factor(seed, stream_id, (uint64_t)(row * T2 + j), drop_p) : pr;
        }
    }
    (void)keep_scale;
}

__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dPd, int T2, int64_t rows,
                                                               float scale, float drop_p, uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id) {
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * T2;
    float* g = dPd + row * T2;
    float pv[ATT_MAXC], gv[ATT_MAXC];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        pv[q] = 0.f; gv[q] = 0.f;
        if (j < T2) {
            pv[q] = p[j];
            gv[q] = g[j];
            if (drop_p > 0.f) gv[q] *= dropout_factor(seed, stream_id, (uint64_t)(row * T2 + j), drop_p);
            dot += pv[q] * gv[q];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int q = 0; q < ATT_MAXC; ++q) {
        const int j = lane + 64 * q;
        if (j < T2) g[j] = scale * pv[q] * (gv[q] - dot);
    }
}

// S: (Z * T1, T2) f32, Z = B * H batch-major; klen: (B) int64 valid key counts.  Pd may be null (no dropout copy).
extern "C" int osp_attn_softmax_fwd(float* S, float* Pd, const int64_t* klen, int64_t B, int64_t H, int64_t T1, int64_t T2, float scale,
                                    float drop_p, int64_t seed, const int64_t* seed_dev, int64_t stream_id, hipStream_t stream) {
    OSP_CHECK_ARG(S && klen && B > 0 && H > 0 && T1 > 0 && T2 > 0 && T2 <= 64 * ATT_MAXC, "bad attention shape (T2 <= 1024)");
    OSP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || Pd), "dropout needs a second output buffer");
    const int64_t rows = B * H * T1;
    hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, S, Pd, klen, (int)H, (int)T1, (int)T2,
                       rows, scale, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_attn_softmax_bwd(const float* P, float* dPd, int64_t rows, int64_t T2, float scale, float drop_p, int64_t seed,
                                    const int64_t* seed_dev, int64_t stream_id, hipStream_t stream) {
    OSP_CHECK_ARG(P && dPd && rows > 0 && T2 > 0 && T2 <= 64 * ATT_MAXC, "bad attention shape (T2 <= 1024)");
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, P, dPd, (int)T2, rows, scale, drop_p,
                       (uint64_t)seed, seed_dev, (uint32_t)stream_id);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
