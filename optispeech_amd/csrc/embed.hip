// Text embedding (reference row A3): TextEmbedding.forward modules/core.py:25-31 with
// ScaledSinusoidalEmbedding modules/layers.py:48-71.
//   out[b,t,:] = (sqrt(dim) * E[tok[b,t], :] + scale * pos[t, :]) * dropout
// pos is the (T, dim) sin/cos table (a constant buffer built once on the host exactly as layers.py:54-70 does).
// One wavefront per (b,t) row, float4 per lane.  Backward scatters into dE with f32 atomics (padding_idx row
// excluded, nn.Embedding(padding_idx) semantics) and reduces d scale per block.
#include "osp_common.h"

__global__ __launch_bounds__(256) void text_embed_fwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ E,
                                                             const float* __restrict__ pos, const float* __restrict__ scale,
                                                             float sqrt_dim, float drop_p, uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id,
                                                             float* __restrict__ out, int64_t rows, int T, int C) {
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int lane = threadIdx.x & 63;
    const float sc = scale[0];
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int t = (int)(row % T);
        const int64_t id = tok[row];
        for (int c = lane * 4; c < C; c += 256) {
            const float4 e = *reinterpret_cast<const float4*>(E + id * C + c);
            const float4 p = *reinterpret_cast<const float4*>(pos + (int64_t)t * C + c);
            float4 o = make_float4(fmaf(sqrt_dim, e.x, sc * p.x), fmaf(sqrt_dim, e.y, sc * p.y),
                                   fmaf(sqrt_dim, e.z, sc * p.z), fmaf(sqrt_dim, e.w, sc * p.w));
            if (drop_p > 0.f) {
                const uint4 r = philox4(seed, ((uint64_t)row * C + c) >> 2, stream_id);
                const float keep = 1.f / (1.f - drop_p);
                o.x *= u32_to_unit(r.x) < drop_p ? 0.f : keep; o.y *= u32_to_unit(r.y) < drop_p ? 0.f : keep;
                o.z *= u32_to_unit(r.z) < drop_p ? 0.f : keep; o.w *= u32_to_unit(r.w) < drop_p ? 0.f : keep;
            }
            *reinterpret_cast<float4*>(out + row * C + c) = o;
        }
    }
}
extern "C" int osp_text_embed_fwd(const int64_t* tok, const float* E, const float* pos, const float* scale, float sqrt_dim,
                                  float drop_p, int64_t seed, const int64_t* seed_dev, int64_t stream_id, float* out, int64_t B, int64_t T,
                                  int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(tok && E && pos && scale && out, "null operand");
    OSP_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    const int64_t rows = B * T;
    hipLaunchKernelGGL(text_embed_fwd_kernel, dim3((unsigned)(cdiv(rows, 4) < 2048 ? cdiv(rows, 4) : 2048)), dim3(256), 0, stream,
                       tok, E, pos, scale, sqrt_dim, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id, out, rows, (int)T, (int)C);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

__global__ __launch_bounds__(256) void text_embed_bwd_kernel(const float* __restrict__ dy, const int64_t* __restrict__ tok,
                                                             const float* __restrict__ pos, float sqrt_dim, float drop_p,
                                                             uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id, int64_t padding_idx,
                                                             float* __restrict__ dE, float* __restrict__ dscale, int64_t rows,
                                                             int T, int C) {
    __shared__ float scratch[16];
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int lane = threadIdx.x & 63;
    float ds = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int t = (int)(row % T);
        const int64_t id = tok[row];
        for (int c = lane * 4; c < C; c += 256) {
            float4 g = *reinterpret_cast<const float4*>(dy + row * C + c);
            if (drop_p > 0.f) {
                const uint4 r = philox4(seed, ((uint64_t)row * C + c) >> 2, stream_id);
                const float keep = 1.f / (1.f - drop_p);
                g.x *= u32_to_unit(r.x) < drop_p ? 0.f : keep; g.y *= u32_to_unit(r.y) < drop_p ? 0.f : keep;
                g.z *= u32_to_unit(r.z) < drop_p ? 0.f : keep; g.w *= u32_to_unit(r.w) < drop_p ? 0.f : keep;
            }
            const float4 p = *reinterpret_cast<const float4*>(pos + (int64_t)t * C + c);
            ds += g.x * p.x + g.y * p.y + g.z * p.z + g.w * p.w;
            if (dE && id != padding_idx) {
                float* d = dE + id * C + c;
                atomicAdd(d + 0, sqrt_dim * g.x); atomicAdd(d + 1, sqrt_dim * g.y);
                atomicAdd(d + 2, sqrt_dim * g.z); atomicAdd(d + 3, sqrt_dim * g.w);
            }
        }
    }
    ds = block_sum(ds, scratch);
    if (dscale && threadIdx.x == 0) atomicAdd(dscale, ds);
}
extern "C" int osp_text_embed_bwd(const float* dy, const int64_t* tok, const float* pos, float sqrt_dim, float drop_p,
                                  int64_t seed, const int64_t* seed_dev, int64_t stream_id, int64_t padding_idx, float* dE, float* dscale,
                                  int64_t B, int64_t T, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(dy && tok && pos, "null operand");
    OSP_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    const int64_t rows = B * T;
    hipLaunchKernelGGL(text_embed_bwd_kernel, dim3((unsigned)(cdiv(rows, 4) < 256 ? cdiv(rows, 4) : 256)), dim3(256), 0, stream,
                       dy, tok, pos, sqrt_dim, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id, padding_idx, dE, dscale, rows,
                       (int)T, (int)C);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
