// weight_norm(Conv2d) packing for the discriminators (torch.nn.utils.weight_norm, dim = 0:
// vocoder/wavenext/disc/_discriminators.py:53-60,154-161):  w[n] = g[n] * v[n] / ||v[n]||.
// One workgroup per output channel computes the norm and writes the normalised filter straight into the kernel-native
// layouts the conv-GEMMs consume (so the ~10 torch ops per conv -- norm, div, mul, permute, contiguous, casts --
// become one launch):
//   reference layout  v (Cout, Cin, P, Q)
//   native            wn (Cout, Q, P, Cin)  bf16 (and optionally f32)         forward / wgrad layout
//   transposed        wt (Cin, Q, P, Cout)  bf16                               dgrad layout
// Backward: from dW in native f32 layout,  s = <dW[n], v[n]>:  dg[n] = s/||v[n]||,  dv = g/||v|| * (dW - v * s/||v||^2).
#include "osp_common.h"

__global__ __launch_bounds__(256) void wnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                        __bf16* __restrict__ wn, float* __restrict__ wn32,
                                                        __bf16* __restrict__ wt, float* __restrict__ inv_norm, int Cout,
                                                        int Cin, int P, int Q) {
    __shared__ float scratch[16];
    const int n = blockIdx.x, E = Cin * P * Q;
    const float* vn = v + (int64_t)n * E;
    float s = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) { const float x = vn[e]; s = fmaf(x, x, s); }
    s = block_sum(s, scratch);
    const float inv = rsqrtf(s), sc = g[n] * inv;
    if (threadIdx.x == 0 && inv_norm) inv_norm[n] = inv;
    for (int e = threadIdx.x; e < E; e += 256) {
        const int c = e / (P * Q), r = e - c * (P * Q), p = r / Q, q = r - p * Q;
        const float w = vn[e] * sc;
        const int64_t o = (((int64_t)n * Q + q) * P + p) * Cin + c;
        if (wn) wn[o] = (__bf16)w;
        if (wn32) wn32[o] = w;
        if (wt) wt[(((int64_t)c * Q + q) * P + p) * Cout + n] = (__bf16)w;
    }
}
extern "C" int osp_wnorm_fwd(const float* v, const float* g, void* wn_bf16, float* wn_f32, void* wt_bf16, float* inv_norm,
                             int64_t Cout, int64_t Cin, int64_t P, int64_t Q, hipStream_t stream) {
    OSP_CHECK_ARG(v && g && Cout > 0 && Cin > 0 && P > 0 && Q > 0, "bad args");
    hipLaunchKernelGGL(wnorm_fwd_kernel, dim3((unsigned)Cout), dim3(256), 0, stream, v, g, (__bf16*)wn_bf16, wn_f32,
                       (__bf16*)wt_bf16, inv_norm, (int)Cout, (int)Cin, (int)P, (int)Q);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

__global__ __launch_bounds__(256) void wnorm_bwd_kernel(const float* __restrict__ dwn, const float* __restrict__ v,
                                                        const float* __restrict__ g, const float* __restrict__ inv_norm,
                                                        float* __restrict__ dv, float* __restrict__ dg, int Cout, int Cin,
                                                        int P, int Q) {
    __shared__ float scratch[16];
    const int n = blockIdx.x, E = Cin * P * Q;
    const float* vn = v + (int64_t)n * E;
    float s = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) {
        const int c = e / (P * Q), r = e - c * (P * Q), p = r / Q, q = r - p * Q;
        s = fmaf(dwn[(((int64_t)n * Q + q) * P + p) * Cin + c], vn[e], s);
    }
    s = block_sum(s, scratch);
    const float inv = inv_norm[n], gn = g[n];
    if (threadIdx.x == 0) dg[n] += s * inv;
    const float a = gn * inv, b = gn * s * inv * inv * inv;
    for (int e = threadIdx.x; e < E; e += 256) {
        const int c = e / (P * Q), r = e - c * (P * Q), p = r / Q, q = r - p * Q;
        dv[(int64_t)n * E + e] += a * dwn[(((int64_t)n * Q + q) * P + p) * Cin + c] - b * vn[e];
    }
}
// dv, dg are ACCUMULATED (gradient arena semantics).
extern "C" int osp_wnorm_bwd(const float* dwn, const float* v, const float* g, const float* inv_norm, float* dv, float* dg,
                             int64_t Cout, int64_t Cin, int64_t P, int64_t Q, hipStream_t stream) {
    OSP_CHECK_ARG(dwn && v && g && inv_norm && dv && dg, "null operand");
    hipLaunchKernelGGL(wnorm_bwd_kernel, dim3((unsigned)Cout), dim3(256), 0, stream, dwn, v, g, inv_norm, dv, dg, (int)Cout,
                       (int)Cin, (int)P, (int)Q);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ fused mean |a - b|
// Feature-matching term (disc/loss.py:71-85): out += scale * sum |a - b| ; backward d a = -d b = sign(a-b) * gscale.
__global__ __launch_bounds__(256) void l1_sum_kernel(const void* __restrict__ a, const void* __restrict__ b, int is_bf16,
                                                     int64_t n, float scale, float* __restrict__ out) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float x, y;
        if (is_bf16) {
            x = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(a)[i]) << 16);
            y = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(b)[i]) << 16);
        } else { x = reinterpret_cast<const float*>(a)[i]; y = reinterpret_cast<const float*>(b)[i]; }
        s += fabsf(x - y);
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(out, s * scale);
}
extern "C" int osp_l1_sum(const void* a, const void* b, int64_t is_bf16, int64_t n, float scale, float* out,
                          hipStream_t stream) {
    OSP_CHECK_ARG(a && b && out && n > 0, "bad args");
    const int64_t blocks = cdiv(n, 256 * 8);
    hipLaunchKernelGGL(l1_sum_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, a, b, (int)is_bf16, n,
                       scale, out);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
// grad_b[i] = gscale[0] * scale * sign(b[i] - a[i])   (a = target, b = the tensor that carries the gradient)
__global__ __launch_bounds__(256) void l1_sign_kernel(const void* __restrict__ a, const void* __restrict__ b, int is_bf16,
                                                      int64_t n, float scale, const float* __restrict__ gscale,
                                                      void* __restrict__ gb) {
    const float gsc = gscale[0] * scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float x, y;
        if (is_bf16) {
            x = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(a)[i]) << 16);
            y = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(b)[i]) << 16);
        } else { x = reinterpret_cast<const float*>(a)[i]; y = reinterpret_cast<const float*>(b)[i]; }
        const float d = y - x, r = d > 0.f ? gsc : (d < 0.f ? -gsc : 0.f);
        if (is_bf16) reinterpret_cast<__bf16*>(gb)[i] = (__bf16)r;
        else reinterpret_cast<float*>(gb)[i] = r;
    }
}
extern "C" int osp_l1_sign(const void* a, const void* b, int64_t is_bf16, int64_t n, float scale, const float* gscale,
                           void* gb, hipStream_t stream) {
    OSP_CHECK_ARG(a && b && gscale && gb && n > 0, "bad args");
    const int64_t blocks = cdiv(n, 256 * 8);
    hipLaunchKernelGGL(l1_sign_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, a, b, (int)is_bf16, n,
                       scale, gscale, gb);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ fused hinge means
// GeneratorLoss / DiscriminatorLoss terms (disc/loss.py:16-65): out += scale * sum max(0, 1 + sgn * x)  (sgn = -1 for the
// "real" / generator terms mean(clamp(1 - x, 0)), +1 for mean(clamp(1 + x, 0))); backward dx = gscale * scale * sgn * [1 + sgn x > 0].
__global__ __launch_bounds__(256) void hinge_sum_kernel(const float* __restrict__ x, int64_t n, float sgn, float scale,
                                                        float* __restrict__ out) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += fmaxf(0.f, 1.f + sgn * x[i]);
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(out, s * scale);
}
extern "C" int osp_hinge_sum(const float* x, int64_t n, float sgn, float scale, float* out, hipStream_t stream) {
    OSP_CHECK_ARG(x && out && n > 0, "bad args");
    const int64_t blocks = cdiv(n, 256 * 8);
    hipLaunchKernelGGL(hinge_sum_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, stream, x, n, sgn, scale, out);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
__global__ __launch_bounds__(256) void hinge_grad_kernel(const float* __restrict__ x, int64_t n, float sgn, float scale,
                                                         const float* __restrict__ gscale, float* __restrict__ dx) {
    const float g = gscale[0] * scale * sgn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = (1.f + sgn * x[i] > 0.f) ? g : 0.f;
}
extern "C" int osp_hinge_grad(const float* x, int64_t n, float sgn, float scale, const float* gscale, float* dx, hipStream_t stream) {
    OSP_CHECK_ARG(x && gscale && dx && n > 0, "bad args");
    const int64_t blocks = cdiv(n, 256 * 8);
    hipLaunchKernelGGL(hinge_grad_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, stream, x, n, sgn, scale, gscale, dx);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
