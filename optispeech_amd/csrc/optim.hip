// Optimiser step on the flat parameter arena (reference row A18: base_lightning_module.py:96-105,116-125 with
// torch.optim.AdamW(2e-4, (0.8, 0.99), wd 1e-2) and clip_grad_norm_(10)).
//   osp_sumsq        sum of squares of the gradient arena -> device scalar (f32 partials, f64 accumulate)
//   osp_adamw_clip   one pass over (p, g, m, v): global-norm clip factor read from the device scalar (no host sync),
//                    decoupled weight decay, bias-corrected Adam update.  HBM-bound: 4 reads + 3 writes per element.
#include "osp_common.h"
#include <stdlib.h>

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    __shared__ float scratch[16];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {               // four 16-byte loads in flight per thread
        const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
        s += (v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w) + (v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w) +
             (v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w) + (v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w);
    }
    for (; i < n4; i += stride) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += g[i] * g[i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(out, (double)s);
}
// out (f64 device scalar) += sum g^2 ; the caller zeroes it (hipMemsetAsync) once per norm.
extern "C" int osp_sumsq(const float* g, int64_t n, double* out, hipStream_t stream) {
    OSP_CHECK_ARG(g && out && n > 0, "bad args");
    OSP_CHECK_ARG((reinterpret_cast<uintptr_t>(g) & 15) == 0, "gradient arena must be 16-byte aligned");
    const int64_t blocks = cdiv(n, 256 * 16);
    static int64_t cap = 0;
    if (!cap) { cap = 256; }         // every workgroup ends with one f64 atomic into `out`: 128 / 256 / 512 / 1024 workgroups = 38 / 28 / 29 / 35 us
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap)), dim3(256), 0, stream, g, n, out);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

struct AdamArgs { float lr, beta1, beta2, eps, wd, bc1, bc2, max_norm, gscale; };

__global__ __launch_bounds__(256) void adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int64_t n, const double* __restrict__ sumsq,
                                                         const float* __restrict__ lr_dev, const int64_t* __restrict__ step_dev, AdamArgs a) {
    if (step_dev) {                                  // step counter kept in device memory (hipGraph replay)
        const double st = (double)*step_dev;
        a.bc1 = (float)(1.0 - pow((double)a.beta1, st));
        a.bc2 = (float)(1.0 - pow((double)a.beta2, st));
    }
    // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    float coef = a.gscale;
    if (sumsq && a.max_norm > 0.f) {
        const float tn = (float)sqrt(*sumsq) * a.gscale;
        coef *= fminf(a.max_norm / (tn + 1e-6f), 1.0f);
    }
    const float lr = lr_dev ? *lr_dev : a.lr;
    const float step = lr / a.bc1, isb2 = rsqrtf(a.bc2), decay = 1.f - lr * a.wd;
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= coef;
        pp *= decay;
        mm = a.beta1 * mm + (1.f - a.beta1) * gg;
        vv = a.beta2 * vv + (1.f - a.beta2) * gg * gg;
        pp -= step * mm / (sqrtf(vv) * isb2 + a.eps);
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        upd(p[i], g[i], m[i], v[i]);
}
// step = 1-based Adam step (read from *step_dev when given; lr from *lr_dev when given).  sumsq: device f64 sum of squares of the (unscaled) gradient or null (no clipping);
// gscale: factor applied to every gradient first (1/world_size for data parallel averaging, 1/accumulation).
extern "C" int osp_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq,
                              const float* lr_dev, const int64_t* step_dev, float lr, float beta1, float beta2, float eps, float wd,
                              int64_t step, float max_norm, float gscale, hipStream_t stream) {
    OSP_CHECK_ARG(p && g && m && v && n > 0 && (step >= 1 || step_dev), "bad args");
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd; a.max_norm = max_norm; a.gscale = gscale;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    const int64_t blocks = cdiv(n, 256 * 8);
    hipLaunchKernelGGL(adamw_clip_kernel, dim3((unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048)), dim3(256), 0, stream, p, g, m,
                       v, n, sumsq, lr_dev, step_dev, a);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
