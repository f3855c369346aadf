// Element-wise plumbing of the TAPED regions of the training step (optispeech_amd/tape.py).
//
// A region of the step that is replayed from a recorded call list may contain nothing but C-ABI calls.  What torch used to launch
// between the kernels of the path -- gradient accumulation adds, scalings by a device scalar, mask conversions, the layout flip
// of the mel batch, the loss assembly -- therefore has entry points of its own here.  All of them are HBM-trivial (a few KB to a
// few MB per launch); 16-byte accesses where the operands allow it, grid-stride otherwise.
//
//   osp_store_i64        one int64 into device memory (the per-step dropout seed: kernels read it through `seed_dev`)
//   osp_ew_axpby         out = a * x + b * y            (y null: a * x + b)      autograd's gradient accumulation, negation, scaling
//   osp_ew_mul           out = x * y, y of the same shape / one value per row / one value per column
//   osp_ew_scale_dev     out = c * s[0] * x             (s: device scalar -- an incoming loss gradient)
//   osp_ew_relu_mask     out = y > 0 ? g : 0            ReLU backward of nn.Conv1d + ReLU (core.py:66-71)
//   osp_transpose_last2  (B, R, C) -> (B, C, R)         mel (B, n_feats, T) -> frames (generator/__init__.py:122)
//   osp_length_masks     keep[b, t] = t < len[b] as f32 and its complement as bool  (utils/model.py:12-16 sequence_mask)
//   osp_sum_scaled       out[0] = scale * sum(x)        (.mean() of the per-utterance loss terms)
//   osp_posenc_fwd / osp_posenc_dalpha   y = x + alpha * pe over the batch and d alpha = sum dy * pe  (_transformer/embedding.py:91-124)
//   osp_permute_0213     (A, B, C, D) -> (A, C, B, D)   the head split / merge of the unfused attention path (attention.py:75-101)
//   osp_dot_multi / osp_scale_vec   loss = sum_i c_i * term_i[0] and its backward (generator/__init__.py:175-181, disc/__init__.py:105-111)
#include "osp_common.h"

__global__ void store_i64_kernel(int64_t* dst, int64_t v) { dst[0] = v; }

extern "C" int osp_store_i64(int64_t* dst, int64_t value, hipStream_t stream) {
    OSP_CHECK_ARG(dst != nullptr, "null destination");
    hipLaunchKernelGGL(store_i64_kernel, dim3(1), dim3(1), 0, stream, dst, value);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int ew_grid(int64_t items) { return (int)(items < 1 ? 1 : (cdiv(items, 256) > 4096 ? 4096 : cdiv(items, 256))); }

__global__ void ew_axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int64_t n, float a,
                                float b, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (; i < n4; i += stride) {
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            float4 o;
            if (y) {
                const float4 yv = reinterpret_cast<const float4*>(y)[i];
                o = make_float4(fmaf(a, xv.x, b * yv.x), fmaf(a, xv.y, b * yv.y), fmaf(a, xv.z, b * yv.z), fmaf(a, xv.w, b * yv.w));
            } else {
                o = make_float4(fmaf(a, xv.x, b), fmaf(a, xv.y, b), fmaf(a, xv.z, b), fmaf(a, xv.w, b));
            }
            reinterpret_cast<float4*>(out)[i] = o;
        }
        return;
    }
    for (; i < n; i += stride) out[i] = y ? fmaf(a, x[i], b * y[i]) : fmaf(a, x[i], b);
}

extern "C" int osp_ew_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, hipStream_t stream) {
    OSP_CHECK_ARG(x && out && n >= 0, "bad args");
    if (n == 0) return OSP_OK;
    const int vec = (n % 4 == 0) && al16(x) && al16(out) && (!y || al16(y));
    hipLaunchKernelGGL(ew_axpby_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(256), 0, stream, x, y, out, n, a, b, vec);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// mode 0: y[i]; 1: y[i / inner] (one factor per row); 2: y[i % inner] (one factor per column)
__global__ void ew_mul_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int64_t n, int64_t inner,
                              int mode, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {                                                   // inner % 4 == 0 for the broadcast modes: a float4 stays inside one row
        const int64_t n4 = n >> 2;
        for (; i < n4; i += stride) {
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            float4 yv;
            if (mode == 0) yv = reinterpret_cast<const float4*>(y)[i];
            else if (mode == 1) { const float s = y[(i << 2) / inner]; yv = make_float4(s, s, s, s); }
            else yv = *reinterpret_cast<const float4*>(y + ((i << 2) % inner));
            reinterpret_cast<float4*>(out)[i] = make_float4(xv.x * yv.x, xv.y * yv.y, xv.z * yv.z, xv.w * yv.w);
        }
        return;
    }
    for (; i < n; i += stride) out[i] = x[i] * (mode == 0 ? y[i] : mode == 1 ? y[i / inner] : y[i % inner]);
}

extern "C" int osp_ew_mul(const float* x, const float* y, float* out, int64_t n, int64_t inner, int64_t mode, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && out && n >= 0 && mode >= 0 && mode <= 2 && (mode == 0 || (inner > 0 && n % inner == 0)), "bad args");
    if (n == 0) return OSP_OK;
    const int vec = (n % 4 == 0) && al16(x) && al16(out) && (mode == 0 ? al16(y) : (inner % 4 == 0 && (mode == 1 || al16(y))));
    hipLaunchKernelGGL(ew_mul_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(256), 0, stream, x, y, out, n, inner, (int)mode, vec);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

__global__ void ew_scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out, int64_t n, float c,
                                    int vec) {
    const float f = c * s[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (; i < n4; i += stride) {
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            reinterpret_cast<float4*>(out)[i] = make_float4(f * xv.x, f * xv.y, f * xv.z, f * xv.w);
        }
        return;
    }
    for (; i < n; i += stride) out[i] = f * x[i];
}

extern "C" int osp_ew_scale_dev(const float* x, const float* s, float* out, int64_t n, float c, hipStream_t stream) {
    OSP_CHECK_ARG(x && s && out && n >= 0, "bad args");
    if (n == 0) return OSP_OK;
    const int vec = (n % 4 == 0) && al16(x) && al16(out);
    hipLaunchKernelGGL(ew_scale_dev_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(256), 0, stream, x, s, out, n, c, vec);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

__global__ void ew_relu_mask_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = y[i] > 0.f ? g[i] : 0.f;
}

extern "C" int osp_ew_relu_mask(const float* g, const float* y, float* out, int64_t n, hipStream_t stream) {
    OSP_CHECK_ARG(g && y && out && n >= 0, "bad args");
    if (n == 0) return OSP_OK;
    hipLaunchKernelGGL(ew_relu_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, g, y, out, n);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// (B, R, C) -> (B, C, R) through a 32 x 33 LDS tile: both sides coalesced
__global__ void transpose_last2_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float* xb = x + (int64_t)b * R * C;
    float* yb = y + (int64_t)b * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < C) ? xb[(int64_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < C && r < R) yb[(int64_t)c * R + r] = tile[threadIdx.x][j];
    }
}

extern "C" int osp_transpose_last2(const float* x, float* y, int64_t B, int64_t R, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && B > 0 && R > 0 && C > 0 && B < 65536, "bad args");
    hipLaunchKernelGGL(transpose_last2_kernel, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32), (unsigned)B), dim3(32, 8), 0, stream, x, y,
                       (int)R, (int)C);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

__global__ void length_masks_kernel(const int64_t* __restrict__ len, int B, int T, float* __restrict__ keep, uint8_t* __restrict__ pad) {
    const int64_t n = (int64_t)B * T;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
        const bool k = t < len[b];
        if (keep) keep[i] = k ? 1.f : 0.f;
        if (pad) pad[i] = k ? 0 : 1;
    }
}

extern "C" int osp_length_masks(const int64_t* len, int64_t B, int64_t T, float* keep, void* pad_bool, hipStream_t stream) {
    OSP_CHECK_ARG(len && B > 0 && T > 0 && (keep || pad_bool), "bad args");
    hipLaunchKernelGGL(length_masks_kernel, dim3(ew_grid(B * T)), dim3(256), 0, stream, len, (int)B, (int)T, keep,
                       reinterpret_cast<uint8_t*>(pad_bool));
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// one workgroup: deterministic order (fixed strides, wave tree, wave 0 adds the partial sums in index order)
__global__ void sum_scaled_kernel(const float* __restrict__ x, int64_t n, float scale, float* __restrict__ out) {
    __shared__ float scratch[16];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += x[i];
    const float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) out[0] = scale * tot;
}

extern "C" int osp_sum_scaled(const float* x, int64_t n, float scale, float* out, hipStream_t stream) {
    OSP_CHECK_ARG(x && out && n > 0, "bad args");
    hipLaunchKernelGGL(sum_scaled_kernel, dim3(1), dim3(n >= 1024 ? 1024 : 256), 0, stream, x, n, scale, out);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

#define DOT_MAX 32
struct DotArgs { const float* x[DOT_MAX]; float c[DOT_MAX]; int n; };

__global__ void dot_multi_kernel(DotArgs a, float* __restrict__ out, float* __restrict__ terms) {
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int i = 0; i < a.n; ++i) {                           // fixed order: the same sum run to run
            const float t = a.x[i][0];
            if (terms) terms[i] = t;
            acc = fmaf(a.c[i], t, acc);
        }
        out[0] = acc;
    }
}

// out[0] = sum_i coeff[i] * x_i[0] over `count` device scalars (addresses in x_host); terms (optional): the gathered scalars
extern "C" int osp_dot_multi(const int64_t* x_host, const float* coeff_host, int64_t count, float* out, float* terms, hipStream_t stream) {
    OSP_CHECK_ARG(x_host && coeff_host && out && count > 0 && count <= DOT_MAX, "1..32 terms");
    DotArgs a;
    a.n = (int)count;
    for (int i = 0; i < a.n; ++i) {
        a.x[i] = reinterpret_cast<const float*>(static_cast<intptr_t>(x_host[i]));
        a.c[i] = coeff_host[i];
        OSP_CHECK_ARG(a.x[i] != nullptr, "null term");
    }
    hipLaunchKernelGGL(dot_multi_kernel, dim3(1), dim3(64), 0, stream, a, out, terms);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

struct VecArgs { float c[DOT_MAX]; int n; };
__global__ void scale_vec_kernel(const float* __restrict__ g, VecArgs a, float* __restrict__ out) {
    if ((int)threadIdx.x < a.n) out[threadIdx.x] = g[0] * a.c[threadIdx.x];
}

// out[i] = g[0] * coeff[i]: the gradients of the terms of osp_dot_multi
extern "C" int osp_scale_vec(const float* g, const float* coeff_host, int64_t count, float* out, hipStream_t stream) {
    OSP_CHECK_ARG(g && coeff_host && out && count > 0 && count <= DOT_MAX, "1..32 terms");
    VecArgs a;
    a.n = (int)count;
    for (int i = 0; i < a.n; ++i) a.c[i] = coeff_host[i];
    hipLaunchKernelGGL(scale_vec_kernel, dim3(1), dim3(64), 0, stream, g, a, out);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}


// ---------------------------------------------------------------------------------------------- scaled positional encoding
// ScaledPositionalEncoding.forward (_transformer/embedding.py:120-124): y[b, i] = x[b, i] + alpha[0] * pe[i], i over the T * C
// entries of the table's first T rows; alpha is a learnt DEVICE scalar (no host read).  The gradient w.r.t. x is dy itself; the
// gradient w.r.t. alpha is sum_{b, i} dy[b, i] * pe[i]: stage 1 here writes one partial per workgroup (fixed grid, fixed order inside
// a workgroup -> bit-reproducible), stage 2 is osp_sum_scaled over the partials.
__global__ void posenc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pe, const float* __restrict__ alpha,
                                  float* __restrict__ y, int64_t total, int64_t n, int vec) {
    const float a = alpha[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {                                                   // n % 4 == 0: a float4 stays inside one utterance
        const int64_t t4 = total >> 2;
        for (; i < t4; i += stride) {
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float4 pv = *reinterpret_cast<const float4*>(pe + ((i << 2) % n));
            reinterpret_cast<float4*>(y)[i] = make_float4(fmaf(a, pv.x, xv.x), fmaf(a, pv.y, xv.y), fmaf(a, pv.z, xv.z), fmaf(a, pv.w, xv.w));
        }
        return;
    }
    for (; i < total; i += stride) y[i] = fmaf(a, pe[i % n], x[i]);
}

extern "C" int osp_posenc_fwd(const float* x, const float* pe, const float* alpha, float* y, int64_t B, int64_t n, hipStream_t stream) {
    OSP_CHECK_ARG(x && pe && alpha && y && B > 0 && n > 0, "bad args");
    const int64_t total = B * n;
    const int vec = (n % 4 == 0) && al16(x) && al16(y) && al16(pe);
    hipLaunchKernelGGL(posenc_fwd_kernel, dim3(ew_grid(vec ? total / 4 : total)), dim3(256), 0, stream, x, pe, alpha, y, total, n, vec);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

__global__ __launch_bounds__(256) void posenc_dalpha_kernel(const float* __restrict__ dy, const float* __restrict__ pe,
                                                            float* __restrict__ partials, int64_t total, int64_t n, int vec) {
    __shared__ float red[256];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    if (vec) {
        const int64_t t4 = total >> 2;
        for (; i < t4; i += stride) {
            const float4 g = reinterpret_cast<const float4*>(dy)[i];
            const float4 pv = *reinterpret_cast<const float4*>(pe + ((i << 2) % n));
            s += g.x * pv.x + g.y * pv.y + g.z * pv.z + g.w * pv.w;
        }
    } else {
        for (; i < total; i += stride) s += dy[i] * pe[i % n];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// partials: nparts floats, one per workgroup (the caller sums them: osp_sum_scaled)
extern "C" int osp_posenc_dalpha(const float* dy, const float* pe, int64_t B, int64_t n, float* partials, int64_t nparts,
                                 hipStream_t stream) {
    OSP_CHECK_ARG(dy && pe && partials && B > 0 && n > 0 && nparts > 0 && nparts <= 4096, "bad args");
    const int64_t total = B * n;
    const int vec = (n % 4 == 0) && al16(dy) && al16(pe);
    hipLaunchKernelGGL(posenc_dalpha_kernel, dim3((unsigned)nparts), dim3(256), 0, stream, dy, pe, partials, total, n, vec);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ---------------------------------------------------------------------------------------------- (A, B, C, D) -> (A, C, B, D)
// y[a, c, b, :] = x[a, b, c, :] for contiguous f32 tensors with D % 4 == 0: rows of D floats move as 16-byte chunks, consecutive
// lanes walk a destination row and then the next one (stores fully coalesced, loads coalesced per row of D).
__global__ void permute_0213_kernel(const float4* __restrict__ x, float4* __restrict__ y, int64_t total4, int Bd, int Cd, int D4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total4; o += stride) {
        const int d = (int)(o % D4);
        int64_t r = o / D4;                                      // destination row index (a, c, b)
        const int b = (int)(r % Bd); r /= Bd;
        const int c = (int)(r % Cd); const int64_t a = r / Cd;
        y[o] = x[((a * Bd + b) * Cd + c) * D4 + d];
    }
}

extern "C" int osp_permute_0213(const float* x, float* y, int64_t A, int64_t B, int64_t C, int64_t D, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && A > 0 && B > 0 && C > 0 && D > 0 && D % 4 == 0, "D must be a positive multiple of 4");
    OSP_CHECK_ARG(al16(x) && al16(y) && B < (1ll << 31) && C < (1ll << 31) && D < (1ll << 31), "operands must be 16-byte aligned");
    const int64_t total4 = A * B * C * (D / 4);
    hipLaunchKernelGGL(permute_0213_kernel, dim3(ew_grid(total4)), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(y), total4, (int)B, (int)C, (int)(D / 4));
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
