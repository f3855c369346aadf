// Batched ("multi") forms of the small reductions and packs of the discriminator phase, and the element-wise pieces of the
// ConvNeXt block backward that used to be torch glue.  One launch handles a whole LIST of tensors: the descriptors travel in the
// kernel-argument segment (<= 32 items, ~1-2 KB), a workgroup finds its item from a prefix table of chunk counts.  Rationale
// (VERDICT r01 item 5 / profiles/r02_*): a training step issued 43 + 43 feature-matching launches (22 us each for 2-byte
// loads), 24 + 24 hinge launches, 48 + 48 weight-norm launches and ~60 torch element-wise launches in the block backward --
// ~3 ms of serialised kernel time and ~300 launches for ~0.3 ms of memory traffic.
//
//   osp_l1_sum_multi / osp_l1_sign_multi     FeatureMatchingLoss (disc/loss.py:68-85) over all feature-map pairs of a family
//   osp_hinge_sum_multi / osp_hinge_grad_multi   GeneratorLoss / DiscriminatorLoss hinge terms (disc/loss.py:11-65)
//   osp_wnorm_fwd_multi / osp_wnorm_bwd_multi    weight_norm packs of up to 32 convs per launch (see wnorm.hip)
//   osp_colsum_prod     dgamma[c] += sum_m rowf[m] * dy[m,c] * z[m,c]     (layer-scale gradient, convnext.py:45-46)
//   osp_cast_bf16_rows  y[m,c] = bf16(rowscale[m] * x[m,c])
// Host-side descriptor arrays are `*_host` (plain host memory, read during the call).
#include "osp_common.h"

#define MULTI_MAX 32
#define CHUNK 8192                     // elements per workgroup: 256 threads x 8 elements x 4 iterations

typedef __bf16 bf16_t;

struct MultiPtr {
    const void* a[MULTI_MAX]; const void* b[MULTI_MAX]; void* o[MULTI_MAX];
    long long n[MULTI_MAX]; float scale[MULTI_MAX]; float sgn[MULTI_MAX]; int bf[MULTI_MAX]; int first[MULTI_MAX + 1]; int count;
};

__device__ __forceinline__ int find_item(const int* first, int count, int blk) {
    int it = 0;
#pragma unroll 1
    for (int i = 1; i < count; ++i) it = blk >= first[i] ? i : it;
    return it;
}
__device__ __forceinline__ void ld8(const void* p, int is_bf16, long long i, float (&f)[8]) {
    if (is_bf16) {
        const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + i);
        f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
        f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
        f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xffff0000u);
        f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xffff0000u);
    } else {
        const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i + 4);
        f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
    }
}
__device__ __forceinline__ float ld1(const void* p, int is_bf16, long long i) {
    return is_bf16 ? __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(p)[i]) << 16) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ unsigned pk2b(float a, float b) {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    v2 r; r[0] = (__bf16)a; r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}

// MODE 0: out[0] += scale * sum |a - b|           MODE 1: o[i] = gscale * scale * sign(b - a)
// MODE 0 runs with a bounded grid: a workgroup walks chunks blockIdx.x, + gridDim.x, ... and adds ONE value to out[0] at its end.
// One atomic per 8 192-element chunk meant ~90 k same-address atomics for a family's feature maps; they serialise at the memory
// side and the sum took 212 us where the sign kernel (MODE 1: same reads PLUS a write, no atomic) takes 130.
template <int MODE>
__global__ __launch_bounds__(256) void l1_multi_kernel(MultiPtr d, float* __restrict__ out, const float* __restrict__ gscale) {
    __shared__ float scratch[16];
    float wg_total = 0.f;
  for (int blk = blockIdx.x; blk < d.first[d.count]; blk += gridDim.x) {
    const int it = find_item(d.first, d.count, blk);
    const long long n = d.n[it], base = (long long)(blk - d.first[it]) * CHUNK;
    const void* a = d.a[it]; const void* b = d.b[it];
    const int bf = d.bf[it], es = bf ? 2 : 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | (MODE ? reinterpret_cast<uintptr_t>(d.o[it]) : 0)) & 15) == 0;
    const long long end = base + CHUNK < n ? base + CHUNK : n;
    float s = 0.f;
    const float gsc = MODE ? gscale[0] * d.scale[it] : 0.f;
    (void)es;
    if (vec) {
        long long i = base + threadIdx.x * 8;
        if (MODE == 0) {
            // four 16-byte pairs per trip, all eight loads requested before the first subtraction (one pair per trip kept 32 bytes
            // per thread in flight: 1.4 TB/s over the ~0.6 GB of feature maps of a family)
            for (; i + 3 * 2048 + 8 <= end; i += 4 * 2048) {
                float x[4][8], y[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { ld8(a, bf, i + q * 2048, x[q]); ld8(b, bf, i + q * 2048, y[q]); }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += fabsf(x[q][k] - y[q][k]);
            }
        }
        for (; i + 8 <= end; i += 256 * 8) {
            float x[8], y[8];
            ld8(a, bf, i, x); ld8(b, bf, i, y);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s += fabsf(x[k] - y[k]);
            } else {
                float r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float dd = y[k] - x[k]; r[k] = dd > 0.f ? gsc : (dd < 0.f ? -gsc : 0.f); }
                if (bf) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(d.o[it]) + i) =
                            make_uint4(pk2b(r[0], r[1]), pk2b(r[2], r[3]), pk2b(r[4], r[5]), pk2b(r[6], r[7]));
                else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.o[it]) + i) = make_float4(r[0], r[1], r[2], r[3]);
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.o[it]) + i + 4) = make_float4(r[4], r[5], r[6], r[7]);
                }
            }
        }
    }
    // tail of the chunk (the last < 8 elements of the tensor), or the whole chunk when a pointer is not 16-byte aligned
    const long long t0 = vec ? base + ((end - base) & ~7ll) : base;
    for (long long i = t0 + threadIdx.x; i < end; i += 256) {
        const float x = ld1(a, bf, i), y = ld1(b, bf, i);
        if (MODE == 0) s += fabsf(x - y);
        else {
            const float dd = y - x, r = dd > 0.f ? gsc : (dd < 0.f ? -gsc : 0.f);
            if (bf) reinterpret_cast<bf16_t*>(d.o[it])[i] = (bf16_t)r; else reinterpret_cast<float*>(d.o[it])[i] = r;
        }
    }
    if (MODE == 0) wg_total += s * d.scale[it];                // per-thread partial, already scaled for its item
  }
    if (MODE == 0) {
        wg_total = block_sum(wg_total, scratch);
        if (threadIdx.x == 0) atomicAdd(out, wg_total);
    }
}

// MODE 0: out += scale * sum max(0, 1 + sgn * x)     MODE 1: dx = gscale * scale * sgn * [1 + sgn * x > 0]      (f32)
template <int MODE>
__global__ __launch_bounds__(256) void hinge_multi_kernel(MultiPtr d, float* __restrict__ out, const float* __restrict__ gscale) {
    __shared__ float scratch[16];
    const int it = find_item(d.first, d.count, blockIdx.x);
    const long long n = d.n[it], base = (long long)(blockIdx.x - d.first[it]) * CHUNK;
    const long long end = base + CHUNK < n ? base + CHUNK : n;
    const float* x = reinterpret_cast<const float*>(d.a[it]);
    const float sgn = d.sgn[it];
    float s = 0.f;
    const float g = MODE ? gscale[0] * d.scale[it] * sgn : 0.f;
    for (long long i = base + threadIdx.x; i < end; i += 256) {
        const float h = 1.f + sgn * x[i];
        if (MODE == 0) s += fmaxf(0.f, h);
        else reinterpret_cast<float*>(d.o[it])[i] = h > 0.f ? g : 0.f;
    }
    if (MODE == 0) {
        s = block_sum(s, scratch);
        if (threadIdx.x == 0) atomicAdd(out, s * d.scale[it]);
    }
}

static int fill_multi(MultiPtr& d, const int64_t* a, const int64_t* b, const int64_t* o, const int64_t* n, const float* scale,
                      const float* sgn, int64_t lo, int64_t hi, const int64_t* bf) {
    int blocks = 0;
    d.count = (int)(hi - lo);
    for (int64_t i = lo; i < hi; ++i) {
        const int k = (int)(i - lo);
        d.a[k] = (const void*)(intptr_t)a[i]; d.b[k] = b ? (const void*)(intptr_t)b[i] : nullptr; d.o[k] = o ? (void*)(intptr_t)o[i] : nullptr;
        d.n[k] = n[i]; d.scale[k] = scale[i]; d.sgn[k] = sgn ? sgn[i] : 0.f; d.bf[k] = bf ? (int)bf[i] : 0;
        d.first[k] = blocks;
        blocks += (int)cdiv(n[i], CHUNK);
    }
    d.first[d.count] = blocks;
    return blocks;
}

// out[0] += sum_i scale[i] * sum |a_i - b_i|.  a_host / b_host: device addresses of the tensors (as int64), n_host: element counts,
// bf16_host: 1 where pair i is stored bf16 (0 = f32).
extern "C" int osp_l1_sum_multi(const int64_t* a_host, const int64_t* b_host, const int64_t* n_host, const float* scale_host,
                                const int64_t* bf16_host, int64_t count, float* out, hipStream_t stream) {
    OSP_CHECK_ARG(a_host && b_host && n_host && scale_host && bf16_host && out && count > 0, "bad args");
    for (int64_t lo = 0; lo < count; lo += MULTI_MAX) {
        MultiPtr d;
        const int blocks = fill_multi(d, a_host, b_host, nullptr, n_host, scale_host, nullptr, lo, lo + MULTI_MAX < count ? lo + MULTI_MAX : count, bf16_host);
        if (blocks > 0) hipLaunchKernelGGL(l1_multi_kernel<0>, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, d, out, (const float*)nullptr);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
// gb_i = gscale[0] * scale[i] * sign(b_i - a_i)   (a = target, b = the tensor that carries the gradient; same dtype as the inputs)
extern "C" int osp_l1_sign_multi(const int64_t* a_host, const int64_t* b_host, const int64_t* gb_host, const int64_t* n_host,
                                 const float* scale_host, const int64_t* bf16_host, int64_t count, const float* gscale, hipStream_t stream) {
    OSP_CHECK_ARG(a_host && b_host && gb_host && n_host && scale_host && bf16_host && gscale && count > 0, "bad args");
    for (int64_t lo = 0; lo < count; lo += MULTI_MAX) {
        MultiPtr d;
        const int blocks = fill_multi(d, a_host, b_host, gb_host, n_host, scale_host, nullptr, lo, lo + MULTI_MAX < count ? lo + MULTI_MAX : count, bf16_host);
        if (blocks > 0) hipLaunchKernelGGL(l1_multi_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, d, (float*)nullptr, gscale);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
// out[0] += sum_i scale[i] * sum max(0, 1 + sgn[i] * x_i)
extern "C" int osp_hinge_sum_multi(const int64_t* x_host, const int64_t* n_host, const float* sgn_host, const float* scale_host,
                                   int64_t count, float* out, hipStream_t stream) {
    OSP_CHECK_ARG(x_host && n_host && sgn_host && scale_host && out && count > 0, "bad args");
    for (int64_t lo = 0; lo < count; lo += MULTI_MAX) {
        MultiPtr d;
        const int blocks = fill_multi(d, x_host, nullptr, nullptr, n_host, scale_host, sgn_host, lo, lo + MULTI_MAX < count ? lo + MULTI_MAX : count, nullptr);
        if (blocks > 0) hipLaunchKernelGGL(hinge_multi_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, d, out, (const float*)nullptr);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
// dx_i = gscale[0] * scale[i] * sgn[i] * [1 + sgn[i] * x_i > 0]
extern "C" int osp_hinge_grad_multi(const int64_t* x_host, const int64_t* dx_host, const int64_t* n_host, const float* sgn_host,
                                    const float* scale_host, int64_t count, const float* gscale, hipStream_t stream) {
    OSP_CHECK_ARG(x_host && dx_host && n_host && sgn_host && scale_host && gscale && count > 0, "bad args");
    for (int64_t lo = 0; lo < count; lo += MULTI_MAX) {
        MultiPtr d;
        const int blocks = fill_multi(d, x_host, nullptr, dx_host, n_host, scale_host, sgn_host, lo, lo + MULTI_MAX < count ? lo + MULTI_MAX : count, nullptr);
        if (blocks > 0) hipLaunchKernelGGL(hinge_multi_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, d, (float*)nullptr, gscale);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ weight norm, many convs
struct WnItem { const float* v; const float* g; bf16_t* wn; float* wn32; bf16_t* wt; float* inv; const float* dwn; float* dv; float* dg;
                int Cout, Cin, P, Q; };
struct WnMulti { WnItem it[MULTI_MAX]; int first[MULTI_MAX + 1]; int count; };

__global__ __launch_bounds__(256) void wnorm_fwd_multi_kernel(WnMulti d) {
    __shared__ float scratch[16];
    const int k = find_item(d.first, d.count, blockIdx.x);
    const WnItem w = d.it[k];
    const int n = blockIdx.x - d.first[k], PQ = w.P * w.Q, E = w.Cin * PQ;
    const float* vn = w.v + (int64_t)n * E;
    float s = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) { const float x = vn[e]; s = fmaf(x, x, s); }
    s = block_sum(s, scratch);
    const float inv = rsqrtf(s), sc = w.g[n] * inv;
    if (threadIdx.x == 0 && w.inv) w.inv[n] = inv;
    // iterate in OUTPUT order (q, p, c): the bf16 native copy is written contiguously, the f32 source reads stride P*Q
    for (int o = threadIdx.x; o < E; o += 256) {
        const int c = o % w.Cin, r = o / w.Cin, p = r % w.P, q = r / w.P;
        const float val = vn[(c * w.P + p) * w.Q + q] * sc;
        const int64_t on = (int64_t)n * E + o;
        if (w.wn) w.wn[on] = (bf16_t)val;
        if (w.wn32) w.wn32[on] = val;
    }
}
// dgrad layout wt (Cin, Q, P, Cout) from the native pack wn (Cout, Q, P, Cin): per (q, p) a Cout x Cin transpose, 32 x 32 tiles
// through LDS so that both the reads (along Cin) and the writes (along Cout) are contiguous.  (Written from the normalising
// kernel, one output channel per workgroup, these were 2-byte stores Cout apart: 0.6 ms per step for 0.1 ms of traffic.)
struct WnTMulti { const bf16_t* wn[MULTI_MAX]; bf16_t* wt[MULTI_MAX]; int Cout[MULTI_MAX], Cin[MULTI_MAX], QP[MULTI_MAX], tn[MULTI_MAX], tc[MULTI_MAX];
                  int first[MULTI_MAX + 1]; int count; };
__global__ __launch_bounds__(256) void wnorm_transpose_multi_kernel(WnTMulti d) {
    __shared__ unsigned short tile[32][33];
    const int k = find_item(d.first, d.count, blockIdx.x);
    const int Cout = d.Cout[k], Cin = d.Cin[k], QP = d.QP[k];
    int b = blockIdx.x - d.first[k];
    const int ct = b % d.tc[k]; b /= d.tc[k];
    const int nt = b % d.tn[k], qp = b / d.tn[k];
    const unsigned short* src = reinterpret_cast<const unsigned short*>(d.wn[k]);
    unsigned short* dst = reinterpret_cast<unsigned short*>(d.wt[k]);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = nt * 32 + r, c = ct * 32 + tx;
        tile[r][tx] = (n < Cout && c < Cin) ? src[((int64_t)n * QP + qp) * Cin + c] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int c = ct * 32 + r, n = nt * 32 + tx;
        if (c < Cin && n < Cout) dst[((int64_t)c * QP + qp) * Cout + n] = tile[tx][r];
    }
}
__global__ __launch_bounds__(256) void wnorm_bwd_multi_kernel(WnMulti d) {
    __shared__ float scratch[16];
    const int k = find_item(d.first, d.count, blockIdx.x);
    const WnItem w = d.it[k];
    const int n = blockIdx.x - d.first[k], E = w.Cin * w.P * w.Q;
    const float* vn = w.v + (int64_t)n * E;
    const float* dn = w.dwn + (int64_t)n * E;              // native (q, p, c) order
    float s = 0.f;
    for (int o = threadIdx.x; o < E; o += 256) {
        const int c = o % w.Cin, r = o / w.Cin, p = r % w.P, q = r / w.P;
        s = fmaf(dn[o], vn[(c * w.P + p) * w.Q + q], s);
    }
    s = block_sum(s, scratch);
    const float inv = w.inv[n], gn = w.g[n];
    if (threadIdx.x == 0) w.dg[n] += s * inv;
    const float a = gn * inv, b = gn * s * inv * inv * inv;
    for (int e = threadIdx.x; e < E; e += 256) {
        const int c = e / (w.P * w.Q), r = e - c * (w.P * w.Q), p = r / w.Q, q = r - p * w.Q;
        w.dv[(int64_t)n * E + e] += a * dn[((q * w.P + p) * w.Cin) + c] - b * vn[e];
    }
}
// desc_host: count rows of 13 int64: {v, g, wn_bf16, wn_f32, wt_bf16, inv_norm, dwn, dv, dg, Cout, Cin, P, Q} (addresses; 0 = absent).
// Forward uses v, g -> wn / wn32 / wt / inv (see osp_wnorm_fwd); backward uses dwn, v, g, inv -> dv, dg (accumulated).
static int wn_launch(const int64_t* desc, int64_t count, bool bwd, hipStream_t stream) {
    for (int64_t lo = 0; lo < count; lo += MULTI_MAX) {
        WnMulti d;
        const int64_t hi = lo + MULTI_MAX < count ? lo + MULTI_MAX : count;
        int blocks = 0;
        d.count = (int)(hi - lo);
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t* r = desc + 13 * i;
            WnItem& w = d.it[i - lo];
            w.v = (const float*)(intptr_t)r[0]; w.g = (const float*)(intptr_t)r[1]; w.wn = (bf16_t*)(intptr_t)r[2];
            w.wn32 = (float*)(intptr_t)r[3]; w.wt = (bf16_t*)(intptr_t)r[4]; w.inv = (float*)(intptr_t)r[5];
            w.dwn = (const float*)(intptr_t)r[6]; w.dv = (float*)(intptr_t)r[7]; w.dg = (float*)(intptr_t)r[8];
            w.Cout = (int)r[9]; w.Cin = (int)r[10]; w.P = (int)r[11]; w.Q = (int)r[12];
            if (!w.v || !w.g || w.Cout <= 0 || (bwd && (!w.dwn || !w.dv || !w.dg || !w.inv))) return -1;
            d.first[i - lo] = blocks;
            blocks += w.Cout;
        }
        d.first[d.count] = blocks;
        if (bwd) hipLaunchKernelGGL(wnorm_bwd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d);
        else {
            hipLaunchKernelGGL(wnorm_fwd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d);
            WnTMulti t;
            int tb = 0, nt = 0;
            for (int i = 0; i < d.count; ++i) {
                const WnItem& w = d.it[i];
                if (!w.wt) continue;
                if (!w.wn) return -1;                        // the transposed copy is made from the native bf16 pack
                t.wn[nt] = w.wn; t.wt[nt] = w.wt; t.Cout[nt] = w.Cout; t.Cin[nt] = w.Cin; t.QP[nt] = w.P * w.Q;
                t.tn[nt] = (w.Cout + 31) / 32; t.tc[nt] = (w.Cin + 31) / 32;
                t.first[nt] = tb;
                tb += t.tn[nt] * t.tc[nt] * t.QP[nt];
                ++nt;
            }
            t.first[nt] = tb; t.count = nt;
            if (nt > 0) hipLaunchKernelGGL(wnorm_transpose_multi_kernel, dim3((unsigned)tb), dim3(256), 0, stream, t);
        }
    }
    return 0;
}
extern "C" int osp_wnorm_fwd_multi(const int64_t* desc_host, int64_t count, hipStream_t stream) {
    OSP_CHECK_ARG(desc_host && count > 0, "bad args");
    OSP_CHECK_ARG(wn_launch(desc_host, count, false, stream) == 0, "bad descriptor");
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
extern "C" int osp_wnorm_bwd_multi(const int64_t* desc_host, int64_t count, hipStream_t stream) {
    OSP_CHECK_ARG(desc_host && count > 0, "bad args");
    OSP_CHECK_ARG(wn_launch(desc_host, count, true, stream) == 0, "bad descriptor");
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ ConvNeXt backward pieces
// out[c] += sum_m rowf[m] * a[m, c] * b[m, c]      (C % 4 == 0; rowf may be null; b null = 1: a plain column sum)
__global__ __launch_bounds__(256) void colsum_prod_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ rowf, float* __restrict__ out, int64_t M, int C,
                                                          int rows_per_block) {
    __shared__ float red[256];
    const int c4 = C >> 2, col = threadIdx.x % c4, sub = threadIdx.x / c4, nsub = 256 / c4;   // c4 <= 256
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block, m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub < nsub) {
        int64_t m = m0 + sub;
        // four rows per trip, their loads requested before the first FMA (one row per trip = two dependent 16-byte loads in flight)
        for (; m + 3 * nsub < m1; m += 4 * nsub) {
            float4 x[4], y[4]; float r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t mq = m + q * nsub;
                x[q] = *reinterpret_cast<const float4*>(a + mq * C + col * 4);
                y[q] = b ? *reinterpret_cast<const float4*>(b + mq * C + col * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                r[q] = rowf ? rowf[mq] : 1.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc.x = fmaf(r[q] * x[q].x, y[q].x, acc.x); acc.y = fmaf(r[q] * x[q].y, y[q].y, acc.y);
                acc.z = fmaf(r[q] * x[q].z, y[q].z, acc.z); acc.w = fmaf(r[q] * x[q].w, y[q].w, acc.w);
            }
        }
        for (; m < m1; m += nsub) {
            const float4 x = *reinterpret_cast<const float4*>(a + m * C + col * 4);
            const float4 y = b ? *reinterpret_cast<const float4*>(b + m * C + col * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float r = rowf ? rowf[m] : 1.f;
            acc.x = fmaf(r * x.x, y.x, acc.x); acc.y = fmaf(r * x.y, y.y, acc.y); acc.z = fmaf(r * x.z, y.z, acc.z); acc.w = fmaf(r * x.w, y.w, acc.w);
        }
    }
    // combine the nsub row groups through LDS, one component at a time
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x] = k == 0 ? acc.x : k == 1 ? acc.y : k == 2 ? acc.z : acc.w;
        __syncthreads();
        if (threadIdx.x < c4) {
            float s = 0.f;
            for (int j = 0; j < nsub; ++j) s += red[j * c4 + threadIdx.x];
            atomicAdd(out + threadIdx.x * 4 + k, s);
        }
        __syncthreads();
    }
}
extern "C" int osp_colsum_prod(const float* a, const float* b, const float* rowf, float* out, int64_t M, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(a && out && M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "C must be a multiple of 4, <= 1024");
    // ~1024 workgroups whatever M is (the encoder / vocoder calls have M = 2-4k rows: 256 rows per block left 8-16 blocks on a
    // 256-CU part, 54 us per launch); at least 8 rows per block so the atomics stay few
    // (round 3: ~64 workgroups, not 1024 -- 1024 / 256 / 128 / 64 / 32 workgroups: 35.4 / 27.6 / 15.7 / 11.3 / 11.7 us per launch averaged over a step's 12 launches: every workgroup adds into the SAME C addresses, and same-address atomics serialise at the
    // memory side -- the kernel's tail was proportional to the number of workgroups, see smallcin.hip's two-stage note)
    static int64_t wg_target = 0;
    if (!wg_target) { wg_target = 64; }
    int64_t rpb = cdiv(M, wg_target);
    rpb = rpb < 8 ? 8 : rpb;
    hipLaunchKernelGGL(colsum_prod_kernel, dim3((unsigned)cdiv(M, rpb)), dim3(256), 0, stream, a, b, rowf, out, M, (int)C, (int)rpb);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// y[m, c] = bf16(rowscale[m] * x[m, c])      (C % 4 == 0)
__global__ __launch_bounds__(256) void cast_bf16_rows_kernel(const float* __restrict__ x, const float* __restrict__ rowscale,
                                                             unsigned short* __restrict__ y, int64_t M, int C) {
    const int c4 = C >> 2;
    const int64_t n4 = M * c4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float r = rowscale ? rowscale[i / c4] : 1.f;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<uint2*>(y)[i] = make_uint2(pk2b(r * v.x, r * v.y), pk2b(r * v.z, r * v.w));
    }
}
extern "C" int osp_cast_bf16_rows(const float* x, const float* rowscale, void* y, int64_t M, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && M > 0 && C > 0 && C % 4 == 0, "C must be a multiple of 4");
    const int64_t blocks = cdiv(M * (C / 4), 256 * 4);
    hipLaunchKernelGGL(cast_bf16_rows_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, x, rowscale,
                       (unsigned short*)y, M, (int)C);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}


// ------------------------------------------------------------------------------------------------ spectral loss reductions
// STFTLoss / MelSpecReconstructionLoss (vocoder/wavenext/disc/loss.py:88-120,197-270) over magnitudes x (prediction) and y (target):
//   sums[0] += sum (y - x)^2      sums[1] += sum y^2      sums[2] += sum |log max(y, clip) - log max(x, clip)|
// -> spectral convergence = sqrt(sums[0] / sums[1]), log-magnitude L1 = sums[2] / n.  One pass instead of ~8 torch launches.
__global__ __launch_bounds__(256) void spectral_sums_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                            float clip, float* __restrict__ sums) {
    __shared__ float scratch[16];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(y)[i];
        const float xs[4] = {a.x, a.y, a.z, a.w}, ys[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = ys[k] - xs[k];
            s0 = fmaf(d, d, s0); s1 = fmaf(ys[k], ys[k], s1);
            s2 += fabsf(logf(fmaxf(ys[k], clip)) - logf(fmaxf(xs[k], clip)));
        }
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = y[i] - x[i];
        s0 = fmaf(d, d, s0); s1 = fmaf(y[i], y[i], s1);
        s2 += fabsf(logf(fmaxf(y[i], clip)) - logf(fmaxf(x[i], clip)));
    }
    s0 = block_sum(s0, scratch); s1 = block_sum(s1, scratch); s2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) { atomicAdd(sums, s0); atomicAdd(sums + 1, s1); atomicAdd(sums + 2, s2); }
}
extern "C" int osp_spectral_loss_sums(const float* x, const float* y, int64_t n, float clip, float* sums, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && sums && n > 0, "bad args");
    OSP_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "operands must be 16-byte aligned");
    const int64_t blocks = cdiv(n, 256 * 16);
    hipLaunchKernelGGL(spectral_sums_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, x, y, n, clip, sums);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
// dx = g[0] * d(sqrt(s0/s1))/dx + g[1] * d(s2/n)/dx  with the sums of the forward:
//   d sc / dx = -(y - x) / (sqrt(s0) * sqrt(s1))          d mag / dx = -sign(log yc - log xc) * [x > clip] / (x * n)
__global__ __launch_bounds__(256) void spectral_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n, float clip,
                                                           const float* __restrict__ sums, const float* __restrict__ g,
                                                           float* __restrict__ dx) {
    const float s0 = sums[0], s1 = sums[1];
    const float ksc = (s0 > 0.f && s1 > 0.f) ? -g[0] * rsqrtf(s0) * rsqrtf(s1) : 0.f, kmag = -g[1] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float xv = x[i], yv = y[i];
        const float dl = logf(fmaxf(yv, clip)) - logf(fmaxf(xv, clip));
        const float sg = dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f);
        dx[i] = ksc * (yv - xv) + (xv > clip ? kmag * sg / xv : 0.f);
    }
}
extern "C" int osp_spectral_loss_bwd(const float* x, const float* y, int64_t n, float clip, const float* sums, const float* g, float* dx,
                                     hipStream_t stream) {
    OSP_CHECK_ARG(x && y && sums && g && dx && n > 0, "bad args");
    const int64_t blocks = cdiv(n, 256 * 8);
    hipLaunchKernelGGL(spectral_bwd_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, x, y, n, clip, sums, g, dx);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}


// ------------------------------------------------------------------------------------------------ random segment starts
// get_random_segments (utils/segments.py:12-38, caller generator/__init__.py:147-153): with num_frames = float(len - 4),
//   max_start = clamp(num_frames - segment_size, min = 0);  start = long(rand01 * max_start)        (all in f32, as upstream)
__global__ void segment_starts_kernel(const float* __restrict__ r01, const int64_t* __restrict__ lens, int B, float lead, float seg,
                                      int64_t* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float max_start = fmaxf((float)(lens[b] - (int64_t)lead) - seg, 0.f);
    out[b] = (int64_t)(r01[b] * max_start);
}
extern "C" int osp_segment_starts(const float* r01, const int64_t* lens, int64_t B, int64_t lead, int64_t segment_size, int64_t* out,
                                  hipStream_t stream) {
    OSP_CHECK_ARG(r01 && lens && out && B > 0 && segment_size >= 0, "bad args");
    hipLaunchKernelGGL(segment_starts_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, stream, r01, lens, (int)B, (float)lead,
                       (float)segment_size, out);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}


// ------------------------------------------------------------------------------------------------ DropPath row factors
// DropPath of every block of a ConvNeXt backbone (generator/modules/convnext.py:121-129: bernoulli(keep) / keep per block and
// utterance) from the package's counter-based RNG -- Philox key (seed, stream), counter = block * B + utterance -- so that a step
// replayed from a hipGraph draws fresh factors (seed in device memory) without torch's generator:
//   scale[l, b*T + t] = 0 with probability p_l, else 1 / (1 - p_l);     rowf[l, m] = scale[l, m] * rowmask[m]
#define DROP_PATH_MAXL 64
struct DropPathP { float p[DROP_PATH_MAXL]; };
__global__ __launch_bounds__(256) void drop_path_rows_kernel(const DropPathP dp, const float* __restrict__ rowmask, uint64_t seed,
                                                             const int64_t* __restrict__ seed_dev, uint32_t stream, float* __restrict__ scale,
                                                             float* __restrict__ rowf, int B, int T) {
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int l = blockIdx.y;
    const int64_t BT = (int64_t)B * T;
    const float p = dp.p[l];
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < BT; m += (int64_t)gridDim.x * 256) {
        const int b = (int)(m / T);
        const float f = p > 0.f ? dropout_factor(seed, stream, (uint64_t)l * B + b, p) : 1.f;
        scale[l * BT + m] = f;
        if (rowf) rowf[l * BT + m] = f * rowmask[m];
    }
}
extern "C" int osp_drop_path_rows(const float* drop_p_host, const float* rowmask, int64_t L, int64_t B, int64_t T, int64_t seed,
                                  const int64_t* seed_dev, int64_t stream_id, float* scale, float* rowf, hipStream_t stream) {
    OSP_CHECK_ARG(drop_p_host && scale && L > 0 && L <= DROP_PATH_MAXL && B > 0 && T > 0, "bad args");
    OSP_CHECK_ARG((rowf != nullptr) == (rowmask != nullptr), "rowf needs rowmask (and vice versa)");
    DropPathP dp;
    for (int64_t l = 0; l < L; ++l) {
        OSP_CHECK_ARG(drop_p_host[l] >= 0.f && drop_p_host[l] < 1.f, "drop probability outside [0, 1)");
        dp.p[l] = drop_p_host[l];
    }
    const int64_t blocks = cdiv(B * T, 256);
    hipLaunchKernelGGL(drop_path_rows_kernel, dim3((unsigned)(blocks < 64 ? blocks : 64), (unsigned)L), dim3(256), 0, stream, dp, rowmask,
                       (uint64_t)seed, seed_dev, (uint32_t)stream_id, scale, rowf, (int)B, (int)T);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}


// ------------------------------------------------------------------------------------------------ period folding (DiscriminatorP)
// DiscriminatorP.forward (_discriminators.py:63-72): reflect-pad the wave on the right to a multiple of the period, view it as
// (b, t / p, p) and hand every period column to the (k, 1) convs -- here as channels-last sequences (b * p, t / p):
//   seq[b * p + w, h] = xpad[b, h * p + w],   xpad[b, i] = x[b, i] (i < T),  x[b, 2 T - 2 - i] (T <= i < Tp * p)
// One launch instead of reflection_pad1d + a strided copy; the gradient (dx[i] = g(i) + g(2 T - 2 - i) where the mirror image lies in
// the pad) is one launch instead of a fill + reflection_pad1d_backward + a strided copy + an add.
__global__ __launch_bounds__(256) void period_fold_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int T, int p, int Tp,
                                                          int backward) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        if (!backward) {                                       // e indexes seq: (b, w, h)
            const int h = (int)(e % Tp);
            const int64_t bw = e / Tp;
            const int w = (int)(bw % p);
            const int64_t b = bw / p;
            const int i = h * p + w;
            y[e] = x[b * T + (i < T ? i : 2 * T - 2 - i)];
        } else {                                               // e indexes dx: (b, i);  x = dseq
            const int i = (int)(e % T);
            const int64_t b = e / T;
            const float* g = x + b * (int64_t)p * Tp;
            float v = g[(int64_t)(i % p) * Tp + i / p];
            const int m = 2 * T - 2 - i;                        // mirror image of i in the pad, if any
            if (m >= T && m < Tp * p) v += g[(int64_t)(m % p) * Tp + m / p];
            y[e] = v;
        }
    }
}
extern "C" int osp_period_fold(const float* x, float* y, int64_t B, int64_t T, int64_t period, int64_t backward, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && B > 0 && T > 1 && period > 0 && period < T, "bad args");
    const int64_t Tp = cdiv(T, period);
    const int64_t n = backward ? B * T : B * period * Tp;
    const int64_t blocks = cdiv(n, 256 * 4);
    hipLaunchKernelGGL(period_fold_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, x, y, n, (int)T, (int)period,
                       (int)Tp, (int)(backward != 0));
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ clip (WaveNeXtHead, A12)
// audio = clip(x, -1, 1)  (vocoder/wavenext/__init__.py:47) and its backward dx = dy where lo <= x <= hi (torch.clamp's rule),
// 16-byte accesses; dy == null selects the forward.
__global__ __launch_bounds__(256) void clip_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n,
                                                   float lo, float hi) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        if (!dy) {
            o = make_float4(fminf(fmaxf(v.x, lo), hi), fminf(fmaxf(v.y, lo), hi), fminf(fmaxf(v.z, lo), hi), fminf(fmaxf(v.w, lo), hi));
        } else {
            const float4 g = reinterpret_cast<const float4*>(dy)[i];
            o = make_float4((v.x >= lo && v.x <= hi) ? g.x : 0.f, (v.y >= lo && v.y <= hi) ? g.y : 0.f, (v.z >= lo && v.z <= hi) ? g.z : 0.f,
                            (v.w >= lo && v.w <= hi) ? g.w : 0.f);
        }
        reinterpret_cast<float4*>(out)[i] = o;
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256)
            out[i] = dy ? ((x[i] >= lo && x[i] <= hi) ? dy[i] : 0.f) : fminf(fmaxf(x[i], lo), hi);
}
extern "C" int osp_clip(const float* x, const float* dy, float* out, int64_t n, float lo, float hi, hipStream_t stream) {
    OSP_CHECK_ARG(x && out && n > 0 && lo <= hi, "bad args");
    OSP_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15) == 0, "16-byte aligned operands");
    const int64_t blocks = cdiv(n, 256 * 4 * 4);
    hipLaunchKernelGGL(clip_kernel, dim3((unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048)), dim3(256), 0, stream, x, dy, out, n, lo, hi);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
