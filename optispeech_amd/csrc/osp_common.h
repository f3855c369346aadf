// Shared device/host helpers for libosp_hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libosp_hip is written for gfx950 (CDNA4) only: DPP row_bcast / wave_shr controls, ds_read_b64_tr_b16, 32x32x16 bf16 MFMA, LDS-DMA"
#endif

#define OSP_OK 0
#define OSP_ERR_ARG -1
#define OSP_ERR_HIP -2
#define OSP_ERR_UNSUPPORTED -3

extern "C" const char* osp_last_error();
void osp_set_error(const char* fmt, ...);
// measurement aid (api.cpp): symbol of the matrix-core kernel a launch helper picked, algorithmic flops of the launch
void osp_note_symbol(const char* sym);
void osp_note_flops(double flops);
void osp_note_bytes(double bytes);

#define OSP_CHECK_ARG(cond, msg)                                      \
    do {                                                              \
        if (!(cond)) {                                                \
            osp_set_error("%s: %s (%s)", __func__, msg, #cond);       \
            return OSP_ERR_ARG;                                       \
        }                                                             \
    } while (0)

#define OSP_LAUNCH_CHECK()                                            \
    do {                                                              \
        hipError_t e__ = hipGetLastError();                           \
        if (e__ != hipSuccess) {                                      \
            osp_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return OSP_ERR_HIP;                                       \
        }                                                             \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- wave / block reductions
// Round 3: DPP row operations + one v_readlane instead of six ds_bpermute_b32 round trips (what __shfl_xor compiles to: ~100
// cycles of LDS-pipe latency each, 1 296 of them in convnext.hip alone; a LayerNorm row = two DEPENDENT reductions).  The tree
// is the xor butterfly's -- quads, 8s (row_half_mirror), 16s (row_mirror), then the rows through row_bcast:15 / row_bcast:31 --
// so the sums are bit-identical to the round-2 ones.  The total lands in lane 63 and comes back wave-uniform (SGPR).
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float dpp_get(float oldv, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
// (a wave that is only partly active -- a 32-thread workgroup, a divergent caller -- takes the shuffle butterfly: the DPP tree
// ends in lane 63, which must have executed it)
__device__ __forceinline__ bool wave_is_full() { return __builtin_amdgcn_read_exec() == ~0ull; }
__device__ __forceinline__ float wave_sum(float v) {
    if (!wave_is_full()) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    v += dpp_get<0xB1>(0.f, v);            // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(0.f, v);            // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(0.f, v);           // row_half_mirror
    v += dpp_get<0x140>(0.f, v);           // row_mirror: every lane of a 16-lane row holds the row's sum
    v += dpp_get<0x142, 0xA>(0.f, v);      // row_bcast:15 into rows 1 and 3
    v += dpp_get<0x143, 0xC>(0.f, v);      // row_bcast:31 into rows 2 and 3: lane 63 = total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    if (!wave_is_full()) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        return v;
    }
    v = fmaxf(v, dpp_get<0xB1>(v, v));
    v = fmaxf(v, dpp_get<0x4E>(v, v));
    v = fmaxf(v, dpp_get<0x141>(v, v));
    v = fmaxf(v, dpp_get<0x140>(v, v));
    v = fmaxf(v, dpp_get<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_get<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// block reduction through LDS scratch (>= 16 floats); result valid in all threads
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// ---------------------------------------------------------------- activations (exact erf GELU == nn.GELU())
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// erf-GELU for epilogues whose RESULT IS ROUNDED TO bf16 (the performance mode's I-wide intermediates): Abramowitz-Stegun 7.1.26,
// |erf error| <= 1.5e-7 -- four orders below the bf16 unit round-off -- on one v_rcp_f32 and one v_exp_f32 plus 8 FMAs instead of the
// ~50-instruction branchy library erff.  At the decoder shape (26 M GELUs per block and direction) the library call alone
// was 37 of the 87 us of the pwconv1 launch (tools/convnext_pw_probe.py (git history)).  Phi(x) is formed without cancellation on the
// negative side (Phi(x) = q / 2, x < 0; 1 - q / 2 otherwise; q = poly(t) exp(-x^2 / 2)); exp(-x^2 / 2) is shared with the pdf.
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& e) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    // v_rcp_f32 (1 ulp) -- `__frcp_rn` / `1.0f / x` compile to the correctly rounded division (v_div_scale / v_div_fmas / v_div_fixup:
    // ten more instructions per element, found in round 4 in the ISA of the fused MLP's first version)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    e = __expf(-ax * ax);
    const float hq = 0.5f * poly * e;
    cdf = x < 0.f ? hq : 1.0f - hq;
}
__device__ __forceinline__ float gelu_fast_f(float x) { float c, e; gelu_fast_parts(x, c, e); return x * c; }
__device__ __forceinline__ float gelu_grad_fast_f(float x) { float c, e; gelu_fast_parts(x, c, e); return fmaf(x * 0.39894228040143267794f, e, c); }

// ---------------------------------------------------------------- counter-based RNG (dropout / drop-path)
// Philox-4x32-10: (seed, stream) key, 64-bit counter -> 4 uniform u32.  Stateless, so a backward pass
// regenerates the forward mask from (seed, stream, element index) instead of storing it.
__device__ __forceinline__ uint4 philox4(uint64_t seed, uint64_t ctr, uint32_t stream) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream, c3 = 0x9E3779B9u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// keep/scale factor for element idx: 0 with prob p, else 1/(1-p)
__device__ __forceinline__ float dropout_factor(uint64_t seed, uint32_t stream, uint64_t idx, float p) {
    const uint4 r = philox4(seed, idx >> 2, stream);
    const uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
    return u32_to_unit(w) < p ? 0.0f : 1.0f / (1.0f - p);
}
