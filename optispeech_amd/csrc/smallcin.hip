// Direct (VALU) kernels for the Cin = 1 first layers of the discriminators: DiscriminatorP convs[0] (1 -> 32, k(5,1),
// _discriminators.py:53) and DiscriminatorR convs[0] (1 -> 64, k(7,5), _discriminators.py:155).  With one input channel
// the contraction depth is only `taps` (5 / 35): a GEMM tile would be > 90 % padding, and the layer is HBM-bound
// (it writes 32-64 channels per input sample).  One lane per output channel, the input sample is wave-uniform.
//   forward : y[m, n] = lrelu(b[n] + sum_j w[n, j] * x[in(m, j)])                 algorithmic bytes/row: Cout*2 (bf16 out)
//   wgrad   : dw[n, j] += sum_m dy[m, n] * x[in(m, j)],  db[n] += sum_m dy[m, n]   bytes/row: Cout*2 read
// Row geometry is the 2-D map of gemm_bf16.hip: m -> (u, th, tw); tap j -> (kh, kw);
//   in = ((u*Hin + th*sh + kh - ph) * Win + tw*sw + kw - pw), zero outside [0,Hin) x [0,Win).
#include "osp_common.h"

#define SC_MAXTAPS 40

struct SmallCin {
    const float* x; const void* y; int y_bf16; const float* w; const float* b; float* dw; float* db;
    int M, Trows, Wrows, Hin, Win, Cout, taps, KW, sh, sw, ph, pw; float slope; int lrelu;
};

// Tap samples of a row: lane j (< taps) of each Cout-lane group fetches x for tap j once; the FMA loop then reads it
// with a lane broadcast (the sample is uniform over the group's output channels).
__device__ __forceinline__ float sc_tap_sample(const SmallCin& p, int64_t m, int jl, int dh, int dw) {
    if (m >= p.M || jl >= p.taps) return 0.f;
    const int u = (int)(m / p.Trows), t = (int)(m - (int64_t)u * p.Trows), th = t / p.Wrows, tw = t - th * p.Wrows;
    const int hh = th * p.sh + dh, ww = tw * p.sw + dw;
    if (hh < 0 || hh >= p.Hin || ww < 0 || ww >= p.Win) return 0.f;
    return p.x[((int64_t)u * p.Hin + hh) * p.Win + ww];
}

// Broadcast of tap sample j inside a Cout-lane group.  With one row per wave (Cout == 64) the source lane is a
// compile-time constant -> v_readlane_b32 (scalar broadcast, no LDS traffic); otherwise ds_bpermute via __shfl.
template <int J>
__device__ __forceinline__ float sc_bcast(float xs, int gbase, bool one_row) {
    if (one_row) return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xs), J));
    return __shfl(xs, gbase + J, 64);
}
template <int J, int JMAX>
struct ScFma {
    static __device__ __forceinline__ void fwd(const float (&w)[SC_MAXTAPS], float xs, int gbase, bool one_row, int taps, float& acc) {
        if (J < taps) acc = fmaf(w[J], sc_bcast<J>(xs, gbase, one_row), acc);
        ScFma<J + 1, JMAX>::fwd(w, xs, gbase, one_row, taps, acc);
    }
    static __device__ __forceinline__ void bwd(float (&acc)[SC_MAXTAPS], float g, float xs, int gbase, bool one_row, int taps) {
        if (J < taps) acc[J] = fmaf(g, sc_bcast<J>(xs, gbase, one_row), acc[J]);
        ScFma<J + 1, JMAX>::bwd(acc, g, xs, gbase, one_row, taps);
    }
};
template <int JMAX>
struct ScFma<JMAX, JMAX> {
    static __device__ __forceinline__ void fwd(const float (&)[SC_MAXTAPS], float, int, bool, int, float&) {}
    static __device__ __forceinline__ void bwd(float (&)[SC_MAXTAPS], float, float, int, bool, int) {}
};

// rows per wave iteration RPW = 64 / Cout (Cout in {16, 32, 64}); requires taps <= Cout
__global__ __launch_bounds__(256) void smallcin_fwd_kernel(SmallCin p) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / p.Cout, n = lane % p.Cout, sub = lane / p.Cout, gbase = sub * p.Cout;
    float w[SC_MAXTAPS];
#pragma unroll
    for (int j = 0; j < SC_MAXTAPS; ++j) w[j] = j < p.taps ? p.w[n * p.taps + j] : 0.f;
    const float bias = p.b ? p.b[n] : 0.f;
    const int kh = n / p.KW, dh = kh - p.ph, dw = (n - kh * p.KW) - p.pw;      // this lane's tap (n doubles as tap id)
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t m0 = wave_id * rpw; m0 < p.M; m0 += nwaves * rpw) {
        const int64_t m = m0 + sub;
        const float xs = sc_tap_sample(p, m, n, dh, dw);
        float acc = bias;
        ScFma<0, SC_MAXTAPS>::fwd(w, xs, gbase, rpw == 1, p.taps, acc);
        if (m >= p.M) continue;
        if (p.lrelu) acc = acc > 0.f ? acc : acc * p.slope;
        if (p.y_bf16) reinterpret_cast<__bf16*>(const_cast<void*>(p.y))[m * p.Cout + n] = (__bf16)acc;
        else reinterpret_cast<float*>(const_cast<void*>(p.y))[m * p.Cout + n] = acc;
    }
}

__global__ __launch_bounds__(256) void smallcin_wgrad_kernel(SmallCin p) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rpw = 64 / p.Cout, n = lane % p.Cout, sub = lane / p.Cout, gbase = sub * p.Cout;
    float acc[SC_MAXTAPS];
#pragma unroll
    for (int j = 0; j < SC_MAXTAPS; ++j) acc[j] = 0.f;
    float bsum = 0.f;
    const int kh = n / p.KW, dh = kh - p.ph, dw = (n - kh * p.KW) - p.pw;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t m0 = wave_id * rpw; m0 < p.M; m0 += nwaves * rpw) {
        const int64_t m = m0 + sub;
        const float xs = sc_tap_sample(p, m, n, dh, dw);
        float g = 0.f;
        if (m < p.M)
            g = p.y_bf16 ? __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(p.y)[m * p.Cout + n]) << 16)
                         : reinterpret_cast<const float*>(p.y)[m * p.Cout + n];
        bsum += g;
        ScFma<0, SC_MAXTAPS>::bwd(acc, g, xs, gbase, rpw == 1, p.taps);
    }
    // combine the `rpw` sub-rows of a wave (lanes n, n+Cout, ...), then the 4 waves, then one atomic per (n, j)
    for (int j = 0; j <= p.taps; ++j) {
        float v = 0.f;
        if (j < p.taps) {
#pragma unroll
            for (int q = 0; q < SC_MAXTAPS; ++q) v = (q == j) ? acc[q] : v;
        } else v = bsum;
        for (int o = p.Cout; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        red[wv][lane] = v;
        __syncthreads();
        if (threadIdx.x < p.Cout) {
            const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (j < p.taps) atomicAdd(p.dw + threadIdx.x * p.taps + j, s);
            else if (p.db) atomicAdd(p.db + threadIdx.x, s);
        }
    }
}

static int smallcin_fill(SmallCin& p, int64_t M, int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t Cout,
                         int64_t taps, int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw) {
    if (!(Cout == 16 || Cout == 32 || Cout == 64) || taps < 1 || taps > SC_MAXTAPS || taps > Cout || taps % KW != 0 || M <= 0 ||
        M % Trows != 0 || Trows % Wrows != 0)
        return 0;
    p.M = (int)M; p.Trows = (int)Trows; p.Wrows = (int)Wrows; p.Hin = (int)Hin; p.Win = (int)Win; p.Cout = (int)Cout;
    p.taps = (int)taps; p.KW = (int)KW; p.sh = (int)sh; p.sw = (int)sw; p.ph = (int)ph; p.pw = (int)pw;
    return 1;
}

// x: (U, Hin, Win) f32 single-channel input; w: (Cout, taps) f32; y: (M, Cout) f32|bf16
extern "C" int osp_smallcin_conv_fwd(const float* x, const float* w, const float* b, void* y, int64_t y_bf16, int64_t M,
                                     int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t Cout, int64_t taps,
                                     int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t lrelu, float slope,
                                     hipStream_t stream) {
    OSP_CHECK_ARG(x && w && y, "null operand");
    SmallCin p;
    OSP_CHECK_ARG(smallcin_fill(p, M, Trows, Wrows, Hin, Win, Cout, taps, KW, sh, sw, ph, pw), "unsupported small-Cin geometry");
    p.x = x; p.w = w; p.b = b; p.y = y; p.y_bf16 = (int)y_bf16; p.dw = nullptr; p.db = nullptr; p.lrelu = (int)lrelu; p.slope = slope;
    const int64_t blocks = cdiv(M, 4 * (64 / Cout) * 8);
    hipLaunchKernelGGL(smallcin_fwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// dw (Cout, taps) and db (Cout) are accumulated (f32 atomics).
extern "C" int osp_smallcin_conv_wgrad(const float* x, const void* dy, int64_t y_bf16, float* dw, float* db, int64_t M,
                                       int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t Cout, int64_t taps,
                                       int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw, hipStream_t stream) {
    OSP_CHECK_ARG(x && dy && dw, "null operand");
    SmallCin p;
    OSP_CHECK_ARG(smallcin_fill(p, M, Trows, Wrows, Hin, Win, Cout, taps, KW, sh, sw, ph, pw), "unsupported small-Cin geometry");
    p.x = x; p.w = nullptr; p.b = nullptr; p.y = dy; p.y_bf16 = (int)y_bf16; p.dw = dw; p.db = db; p.lrelu = 0; p.slope = 0.f;
    const int64_t blocks = cdiv(M, 4 * (64 / Cout) * 64);
    hipLaunchKernelGGL(smallcin_wgrad_kernel, dim3((unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024)), dim3(256), 0, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
