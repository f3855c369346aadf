// HBM-bound halves of the ConvNeXt block and the LayerNorm family (reference rows A1a, A2, A4, A11):
//   dwconv7 + bias + LayerNorm(C) fused forward      <- generator/modules/convnext.py:36-38
//   LayerNorm forward / backward (eps argument)       <- convnext.py:85,102; modules/layers.py:26-45; wavenext:84
//   depthwise-conv backward (dx, dw, db)              <- autograd of convnext.py:36
//
// Layout: channels-last (B, T, C) f32, C % 4 == 0, C <= 1024.  One 64-lane wavefront owns one frame at a
// time: lane l holds channels {256k + 4l .. 256k + 4l + 3}, so every global access is a coalesced float4 row
// segment (1 KiB per wave instruction) and the LayerNorm statistics are intra-wave shuffles (no LDS, no barrier).
// A wave walks a run of FRAMES consecutive frames of one utterance with the 7-row window in registers, so each
// input row is fetched (7 + FRAMES - 1) / FRAMES times (from L2) instead of 7.
// Algorithmic HBM bytes per frame: forward 2*C*4 (+C*4 when xhat is saved for training); see DESIGN.md.
#include "osp_common.h"
#include <stdlib.h>

#define FRAMES 8
#define MAXCH 4   // chunks of 256 channels per lane => C <= 1024

// streaming stores: the activations these kernels write are consumed by a LATER kernel, never re-read by this one -- a
// non-temporal store keeps them from evicting the halo rows / taps the neighbouring waves still want from L2
#ifndef OSP_NT_STORES
#define OSP_NT_STORES 1
#endif
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
#if OSP_NT_STORES
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(uint2* p, uint2 v) {
#if OSP_NT_STORES
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
#else
    *p = v;
#endif
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float f4sum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// ------------------------------------------------------------------------------------------------------------
// Round 2: a wave owns a run of FR consecutive frames and requests ALL FR + 6 input rows of the run before it computes
// anything (FR + 6 independent 1 KiB loads in flight per wave instead of one dependent load per frame: the round-1 loop
// used the row it had just requested and exposed a full memory latency per frame -- 39 % of the HBM roofline at the
// decoder shape).  The 6 halo rows are shared with the neighbouring runs and come from L2; with FR = 16 the
// request amplification is 22 / 16.  Everything after the loads runs out of registers: 7 FMAs per channel, two wave
// reductions, one or two 1 KiB stores per frame.
// launch bound: NCH = 1, FR = 8 needs 130 VGPRs unconstrained = 3 workgroups / CU = 768 resident, and the decoder shape has 800:
// a second round of 32 workgroups ran alone at the end.  Capped at 128 VGPRs (4 / CU) the whole grid is one round.
template <int NCH, int FR>
__global__ __launch_bounds__(256, (NCH == 1 && FR <= 8) ? 4 : 1) void dwconv7_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ dw,
                                                             const float* __restrict__ dwb, const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, float eps,
                                                             void* __restrict__ h, int h_bf16, float* __restrict__ xhat,
                                                             float* __restrict__ rstd_out, int B, int T, int C, int runs_per_utt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int run = blockIdx.x * 4 + wave;
    if (run >= B * runs_per_utt) return;
    // run r of an utterance owns frames [r T / R, (r + 1) T / R): at most FR of them (the host picks R >= T / FR), ragged by one
    // when R does not divide T -- R is chosen so that B * R fills whole rounds of resident waves (see the launcher)
    const int b = run / runs_per_utt, rr = run - b * runs_per_utt;
    const int t0 = (int)((int64_t)rr * T / runs_per_utt), t1 = (int)((int64_t)(rr + 1) * T / runs_per_utt);
    const float* xb = x + (int64_t)b * T * C;
    bool act[NCH];
    float4 rows[NCH][FR + 6];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = k * 256 + lane * 4;
        act[k] = ch < C;
#pragma unroll
        for (int r = 0; r < FR + 6; ++r) {
            const int t = t0 + r - 3;
            const bool ok = act[k] && t >= 0 && t < T;
            const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)(ok ? t : t0) * C + (act[k] ? ch : 0));   // unconditional load, index select
            rows[k][r] = ok ? v : f4zero();
        }
    }
    float4 w[NCH][7], bias[NCH], gw[NCH], gb[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = act[k] ? k * 256 + lane * 4 : 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) w[k][j] = *reinterpret_cast<const float4*>(dw + (int64_t)j * C + ch);
        bias[k] = *reinterpret_cast<const float4*>(dwb + ch);
        gw[k] = *reinterpret_cast<const float4*>(lnw + ch);
        gb[k] = *reinterpret_cast<const float4*>(lnb + ch);
    }
    const float invC = 1.0f / (float)C;
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int t = t0 + f;
        if (t < t1) {                                           // wave-uniform
            float4 c[NCH];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                float4 a = act[k] ? bias[k] : f4zero();
#pragma unroll
                for (int j = 0; j < 7; ++j) a = f4fma(w[k][j], rows[k][f + j], a);
                c[k] = act[k] ? a : f4zero();
                s += f4sum(c[k]);
            }
            const float mean = wave_sum(s) * invC;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (act[k]) {
                    c[k].x -= mean; c[k].y -= mean; c[k].z -= mean; c[k].w -= mean;
                    v += c[k].x * c[k].x + c[k].y * c[k].y + c[k].z * c[k].z + c[k].w * c[k].w;
                }
            const float rstd = rsqrtf(wave_sum(v) * invC + eps);
            const int64_t row = ((int64_t)b * T + t) * C;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (act[k]) {
                    const float4 n = make_float4(c[k].x * rstd, c[k].y * rstd, c[k].z * rstd, c[k].w * rstd);
                    const int ch = k * 256 + lane * 4;
                    if (xhat) st_stream(reinterpret_cast<float4*>(xhat + row + ch), n);
                    const float4 o = f4fma(n, gw[k], gb[k]);
                    if (h_bf16) {                               // the consumer is the bf16 GEMM: half the bytes, no cast launch later
                        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
                        v2 p0, p1; p0[0] = (__bf16)o.x; p0[1] = (__bf16)o.y; p1[0] = (__bf16)o.z; p1[1] = (__bf16)o.w;
                        st_stream(reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(h) + row + ch),
                                  make_uint2(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)));
                    } else {
                        st_stream(reinterpret_cast<float4*>(reinterpret_cast<float*>(h) + row + ch), o);
                    }
                }
            if (rstd_out && lane == 0) rstd_out[(int64_t)b * T + t] = rstd;
        }
    }
}

// C = 384 (the WaveNeXt trunk) on the kernel above is two 256-channel chunks with the upper half of the second one idle: 25 % of the
// lanes' registers, loads and FMAs are spent on nothing (the loads are unconditional).  Here a lane owns SIX channels -- 4 lane .. + 3
// of the first 256 as a float4 and 256 + 2 lane, + 1 as a float2 -- so every lane is busy on every row, a row is 6 registers instead
// of 8, and runs of FR = 16 frames (22 rows requested up front, request amplification 22 / 16 instead of 14 / 8) fit.  Round 6.
__device__ __forceinline__ float2 f2fma(float2 a, float2 b, float2 c) { return make_float2(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }
template <int FR>
__global__ __launch_bounds__(256) void dwconv7_ln_fwd_c384_kernel(const float* __restrict__ x, const float* __restrict__ dw,
                                                                  const float* __restrict__ dwb, const float* __restrict__ lnw,
                                                                  const float* __restrict__ lnb, float eps,
                                                                  void* __restrict__ h, int h_bf16, float* __restrict__ xhat,
                                                                  float* __restrict__ rstd_out, int B, int T, int runs_per_utt) {
    constexpr int C = 384;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int run = blockIdx.x * 4 + wave;
    if (run >= B * runs_per_utt) return;
    const int b = run / runs_per_utt, rr = run - b * runs_per_utt;
    const int t0 = (int)((int64_t)rr * T / runs_per_utt), t1 = (int)((int64_t)(rr + 1) * T / runs_per_utt);
    const float* xb = x + (int64_t)b * T * C;
    const int ca = 4 * lane, cb = 256 + 2 * lane;
    float4 ra[FR + 6];
    float2 rb[FR + 6];
#pragma unroll
    for (int r = 0; r < FR + 6; ++r) {
        const int t = t0 + r - 3;
        const bool ok = t >= 0 && t < T;
        const float* row = xb + (int64_t)(ok ? t : t0) * C;                      // unconditional loads, index select
        const float4 va = *reinterpret_cast<const float4*>(row + ca);
        const float2 vb = *reinterpret_cast<const float2*>(row + cb);
        ra[r] = ok ? va : f4zero();
        rb[r] = ok ? vb : make_float2(0.f, 0.f);
    }
    float4 wa[7]; float2 wb[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        wa[j] = *reinterpret_cast<const float4*>(dw + (int64_t)j * C + ca);
        wb[j] = *reinterpret_cast<const float2*>(dw + (int64_t)j * C + cb);
    }
    const float4 ba = *reinterpret_cast<const float4*>(dwb + ca), gwa = *reinterpret_cast<const float4*>(lnw + ca), gba = *reinterpret_cast<const float4*>(lnb + ca);
    const float2 bb = *reinterpret_cast<const float2*>(dwb + cb), gwb = *reinterpret_cast<const float2*>(lnw + cb), gbb = *reinterpret_cast<const float2*>(lnb + cb);
    const float invC = 1.0f / (float)C;
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int t = t0 + f;
        if (t < t1) {                                           // wave-uniform
            float4 a = ba; float2 c2 = bb;
#pragma unroll
            for (int j = 0; j < 7; ++j) { a = f4fma(wa[j], ra[f + j], a); c2 = f2fma(wb[j], rb[f + j], c2); }
            const float mean = wave_sum(f4sum(a) + (c2.x + c2.y)) * invC;
            a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean; c2.x -= mean; c2.y -= mean;
            const float v = (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w) + (c2.x * c2.x + c2.y * c2.y);
            const float rstd = rsqrtf(wave_sum(v) * invC + eps);
            const int64_t row = ((int64_t)b * T + t) * C;
            const float4 na = make_float4(a.x * rstd, a.y * rstd, a.z * rstd, a.w * rstd);
            const float2 nb = make_float2(c2.x * rstd, c2.y * rstd);
            if (xhat) {
                st_stream(reinterpret_cast<float4*>(xhat + row + ca), na);
                *reinterpret_cast<float2*>(xhat + row + cb) = nb;
            }
            const float4 oa = f4fma(na, gwa, gba);
            const float2 ob = f2fma(nb, gwb, gbb);
            if (h_bf16) {
                typedef __bf16 v2 __attribute__((ext_vector_type(2)));
                v2 p0, p1, p2; p0[0] = (__bf16)oa.x; p0[1] = (__bf16)oa.y; p1[0] = (__bf16)oa.z; p1[1] = (__bf16)oa.w; p2[0] = (__bf16)ob.x; p2[1] = (__bf16)ob.y;
                unsigned short* hp = reinterpret_cast<unsigned short*>(h) + row;
                st_stream(reinterpret_cast<uint2*>(hp + ca), make_uint2(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)));
                *reinterpret_cast<unsigned*>(hp + cb) = __builtin_bit_cast(unsigned, p2);
            } else {
                float* hp = reinterpret_cast<float*>(h) + row;
                st_stream(reinterpret_cast<float4*>(hp + ca), oa);
                *reinterpret_cast<float2*>(hp + cb) = ob;
            }
            if (rstd_out && lane == 0) rstd_out[(int64_t)b * T + t] = rstd;
        }
    }
}

// h_bf16 != 0: h is written as bf16 (the operand the bf16 pointwise GEMM reads); xhat / rstd stay f32.
extern "C" int osp_dwconv7_ln_fwd(const float* x, const float* dw, const float* dwb, const float* lnw,
                                  const float* lnb, float eps, void* h, int64_t h_bf16, float* xhat, float* rstd, int64_t B,
                                  int64_t T, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(x && dw && dwb && lnw && lnb && h, "null operand");
    OSP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXCH, "C must be a multiple of 4, <= 1024");
    const int nch = (int)cdiv(C, 256);
    // frames per wave: 8 (one or two 256-channel chunks; 16 halves the occupancy: 174 VGPRs), 4 beyond.
    // Runs per utterance R >= ceil(T / FR).  The C <= 256 kernel keeps 4 workgroups = 16 waves per CU resident (4096 waves on the
    // chip): when B * ceil(T / FR) is within reach of a whole number of such rounds, R is raised so that B * R IS one -- the
    // decoder shape (32 x 800, FR 8) has 3200 runs = 800 workgroups = 3.1 per CU, i.e. the CUs holding four set the time; with
    // R = 128 (runs of 6 or 7 frames) every CU holds exactly four (tools/lndw_probe.py: 15.0 -> 14.5 us).
    auto pick_runs = [&](int fr) -> int {
        const int64_t rmin = cdiv(T, fr);
        static int balance = -1;
        if (balance < 0) { balance = 1; }
        if (!balance || nch != 1) return (int)rmin;
        const int64_t resident = 4096, total = B * rmin;
        const int64_t rounds = cdiv(total, resident);
        const int64_t r = (rounds * resident) / B;                       // largest R with B * R <= rounds * resident
        return (int)((r >= rmin && r * 10 <= rmin * 13 && r <= T) ? r : rmin);   // at most 30 % more (shorter) runs
    };
#define L(N, F) do { const int R_ = pick_runs(F); hipLaunchKernelGGL((dwconv7_ln_fwd_kernel<N, F>), dim3((unsigned)cdiv(B * (int64_t)R_, 4)), dim3(256), 0, stream, x, dw, dwb, lnw, lnb, eps, h, (int)h_bf16, xhat, rstd, (int)B, (int)T, (int)C, R_); } while (0)
    static int fr1 = -1;
    if (fr1 < 0) { fr1 = 8; }      // measured at 32 x 800 x 256 (tools/dwconv_probe.py (git history)): FR 4 / 8 / 16 = 18.7 / 17.2 / 21.5 us with xhat saved
    // C = 384: six channels per lane (every lane busy), runs of 16 frames from 16 k rows (64 x 772 frames: 36.9 -> 32.5 us, 32 x 800:
    // 20.7 -> 16.1 us), of 8 below (32 x 64: 8.4 -> 8.2 us; runs of 16 leave SIMDs empty there: 9.9 us).  OSP_DWLN_C384=0: the
    // two-chunk kernel (A/B runs), = 8 / 12 / 16: that run length.
    // Taken by the no-grad form (nothing saved: synthesise, the tape-less decoder); a forward that saves x-hat / rstd for a backward stays
    // on the two-chunk kernel unless the switch names a run length: the two agree to f32 rounding, and the vocoder's gradient norms at
    // the benchmark size move by ~1 % under ANY such perturbation (bf16 discriminators behind it; tests/test_gpu_fullsize_golden.py
    // bounds them at 6 % with 5.1 % measured) -- the training step's numbers stay the ones every fixture was accepted on.
    if (C == 384) {
        const char* e = getenv("OSP_DWLN_C384");
        const int sel = e ? atoi(e) : (xhat ? 0 : (B * T >= 16384 && T >= 16 ? 16 : 8));
#define L6(F) do { const int R_ = (int)cdiv(T, F); hipLaunchKernelGGL((dwconv7_ln_fwd_c384_kernel<F>), dim3((unsigned)cdiv(B * (int64_t)R_, 4)), dim3(256), 0, stream, x, dw, dwb, lnw, lnb, eps, h, (int)h_bf16, xhat, rstd, (int)B, (int)T, R_); } while (0)
        if (sel == 16) { L6(16); OSP_LAUNCH_CHECK(); return OSP_OK; }
        if (sel == 12) { L6(12); OSP_LAUNCH_CHECK(); return OSP_OK; }
        if (sel == 8) { L6(8); OSP_LAUNCH_CHECK(); return OSP_OK; }
#undef L6
    }
    if (nch == 1) { if (fr1 == 8) L(1, 8); else if (fr1 == 4) L(1, 4); else L(1, 16); } else if (nch == 2) L(2, 8); else if (nch == 3) L(3, 4); else L(4, 4);
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm forward over the last dim.  Optional fused tail:  y = (LN(x)*w + b) * dropout * rowmask.
// mean/rstd are saved (2 floats per row) so the backward recomputes xhat from x.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            const float* __restrict__ rowmask, float drop_p,
                                                            uint64_t seed, const int64_t* __restrict__ seed_dev, uint32_t stream_id, int64_t rows, int C) {
    if (seed_dev) seed += (uint64_t)*seed_dev;      // per-step seed kept in device memory (hipGraph replay)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invC = 1.0f / (float)C;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        float4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = k * 256 + lane * 4;
            v[k] = ch < C ? *reinterpret_cast<const float4*>(x + row * C + ch) : f4zero();
            s += f4sum(v[k]);
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (k * 256 + lane * 4 < C) {
                v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
                q += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
            }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
        const float rm = rowmask ? rowmask[row] : 1.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = k * 256 + lane * 4;
            if (ch < C) {
                const float4 gw = *reinterpret_cast<const float4*>(w + ch), gb = *reinterpret_cast<const float4*>(b + ch);
                float4 o = make_float4(fmaf(v[k].x * rstd, gw.x, gb.x), fmaf(v[k].y * rstd, gw.y, gb.y),
                                       fmaf(v[k].z * rstd, gw.z, gb.z), fmaf(v[k].w * rstd, gw.w, gb.w));
                if (drop_p > 0.f) {
                    const uint64_t e = (uint64_t)row * C + ch;   // multiple of 4: one Philox call covers the float4
                    const uint4 r = philox4(seed, e >> 2, stream_id);
                    const float keep = 1.0f / (1.0f - drop_p);
                    o.x *= u32_to_unit(r.x) < drop_p ? 0.f : keep;
                    o.y *= u32_to_unit(r.y) < drop_p ? 0.f : keep;
                    o.z *= u32_to_unit(r.z) < drop_p ? 0.f : keep;
                    o.w *= u32_to_unit(r.w) < drop_p ? 0.f : keep;
                }
                o.x *= rm; o.y *= rm; o.z *= rm; o.w *= rm;
                st_stream(reinterpret_cast<float4*>(y + row * C + ch), o);
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

extern "C" int osp_layernorm_fwd(const float* x, const float* w, const float* b, float eps, float* y, float* mean,
                                 float* rstd, const float* rowmask, float drop_p, int64_t seed, const int64_t* seed_dev, int64_t stream_id,
                                 int64_t rows, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(x && w && b && y, "null operand");
    OSP_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXCH, "C must be a multiple of 4, <= 1024");
    OSP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "dropout rate");
    dim3 grid((unsigned)(cdiv(rows, 4) < 4096 ? cdiv(rows, 4) : 4096));
    const int nch = (int)cdiv(C, 256);
#define L(N) hipLaunchKernelGGL((layernorm_fwd_kernel<N>), grid, dim3(256), 0, stream, x, w, b, eps, y, mean, rstd, rowmask, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id, rows, (int)C)
    if (nch == 1) L(1); else if (nch == 2) L(2); else if (nch == 3) L(3); else L(4);
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Two-stage reduction of the parameter-gradient rows of the backward kernels below (round 3).
// Round 2 let every persistent workgroup add its partial rows into the gradient arena with device-scope float atomics: 256
// workgroups x 512 (LayerNorm) / 2 560 (fused LN + dwconv) addresses.  The L2s of the 8 XCDs are not coherent with each other,
// so those atomics execute at the memory side: 8-12 us of a 25-41 us kernel, and -- worse -- the reason the grids were capped
// at one workgroup per CU (4-8 waves per CU: a streaming kernel needs several times that to cover HBM latency).
// Now: stage 1, every workgroup STORES its partial rows (plain streaming stores) to ws[workgroup][ncols]; stage 2, a small
// kernel sums 64 partials per workgroup and column with 16-byte loads and adds the result to the destination rows with one
// atomic per column and 64 partials (ncols * parts / 64 atomics in total: a few thousand instead of 131 k - 655 k).  The kernel
// boundary is what makes stage 1's stores visible: no fences, no tickets.
struct PartialDst { float* p[4]; int n[4]; };           // columns [0, n0) -> p[0], [n0, n0 + n1) -> p[1], ...
// Stage 2: grid (column blocks of 64, groups of RED_PARTS partial rows); a workgroup = 16 column quads x 16 part lanes, each thread
// sums RED_PARTS / 16 partials of its quad with 16-byte loads, the part lanes meet in LDS, one atomic per column and group.
#define RED_PARTS 32
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ ws, int nparts, int ncols, PartialDst d) {
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int p0 = blockIdx.y * RED_PARTS, p1 = min(nparts, p0 + RED_PARTS);
    float4 acc = f4zero();
    if (c < ncols)
        for (int q = p0 + pl; q < p1; q += 16) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)q * ncols + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    red[pl][cq] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int q4 = threadIdx.x >> 2, e = threadIdx.x & 3, col = blockIdx.x * 64 + threadIdx.x;
        if (col < ncols) {
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float4 v = red[k][q4]; r += e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
            int seg = 0, off = 0;
            while (seg < 3 && col >= off + d.n[seg]) { off += d.n[seg]; ++seg; }
            if (d.p[seg]) atomicAdd(d.p[seg] + (col - off), r);
        }
    }
}
static void launch_reduce_partials(const float* ws, int nparts, int ncols, const PartialDst& d, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(ncols, 64), (unsigned)cdiv(nparts, RED_PARTS)), dim3(256), 0, stream,
                       ws, nparts, ncols, d);
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  g = dy * dropout * rowmask;  xhat = xin (mean == null) or (xin - mean)*rstd;
//   dx = rstd * (g*w - mean_C(g*w) - xhat * mean_C(g*w*xhat))  [* (relu_src > 0) if given]
//   dlnw += sum_rows g * xhat ; dlnb += sum_rows g
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xin,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ w, const float* __restrict__ relu_src,
                                                            const float* __restrict__ rowmask, float drop_p, uint64_t seed,
                                                            const int64_t* __restrict__ seed_dev, uint32_t stream_id, float* __restrict__ dx,
                                                            float* __restrict__ dlnw, float* __restrict__ dlnb,
                                                            float* __restrict__ ws, int64_t rows, int C) {
    __shared__ float red[2][4][256 * NCH];   // [w|b][wave][channel-in-lane-order]
    if (seed_dev) seed += (uint64_t)*seed_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invC = 1.0f / (float)C;
    float4 aw[NCH], ab[NCH], gw[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        aw[k] = f4zero(); ab[k] = f4zero();
        const int ch = k * 256 + lane * 4;
        gw[k] = ch < C ? *reinterpret_cast<const float4*>(w + ch) : f4zero();
    }
    // two rows per trip, both requested before either is reduced: a wave otherwise exposes a full memory latency per row
    // (one dependent load -> two wave reductions -> store chain at a time)
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t row0 = (int64_t)blockIdx.x * 4 + wave; row0 < rows; row0 += 2 * stride) {
        float4 dv[2][NCH], xv_[2][NCH], rl[2][NCH];
        float mu_[2], rs_[2], rm_[2];
        bool live[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t row = row0 + q * stride;
            live[q] = row < rows;
            const int64_t rsafe = live[q] ? row : row0;
            mu_[q] = mean ? mean[rsafe] : 0.f; rs_[q] = rstd[rsafe]; rm_[q] = rowmask ? rowmask[rsafe] : 1.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int ch = k * 256 + lane * 4;
                const int cs = ch < C ? ch : 0;
                dv[q][k] = *reinterpret_cast<const float4*>(dy + rsafe * C + cs);
                xv_[q][k] = *reinterpret_cast<const float4*>(xin + rsafe * C + cs);
                rl[q][k] = relu_src ? *reinterpret_cast<const float4*>(relu_src + rsafe * C + cs) : f4zero();
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!live[q]) continue;                              // wave-uniform
            const int64_t row = row0 + q * stride;
            const float mu = mu_[q], rs = rs_[q], rm = rm_[q];
            float4 g[NCH], xh[NCH];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int ch = k * 256 + lane * 4;
                if (ch < C) {
                    float4 d = dv[q][k];
                    float4 xv = xv_[q][k];
                    if (mean) { xv.x = (xv.x - mu) * rs; xv.y = (xv.y - mu) * rs; xv.z = (xv.z - mu) * rs; xv.w = (xv.w - mu) * rs; }
                    if (drop_p > 0.f) {
                        const uint64_t e = (uint64_t)row * C + ch;
                        const uint4 r = philox4(seed, e >> 2, stream_id);
                        const float keep = 1.0f / (1.0f - drop_p);
                        d.x *= u32_to_unit(r.x) < drop_p ? 0.f : keep;
                        d.y *= u32_to_unit(r.y) < drop_p ? 0.f : keep;
                        d.z *= u32_to_unit(r.z) < drop_p ? 0.f : keep;
                        d.w *= u32_to_unit(r.w) < drop_p ? 0.f : keep;
                    }
                    d.x *= rm; d.y *= rm; d.z *= rm; d.w *= rm;
                    aw[k] = f4fma(d, xv, aw[k]);
                    ab[k].x += d.x; ab[k].y += d.y; ab[k].z += d.z; ab[k].w += d.w;
                    d.x *= gw[k].x; d.y *= gw[k].y; d.z *= gw[k].z; d.w *= gw[k].w;
                    g[k] = d; xh[k] = xv;
                    s1 += f4sum(d);
                    s2 += d.x * xv.x + d.y * xv.y + d.z * xv.z + d.w * xv.w;
                } else { g[k] = f4zero(); xh[k] = f4zero(); }
            }
            const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int ch = k * 256 + lane * 4;
                if (ch < C) {
                    float4 o = make_float4(rs * (g[k].x - m1 - xh[k].x * m2), rs * (g[k].y - m1 - xh[k].y * m2),
                                           rs * (g[k].z - m1 - xh[k].z * m2), rs * (g[k].w - m1 - xh[k].w * m2));
                    if (relu_src) {
                        const float4 r = rl[q][k];
                        o.x = r.x > 0.f ? o.x : 0.f; o.y = r.y > 0.f ? o.y : 0.f;
                        o.z = r.z > 0.f ? o.z : 0.f; o.w = r.w > 0.f ? o.w : 0.f;
                    }
                    st_stream(reinterpret_cast<float4*>(dx + row * C + ch), o);
                }
            }
        }
    }
    if (!dlnw) return;
    // block-level combine of the per-wave parameter-gradient partials, then one atomic per channel per block
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        *reinterpret_cast<float4*>(&red[0][wave][k * 256 + lane * 4]) = aw[k];
        *reinterpret_cast<float4*>(&red[1][wave][k * 256 + lane * 4]) = ab[k];
    }
    __syncthreads();
    float* part = ws ? ws + (int64_t)blockIdx.x * 2 * C : nullptr;       // stage 1 of the two-stage reduction (see above)
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        const float a = red[0][0][ch] + red[0][1][ch] + red[0][2][ch] + red[0][3][ch];
        const float bsum = red[1][0][ch] + red[1][1][ch] + red[1][2][ch] + red[1][3][ch];
        if (part) { part[ch] = a; part[C + ch] = bsum; }
        else { atomicAdd(dlnw + ch, a); atomicAdd(dlnb + ch, bsum); }
    }
}

// ws (optional): ws_blocks * 2 * C floats of scratch for the two-stage parameter-gradient reduction; with it the grid is
// min(rows / (4 * FRAMES), ws_blocks) workgroups (many waves per CU), without it at most 256 (device-scope atomics per workgroup).
extern "C" int osp_layernorm_bwd(const float* dy, const float* xin, const float* mean, const float* rstd,
                                 const float* w, const float* relu_src, const float* rowmask, float drop_p,
                                 int64_t seed, const int64_t* seed_dev, int64_t stream_id, float* dx, float* dlnw, float* dlnb, int64_t rows,
                                 int64_t C, float* ws, int64_t ws_blocks, hipStream_t stream) {
    OSP_CHECK_ARG(dy && xin && rstd && w && dx, "null operand");
    OSP_CHECK_ARG((dlnw == nullptr) == (dlnb == nullptr), "dlnw/dlnb come together");
    OSP_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXCH, "C must be a multiple of 4, <= 1024");
    OSP_CHECK_ARG(!ws || ws_blocks > 0, "ws needs ws_blocks > 0");
    const int64_t blocks = cdiv(rows, 4 * FRAMES);   // FRAMES rows per wave
    static int64_t lnb_cap = 0;
    if (!lnb_cap) { lnb_cap = 256; }      // atomics path: tools/lndw_probe.py at 32 x 800 x 256, caps 256 / 512 / 1024 = 25.0 / 26.8 / 31.3 us
    const bool two_stage = ws && dlnw;
    const int64_t cap = two_stage ? ws_blocks : (dlnw ? lnb_cap : 4096);
    dim3 grid((unsigned)(blocks < cap ? blocks : cap));
    const int nch = (int)cdiv(C, 256);
#define L(N) hipLaunchKernelGGL((layernorm_bwd_kernel<N>), grid, dim3(256), 0, stream, dy, xin, mean, rstd, w, relu_src, rowmask, drop_p, (uint64_t)seed, seed_dev, (uint32_t)stream_id, dx, dlnw, dlnb, two_stage ? ws : nullptr, rows, (int)C)
    if (nch == 1) L(1); else if (nch == 2) L(2); else if (nch == 3) L(3); else L(4);
#undef L
    if (two_stage) {
        PartialDst d = {{dlnw, dlnb, nullptr, nullptr}, {(int)C, (int)C, 0, 0}};
        launch_reduce_partials(ws, (int)grid.x, 2 * (int)C, d, stream);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Depthwise conv backward:  dx[t] = dres[t]*dres_rowmask[t] + sum_j w[j] * dc[t - j + 3];  ddw[j] += sum_t dc[t] * x[t + j - 3];
// ddb += sum_t dc[t].  Same wave-per-run walk with two register windows (dc for dx, x for ddw).
template <int NCH>
__global__ __launch_bounds__(256) void dwconv7_bwd_kernel(const float* __restrict__ dc, const float* __restrict__ x,
                                                          const float* __restrict__ dw, const float* __restrict__ dres,
                                                          const float* __restrict__ dres_rowmask,
                                                          float* __restrict__ dx, float* __restrict__ ddw,
                                                          float* __restrict__ ddb, int B, int T, int C) {
    __shared__ float red[4][256 * NCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int runs_per_utt = (T + FRAMES - 1) / FRAMES;
    const int run = blockIdx.x * 4 + wave;
    const bool live = run < B * runs_per_utt;
    const int b = live ? run / runs_per_utt : 0, t0 = live ? (run - b * runs_per_utt) * FRAMES : 0;
    const float* xb = x + (int64_t)b * T * C;
    const float* db_ = dc + (int64_t)b * T * C;
    bool act[NCH];
    float4 w[NCH][7], gw[NCH][7], gb[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = k * 256 + lane * 4;
        act[k] = live && ch < C;
        gb[k] = f4zero();
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            gw[k][j] = f4zero();
            w[k][j] = act[k] ? *reinterpret_cast<const float4*>(dw + (int64_t)j * C + ch) : f4zero();
        }
    }
    auto ld = [&](const float* base, int k, int t) -> float4 {
        if (!act[k] || t < 0 || t >= T) return f4zero();
        return *reinterpret_cast<const float4*>(base + (int64_t)t * C + k * 256 + lane * 4);
    };
    float4 wx[NCH][7], wd[NCH][7];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int j = 0; j < 6; ++j) { wx[k][j + 1] = ld(xb, k, t0 + j - 3); wd[k][j + 1] = ld(db_, k, t0 + j - 3); }
    if (live)
        for (int f = 0; f < FRAMES; ++f) {
            const int t = t0 + f;
            if (t >= T) break;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { wx[k][j] = wx[k][j + 1]; wd[k][j] = wd[k][j + 1]; }
                wx[k][6] = ld(xb, k, t + 3);
                wd[k][6] = ld(db_, k, t + 3);
                if (!act[k]) continue;
                const int64_t off = ((int64_t)b * T + t) * C + k * 256 + lane * 4;
                float4 a = dres ? *reinterpret_cast<const float4*>(dres + off) : f4zero();
                if (dres && dres_rowmask) {
                    const float rm = dres_rowmask[(int64_t)b * T + t];
                    a.x *= rm; a.y *= rm; a.z *= rm; a.w *= rm;
                }
                const float4 d0 = wd[k][3];   // dc[t]
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    a = f4fma(w[k][j], wd[k][6 - j], a);          // dc[t - j + 3]
                    gw[k][j] = f4fma(d0, wx[k][j], gw[k][j]);     // x[t + j - 3]
                }
                gb[k].x += d0.x; gb[k].y += d0.y; gb[k].z += d0.z; gb[k].w += d0.w;
                st_stream(reinterpret_cast<float4*>(dx + off), a);
            }
        }
    if (!ddw) return;
    for (int j = 0; j < 8; ++j) {   // 7 taps + bias, one LDS pass each
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            *reinterpret_cast<float4*>(&red[wave][k * 256 + lane * 4]) = j < 7 ? gw[k][j] : gb[k];
        __syncthreads();
        for (int ch = threadIdx.x; ch < C; ch += 256) {
            const float s = red[0][ch] + red[1][ch] + red[2][ch] + red[3][ch];
            atomicAdd(j < 7 ? ddw + (int64_t)j * C + ch : ddb + ch, s);
        }
    }
}

extern "C" int osp_dwconv7_bwd(const float* dc, const float* x, const float* dw, const float* dres,
                               const float* dres_rowmask, float* dx, float* ddw, float* ddb, int64_t B, int64_t T, int64_t C, hipStream_t stream) {
    OSP_CHECK_ARG(dc && x && dw && dx, "null operand");
    OSP_CHECK_ARG((ddw == nullptr) == (ddb == nullptr), "ddw/ddb come together");
    OSP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXCH, "C must be a multiple of 4, <= 1024");
    const int64_t runs = B * cdiv(T, FRAMES);
    dim3 grid((unsigned)cdiv(runs, 4));
    const int nch = (int)cdiv(C, 256);
#define L(N) hipLaunchKernelGGL((dwconv7_bwd_kernel<N>), grid, dim3(256), 0, stream, dc, x, dw, dres, dres_rowmask, dx, ddw, ddb, (int)B, (int)T, (int)C)
    if (nch == 1) L(1); else if (nch == 2) L(2); else if (nch == 3) L(3); else L(4);
#undef L
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}


// ------------------------------------------------------------------------------------------------------------
// LayerNorm backward + depthwise-conv backward of the ConvNeXt block in ONE pass (C <= 256: one 256-channel chunk per lane).
//   dc[t]  = rstd[t] * (g - mean_C(g) - xhat[t] * mean_C(g * xhat[t])),  g = dh[t] * lnw          (LayerNorm backward, xhat saved)
//   dx[t]  = dres[t] * rowmask[t] + sum_j w[j] * dc[t - j + 3]                                     (depthwise conv backward + residual)
//   dlnw += sum_t dh[t] * xhat[t];  dlnb += sum_t dh[t];  ddw[j] += sum_t dc[t] * x[t + j - 3];  ddb += sum_t dc[t]
// The unfused pair writes dc (M x C f32) to HBM and reads it back through a 7-row window: 2 x 4 bytes per element of the
// 9 x 4 the pair moves (dh, xhat, dc | dc, x, dres, dx ...).  Here a wave owns a run of FR frames and rebuilds the dc rows of
// its FR + 6 window from dh / xhat (the 6 halo rows are shared with the neighbouring runs: L2 hits, two extra wave reductions
// each), so dc never leaves the registers.  Algorithmic HBM bytes per frame: (dh + xhat + x + dres) reads + dx write = 5 * C * 4.
// Loads are requested in two bursts (dh + xhat rows, then x + dres rows) to stay under 256 VGPRs with the 7 + 3 parameter-
// gradient accumulators live.
template <int FR, int NWV>
__global__ __launch_bounds__(64 * NWV, 8 / NWV) void ln_dwconv7_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ xhat,
                                                             const float* __restrict__ rstd, const float* __restrict__ lnw,
                                                             const float* __restrict__ x, const float* __restrict__ dw,
                                                             const float* __restrict__ dres, const float* __restrict__ dres_rowmask,
                                                             float* __restrict__ dx, float* __restrict__ dlnw, float* __restrict__ dlnb,
                                                             float* __restrict__ ddw, float* __restrict__ ddb, float* __restrict__ ws,
                                                             int B, int T, int C) {
    __shared__ float red[NWV][10][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int runs_per_utt = (T + FR - 1) / FR, nruns = B * runs_per_utt;
    const int ch = lane * 4;
    const bool chan = ch < C;
    const int chs = chan ? ch : 0;
    const float invC = 1.0f / (float)C;
    const float4 gw = chan ? *reinterpret_cast<const float4*>(lnw + chs) : f4zero();
    float4 gwt[7], gb = f4zero(), aw = f4zero(), ab = f4zero();
#pragma unroll
    for (int j = 0; j < 7; ++j) gwt[j] = f4zero();
    // persistent workgroups (the grid is capped at two per CU): the parameter-gradient partials stay in registers over all the
    // runs of a wave, so every channel takes one atomic per WORKGROUP, not per run.  Three load bursts per run, fenced against
    // each other (the scheduler otherwise hoists all of them to the top and the kernel spills): dh + xhat -> dc window;
    // dres + taps -> dx; x window -> tap gradients.
    for (int run = blockIdx.x * NWV + wave; run < nruns; run += gridDim.x * NWV) {
        const int b = run / runs_per_utt, t0 = (run - b * runs_per_utt) * FR;
        const int64_t base = (int64_t)b * T * C;
        float4 dc[FR + 6];
        {
            float4 xh[FR + 6];
            float rs[FR + 6];
#pragma unroll
            for (int r = 0; r < FR + 6; ++r) {
                const int t = t0 + r - 3;
                const bool in = t >= 0 && t < T;
                const int64_t off = base + (int64_t)(in ? t : t0) * C + chs;
                const float4 d = *reinterpret_cast<const float4*>(dh + off);
                const float4 v = *reinterpret_cast<const float4*>(xhat + off);
                rs[r] = in ? rstd[(int64_t)b * T + t] : 0.f;         // rows outside the utterance are the conv's zero padding
                dc[r] = (in && chan) ? d : f4zero();
                xh[r] = (in && chan) ? v : f4zero();
            }
#pragma unroll
            for (int r = 0; r < FR + 6; ++r) {
                const float4 d = dc[r], v = xh[r];
                if (r >= 3 && r < FR + 3) {                          // owned frames (zero beyond T): LayerNorm parameter gradients
                    aw = f4fma(d, v, aw);
                    ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
                }
                const float4 g = make_float4(d.x * gw.x, d.y * gw.y, d.z * gw.z, d.w * gw.w);
                const float m1 = wave_sum(f4sum(g)) * invC;
                const float m2 = wave_sum(g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w) * invC;
                const float sc = rs[r];
                dc[r] = chan ? make_float4(sc * (g.x - m1 - v.x * m2), sc * (g.y - m1 - v.y * m2), sc * (g.z - m1 - v.z * m2),
                                           sc * (g.w - m1 - v.w * m2)) : f4zero();
            }
        }
        asm volatile("" ::: "memory");
        {
            // input gradient: residual rows of the owned frames + the 7 taps (L1 / L2 resident)
            float4 dr[FR], w[7];
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const int t = t0 + f;
                const bool in = t < T;
                const float4 v = dres ? *reinterpret_cast<const float4*>(dres + base + (int64_t)(in ? t : t0) * C + chs) : f4zero();
                const float rm = (dres_rowmask && in) ? dres_rowmask[(int64_t)b * T + t] : 1.f;
                dr[f] = (in && chan && dres) ? make_float4(v.x * rm, v.y * rm, v.z * rm, v.w * rm) : f4zero();
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) w[j] = *reinterpret_cast<const float4*>(dw + (int64_t)j * C + chs);
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const int t = t0 + f;
                if (t < T && chan) {
                    float4 a = dr[f];
#pragma unroll
                    for (int j = 0; j < 7; ++j) a = f4fma(w[j], dc[f + 6 - j], a);      // dc[t - j + 3]
                    st_stream(reinterpret_cast<float4*>(dx + base + (int64_t)t * C + ch), a);
                }
            }
        }
        asm volatile("" ::: "memory");
        if (ddw) {                                                    // kernel-uniform
            // tap gradients: x rows of the window against dc of the owned frames
            float4 xr[FR + 6];
#pragma unroll
            for (int r = 0; r < FR + 6; ++r) {
                const int t = t0 + r - 3;
                const bool in = t >= 0 && t < T;
                const float4 v = *reinterpret_cast<const float4*>(x + base + (int64_t)(in ? t : t0) * C + chs);
                xr[r] = (in && chan) ? v : f4zero();
            }
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const float4 d0 = (t0 + f < T) ? dc[f + 3] : f4zero();            // dc[t]
#pragma unroll
                for (int j = 0; j < 7; ++j) gwt[j] = f4fma(d0, xr[f + j], gwt[j]);   // x[t + j - 3]
                gb.x += d0.x; gb.y += d0.y; gb.z += d0.z; gb.w += d0.w;
            }
        }
        asm volatile("" ::: "memory");
    }
    // parameter gradients: per-wave partials of all ten rows (7 taps, conv bias, LN weight, LN bias) -> LDS, one barrier, then one
    // atomic per channel and workgroup
    if (!ddw && !dlnw) return;                                          // kernel-uniform
#pragma unroll
    for (int j = 0; j < 10; ++j)
        *reinterpret_cast<float4*>(&red[wave][j][lane * 4]) = j < 7 ? gwt[j] : j == 7 ? gb : j == 8 ? aw : ab;
    __syncthreads();
    float* part = ws ? ws + (int64_t)blockIdx.x * 10 * C : nullptr;      // stage 1 of the two-stage reduction: rows [7 taps | ddb | dlnw | dlnb]
    for (int i = threadIdx.x; i < 10 * 256; i += 64 * NWV) {
        const int j = i >> 8, c = i & 255;
        float* dst = j < 7 ? (ddw ? ddw + (int64_t)j * C : nullptr) : j == 7 ? ddb : j == 8 ? dlnw : dlnb;
        if (c < C && (dst || part)) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < NWV; ++q) sum += red[q][j][c];
            if (part) part[(int64_t)j * C + c] = sum;
            else atomicAdd(dst + c, sum);
        }
    }
}


// ---- the same pass for 256 < C <= 384 (the WaveNeXt trunk: C = 384).  A lane owns six channels: a float4 at 4 * lane (channels
// 0..255) and a float2 at 256 + 2 * lane (channels 256..383): both loads are coalesced, the wave reductions run over both parts.
struct V6 { float4 a; float2 b; };
__device__ __forceinline__ V6 v6zero() { V6 r; r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = make_float2(0.f, 0.f); return r; }
__device__ __forceinline__ V6 v6load(const float* row, int lane, bool hasb) {
    V6 r; r.a = *reinterpret_cast<const float4*>(row + 4 * lane);
    r.b = hasb ? *reinterpret_cast<const float2*>(row + 256 + 2 * lane) : make_float2(0.f, 0.f);
    return r;
}
__device__ __forceinline__ V6 v6fma(V6 x, V6 y, V6 c) {
    V6 r; r.a = make_float4(fmaf(x.a.x, y.a.x, c.a.x), fmaf(x.a.y, y.a.y, c.a.y), fmaf(x.a.z, y.a.z, c.a.z), fmaf(x.a.w, y.a.w, c.a.w));
    r.b = make_float2(fmaf(x.b.x, y.b.x, c.b.x), fmaf(x.b.y, y.b.y, c.b.y));
    return r;
}
__device__ __forceinline__ V6 v6mul(V6 x, V6 y) { return v6fma(x, y, v6zero()); }
__device__ __forceinline__ V6 v6add(V6 x, V6 y) {
    V6 r; r.a = make_float4(x.a.x + y.a.x, x.a.y + y.a.y, x.a.z + y.a.z, x.a.w + y.a.w); r.b = make_float2(x.b.x + y.b.x, x.b.y + y.b.y); return r;
}
__device__ __forceinline__ float v6sum(V6 x) { return ((x.a.x + x.a.y) + (x.a.z + x.a.w)) + (x.b.x + x.b.y); }
__device__ __forceinline__ V6 v6scale(V6 x, float s) {
    V6 r; r.a = make_float4(x.a.x * s, x.a.y * s, x.a.z * s, x.a.w * s); r.b = make_float2(x.b.x * s, x.b.y * s); return r;
}

template <int FR, int NWV>
__global__ __launch_bounds__(64 * NWV) void ln_dwconv7_bwd_wide_kernel(const float* __restrict__ dh, const float* __restrict__ xhat,
                                                             const float* __restrict__ rstd, const float* __restrict__ lnw,
                                                             const float* __restrict__ x, const float* __restrict__ dw,
                                                             const float* __restrict__ dres, const float* __restrict__ dres_rowmask,
                                                             float* __restrict__ dx, float* __restrict__ dlnw, float* __restrict__ dlnb,
                                                             float* __restrict__ ddw, float* __restrict__ ddb, float* __restrict__ ws,
                                                             int B, int T, int C) {
    __shared__ float red[NWV][10][384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int runs_per_utt = (T + FR - 1) / FR, nruns = B * runs_per_utt;
    const bool hasb = 256 + 2 * lane < C;
    const float invC = 1.0f / (float)C;
    const V6 gw = v6load(lnw, lane, hasb);
    V6 gwt[7], gb = v6zero(), aw = v6zero(), ab = v6zero();
#pragma unroll
    for (int j = 0; j < 7; ++j) gwt[j] = v6zero();
    for (int run = blockIdx.x * NWV + wave; run < nruns; run += gridDim.x * NWV) {
        const int b = run / runs_per_utt, t0 = (run - b * runs_per_utt) * FR;
        const int64_t base = (int64_t)b * T * C;
        V6 dc[FR + 6];
        {
            V6 xh[FR + 6];
            float rs[FR + 6];
#pragma unroll
            for (int r = 0; r < FR + 6; ++r) {
                const int t = t0 + r - 3;
                const bool in = t >= 0 && t < T;
                const int64_t off = base + (int64_t)(in ? t : t0) * C;
                const V6 d = v6load(dh + off, lane, hasb), v = v6load(xhat + off, lane, hasb);
                rs[r] = in ? rstd[(int64_t)b * T + t] : 0.f;
                dc[r] = in ? d : v6zero();
                xh[r] = in ? v : v6zero();
            }
#pragma unroll
            for (int r = 0; r < FR + 6; ++r) {
                const V6 d = dc[r], v = xh[r];
                if (r >= 3 && r < FR + 3) { aw = v6fma(d, v, aw); ab = v6add(ab, d); }
                const V6 g = v6mul(d, gw);
                const float m1 = wave_sum(v6sum(g)) * invC;
                const float m2 = wave_sum(v6sum(v6mul(g, v))) * invC;
                const float sc = rs[r];
                V6 o;
                o.a = make_float4(sc * (g.a.x - m1 - v.a.x * m2), sc * (g.a.y - m1 - v.a.y * m2), sc * (g.a.z - m1 - v.a.z * m2), sc * (g.a.w - m1 - v.a.w * m2));
                o.b = hasb ? make_float2(sc * (g.b.x - m1 - v.b.x * m2), sc * (g.b.y - m1 - v.b.y * m2)) : make_float2(0.f, 0.f);
                dc[r] = o;
            }
        }
        asm volatile("" ::: "memory");
        {
            V6 w[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) w[j] = v6load(dw + (int64_t)j * C, lane, hasb);
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const int t = t0 + f;
                if (t < T) {
                    V6 a = v6zero();
                    if (dres) {
                        const float rm = dres_rowmask ? dres_rowmask[(int64_t)b * T + t] : 1.f;
                        a = v6scale(v6load(dres + base + (int64_t)t * C, lane, hasb), rm);
                    }
#pragma unroll
                    for (int j = 0; j < 7; ++j) a = v6fma(w[j], dc[f + 6 - j], a);
                    float* o = dx + base + (int64_t)t * C;
                    *reinterpret_cast<float4*>(o + 4 * lane) = a.a;
                    if (hasb) *reinterpret_cast<float2*>(o + 256 + 2 * lane) = a.b;
                }
            }
        }
        asm volatile("" ::: "memory");
        if (ddw) {
#pragma unroll
            for (int r = 0; r < FR + 6; ++r) {
                const int t = t0 + r - 3;
                const bool in = t >= 0 && t < T;
                const V6 xr = in ? v6load(x + base + (int64_t)t * C, lane, hasb) : v6zero();
                // x[t0 + r - 3] meets dc of the owned frame f with tap j = r - f
#pragma unroll
                for (int f = 0; f < FR; ++f) {
                    const int j = r - f;
                    if (j >= 0 && j < 7) {
                        const V6 d0 = (t0 + f < T) ? dc[f + 3] : v6zero();
                        gwt[j] = v6fma(d0, xr, gwt[j]);
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < FR; ++f)
                if (t0 + f < T) gb = v6add(gb, dc[f + 3]);
        }
        asm volatile("" ::: "memory");
    }
    if (!ddw && !dlnw) return;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const V6 v = j < 7 ? gwt[j] : j == 7 ? gb : j == 8 ? aw : ab;
        *reinterpret_cast<float4*>(&red[wave][j][lane * 4]) = v.a;
        *reinterpret_cast<float2*>(&red[wave][j][256 + lane * 2]) = v.b;
    }
    __syncthreads();
    float* part = ws ? ws + (int64_t)blockIdx.x * 10 * C : nullptr;
    for (int i = threadIdx.x; i < 10 * 384; i += 64 * NWV) {
        const int j = i / 384, c = i - j * 384;
        float* dst = j < 7 ? (ddw ? ddw + (int64_t)j * C : nullptr) : j == 7 ? ddb : j == 8 ? dlnw : dlnb;
        if (c < C && (dst || part)) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < NWV; ++q) sum += red[q][j][c];
            if (part) part[(int64_t)j * C + c] = sum;
            else atomicAdd(dst + c, sum);
        }
    }
}

extern "C" int osp_ln_dwconv7_bwd(const float* dh, const float* xhat, const float* rstd, const float* lnw, const float* x,
                                  const float* dw, const float* dres, const float* dres_rowmask, float* dx, float* dlnw, float* dlnb,
                                  float* ddw, float* ddb, int64_t B, int64_t T, int64_t C, float* ws, int64_t ws_blocks, hipStream_t stream) {
    OSP_CHECK_ARG(dh && xhat && rstd && lnw && x && dw && dx, "null operand");
    OSP_CHECK_ARG((dlnw == nullptr) == (dlnb == nullptr) && (ddw == nullptr) == (ddb == nullptr), "dlnw/dlnb and ddw/ddb come in pairs");
    OSP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && C <= 384, "C must be a multiple of 4, <= 384 (wider blocks: osp_layernorm_bwd + osp_dwconv7_bwd)");
    OSP_CHECK_ARG(!ws || ws_blocks > 0, "ws needs ws_blocks > 0");
    if (C > 256) {                                                // the WaveNeXt trunk: six channels per lane, four waves per workgroup
        const bool two = ws && (dlnw || ddw);
        const int64_t want = cdiv(B * cdiv(T, 4), 4), capw = two ? (ws_blocks < 512 ? ws_blocks : 512) : 512;
        const unsigned nb = (unsigned)(want < capw ? want : capw);
        hipLaunchKernelGGL((ln_dwconv7_bwd_wide_kernel<4, 4>), dim3(nb), dim3(256), 0, stream, dh, xhat, rstd, lnw, x, dw, dres, dres_rowmask, dx,
                           dlnw, dlnb, ddw, ddb, two ? ws : nullptr, (int)B, (int)T, (int)C);
        if (two) {
            PartialDst d = {{ddw, ddb, dlnw, dlnb}, {7 * (int)C, (int)C, (int)C, (int)C}};
            launch_reduce_partials(ws, (int)nb, 10 * (int)C, d, stream);
        }
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    static int fr = -1;
    if (fr < 0) { fr = 4; }      // tools/lndw_probe.py at 32 x 800 x 256: FR 4 / 6 / 8 = 41.9 / 49.7 / 50.6 us (8 spills: 256-VGPR cap for two workgroups / CU)
    // 256 VGPRs per thread: 8 waves per CU are resident, as one workgroup of 8 waves (256 workgroups: half the atomics per address)
    // (two of 4 measured the same within noise)
    static int nwv = -1, maxwg = -1;
    if (nwv < 0) { nwv = 8; }
    if (maxwg < 0) { maxwg = 2048 / nwv; }
    const bool two_stage = ws && (dlnw || ddw);
    const int64_t cap = two_stage ? (ws_blocks < maxwg ? ws_blocks : maxwg) : maxwg;
    unsigned nblk = 0;
#define L(F, W) do { nblk = (unsigned)(cdiv(B * cdiv(T, F), W) < cap ? cdiv(B * cdiv(T, F), W) : cap); \
        hipLaunchKernelGGL((ln_dwconv7_bwd_kernel<F, W>), dim3(nblk), dim3(64 * W), 0, stream, dh, xhat, rstd, lnw, x, dw, dres, dres_rowmask, dx, dlnw, dlnb, ddw, ddb, two_stage ? ws : nullptr, (int)B, (int)T, (int)C); } while (0)
    if (nwv == 8) { if (fr == 8) L(8, 8); else L(4, 8); } else { if (fr == 8) L(8, 4); else L(4, 4); }
#undef L
    if (two_stage) {
        PartialDst d = {{ddw, ddb, dlnw, dlnb}, {7 * (int)C, (int)C, (int)C, (int)C}};
        launch_reduce_partials(ws, (int)nblk, 10 * (int)C, d, stream);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
