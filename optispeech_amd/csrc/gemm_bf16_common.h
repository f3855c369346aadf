// Shared device helpers of the bf16 conv-GEMM family (gemm_bf16.hip: forward / dgrad, wgrad_bf16.hip: weight gradients,
// weight packing).  Header-only; every translation unit gets its own copies of the inline functions.
#pragma once
#include "osp_common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define TBM 128
#define TBN 128
#define TBK 64
#define LDK (TBK + 8)

// Division by a run-time constant via multiply-high (round-up method, exact for all 32-bit numerators < 2^31):
// integer division costs ~40 VALU instructions on CDNA; the conv row maps need several per loaded row.
struct FastDiv { unsigned magic, shift, d; };
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f; f.d = d;
    if (d <= 1) { f.magic = 0; f.shift = 0; return f; }
    unsigned s = 0; while ((1u << s) < d) ++s;
    f.shift = s;
    f.magic = (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ int fd_div(int m, const FastDiv f) {
    if (f.d <= 1) return m;
    const unsigned hi = __umulhi((unsigned)m, f.magic);
    return (int)((hi + (unsigned)m) >> f.shift);
}

enum { BEPI_NONE = 0, BEPI_RELU = 1, BEPI_GELU = 2, BEPI_SCALE_RES_MASK = 3, BEPI_GELU_BWD = 4, BEPI_RELU_BWD = 5,
       BEPI_AXMY = 6, BEPI_MASK = 7, BEPI_LRELU = 8, BEPI_LRELU_BWD = 9 };

struct GemmB {
    const void* A; int a_bf16; int64_t lda; int M, Trows, Tin, Cin, taps, a_step, a_tapstep, a_off;
    const float* a_rowscale;
    const void* B; int b_bf16; int64_t sBn, sBtap, sBk; int N;
    void* C; int c_bf16; int64_t ldc; int Tc, c_step, c_off;
    int epi; const float *bias, *gamma, *res; int64_t ldr; const float *rowmask, *rowscale;
    void* aux_out; const void* aux_in; int aux_bf16; int64_t ld_aux; float slope;   // aux_bf16 describes whichever aux is used
    const void* res_any; int res_bf16;   // LRELU_BWD extra addend (f32 or bf16)
    // 2-D (conv2d over channels-last (U,H,W,C)) extension; the 1-D case has Hin = 1, Wrows = Trows, KW = taps
    int Wrows, Hin, KW, a_step_h, a_tapstep_h, a_off_h, Wc, c_step_h, c_off_h; int64_t sBtap_h;
    int64_t sAb, sBb, sCb, sXb; int accumulate;
    int nt_out;                                      // bf16 row stores of the epilogue as non-temporal (streaming) stores
    FastDiv fd_trows, fd_wrows;
    // output phases of a strided-conv dgrad fused into one launch (blockIdx.z = phase; batch must be 1): the fields a phase
    // overrides -- its row count / geometry, tap subset (count, KW, first-tap offsets into dy and into the weights) and the
    // output offsets.  M of the struct itself is the maximum over the phases (grid size).
    int nphase;
    struct Phase { int M, Trows, Wrows, taps, KW, a_off_h, a_off, c_off_h, c_off; int64_t b_off; FastDiv fd_trows, fd_wrows; } ph[4];
};

// Where a workgroup stands in its problem: linear tile id, tile-grid extent, z (batch index, or output phase of a fused dgrad).
// A plain launch derives it from blockIdx / gridDim; a GROUPED launch (several problems -- the five period discriminators' copies
// of one layer -- in one grid, gemm_bf16.hip) from the group's prefix table.
struct TileCtx { int lin, NB, MB, z; };
__device__ __forceinline__ TileCtx grid_tile_ctx() {
    TileCtx t; t.NB = gridDim.x; t.MB = gridDim.y; t.lin = blockIdx.y * gridDim.x + blockIdx.x; t.z = blockIdx.z; return t;
}

// effective parameters of this workgroup (wave-uniform: stays in SGPRs)
__device__ __forceinline__ GemmB gemm_select_phase(const GemmB& pin, int z) {
    GemmB pp = pin;
    if (pin.nphase > 0) {
        const GemmB::Phase q = pin.ph[z];
        pp.M = q.M; pp.Trows = q.Trows; pp.Wrows = q.Wrows; pp.taps = q.taps; pp.KW = q.KW; pp.a_off_h = q.a_off_h; pp.a_off = q.a_off;
        pp.c_off_h = q.c_off_h; pp.c_off = q.c_off; pp.fd_trows = q.fd_trows; pp.fd_wrows = q.fd_wrows;
        pp.B = reinterpret_cast<const char*>(pin.B) + q.b_off * (pin.b_bf16 ? 2 : 4);
    }
    return pp;
}
__device__ __forceinline__ GemmB gemm_select_phase(const GemmB& pin) { return gemm_select_phase(pin, (int)blockIdx.z); }

// 16-byte output row chunk; nt (kernel-uniform): the consumer is a later kernel, keep the operand panels in L2 instead
__device__ __forceinline__ void st_rows(uint4* p, uint4 v, int nt) {
    if (nt) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}
__device__ __forceinline__ unsigned pk2(float a, float b) {
    bf16x2 r; r[0] = (__bf16)a; r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float ld_elem(const void* p, int is_bf16, int64_t off) {
    return is_bf16 ? bf2f(reinterpret_cast<const unsigned short*>(p)[off]) : reinterpret_cast<const float*>(p)[off];
}
// 8 consecutive elements starting at element offset `off` -> packed bf16x8
__device__ __forceinline__ uint4 ld8_contig(const void* p, int is_bf16, int64_t off, bool vec) {
    if (is_bf16) {
        if (vec) return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + off);
        const unsigned short* h = reinterpret_cast<const unsigned short*>(p) + off;
        return make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16),
                          h[6] | ((unsigned)h[7] << 16));
    }
    const float* f = reinterpret_cast<const float*>(p) + off;
    if (vec) {
        const float4 a = *reinterpret_cast<const float4*>(f), b = *reinterpret_cast<const float4*>(f + 4);
        return make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
    }
    return make_uint4(pk2(f[0], f[1]), pk2(f[2], f[3]), pk2(f[4], f[5]), pk2(f[6], f[7]));
}

template <int TM_, int TN_, int BK_ = TBK>
__device__ __forceinline__ void mma_tile_bf16(const unsigned short* __restrict__ As, const unsigned short* __restrict__ Bs,
                                              int wm0, int wn0, int lane, f32x16 (&acc)[TM_][TN_]) {
    constexpr int LD_ = BK_ + 8;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK_ / 16; ++ks) {
        bf16x8 a[TM_], b[TN_];
#pragma unroll
        for (int i = 0; i < TM_; ++i)
            a[i] = *reinterpret_cast<const bf16x8*>(As + (wm0 + 32 * i + l31) * LD_ + ks * 16 + 8 * lh);
#pragma unroll
        for (int j = 0; j < TN_; ++j)
            b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn0 + 32 * j + l31) * LD_ + ks * 16 + 8 * lh);
#pragma unroll
        for (int i = 0; i < TM_; ++i)
#pragma unroll
            for (int j = 0; j < TN_; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}



// ---- shared epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
// The epilogue kind is a template parameter so that every instantiation is a small, fully unrolled, statically indexed
// loop over the 64 accumulator values: a run-time `switch` inside the loop kept it from unrolling and pushed the
// accumulators to scratch (tens of microseconds per workgroup on the short-K convolutions).
// lane <-> lane^1 exchange (DPP quad_perm [1,0,3,2])
__device__ __forceinline__ float dpp_swap1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}

__device__ __forceinline__ void st_aux(void* p, int is_bf16, int64_t idx, float v) {
    if (is_bf16) reinterpret_cast<__bf16*>(p)[idx] = (__bf16)v;
    else reinterpret_cast<float*>(p)[idx] = v;
}

template <int EPI>
__device__ __forceinline__ float gemm_bf16_epi_value(const GemmB& pp, float v, int64_t mr, int64_t crow, int n, float gam,
                                                     const float* res, const char* aux_in, char* aux_out) {
    float out = v;
    if constexpr (EPI == BEPI_RELU) out = fmaxf(v, 0.f);
    if constexpr (EPI == BEPI_LRELU) out = v > 0.f ? v : v * pp.slope;
    if constexpr (EPI == BEPI_GELU) {
        if (aux_out) st_aux(aux_out, pp.aux_bf16, crow * pp.ld_aux + n, v);
        if (pp.c_bf16) out = gelu_fast_f(v);                    // bf16 destination: osp_common.h gelu_fast_parts
        else out = gelu_f(v);
    }
    if constexpr (EPI == BEPI_SCALE_RES_MASK) {
        if (aux_out) st_aux(aux_out, pp.aux_bf16, crow * pp.ld_aux + n, v);
        const float rs = pp.rowscale ? pp.rowscale[mr] : 1.f, mk = pp.rowmask ? pp.rowmask[mr] : 1.f;
        out = (res[crow * pp.ldr + n] + rs * gam * v) * mk;
    }
    if constexpr (EPI == BEPI_GELU_BWD)
    {
        const float uu = ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n);
        float gp;
        if (pp.c_bf16) gp = gelu_grad_fast_f(uu);
        else gp = gelu_grad_f(uu);
        out = (pp.rowscale ? pp.rowscale[mr] : 1.f) * v * gp;
    }
    if constexpr (EPI == BEPI_RELU_BWD) out = ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n) > 0.f ? v : 0.f;
    if constexpr (EPI == BEPI_LRELU_BWD) {   // (acc + extra) * lrelu'(y)
        const float e = pp.res_any ? ld_elem(pp.res_any, pp.res_bf16, crow * pp.ldr + n) : 0.f;
        out = ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n) > 0.f ? (v + e) : (v + e) * pp.slope;
    }
    if constexpr (EPI == BEPI_AXMY)
        out = (pp.rowscale ? pp.rowscale[mr] : 1.f) * ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n) - v;
    if constexpr (EPI == BEPI_MASK) out = v * (pp.rowmask ? pp.rowmask[mr] : 1.f);
    return out;
}

// ---- row-domain epilogue for the kinds that READ a tensor of the output's shape (LeakyReLU' needs y, plus the optional
// feature-matching gradient; GELU' needs u).  In the MFMA layout a lane owns one column, so those reads were 2-byte loads
// (64 per operand and wave tile) touching 64-byte fragments of 4 lines each: on the DiscriminatorP dgrad launches they
// doubled the kernel time (67 -> 139 us at 1024 -> 512, stride 3).  Here the f32 accumulators of one 32-row block go
// through the wave-private LDS tile first; afterwards a lane owns 8 consecutive channels of a row, reads y / extra / u with
// one 16-byte load each (8 full lines per wave instruction), applies the epilogue and stores 16 bytes.
__device__ __forceinline__ void unpack8_bf16(const uint4 q, float (&f)[8]) {
    f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
    f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xffff0000u);
    f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xffff0000u);
}
__device__ __forceinline__ void ld8_f32(const void* p, int is_bf16, int64_t off, float (&f)[8]) {
    if (is_bf16) unpack8_bf16(*reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + off), f);
    else {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + off);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + off + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
}
__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int EPI>
__device__ __forceinline__ bool gemm_bf16_rows_ok(const GemmB& pp) {
    if constexpr (EPI == BEPI_LRELU_BWD)
        return pp.aux_in && aligned16(pp.aux_in) && (pp.ld_aux & 7) == 0 && ((pp.sXb * (pp.aux_bf16 ? 2 : 4)) & 15) == 0 &&
               (!pp.res_any || (aligned16(pp.res_any) && (pp.ldr & 7) == 0));
    if constexpr (EPI == BEPI_GELU_BWD || EPI == BEPI_RELU_BWD)
        return pp.aux_in && aligned16(pp.aux_in) && (pp.ld_aux & 7) == 0 && ((pp.sXb * (pp.aux_bf16 ? 2 : 4)) & 15) == 0;
    if constexpr (EPI == BEPI_GELU)                            // forward GELU that also writes the pre-activation (bf16 or f32 rows)
        return !pp.aux_out || (aligned16(pp.aux_out) && (pp.ld_aux & 7) == 0 && ((pp.sXb * (pp.aux_bf16 ? 2 : 4)) & 15) == 0);
    // kinds that read nothing of the output's shape: the row domain still wins on instruction count -- one ds_write_b32 per
    // accumulator, then 8 channels per lane (bias add, activation, pack, ONE 16-byte store, row geometry once per 8 values)
    // against ~12 VALU + a DPP swap + a 4-byte LDS store per value in the MFMA layout.  The 256x256 tile's epilogue was ~10 of
    // the 21.7 us a one-slab launch takes (tools/epi_probe.py (git history)).
    if constexpr (EPI == BEPI_NONE || EPI == BEPI_RELU || EPI == BEPI_LRELU || EPI == BEPI_MASK) return true;
    if constexpr (EPI == BEPI_SCALE_RES_MASK)                  // ConvNeXt pwconv2: f32 residual rows in, optional z rows out
        return pp.res && aligned16(pp.res) && (pp.ldr & 7) == 0 && ((pp.sXb * 4) & 15) == 0 &&
               (!pp.aux_out || (aligned16(pp.aux_out) && (pp.ld_aux & 7) == 0 && ((pp.sXb * (pp.aux_bf16 ? 2 : 4)) & 15) == 0));
    if constexpr (EPI == BEPI_AXMY)
        return pp.aux_in && aligned16(pp.aux_in) && (pp.ld_aux & 7) == 0 && ((pp.sXb * (pp.aux_bf16 ? 2 : 4)) & 15) == 0;
    return false;
}

// FASTG (GELU kinds only): the Abramowitz-Stegun GELU for bf16 destinations, the library erff otherwise -- a template parameter,
// not a run-time select: under `c_bf16 ? fast : exact` (and even under an if / else around two loops) the compiler evaluated both.
template <int EPI, int TM_, int TN_, bool FASTG = true>
__device__ __forceinline__ void gemm_bf16_epilogue_rows(const GemmB& pp, f32x16 (&acc)[TM_][TN_], int m0, int n0, int wm0, int wn0,
                                                        int lane, int64_t bz, float* stage) {
    constexpr int SPF = 32 * TN_ + 8;                          // f32 staging pitch: 4 rows apart = 32 banks apart
    constexpr int CPR = 4 * TN_, RPI = 64 / CPR;               // 8-channel chunks per row, rows per wave instruction
    const bool c_bf16 = pp.c_bf16 != 0;
    char* Cb = reinterpret_cast<char*>(pp.C) + bz * pp.sCb * (c_bf16 ? 2 : 4);
    const char* aux_in = reinterpret_cast<const char*>(pp.aux_in) + bz * pp.sXb * (pp.aux_bf16 ? 2 : 4);
    char* aux_out = pp.aux_out ? reinterpret_cast<char*>(pp.aux_out) + bz * pp.sXb * (pp.aux_bf16 ? 2 : 4) : nullptr;
    const float* res = pp.res ? pp.res + bz * pp.sXb : nullptr;
    const int l31 = lane & 31, lh = lane >> 5, cc = lane % CPR, rr = lane / CPR;
    const int Trows = pp.Trows, Wrows = pp.Wrows, Tc = pp.Tc, Wc = pp.Wc, c_step = pp.c_step, c_off = pp.c_off,
              c_step_h = pp.c_step_h, c_off_h = pp.c_off_h, M = pp.M, N = pp.N;
    const FastDiv fd_trows = pp.fd_trows, fd_wrows = pp.fd_wrows;
    const int n = n0 + wn0 + cc * 8;
    const bool n_ok = n < N;                                   // N % 8 == 0: a chunk is inside or outside as a whole
    float bias[8], gam[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bias[k] = (pp.bias && n_ok) ? pp.bias[n + k] : 0.f;
        gam[k] = (EPI == BEPI_SCALE_RES_MASK && pp.gamma && n_ok) ? pp.gamma[n + k] : 1.f;
    }
    const float slope = pp.slope;
#pragma unroll
    for (int i = 0; i < TM_; ++i) {
#pragma unroll
        for (int j = 0; j < TN_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SPF + 32 * j + l31] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int lrow = it * RPI + rr, m = m0 + wm0 + 32 * i + lrow;
            if (m < M && n_ok) {
                const int u = fd_div(m, fd_trows), t = m - u * Trows, th = fd_div(t, fd_wrows), tw = t - th * Wrows;
                const int64_t crow = (int64_t)u * Tc + (int64_t)(th * c_step_h + c_off_h) * Wc + (int64_t)tw * c_step + c_off;
                const float4 v0 = *reinterpret_cast<const float4*>(stage + lrow * SPF + cc * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(stage + lrow * SPF + cc * 8 + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w}, y[8], o[8];
                if constexpr (EPI == BEPI_GELU) {
                    // pre-activation rows (the backward's GELU' operand) as 16-byte chunks: in the MFMA layout they were 2-byte
                    // stores, 64 per lane and wave tile (+14 us on the 87 us decoder pwconv1 launch)
#pragma unroll
                    for (int k = 0; k < 8; ++k) y[k] = v[k] + bias[k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { if constexpr (FASTG) o[k] = gelu_fast_f(y[k]); else o[k] = gelu_f(y[k]); }
                    if (aux_out) {
                        if (pp.aux_bf16)
                            st_rows(reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(aux_out) + crow * pp.ld_aux + n),
                                    make_uint4(pk2(y[0], y[1]), pk2(y[2], y[3]), pk2(y[4], y[5]), pk2(y[6], y[7])), pp.nt_out);
                        else {
                            float* up = reinterpret_cast<float*>(aux_out) + crow * pp.ld_aux + n;
                            *reinterpret_cast<float4*>(up) = make_float4(y[0], y[1], y[2], y[3]);
                            *reinterpret_cast<float4*>(up + 4) = make_float4(y[4], y[5], y[6], y[7]);
                        }
                    }
                } else if constexpr (EPI == BEPI_SCALE_RES_MASK) {
                    const int64_t mr = bz * M + m;
                    const float rs = pp.rowscale ? pp.rowscale[mr] : 1.f, mk = pp.rowmask ? pp.rowmask[mr] : 1.f;
                    ld8_f32(res, 0, crow * pp.ldr + n, y);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float sv = v[k] + bias[k]; v[k] = sv; o[k] = (y[k] + rs * gam[k] * sv) * mk; }
                    if (aux_out) {                              // z = W2 g + b2: the operand of the layer-scale gradient
                        if (pp.aux_bf16)
                            st_rows(reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(aux_out) + crow * pp.ld_aux + n),
                                    make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7])), pp.nt_out);
                        else {
                            float* up = reinterpret_cast<float*>(aux_out) + crow * pp.ld_aux + n;
                            *reinterpret_cast<float4*>(up) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(up + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                    }
                } else if constexpr (EPI == BEPI_NONE || EPI == BEPI_RELU || EPI == BEPI_LRELU || EPI == BEPI_MASK) {
                    const float mk = (EPI == BEPI_MASK && pp.rowmask) ? pp.rowmask[bz * M + m] : 1.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float sv = v[k] + bias[k];
                        if constexpr (EPI == BEPI_NONE) o[k] = sv;
                        if constexpr (EPI == BEPI_RELU) o[k] = fmaxf(sv, 0.f);
                        if constexpr (EPI == BEPI_LRELU) o[k] = sv > 0.f ? sv : sv * slope;
                        if constexpr (EPI == BEPI_MASK) o[k] = sv * mk;
                    }
                } else {
                    ld8_f32(aux_in, pp.aux_bf16, crow * pp.ld_aux + n, y);
                }
                if constexpr (EPI == BEPI_LRELU_BWD) {
                    float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (pp.res_any) ld8_f32(pp.res_any, pp.res_bf16, crow * pp.ldr + n, e);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float s = v[k] + bias[k] + e[k]; o[k] = y[k] > 0.f ? s : s * slope; }
                }
                if constexpr (EPI == BEPI_RELU_BWD) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = y[k] > 0.f ? v[k] + bias[k] : 0.f;
                }
                if constexpr (EPI == BEPI_GELU_BWD) {
                    const float rs = pp.rowscale ? pp.rowscale[bz * M + m] : 1.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if constexpr (FASTG) o[k] = rs * (v[k] + bias[k]) * gelu_grad_fast_f(y[k]);
                        else o[k] = rs * (v[k] + bias[k]) * gelu_grad_f(y[k]);
                    }
                }
                if constexpr (EPI == BEPI_AXMY) {
                    const float rs = pp.rowscale ? pp.rowscale[bz * M + m] : 1.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = rs * y[k] - (v[k] + bias[k]);
                }
                if (c_bf16) {
                    st_rows(reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(Cb) + crow * pp.ldc + n),
                            make_uint4(pk2(o[0], o[1]), pk2(o[2], o[3]), pk2(o[4], o[5]), pk2(o[6], o[7])), pp.nt_out);
                } else {                                        // f32 rows: two 16-byte stores per lane (32 B x 8 lanes = a 256-byte run)
                    float* cp = reinterpret_cast<float*>(Cb) + crow * pp.ldc + n;
                    if (pp.accumulate) {
                        const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
                        o[0] += c0.x; o[1] += c0.y; o[2] += c0.z; o[3] += c0.w; o[4] += c1.x; o[5] += c1.y; o[6] += c1.z; o[7] += c1.w;
                    }
                    *reinterpret_cast<float4*>(cp) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(cp + 4) = make_float4(o[4], o[5], o[6], o[7]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// Row geometry (two divisions per row, done with the multiply-high dividers) is computed once per accumulator row and
// shared by the TN_ column tiles.  bf16 destinations are written as 4-byte pairs: lanes n / n+1 swap the values of two
// consecutive rows (DPP), the even lane stores (row r, cols n..n+1), the odd lane (row r+1, cols n-1..n).
template <int EPI, int TM_, int TN_>
__device__ __forceinline__ void gemm_bf16_epilogue_t(const GemmB& pp, f32x16 (&acc)[TM_][TN_], int m0, int n0, int wm0, int wn0,
                                                     int lane, int64_t bz, unsigned short* stage) {
    const int esC = pp.c_bf16 ? 2 : 4;
    char* Cb = reinterpret_cast<char*>(pp.C) + bz * pp.sCb * esC;
    const float* res = pp.res ? pp.res + bz * pp.sXb : nullptr;
    const char* aux_in = pp.aux_in ? reinterpret_cast<const char*>(pp.aux_in) + bz * pp.sXb * (pp.aux_bf16 ? 2 : 4) : nullptr;
    char* aux_out = pp.aux_out ? reinterpret_cast<char*>(pp.aux_out) + bz * pp.sXb * (pp.aux_bf16 ? 2 : 4) : nullptr;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool c_bf16 = pp.c_bf16 != 0, accumulate = pp.accumulate != 0;
    const int Trows = pp.Trows, Wrows = pp.Wrows, Tc = pp.Tc, Wc = pp.Wc, c_step = pp.c_step, c_off = pp.c_off,
              c_step_h = pp.c_step_h, c_off_h = pp.c_off_h, M = pp.M, N = pp.N;
    const int64_t ldc = pp.ldc;
    const FastDiv fd_trows = pp.fd_trows, fd_wrows = pp.fd_wrows;
    const bool pair_ok = c_bf16 && (ldc & 1) == 0 && ((reinterpret_cast<uintptr_t>(Cb) & 3) == 0) && (N & 1) == 0;
    // bf16 destinations with 16-byte addressable rows go through a wave-private LDS tile (the operand buffers are dead by
    // now): the MFMA layout (lane = column) is turned into 16-byte row chunks, so every store instruction writes 8 full
    // 128-byte lines instead of 64-byte fragments of 4 different lines.
    constexpr int SP = 32 * TN_ + 8;                                              // staging pitch (elements)
    const bool staged = stage != nullptr && pair_ok && (ldc & 7) == 0 && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0) && (N & 7) == 0;
    // (callers reserve max(32 * TM_, 64) bf16 rows per wave: one 32-row block of f32)
    // row-domain path: bf16 rows as before; f32 rows (pwconv2 / dh of the ConvNeXt block, the accumulate form) when 8-channel chunks
    // are 16-byte addressable
    const bool rows_f32 = stage != nullptr && !c_bf16 && (ldc & 7) == 0 && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0) && (N & 7) == 0;
    {
        if ((staged || rows_f32) && gemm_bf16_rows_ok<EPI>(pp)) {                                   // kernel-uniform
            if constexpr (EPI == BEPI_GELU || EPI == BEPI_GELU_BWD) {
                if (c_bf16) gemm_bf16_epilogue_rows<EPI, TM_, TN_, true>(pp, acc, m0, n0, wm0, wn0, lane, bz, reinterpret_cast<float*>(stage));
                else gemm_bf16_epilogue_rows<EPI, TM_, TN_, false>(pp, acc, m0, n0, wm0, wn0, lane, bz, reinterpret_cast<float*>(stage));
            } else {
                gemm_bf16_epilogue_rows<EPI, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, reinterpret_cast<float*>(stage));
            }
            return;
        }
    }
    float bias[TN_], gam[TN_];
    int ncol[TN_];
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
        ncol[j] = n0 + wn0 + 32 * j + l31;
        const bool n_ok = ncol[j] < N;
        bias[j] = (pp.bias && n_ok) ? pp.bias[ncol[j]] : 0.f;
        gam[j] = (EPI == BEPI_SCALE_RES_MASK && pp.gamma && n_ok) ? pp.gamma[ncol[j]] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < TM_; ++i)
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {                       // row pair (r, r + 1): consecutive rows m, m + 1
            int64_t crow[2]; int mrow[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 2 * rp + h;
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                mrow[h] = m;
                const int u = fd_div(m, fd_trows), t = m - u * Trows, th = fd_div(t, fd_wrows), tw = t - th * Wrows;
                crow[h] = (int64_t)u * Tc + (int64_t)(th * c_step_h + c_off_h) * Wc + (int64_t)tw * c_step + c_off;
            }
#pragma unroll
            for (int j = 0; j < TN_; ++j) {
                const int n = ncol[j];
                const bool n_ok = n < N;
                float out[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    out[h] = 0.f;
                    if (n_ok && mrow[h] < M)
                        out[h] = gemm_bf16_epi_value<EPI>(pp, acc[i][j][2 * rp + h] + bias[j], bz * M + mrow[h], crow[h], n, gam[j],
                                                          res, aux_in, aux_out);
                }
                if (pair_ok) {
                    const bool odd = (lane & 1) != 0;
                    const float give = odd ? out[0] : out[1], got = dpp_swap1(give);
                    const int h = odd ? 1 : 0;
                    if (staged) {
                        const int r = 2 * rp + h, lrow = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        *reinterpret_cast<unsigned*>(stage + lrow * SP + 32 * j + (l31 & ~1)) = odd ? pk2(got, out[1]) : pk2(out[0], got);
                    } else if (n_ok && mrow[h] < M) {
                        const unsigned pk = odd ? pk2(got, out[1]) : pk2(out[0], got);
                        *reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(Cb) + crow[h] * ldc + (n & ~1)) = pk;
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (n_ok && mrow[h] < M) {
                            if (c_bf16) reinterpret_cast<__bf16*>(Cb)[crow[h] * ldc + n] = (__bf16)out[h];
                            else {
                                float* dst = reinterpret_cast<float*>(Cb) + crow[h] * ldc + n;
                                *dst = accumulate ? (*dst + out[h]) : out[h];
                            }
                        }
                }
            }
        }
    if (staged) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int CPR = 4 * TN_;                                              // 16-byte chunks per staged row
        constexpr int RPI = 64 / CPR;                                             // rows per wave instruction
        const int cc = lane % CPR, rr = lane / CPR;
#pragma unroll
        for (int it = 0; it < 32 * TM_ / RPI; ++it) {
            const int lrow = it * RPI + rr, m = m0 + wm0 + lrow, n = n0 + wn0 + cc * 8;
            if (m < M && n < N) {
                const int u = fd_div(m, fd_trows), t = m - u * Trows, th = fd_div(t, fd_wrows), tw = t - th * Wrows;
                const int64_t crow = (int64_t)u * Tc + (int64_t)(th * c_step_h + c_off_h) * Wc + (int64_t)tw * c_step + c_off;
                st_rows(reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(Cb) + crow * ldc + n),
                        *reinterpret_cast<const uint4*>(stage + lrow * SP + cc * 8), pp.nt_out);
            }
        }
    }
}

template <int TM_, int TN_>
__device__ __forceinline__ void gemm_bf16_epilogue(const GemmB& pp, f32x16 (&acc)[TM_][TN_], int m0, int n0, int wm0, int wn0,
                                                   int lane, int64_t bz, unsigned short* stage = nullptr) {
    switch (pp.epi) {
        case BEPI_RELU: gemm_bf16_epilogue_t<BEPI_RELU, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_GELU: gemm_bf16_epilogue_t<BEPI_GELU, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_SCALE_RES_MASK: gemm_bf16_epilogue_t<BEPI_SCALE_RES_MASK, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_GELU_BWD: gemm_bf16_epilogue_t<BEPI_GELU_BWD, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_RELU_BWD: gemm_bf16_epilogue_t<BEPI_RELU_BWD, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_AXMY: gemm_bf16_epilogue_t<BEPI_AXMY, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_MASK: gemm_bf16_epilogue_t<BEPI_MASK, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_LRELU: gemm_bf16_epilogue_t<BEPI_LRELU, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        case BEPI_LRELU_BWD: gemm_bf16_epilogue_t<BEPI_LRELU_BWD, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage); break;
        default: gemm_bf16_epilogue_t<BEPI_NONE, TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, stage);
    }
}

// XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs in linear block order (x fastest), and each
// XCD has its own 4 MB L2.  The linear id is first folded so that every XCD owns one contiguous range of tile ids, then
// tiles are ordered in groups of 8 row blocks x all column blocks: the ~64 workgroups resident on one XCD share 8 A row
// panels and the B column panels through that XCD's L2 instead of streaming 64 different A panels from HBM.
__device__ __forceinline__ void xcd_tile(const TileCtx& tc, int& mb, int& nb) {
    const int NB = tc.NB, MB = tc.MB, total = NB * MB;
    const int lin = tc.lin;
    const int xcd = lin & 7, local = lin >> 3;
    const int per = total >> 3, rem = total & 7;               // XCDs < rem own per + 1 tiles
    const int pid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + local;
    constexpr int GM = 8;
    const int gsize = GM * NB, group = pid / gsize, first = group * GM;
    const int gm = MB - first < GM ? MB - first : GM;
    const int in_g = pid - group * gsize;
    mb = first + in_g % gm;
    nb = in_g / gm;
}
__device__ __forceinline__ void xcd_tile(int& mb, int& nb) { xcd_tile(grid_tile_ctx(), mb, nb); }


// register-staged kernel family (gemm_bf16_reg.hip): tile (bm x bn), k-contiguous / k-strided B, fast / generic loaders
int osp_launch_gemm_reg(const GemmB& p, dim3 grid, int bm, int bn, bool b_kcontig, bool fast, bool bk32, hipStream_t stream);
// small-problem kernel (gemm_bf16_small.hip): 64x64 tiles, 2 waves, 3-4 stage LDS-DMA ring; bf16 or f32 A, bf16 k-contiguous B,
// Cin % 64 == 0, 16-byte addressable rows, no per-row A scale, no fused dgrad phases
int osp_launch_gemm_small(const GemmB& p, int64_t batch, hipStream_t stream);
