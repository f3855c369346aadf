// bf16-MFMA conv-GEMM family (performance mode; f32 accumulate): forward / dgrad kernel and weight-gradient kernel.
//
// Serves the same call sites as gemm.hip (pointwise Linear, k-tap Conv1d on channels-last frames, batched products)
// plus the STRIDED (k,1) convolutions of the multi-period discriminator (DiscriminatorP,
// vocoder/wavenext/disc/_discriminators.py:51-60): in channels-last layout every (utterance, period-column) is an
// independent 1-D sequence, so Conv2d((5,1), stride (3,1)) is a strided k-tap Conv1d = a GEMM over K = taps*Cin whose
// operand row for output frame t and tap j is input frame t*stride + j - pad (no im2col, no layout change).
//
// Operands may be stored f32 or bf16 in HBM; they are converted (v_cvt_pk_bf16_f32, RNE) while being staged into LDS.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles (64 accumulator
// VGPRs).  LDS image is row-major with 8 consecutive k per 16-byte slot and rows padded to 72 bf16 (144 B): the
// fragment ds_read_b128 of a 16-lane group then touches 16 distinct slots (conflict-free).  Register-staged double
// buffering: the global loads of tile i+1 are in flight while tile i feeds the matrix pipe; one barrier per tile.
// Sources whose reduction index is NOT the contiguous one (dgrad weights, both wgrad operands) go through a
// transposing loader: 8 strided rows x float4 per thread, packed to k-contiguous 16-byte LDS slots.
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
    FastDiv fd_trows, fd_wrows;
    // output phases of a strided-conv dgrad fused into one launch (blockIdx.z = phase; batch must be 1): the fields a phase
    // overrides -- its row count / geometry, tap subset (count, KW, first-tap offsets into dy and into the weights) and the
    // output offsets.  M of the struct itself is the maximum over the phases (grid size).
    int nphase;
    struct Phase { int M, Trows, Wrows, taps, KW, a_off_h, a_off, c_off_h, c_off; int64_t b_off; FastDiv fd_trows, fd_wrows; } ph[4];
};

// effective parameters of this workgroup (wave-uniform: stays in SGPRs)
__device__ __forceinline__ GemmB gemm_select_phase(const GemmB& pin) {
    GemmB pp = pin;
    if (pin.nphase > 0) {
        const GemmB::Phase q = pin.ph[blockIdx.z];
        pp.M = q.M; pp.Trows = q.Trows; pp.Wrows = q.Wrows; pp.taps = q.taps; pp.KW = q.KW; pp.a_off_h = q.a_off_h; pp.a_off = q.a_off;
        pp.c_off_h = q.c_off_h; pp.c_off = q.c_off; pp.fd_trows = q.fd_trows; pp.fd_wrows = q.fd_wrows;
        pp.B = reinterpret_cast<const char*>(pin.B) + q.b_off * (pin.b_bf16 ? 2 : 4);
    }
    return pp;
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
        out = gelu_f(v);
    }
    if constexpr (EPI == BEPI_SCALE_RES_MASK) {
        if (aux_out) st_aux(aux_out, pp.aux_bf16, crow * pp.ld_aux + n, v);
        const float rs = pp.rowscale ? pp.rowscale[mr] : 1.f, mk = pp.rowmask ? pp.rowmask[mr] : 1.f;
        out = (res[crow * pp.ldr + n] + rs * gam * v) * mk;
    }
    if constexpr (EPI == BEPI_GELU_BWD)
        out = (pp.rowscale ? pp.rowscale[mr] : 1.f) * v * gelu_grad_f(ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n));
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
                *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(Cb) + crow * ldc + n) =
                    *reinterpret_cast<const uint4*>(stage + lrow * SP + cc * 8);
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
__device__ __forceinline__ void xcd_tile(int& mb, int& nb) {
    const int NB = gridDim.x, MB = gridDim.y, total = NB * MB;
    const int lin = blockIdx.y * NB + blockIdx.x;
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

// ------------------------------------------------------------------------------------------------ forward / dgrad
// FAST: every operand row is 16-byte addressable (Cin % 8 == 0, aligned strides, no per-row A scale) -- the generic
// element-wise loaders are not even compiled into that instantiation (they bloat the loop past the I-cache).
template <bool B_KCONTIG, int BKT, bool FAST, int BM_, int BN_>
__global__ __launch_bounds__(256) void conv_gemm_bf16_kernel(const GemmB pin) {
    const GemmB pp = gemm_select_phase(pin);
    // hot-loop scalars in registers (the by-value struct must not be addressed inside the K loop)
    struct { int M, Trows, Wrows, Tin, Hin, Cin, taps, KW, a_step, a_step_h, a_off, a_off_h, a_tapstep, a_tapstep_h, N, a_bf16, b_bf16;
             int64_t lda, sBn, sBtap, sBtap_h, sBk; const float* a_rowscale; } p;
    p.M = pp.M; p.Trows = pp.Trows; p.Wrows = pp.Wrows; p.Tin = pp.Tin; p.Hin = pp.Hin; p.Cin = pp.Cin; p.taps = pp.taps; p.KW = pp.KW;
    p.a_step = pp.a_step; p.a_step_h = pp.a_step_h; p.a_off = pp.a_off; p.a_off_h = pp.a_off_h; p.a_tapstep = pp.a_tapstep;
    p.a_tapstep_h = pp.a_tapstep_h; p.N = pp.N; p.a_bf16 = pp.a_bf16; p.b_bf16 = pp.b_bf16; p.lda = pp.lda; p.sBn = pp.sBn;
    p.sBtap = pp.sBtap; p.sBtap_h = pp.sBtap_h; p.sBk = pp.sBk; p.a_rowscale = pp.a_rowscale;
    constexpr int LDK_ = BKT + 8, KG = BKT / 8, NI = BM_ * KG / 256, NJ = BN_ * KG / 256, RSTEP = 256 / KG;
    constexpr int TM_ = BM_ / 64, TN_ = BN_ / 64;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (BM_ + BN_) * LDK_];
    unsigned short* As = smem;                       // [2][BM_][LDK_]
    unsigned short* Bs = smem + 2 * BM_ * LDK_;      // [2][BN_][LDK_]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM_ / 2), wn0 = (wave & 1) * (BN_ / 2);
    int mb_, nb_;
    xcd_tile(mb_, nb_);
    const int m0 = mb_ * BM_, n0 = nb_ * BN_;
    const int64_t bz = pp.nphase > 0 ? 0 : blockIdx.z;
    const int esA = p.a_bf16 ? 2 : 4, esB = p.b_bf16 ? 2 : 4;
    const char* A = reinterpret_cast<const char*>(pp.A) + bz * pp.sAb * esA;
    const char* B = reinterpret_cast<const char*>(pp.B) + bz * pp.sBb * esB;
    const int K = p.taps * p.Cin;

    // A items: NI per thread: row = tid / KG + RSTEP*i, k-group g = tid % KG
    const int g = tid % KG, r0 = tid / KG;
    int a_t[NI], a_h[NI]; int64_t a_base[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = m0 + r0 + RSTEP * i;
        if (m < p.M) {
            const int u = m / p.Trows, t = m - u * p.Trows, th = t / p.Wrows, tw = t - th * p.Wrows;
            a_t[i] = tw * p.a_step + p.a_off;
            a_h[i] = th * p.a_step_h + p.a_off_h;
            a_base[i] = (int64_t)u * p.Hin * p.Tin;
        } else { a_t[i] = -0x40000000; a_h[i] = 0; a_base[i] = 0; }
    }
    f32x16 acc[TM_][TN_];
#pragma unroll
    for (int i = 0; i < TM_; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[NI], rb[NJ > 4 ? NJ : 4];
    auto a_elem = [&](int i, int k) -> float {
        if (k >= K) return 0.f;
        const int j = k / p.Cin, c = k - j * p.Cin, kh = j / p.KW, kw = j - kh * p.KW;
        const int tt = a_t[i] + kw * p.a_tapstep, hh = a_h[i] + kh * p.a_tapstep_h;
        if (tt < 0 || tt >= p.Tin || hh < 0 || hh >= p.Hin) return 0.f;
        const int64_t row = a_base[i] + (int64_t)hh * p.Tin + tt;
        float v = ld_elem(A, p.a_bf16, row * p.lda + c);
        if (p.a_rowscale) v *= p.a_rowscale[bz * p.M + row];
        return v;
    };
    auto gload = [&](int kt) {
        const int k0 = kt * BKT + g * 8;
        const int j0k = k0 / p.Cin, c0k = k0 - j0k * p.Cin, kh0 = (p.KW == p.taps) ? 0 : j0k / p.KW, kw0 = j0k - kh0 * p.KW;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if constexpr (FAST) {
                if (k0 < K) {   // (j, c, kh, kw) of this thread's k-group: hoisted, one division pair per k-tile
                    const int tt = a_t[i] + kw0 * p.a_tapstep, hh = a_h[i] + kh0 * p.a_tapstep_h;
                    if (tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin)
                        v = ld8_contig(A, p.a_bf16, (a_base[i] + (int64_t)hh * p.Tin + tt) * p.lda + c0k, true);
                }
            } else {
                v = make_uint4(pk2(a_elem(i, k0), a_elem(i, k0 + 1)), pk2(a_elem(i, k0 + 2), a_elem(i, k0 + 3)),
                               pk2(a_elem(i, k0 + 4), a_elem(i, k0 + 5)), pk2(a_elem(i, k0 + 6), a_elem(i, k0 + 7)));
            }
            ra[i] = v;
        }
        if constexpr (B_KCONTIG) {
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const int n = n0 + r0 + RSTEP * i;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (n < p.N && k0 < K) {
                    if constexpr (FAST) {
                        v = ld8_contig(B, p.b_bf16, (int64_t)n * p.sBn + (int64_t)kh0 * p.sBtap_h + (int64_t)kw0 * p.sBtap + c0k, true);
                    } else {
                        float e[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int k = k0 + q;
                            float x = 0.f;
                            if (k < K) { const int j = k / p.Cin, c = k - j * p.Cin, kh = j / p.KW, kw = j - kh * p.KW;
                                x = ld_elem(B, p.b_bf16, (int64_t)n * p.sBn + (int64_t)kh * p.sBtap_h + (int64_t)kw * p.sBtap + (int64_t)c * p.sBk); }
                            e[q] = x;
                        }
                        v = make_uint4(pk2(e[0], e[1]), pk2(e[2], e[3]), pk2(e[4], e[5]), pk2(e[6], e[7]));
                    }
                }
                rb[i] = v;
            }
        } else {
            // transposing loader: k-group kg (8 reduction rows) x 4 output columns per thread
            static_assert(BKT == 64, "the k-strided B loader is laid out for BK = 64");
            constexpr int N4 = BN_ / 4;                                  // column groups per tile
            const int kg = tid / N4, n4 = tid % N4, nn = n0 + 4 * n4;     // kg >= 8 (only when BN_ < 128): idle
            float4 rows[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = kt * BKT + kg * 8 + q;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kg < 8 && k < K && nn < p.N) {
                    const int j = k / p.Cin, c = k - j * p.Cin, kh = j / p.KW, kw = j - kh * p.KW;
                    const int64_t off = (int64_t)c * p.sBk + (int64_t)kh * p.sBtap_h + (int64_t)kw * p.sBtap + nn;
                    if (FAST && nn + 3 < p.N) x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(B) + off);
                    else {
                        x.x = ld_elem(B, p.b_bf16, off);
                        if (nn + 1 < p.N) x.y = ld_elem(B, p.b_bf16, off + 1);
                        if (nn + 2 < p.N) x.z = ld_elem(B, p.b_bf16, off + 2);
                        if (nn + 3 < p.N) x.w = ld_elem(B, p.b_bf16, off + 3);
                    }
                }
                rows[q] = x;
            }
            rb[0] = make_uint4(pk2(rows[0].x, rows[1].x), pk2(rows[2].x, rows[3].x), pk2(rows[4].x, rows[5].x), pk2(rows[6].x, rows[7].x));
            rb[1] = make_uint4(pk2(rows[0].y, rows[1].y), pk2(rows[2].y, rows[3].y), pk2(rows[4].y, rows[5].y), pk2(rows[6].y, rows[7].y));
            rb[2] = make_uint4(pk2(rows[0].z, rows[1].z), pk2(rows[2].z, rows[3].z), pk2(rows[4].z, rows[5].z), pk2(rows[6].z, rows[7].z));
            rb[3] = make_uint4(pk2(rows[0].w, rows[1].w), pk2(rows[2].w, rows[3].w), pk2(rows[4].w, rows[5].w), pk2(rows[6].w, rows[7].w));
        }
    };
    auto sstore = [&](int buf) {
        unsigned short* as = As + buf * BM_ * LDK_;
        unsigned short* bs = Bs + buf * BN_ * LDK_;
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(as + (r0 + RSTEP * i) * LDK_ + g * 8) = ra[i];
        if constexpr (B_KCONTIG) {
#pragma unroll
            for (int i = 0; i < NJ; ++i) *reinterpret_cast<uint4*>(bs + (r0 + RSTEP * i) * LDK_ + g * 8) = rb[i];
        } else {
            constexpr int N4 = BN_ / 4;
            const int kg = tid / N4, n4 = tid % N4;
            if (kg < 8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(bs + (4 * n4 + q) * LDK_ + kg * 8) = rb[q];
            }
        }
    };

    const int nk = (K + BKT - 1) / BKT;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        mma_tile_bf16<TM_, TN_, BKT>(As + buf * BM_ * LDK_, Bs + buf * BN_ * LDK_, wm0, wn0, lane, acc);
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    __syncthreads();                                  // operand tiles are dead: reuse them as the epilogue staging tiles
    gemm_bf16_epilogue<TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, smem + wave * (32 * TM_) * (32 * TN_ + 8));
}



// ---- direct-to-LDS variant (bf16 operands in HBM, Cin % 64 == 0): the staging tiles are written by the LDS-DMA path
// (global_load_lds_dwordx4: 64 lanes x 16 B = 8 rows of 128 B per wave instruction), no VGPR round trip and no
// ds_write pass.  The LDS image is unpadded 128-byte rows; bank conflicts are avoided with an XOR swizzle of the
// 16-byte slot, applied on the SOURCE address when staging and on the read address (both-sides rule, guide section 5.4/21):
//   physical slot = logical k-group ^ ((row >> 1) & 7)
// Rows that fall into conv padding (or past M / N) read a zero page instead.
__device__ __attribute__((aligned(256))) unsigned osp_zero_page[64];

// BM_ = 128: 4 waves of 64x64, two 32 KB stages, 2 workgroups / CU (prefetch distance 1; the second workgroup hides the wait).
// BM_ = 256: 4 waves of 128x64 (128 accumulator registers), three 48 KB stages, 1 workgroup / CU, prefetch distance 2.
//   Per k-slab a CU then reads (128 + 64) * 64 * 2 B * 4 waves = 96 KB of fragments for 2 * 256*128*64 flop, i.e. LDS
//   traffic per flop is 2/3 of the 128x128 tile's (which is LDS-bandwidth bound: 96 KB + 32 KB DMA per 512 MFMA clocks).
template <int BM_, int NST, int BN_ = TBN>
__device__ __forceinline__ void conv_gemm_bf16_glds_body(const GemmB& pin, unsigned short* smem) {
    const GemmB pp = gemm_select_phase(pin);
    constexpr int RA = BM_ / 32, RB = BN_ / 32, TM_ = BM_ / 64, TN_ = BN_ / 64;   // rows staged per thread (A, B); 32x32 tiles per wave along M
    unsigned short* As = smem;                       // [NST][BM_][64]
    unsigned short* Bs = smem + NST * BM_ * TBK;     // [NST][BN_][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM_ / 2), wn0 = (wave & 1) * (BN_ / 2);
    int mb_, nb_;
    xcd_tile(mb_, nb_);
    const int m0 = mb_ * BM_, n0 = nb_ * BN_;
    const int64_t bz = pp.nphase > 0 ? 0 : blockIdx.z;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(pp.A) + bz * pp.sAb;
    const unsigned short* B = reinterpret_cast<const unsigned short*>(pp.B) + bz * pp.sBb;
    const int Cin = pp.Cin, Tin = pp.Tin, Hin = pp.Hin, KW = pp.KW, a_tapstep = pp.a_tapstep, a_tapstep_h = pp.a_tapstep_h;
    const int taps = pp.taps;
    const int64_t lda = pp.lda, sBn = pp.sBn, sBtap = pp.sBtap, sBtap_h = pp.sBtap_h;
    const int K = taps * Cin;
    const int rsub = lane >> 3, pslot = lane & 7;
    // wave w stages rows 8 * (w * RA + i) + rsub of A (i < RA) and 8 * (w * RB + i) + rsub of B (i < RB)
    int a_t[RA], a_h[RA]; int64_t a_base[RA]; int64_t b_row[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + 8 * (wave * RA + i) + rsub;
        if (m < pp.M) {
            const int u = fd_div(m, pp.fd_trows), t = m - u * pp.Trows, th = fd_div(t, pp.fd_wrows), tw = t - th * pp.Wrows;
            a_t[i] = tw * pp.a_step + pp.a_off;
            a_h[i] = th * pp.a_step_h + pp.a_off_h;
            a_base[i] = (int64_t)u * Hin * Tin;
        } else { a_t[i] = -0x40000000; a_h[i] = 0; a_base[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + 8 * (wave * RB + i) + rsub;
        b_row[i] = n < pp.N ? (int64_t)n * sBn : -1;
    }
    // accumulators as 64-row halves: the epilogue is instantiated per half with compile-time indices only (one 512-byte
    // array indexed through the epilogue's nested loops stayed a stack object and was stored to scratch every iteration)
    f32x16 acc0[2][TN_], acc1[2][TN_];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(osp_zero_page);

    // Staging state: per owned row a source pointer for the current tap (or the zero page, with a zero channel stride);
    // advancing k inside a tap is one 64-bit add per row, the row / bounds arithmetic runs once per tap.
    const unsigned short* a_src[RA]; const unsigned short* b_src[RB]; int a_inc[RA], b_inc[RB];
    int cur_tap = -1;
    auto set_tap = [&](int j) {
        const int kh = (KW == taps) ? 0 : j / KW, kw = j - kh * KW;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int r = 8 * (wave * RA + i) + rsub;
            const int q = pslot ^ ((r >> 1) & 7);
            const int tt = a_t[i] + kw * a_tapstep, hh = a_h[i] + kh * a_tapstep_h;
            const bool ok = tt >= 0 && tt < Tin && hh >= 0 && hh < Hin;
            a_src[i] = ok ? A + (a_base[i] + (int64_t)hh * Tin + tt) * lda + q * 8 : zero;
            a_inc[i] = ok ? 1 : 0;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = 8 * (wave * RB + i) + rsub;
            const int q = pslot ^ ((r >> 1) & 7);
            b_src[i] = b_row[i] >= 0 ? B + b_row[i] + (int64_t)kh * sBtap_h + (int64_t)kw * sBtap + q * 8 : zero;
            b_inc[i] = b_row[i] >= 0 ? 1 : 0;
        }
        cur_tap = j;
    };
    int is_j = 0, is_cb = 0;                                   // (tap, channel offset) of the next k-slab to stage
    // one of the RA + RB row loads of a slab (compile-time index): the loads are spread over the 4 k-steps of the MFMA
    // phase -- issued back to back at the top of an iteration they queue behind each other in the texture-address unit
    // (4 waves x 12 x 1 KB at 64 B/clk) and the MFMA pipe idles until the last one has been accepted.
    auto issue_one = [&](int buf, auto idx) {
        constexpr int I = decltype(idx)::value;
        if constexpr (I < RA) {
            unsigned short* dst = As + buf * BM_ * TBK + (wave * RA + I) * 8 * TBK;      // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[I] + is_cb * a_inc[I]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            constexpr int J = I - RA;
            unsigned short* dst = Bs + buf * BN_ * TBK + (wave * RB + J) * 8 * TBK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[J] + is_cb * b_inc[J]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto issue_quarter = [&](int buf, auto qidx) {             // loads [Q * NL / 4, (Q + 1) * NL / 4)
        constexpr int Q = decltype(qidx)::value, NL = RA + RB, L0 = Q * NL / 4, L1 = (Q + 1) * NL / 4;
        if constexpr (L1 - L0 > 0) issue_one(buf, std::integral_constant<int, L0>{});
        if constexpr (L1 - L0 > 1) issue_one(buf, std::integral_constant<int, L0 + 1>{});
        if constexpr (L1 - L0 > 2) issue_one(buf, std::integral_constant<int, L0 + 2>{});
    };
    auto issue_begin = [&]() { if (is_j != cur_tap) set_tap(is_j); };          // wave-uniform branch
    auto issue_end = [&]() { is_cb += TBK; if (is_cb == Cin) { is_cb = 0; ++is_j; } };
    auto issue = [&](int buf) {
        issue_begin();
        issue_quarter(buf, std::integral_constant<int, 0>{}); issue_quarter(buf, std::integral_constant<int, 1>{});
        issue_quarter(buf, std::integral_constant<int, 2>{}); issue_quarter(buf, std::integral_constant<int, 3>{});
        issue_end();
    };
    // MFMA phase over slab `buf`; when `ld` >= 0 the next slab's loads go to buffer `ld`, a quarter per k-step
    auto mma = [&](int buf, int ld) {
        const unsigned short* as = As + buf * BM_ * TBK;
        const unsigned short* bs = Bs + buf * BN_ * TBK;
        const int l31 = lane & 31, lh = lane >> 5;
        if (ld >= 0) issue_begin();
        auto kstep = [&](auto ksidx) {
            constexpr int ks = decltype(ksidx)::value;
            bf16x8 a[TM_], b[TN_];
#pragma unroll
            for (int i = 0; i < TM_; ++i) {
                const int row = wm0 + 32 * i + l31;
                a[i] = *reinterpret_cast<const bf16x8*>(as + row * TBK + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN_; ++j) {
                const int row = wn0 + 32 * j + l31;
                b[j] = *reinterpret_cast<const bf16x8*>(bs + row * TBK + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 3));
            }
            if (ld >= 0) issue_quarter(ld, ksidx);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN_; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc0[i][j], 0, 0, 0);
                    if constexpr (TM_ == 4) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 + i], b[j], acc1[i][j], 0, 0, 0);
                }
        };
        kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{}); kstep(std::integral_constant<int, 3>{});
        if (ld >= 0) issue_end();
    };
    const int nk = K / TBK;
    if constexpr (NST == 2) {
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            mma(buf, kt + 1 < nk ? (buf ^ 1) : -1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // three stages, prefetch distance 2: slab kt+2 is issued into the buffer slab kt-1 was read from (every wave has
        // passed this iteration's barrier, hence finished computing kt-1); the wait leaves slab kt+1's loads in flight.
        issue(0);
        if (nk > 1) issue(1);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA + RB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // bare s_barrier: __syncthreads() carries a workgroup fence that the compiler lowers to vmcnt(0), which would
            // drain slab kt+1's LDS-DMA loads at every iteration (i.e. no prefetch at all).  Every wave has waited for its
            // own slab-kt loads above, so after the barrier the whole slab is in LDS; all ds_reads of the previous
            // iteration have been consumed by MFMAs (lgkmcnt(0)) before a wave arrives here.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mma(buf, kt + 2 < nk ? (buf >= 1 ? buf - 1 : 2) : -1);
            buf = buf == 2 ? 0 : buf + 1;
        }
        __syncthreads();
    }
    constexpr int SP_ = 32 * TN_ + 8;
    gemm_bf16_epilogue<2, TN_>(pp, acc0, m0, n0, wm0, wn0, lane, bz, smem + wave * (32 * TM_) * SP_);
    if constexpr (TM_ == 4) gemm_bf16_epilogue<2, TN_>(pp, acc1, m0, n0, wm0 + 64, wn0, lane, bz, smem + wave * (32 * TM_) * SP_ + 64 * SP_);
}

extern __shared__ __attribute__((aligned(1024))) unsigned short glds_smem[];
__global__ __launch_bounds__(256) void conv_gemm_bf16_glds_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2>(pp, glds_smem);
}
// narrow outputs (N <= 64: the DiscriminatorR stacks): 128x64 tiles, 48 KB of LDS -> 3 workgroups / CU
__global__ __launch_bounds__(256) void conv_gemm_bf16_glds_n64_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2, 64>(pp, glds_smem);
}
// one workgroup per CU (144 KB of LDS): let the register allocator use the whole 512-entry file of a single wave / SIMD
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_gemm_bf16_glds256_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<256, 3>(pp, glds_smem);
}

// C[u, t*c_step + c_off, n] = epi( sum_{j<taps} sum_{c<Cin} A[u, t*a_step + j*a_tapstep + a_off, c] * Bw(n, j, c) )
// for t < Trows (rows M = utterances * Trows); a tap that leaves [0, Tin) contributes zero.
//   forward strided conv : a_step = stride, a_tapstep = 1, a_off = -pad, Tc = Trows = T_out, c_step = 1, c_off = 0
//   dgrad of a strided conv, phase r : see optispeech_amd/ops.py (MPD) -- rows q, t_in = r + stride*q
// dtype flags: 0 = f32 storage, 1 = bf16 storage.
// ------------------------------------------------------------------------------------------------ degenerate shapes
// The discriminators' post convolutions (Cout = 1, _discriminators.py:60,160) and their dgrad (Cin = 1) are not GEMMs: a
// 128-wide tile would be > 98 % padding.  They are HBM-bound streams over the activation (N = 1: read Cin*2 bytes per row
// and tap; Cin = 1: write N*2 bytes per row), so they get VALU kernels that touch every byte once with 16-byte accesses.

// one output element through the run-time epilogue (same semantics as gemm_bf16_epilogue_t)
__device__ __forceinline__ void gemm_bf16_epi_elem(const GemmB& pp, float acc, int m, int n, int64_t bz) {
    const float v = acc + (pp.bias ? pp.bias[n] : 0.f);
    const int64_t mr = bz * pp.M + m;
    const int u = m / pp.Trows, t = m - u * pp.Trows, th = t / pp.Wrows, tw = t - th * pp.Wrows;
    const int64_t crow = (int64_t)u * pp.Tc + (int64_t)(th * pp.c_step_h + pp.c_off_h) * pp.Wc + (int64_t)tw * pp.c_step + pp.c_off;
    const char* aux_in = pp.aux_in ? reinterpret_cast<const char*>(pp.aux_in) + bz * pp.sXb * (pp.aux_bf16 ? 2 : 4) : nullptr;
    float out = v;
    switch (pp.epi) {
        case BEPI_RELU: out = fmaxf(v, 0.f); break;
        case BEPI_LRELU: out = v > 0.f ? v : v * pp.slope; break;
        case BEPI_GELU:
            if (pp.aux_out) st_aux(reinterpret_cast<char*>(pp.aux_out) + bz * pp.sXb * (pp.aux_bf16 ? 2 : 4), pp.aux_bf16, crow * pp.ld_aux + n, v);
            out = gelu_f(v);
            break;
        case BEPI_SCALE_RES_MASK: {
            if (pp.aux_out) st_aux(reinterpret_cast<char*>(pp.aux_out) + bz * pp.sXb * (pp.aux_bf16 ? 2 : 4), pp.aux_bf16, crow * pp.ld_aux + n, v);
            const float rs = pp.rowscale ? pp.rowscale[mr] : 1.f, mk = pp.rowmask ? pp.rowmask[mr] : 1.f;
            out = ((pp.res + bz * pp.sXb)[crow * pp.ldr + n] + rs * (pp.gamma ? pp.gamma[n] : 1.f) * v) * mk;
            break;
        }
        case BEPI_GELU_BWD:
            out = (pp.rowscale ? pp.rowscale[mr] : 1.f) * v * gelu_grad_f(ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n));
            break;
        case BEPI_RELU_BWD: out = ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n) > 0.f ? v : 0.f; break;
        case BEPI_LRELU_BWD: {
            const float e = pp.res_any ? ld_elem(pp.res_any, pp.res_bf16, crow * pp.ldr + n) : 0.f;
            out = ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n) > 0.f ? (v + e) : (v + e) * pp.slope;
            break;
        }
        case BEPI_AXMY: out = (pp.rowscale ? pp.rowscale[mr] : 1.f) * ld_elem(aux_in, pp.aux_bf16, crow * pp.ld_aux + n) - v; break;
        case BEPI_MASK: out = v * (pp.rowmask ? pp.rowmask[mr] : 1.f); break;
        default: break;
    }
    const int esC = pp.c_bf16 ? 2 : 4;
    char* Cb = reinterpret_cast<char*>(pp.C) + bz * pp.sCb * esC;
    if (pp.c_bf16) reinterpret_cast<__bf16*>(Cb)[crow * pp.ldc + n] = (__bf16)out;
    else {
        float* dst = reinterpret_cast<float*>(Cb) + crow * pp.ldc + n;
        *dst = pp.accumulate ? (*dst + out) : out;
    }
}

__device__ __forceinline__ float dot8_bf16(const uint4 a, const uint4 b, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.x), __builtin_bit_cast(bf16x2, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.y), __builtin_bit_cast(bf16x2, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.z), __builtin_bit_cast(bf16x2, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.w), __builtin_bit_cast(bf16x2, b.w), acc, false);
    return acc;
}

// N == 1: y[m] = epi(bias + sum_{tap, c} A[row(m, tap), c] * B[tap, c]).  L lanes share a row (each owns 16-byte channel
// chunks lane, lane + L, ...), the weights (taps * Cin bf16) sit in LDS, products go through v_dot2c_f32_bf16.
// Algorithmic bytes per row: taps * Cin * 2 read (L2 absorbs the tap overlap: unique bytes = Cin * 2) + 4 written.
template <int L>
__global__ __launch_bounds__(256) void conv_rowdot_bf16_kernel(const GemmB pp) {
    extern __shared__ uint4 rd_w[];
    const int C8 = pp.Cin >> 3, nchunk = pp.taps * C8;
    for (int idx = threadIdx.x; idx < nchunk; idx += 256) {
        const int tap = idx / C8, c8 = idx - tap * C8, kh = tap / pp.KW, kw = tap - kh * pp.KW;
        rd_w[idx] = ld8_contig(pp.B, pp.b_bf16, kh * pp.sBtap_h + kw * pp.sBtap + c8 * 8, false);
    }
    __syncthreads();
    const unsigned short* __restrict__ A = reinterpret_cast<const unsigned short*>(pp.A);
    constexpr int RPB = 256 / L;
    const int sub = threadIdx.x % L, rgrp = threadIdx.x / L;
    const int KH = pp.taps / pp.KW;
    for (int m = blockIdx.x * RPB + rgrp; m < pp.M; m += gridDim.x * RPB) {
        const int u = m / pp.Trows, t = m - u * pp.Trows, th = t / pp.Wrows, tw = t - th * pp.Wrows;
        const int at = tw * pp.a_step + pp.a_off, ah = th * pp.a_step_h + pp.a_off_h;
        const int64_t base = (int64_t)u * pp.Hin * pp.Tin;
        float acc = 0.f;
        for (int kh = 0; kh < KH; ++kh) {
            const int hh = ah + kh * pp.a_tapstep_h;
            for (int kw = 0; kw < pp.KW; ++kw) {
                const int tt = at + kw * pp.a_tapstep;
                const bool ok = (unsigned)hh < (unsigned)pp.Hin && (unsigned)tt < (unsigned)pp.Tin;
                const int64_t row = ok ? base + (int64_t)hh * pp.Tin + tt : 0;               // index select: the load stays unconditional
                const uint4* __restrict__ ar = reinterpret_cast<const uint4*>(A + row * pp.lda);
                const uint4* __restrict__ wr = rd_w + (kh * pp.KW + kw) * C8;
                float d = 0.f;
                for (int c8 = sub; c8 < C8; c8 += L) d = dot8_bf16(ar[c8], wr[c8], d);
                acc += ok ? d : 0.f;
            }
        }
#pragma unroll
        for (int o = L >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0) gemm_bf16_epi_elem(pp, acc, m, 0, 0);
    }
}

// Cin == 1: y[m, n] = epi(bias[n] + sum_tap A[row(m, tap)] * B[n, tap]), taps <= 9.  One thread owns 8 consecutive n
// (weights in registers) and walks rows; the LRELU_BWD epilogue on bf16 operands is vectorised (16-byte aux / res / C).
// Algorithmic bytes per row: N * 2 written (+ N * 2 per bf16 epilogue operand read).
#define OUTER_MAXT 9
__global__ __launch_bounds__(256) void conv_outer_bf16_kernel(const GemmB pp) {
    const int N8 = pp.N >> 3, rpb = 256 / N8 > 0 ? 256 / N8 : 1;
    const int n8 = threadIdx.x % N8, rgrp = threadIdx.x / N8;
    if (rgrp >= rpb) return;
    const int n = n8 * 8;
    float w[OUTER_MAXT][8], bias[8];
#pragma unroll
    for (int tap = 0; tap < OUTER_MAXT; ++tap) {
        const int tp = tap < pp.taps ? tap : 0, kh = tp / pp.KW, kw = tp - kh * pp.KW;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float v = ld_elem(pp.B, pp.b_bf16, (int64_t)(n + q) * pp.sBn + kh * pp.sBtap_h + kw * pp.sBtap);
            w[tap][q] = tap < pp.taps ? v : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) bias[q] = pp.bias ? pp.bias[n + q] : 0.f;
    const bool vec_epi = pp.epi == BEPI_LRELU_BWD && pp.c_bf16 && pp.aux_bf16 && (!pp.res_any || pp.res_bf16) &&
                         (pp.ldc & 7) == 0 && (pp.ld_aux & 7) == 0 && (pp.ldr & 7) == 0 &&
                         ((reinterpret_cast<uintptr_t>(pp.C) | reinterpret_cast<uintptr_t>(pp.aux_in) |
                           reinterpret_cast<uintptr_t>(pp.res_any)) & 15) == 0;
    for (int m = blockIdx.x * rpb + rgrp; m < pp.M; m += gridDim.x * rpb) {
        const int u = m / pp.Trows, t = m - u * pp.Trows, th = t / pp.Wrows, tw = t - th * pp.Wrows;
        const int at = tw * pp.a_step + pp.a_off, ah = th * pp.a_step_h + pp.a_off_h;
        const int64_t base = (int64_t)u * pp.Hin * pp.Tin;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
        for (int tap = 0; tap < OUTER_MAXT; ++tap) {
            const int tp = tap < pp.taps ? tap : 0, kh = tp / pp.KW, kw = tp - kh * pp.KW;
            const int hh = ah + kh * pp.a_tapstep_h, tt = at + kw * pp.a_tapstep;
            const bool ok = tap < pp.taps && (unsigned)hh < (unsigned)pp.Hin && (unsigned)tt < (unsigned)pp.Tin;
            const int64_t row = ok ? base + (int64_t)hh * pp.Tin + tt : 0;
            const float a = ld_elem(pp.A, pp.a_bf16, row * pp.lda);
            const float as = ok ? a : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fmaf(as, w[tap][q], acc[q]);
        }
        if (vec_epi) {
            const int64_t crow = (int64_t)u * pp.Tc + (int64_t)(th * pp.c_step_h + pp.c_off_h) * pp.Wc + (int64_t)tw * pp.c_step + pp.c_off;
            const uint4 y = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(pp.aux_in) + crow * pp.ld_aux + n);
            uint4 e = make_uint4(0, 0, 0, 0);
            if (pp.res_any) e = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(pp.res_any) + crow * pp.ldr + n);
            const unsigned yy[4] = {y.x, y.y, y.z, y.w}, ee[4] = {e.x, e.y, e.z, e.w};
            unsigned oo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float y0 = __uint_as_float(yy[q] << 16), y1 = __uint_as_float(yy[q] & 0xffff0000u);
                const float v0 = acc[2 * q] + bias[2 * q] + __uint_as_float(ee[q] << 16);
                const float v1 = acc[2 * q + 1] + bias[2 * q + 1] + __uint_as_float(ee[q] & 0xffff0000u);
                oo[q] = pk2(y0 > 0.f ? v0 : v0 * pp.slope, y1 > 0.f ? v1 : v1 * pp.slope);
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(pp.C) + crow * pp.ldc + n) = make_uint4(oo[0], oo[1], oo[2], oo[3]);
        } else {
            GemmB q = pp; q.bias = nullptr;
#pragma unroll
            for (int e = 0; e < 8; ++e) gemm_bf16_epi_elem(q, acc[e] + bias[e], m, n + e, 0);
        }
    }
}

// Kernel selection + launch for a filled parameter block.  With p.nphase > 0 (fused dgrad phases) the grid's z dimension
// enumerates the phases and p.M is the largest phase (see GemmB::Phase); batch must then be 1.
static int gemm_launch(GemmB& p, int64_t batch_in, hipStream_t stream) {
    const int64_t M = p.M, N = p.N, Cin = p.Cin, taps = p.taps, lda = p.lda, sBn = p.sBn, sBtap = p.sBtap, sBk = p.sBk,
                  sAb = p.sAb, sBb = p.sBb, a_bf16 = p.a_bf16, b_bf16 = p.b_bf16;
    const int64_t d2_9 = p.sBtap_h;
    const void *A = p.A, *B = p.B;
    const float* a_rowscale = p.a_rowscale;
    const int64_t batch = p.nphase > 0 ? p.nphase : batch_in;          // grid z
    const bool single = p.nphase == 0 && batch_in == 1;                  // the degenerate-shape kernels take one problem
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int64_t ea = a_bf16 ? 2 : 4, eb = b_bf16 ? 2 : 4;
    const bool a_fast = (Cin % 8 == 0) && (lda % 8 == 0) && al16(A) && ((sAb * ea) % 16 == 0) && !a_rowscale;
    bool fast;
    if (sBk == 1)
        fast = a_fast && (sBn % 8 == 0) && (sBtap % 8 == 0) && (d2_9 % 8 == 0) && al16(B) && ((sBb * eb) % 16 == 0);
    else
        fast = a_fast && !b_bf16 && (sBk % 4 == 0) && (sBtap % 4 == 0) && (d2_9 % 4 == 0) && al16(B) && ((sBb * eb) % 16 == 0);
    static int use_degen = -1;
    if (use_degen < 0) { const char* e = getenv("OSP_GEMM_DEGEN"); use_degen = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_degen && N == 1 && single && a_bf16 && a_fast && sBk == 1 && taps * Cin * 2 <= 65536) {
        const int c8 = (int)(Cin / 8);
        const int L = c8 >= 64 ? 64 : (c8 >= 32 ? 32 : (c8 >= 16 ? 16 : (c8 >= 8 ? 8 : (c8 >= 4 ? 4 : (c8 >= 2 ? 2 : 1)))));
        const int64_t nb = cdiv(M, 256 / L);
        const dim3 grid((unsigned)(nb < 4096 ? nb : 4096));
        const size_t lds = (size_t)taps * Cin * 2;
#define OSP_ROWDOT(L_) hipLaunchKernelGGL((conv_rowdot_bf16_kernel<L_>), grid, dim3(256), lds, stream, p)
        switch (L) { case 64: OSP_ROWDOT(64); break; case 32: OSP_ROWDOT(32); break; case 16: OSP_ROWDOT(16); break;
                     case 8: OSP_ROWDOT(8); break; case 4: OSP_ROWDOT(4); break; case 2: OSP_ROWDOT(2); break; default: OSP_ROWDOT(1); }
#undef OSP_ROWDOT
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    if (use_degen && Cin == 1 && single && !a_rowscale && taps <= OUTER_MAXT && N % 8 == 0 && N >= 8 && N <= 2048) {
        const int64_t rpb = 256 / (N / 8) > 0 ? 256 / (N / 8) : 1, nb = cdiv(M, rpb * 4);
        hipLaunchKernelGGL(conv_outer_bf16_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, stream, p);
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    // tile shape: 128x128 by default; 128x64 for narrow outputs; 64x64 when the big tiles cannot fill the 256 CUs
    int bm = 128, bn = 128;
    if (N <= 64) bn = 64;
    else if (cdiv(M, 128) * cdiv(N, 128) * batch < 160) { bm = 64; bn = 64; }
    dim3 grid((unsigned)cdiv(N, bn), (unsigned)cdiv(M, bm), (unsigned)batch);
    static int use_glds = -1;
    if (use_glds < 0) { const char* e = getenv("OSP_GEMM_GLDS"); use_glds = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_glds && fast && sBk == 1 && a_bf16 && b_bf16 && (Cin % TBK == 0) && bm == 128 && bn == 64 && N > 8) {
        static int attr64 = 0;
        if (!attr64) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds_n64_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 64) * TBK * 2);
            attr64 = 1;
        }
        hipLaunchKernelGGL(conv_gemm_bf16_glds_n64_kernel, grid, dim3(256), 2 * (128 + 64) * TBK * 2, stream, p);
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    if (use_glds && fast && sBk == 1 && a_bf16 && b_bf16 && (Cin % TBK == 0) && bm == 128 && bn == 128) {
        static int big = -1, attr_done = 0;
        if (big < 0) { const char* e = getenv("OSP_GEMM_BIG"); big = (e && atoi(e) == 1) ? 1 : 0; }
        if (!attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds256_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 + TBN) * TBK * 2);
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + TBN) * TBK * 2);
            attr_done = 1;
        }
        // 256-row / three-stage variant: opt-in (OSP_GEMM_BIG=1).  Measured on MI355X it does not beat the 128x128
        // kernel at 2 workgroups / CU (M=13056, N=1024, K=5120: 218 vs 209 us): with one wave per SIMD the barrier per
        // k-slab and the LDS latency of the first k-step are exposed.  Kept for the next round's 8-wave version.
        if (big && cdiv(M, 256) * cdiv(N, TBN) * batch >= 200) {
            const dim3 g256((unsigned)cdiv(N, TBN), (unsigned)cdiv(M, 256), (unsigned)batch);
            hipLaunchKernelGGL(conv_gemm_bf16_glds256_kernel, g256, dim3(256), 3 * (256 + TBN) * TBK * 2, stream, p);
        } else {
            hipLaunchKernelGGL(conv_gemm_bf16_glds_kernel, grid, dim3(256), 2 * (128 + TBN) * TBK * 2, stream, p);
        }
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    // narrow outputs (N <= 64, the DiscriminatorR stacks): short K loops are latency-bound at 2 workgroups / CU; the
    // BK = 32 instantiation halves the LDS footprint (5 workgroups / CU) -- pays off once there are many row tiles
    static int bk32_env = -2;
    if (bk32_env == -2) { const char* e = getenv("OSP_GEMM_BK32"); bk32_env = e ? atoi(e) : -1; }
    const bool bk32 = bk32_env >= 0 ? bk32_env != 0 : (M >= 65536);
#define OSP_LAUNCH_TILE(KC, F)                                                                                              \
    do {                                                                                                                    \
        if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, 64, F, 128, 128>), grid, dim3(256), 0, stream, p); \
        else if (bm == 128 && KC && bk32) hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, (KC ? 32 : 64), F, 128, 64>), grid, dim3(256), 0, stream, p);  \
        else if (bm == 128) hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, 64, F, 128, 64>), grid, dim3(256), 0, stream, p);  \
        else hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, 64, F, 64, 64>), grid, dim3(256), 0, stream, p);                 \
    } while (0)
    if (sBk != 1) { if (fast) OSP_LAUNCH_TILE(false, true); else OSP_LAUNCH_TILE(false, false); }
    else { if (fast) OSP_LAUNCH_TILE(true, true); else OSP_LAUNCH_TILE(true, false); }
#undef OSP_LAUNCH_TILE
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

static int conv_gemm_bf16_impl(const int64_t* d2, const void* A, int64_t a_bf16, int64_t lda, int64_t M, int64_t Trows, int64_t Tin,
                                  int64_t Cin, int64_t taps, int64_t a_step, int64_t a_tapstep, int64_t a_off,
                                  const float* a_rowscale, const void* B, int64_t b_bf16, int64_t sBn, int64_t sBtap,
                                  int64_t sBk, int64_t N, void* C, int64_t c_bf16, int64_t ldc, int64_t Tc,
                                  int64_t c_step, int64_t c_off, int64_t epi, const float* bias, const float* gamma,
                                  const void* res, int64_t res_bf16, int64_t ldr, const float* rowmask, const float* rowscale,
                                  void* aux_out, const void* aux_in, int64_t aux_bf16, int64_t ld_aux, float slope,
                                  int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate,
                                  hipStream_t stream) {
    OSP_CHECK_ARG(A && B && C, "null operand");
    OSP_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && taps > 0 && Trows > 0 && Tin > 0 && batch > 0, "bad shape");
    OSP_CHECK_ARG(d2[0] > 0 && d2[1] > 0 && d2[2] > 0 && taps % d2[2] == 0 && Trows % d2[0] == 0, "bad 2-D geometry");
    OSP_CHECK_ARG(M % Trows == 0, "M must be a whole number of utterances");
    OSP_CHECK_ARG(sBk == 1 || sBn == 1, "B must be contiguous along k or along n");
    OSP_CHECK_ARG(epi >= 0 && epi <= BEPI_LRELU_BWD, "unknown epilogue");
    OSP_CHECK_ARG(epi != BEPI_SCALE_RES_MASK || (res && !res_bf16), "epilogue needs an f32 res");
    OSP_CHECK_ARG((epi != BEPI_GELU_BWD && epi != BEPI_RELU_BWD && epi != BEPI_AXMY && epi != BEPI_LRELU_BWD) || aux_in, "epilogue needs aux_in");
    OSP_CHECK_ARG(!(c_bf16 && accumulate), "accumulate needs an f32 destination");
    GemmB p;
    p.A = A; p.a_bf16 = (int)a_bf16; p.lda = lda; p.M = (int)M; p.Trows = (int)Trows; p.Tin = (int)Tin; p.Cin = (int)Cin;
    p.taps = (int)taps; p.a_step = (int)a_step; p.a_tapstep = (int)a_tapstep; p.a_off = (int)a_off; p.a_rowscale = a_rowscale;
    p.B = B; p.b_bf16 = (int)b_bf16; p.sBn = sBn; p.sBtap = sBtap; p.sBk = sBk; p.N = (int)N;
    p.C = C; p.c_bf16 = (int)c_bf16; p.ldc = ldc; p.Tc = (int)Tc; p.c_step = (int)c_step; p.c_off = (int)c_off;
    p.epi = (int)epi; p.bias = bias; p.gamma = gamma; p.res = res_bf16 ? nullptr : (const float*)res; p.ldr = ldr;
    p.res_any = res; p.res_bf16 = (int)res_bf16; p.rowmask = rowmask; p.rowscale = rowscale;
    p.aux_out = aux_out; p.aux_in = aux_in; p.aux_bf16 = (int)aux_bf16; p.ld_aux = ld_aux; p.slope = slope;
    p.sAb = sAb; p.sBb = sBb; p.sCb = sCb; p.sXb = sXb; p.accumulate = (int)accumulate;
    p.nphase = 0;
    p.fd_trows = make_fastdiv((unsigned)Trows); p.fd_wrows = make_fastdiv((unsigned)d2[0]);
    p.Wrows = (int)d2[0]; p.Hin = (int)d2[1]; p.KW = (int)d2[2]; p.a_step_h = (int)d2[3]; p.a_tapstep_h = (int)d2[4];
    p.a_off_h = (int)d2[5]; p.Wc = (int)d2[6]; p.c_step_h = (int)d2[7]; p.c_off_h = (int)d2[8]; p.sBtap_h = d2[9];
    return gemm_launch(p, batch, stream);
}

extern "C" int osp_conv_gemm_bf16(const void* A, int64_t a_bf16, int64_t lda, int64_t M, int64_t Trows, int64_t Tin,
                                  int64_t Cin, int64_t taps, int64_t a_step, int64_t a_tapstep, int64_t a_off,
                                  const float* a_rowscale, const void* B, int64_t b_bf16, int64_t sBn, int64_t sBtap,
                                  int64_t sBk, int64_t N, void* C, int64_t c_bf16, int64_t ldc, int64_t Tc,
                                  int64_t c_step, int64_t c_off, int64_t epi, const float* bias, const float* gamma,
                                  const void* res, int64_t res_bf16, int64_t ldr, const float* rowmask, const float* rowscale,
                                  void* aux_out, const void* aux_in, int64_t aux_bf16, int64_t ld_aux, float slope,
                                  int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate,
                                  hipStream_t stream) {
    const int64_t d2[10] = {Trows, 1, taps, 0, 0, 0, Tc, 0, 0, 0};
    return conv_gemm_bf16_impl(d2, A, a_bf16, lda, M, Trows, Tin, Cin, taps, a_step, a_tapstep, a_off, a_rowscale, B, b_bf16, sBn,
                               sBtap, sBk, N, C, c_bf16, ldc, Tc, c_step, c_off, epi, bias, gamma, res, res_bf16, ldr, rowmask,
                               rowscale, aux_out, aux_in, aux_bf16, ld_aux, slope, batch, sAb, sBb, sCb, sXb, accumulate, stream);
}

// 2-D variant: rows of an utterance are (h, w) positions of a channels-last (U, H, W, C) tensor.
//   t -> (th, tw) = divmod(t, Wrows), tap j -> (kh, kw) = divmod(j, KW)
//   input  position: (th*a_step_h + kh*a_tapstep_h + a_off_h,  tw*a_step + kw*a_tapstep + a_off)  in  [0,Hin) x [0,Win)
//   output position: (th*c_step_h + c_off_h, tw*c_step + c_off) of a (Hc, Wc) map with Hc*Wc = Tc
//   weights: Bw(n, kh, kw, c) = B[n*sBn + kh*sBtap_h + kw*sBtap + c*sBk]
// Serves DiscriminatorR's Conv2d stacks (vocoder/wavenext/disc/_discriminators.py:154-161,174-194) and their dgrad.
extern "C" int osp_conv2d_gemm_bf16(const void* A, int64_t a_bf16, int64_t lda, int64_t M, int64_t Trows, int64_t Wrows,
                                    int64_t Hin, int64_t Win, int64_t Cin, int64_t taps, int64_t KW, int64_t a_step_h,
                                    int64_t a_tapstep_h, int64_t a_off_h, int64_t a_step, int64_t a_tapstep, int64_t a_off,
                                    const void* B, int64_t b_bf16, int64_t sBn, int64_t sBtap_h, int64_t sBtap, int64_t sBk,
                                    int64_t N, void* C, int64_t c_bf16, int64_t ldc, int64_t Tc, int64_t Wc, int64_t c_step_h,
                                    int64_t c_off_h, int64_t c_step, int64_t c_off, int64_t epi, const float* bias,
                                    const void* res, int64_t res_bf16, int64_t ldr, const void* aux_in, int64_t aux_bf16,
                                    int64_t ld_aux, float slope, hipStream_t stream) {
    const int64_t d2[10] = {Wrows, Hin, KW, a_step_h, a_tapstep_h, a_off_h, Wc, c_step_h, c_off_h, sBtap_h};
    return conv_gemm_bf16_impl(d2, A, a_bf16, lda, M, Trows, Win, Cin, taps, a_step, a_tapstep, a_off, nullptr, B, b_bf16, sBn,
                               sBtap, sBk, N, C, c_bf16, ldc, Tc, c_step, c_off, epi, bias, nullptr, res, res_bf16, ldr, nullptr,
                               nullptr, nullptr, aux_in, aux_bf16, ld_aux, slope, 1, 0, 0, 0, 0, 0, stream);
}

// dgrad of a strided channels-last conv2d, all output phases in ONE launch.
//   dx[u, h, w, c] = epi( sum_{kh, kw, n} dy[u, (h + ph - kh) / sh, (w + pw - kw) / sw, n] * Wt[c, kh, kw, n] )   (exact divisions only)
// Output phase (rh, rw) = (h % sh, w % sw) only sees the taps kh = kh0 + i*sh, kw = kw0 + j*sw (kh0 = (rh + ph) % sh, ...),
// i.e. a dense convolution over dy with a sub-sampled kernel; phases differ in tap count, first-tap offsets and output offsets
// (GemmB::Phase) and run as blockIdx.z of one grid instead of sh*sw small launches (DiscriminatorR: 4, DiscriminatorP: 3).
// Wt: (Cin, KH, KW, Cout) = the weights transposed for dgrad.  epi: BEPI_NONE or BEPI_LRELU_BWD (aux_in = forward output y
// of the previous layer, `res` an extra addend: the feature-matching gradient of that layer).
extern "C" int osp_conv2d_dgrad_bf16(const void* dy, int64_t dy_bf16, const void* wt, int64_t w_bf16, void* dx, int64_t dx_bf16,
                                     int64_t U, int64_t H, int64_t W, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, int64_t KH,
                                     int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t epi, const void* aux_in,
                                     int64_t aux_bf16, const void* res, int64_t res_bf16, float slope, hipStream_t stream) {
    OSP_CHECK_ARG(dy && wt && dx, "null operand");
    OSP_CHECK_ARG(U > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && sh > 0 && sw > 0, "bad shape");
    OSP_CHECK_ARG(sh * sw <= 4 && KH >= sh && KW >= sw, "unsupported stride (at most 4 phases, kernel >= stride)");
    OSP_CHECK_ARG(epi == BEPI_NONE || (epi == BEPI_LRELU_BWD && aux_in), "dgrad epilogue is NONE or LRELU_BWD");
    GemmB p;
    p.A = dy; p.a_bf16 = (int)dy_bf16; p.lda = Cout; p.Tin = (int)Wo; p.Hin = (int)Ho; p.Cin = (int)Cout;
    p.a_step = 1; p.a_step_h = 1; p.a_tapstep = -1; p.a_tapstep_h = -1; p.a_rowscale = nullptr;
    p.B = wt; p.b_bf16 = (int)w_bf16; p.sBn = KH * KW * Cout; p.sBtap_h = sh * KW * Cout; p.sBtap = sw * Cout; p.sBk = 1; p.N = (int)Cin;
    p.C = dx; p.c_bf16 = (int)dx_bf16; p.ldc = Cin; p.Tc = (int)(H * W); p.Wc = (int)W; p.c_step_h = (int)sh; p.c_step = (int)sw;
    p.epi = (int)epi; p.bias = nullptr; p.gamma = nullptr; p.res = res_bf16 ? nullptr : (const float*)res; p.ldr = Cin;
    p.res_any = res; p.res_bf16 = (int)res_bf16; p.rowmask = nullptr; p.rowscale = nullptr;
    p.aux_out = nullptr; p.aux_in = aux_in; p.aux_bf16 = (int)aux_bf16; p.ld_aux = Cin; p.slope = slope;
    p.sAb = p.sBb = p.sCb = p.sXb = 0; p.accumulate = 0;
    int np = 0;
    for (int64_t rh = 0; rh < sh; ++rh)
        for (int64_t rw = 0; rw < sw; ++rw) {
            const int64_t qh = (H - rh + sh - 1) / sh, qw = (W - rw + sw - 1) / sw;
            if (qh <= 0 || qw <= 0) continue;
            const int64_t kh0 = (rh + ph) % sh, kw0 = (rw + pw) % sw;
            const int64_t n_h = (KH - kh0 + sh - 1) / sh, n_w = (KW - kw0 + sw - 1) / sw;
            GemmB::Phase& q = p.ph[np++];
            q.M = (int)(U * qh * qw); q.Trows = (int)(qh * qw); q.Wrows = (int)qw; q.taps = (int)(n_h * n_w); q.KW = (int)n_w;
            q.a_off_h = (int)((rh + ph - kh0) / sh); q.a_off = (int)((rw + pw - kw0) / sw); q.c_off_h = (int)rh; q.c_off = (int)rw;
            q.b_off = (kh0 * KW + kw0) * Cout;
            q.fd_trows = make_fastdiv((unsigned)q.Trows); q.fd_wrows = make_fastdiv((unsigned)q.Wrows);
        }
    OSP_CHECK_ARG(np > 0, "empty output");
    // degenerate channel counts (first / last layers) go to the single-problem kernels: one launch per phase
    const bool degenerate = (Cin == 1) || (Cout == 1);
    int rc = OSP_OK;
    for (int i = 0; i < (degenerate ? np : 1) && rc == OSP_OK; ++i) {
        GemmB r = p;
        const GemmB::Phase& q = p.ph[i];
        if (degenerate) {
            r.nphase = 0;
            r.M = q.M; r.Trows = q.Trows; r.Wrows = q.Wrows; r.taps = q.taps; r.KW = q.KW; r.a_off_h = q.a_off_h; r.a_off = q.a_off;
            r.c_off_h = q.c_off_h; r.c_off = q.c_off; r.fd_trows = q.fd_trows; r.fd_wrows = q.fd_wrows;
            r.B = reinterpret_cast<const char*>(wt) + q.b_off * (w_bf16 ? 2 : 4);
        } else {
            r.nphase = np;
            int mmax = 0, tmax = 0;
            for (int k = 0; k < np; ++k) { mmax = p.ph[k].M > mmax ? p.ph[k].M : mmax; tmax = p.ph[k].taps > tmax ? p.ph[k].taps : tmax; }
            r.M = mmax; r.Trows = p.ph[0].Trows; r.Wrows = p.ph[0].Wrows; r.taps = tmax; r.KW = p.ph[0].KW;
            r.a_off_h = r.a_off = r.c_off_h = r.c_off = 0; r.fd_trows = p.ph[0].fd_trows; r.fd_wrows = p.ph[0].fd_wrows;
        }
        rc = gemm_launch(r, 1, stream);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------ wgrad
// dW[n, j, c] += oscale[n] * sum_{u,t} arow * dY[u, t, n] * X[u, t*x_step + j - pad, c];  db[n] likewise.
// Both operands are reduction-major -> transposing loader for both.  Split over the frame dimension, f32 atomics.
struct WgradB {
    const void* dY; int y_bf16; int64_t ldy; const void* X; int x_bf16; int64_t ldx;
    int M, Trows, Tin, N, Cin, taps, pad, x_step;
    int Wrows, Hin, KW, x_step_h, pad_h;              // 2-D extension (1-D: Wrows = Trows, Hin = 1, KW = taps)
    FastDiv fd_trows, fd_wrows;
    const float *arow, *oscale; float* dW; int64_t ldw; float* db; int chunk, splits;
    int64_t sYb, sXb, sWb, sDb;
};

__device__ __forceinline__ float4 ld4_any(const void* p, int is_bf16, int64_t off, int lim, bool vec) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lim >= 4 && vec) {
        if (is_bf16) {
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + off);
            x = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16),
                            __uint_as_float(h.y & 0xffff0000u));
        } else x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + off);
    } else {
        if (lim > 0) x.x = ld_elem(p, is_bf16, off);
        if (lim > 1) x.y = ld_elem(p, is_bf16, off + 1);
        if (lim > 2) x.z = ld_elem(p, is_bf16, off + 2);
        if (lim > 3) x.w = ld_elem(p, is_bf16, off + 3);
    }
    return x;
}

template <bool FAST>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(WgradB p) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (TBM + TBN) * LDK];
    unsigned short* As = smem;
    unsigned short* Bs = smem + 2 * TBM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int ctiles = (p.Cin + TBN - 1) / TBN;
    const int j = blockIdx.y / ctiles, c0 = (blockIdx.y - j * ctiles) * TBN;
    const int n0 = blockIdx.x * TBM;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z - bz * p.splits;
    const char* dY = reinterpret_cast<const char*>(p.dY) + (int64_t)bz * p.sYb * (p.y_bf16 ? 2 : 4);
    const char* X = reinterpret_cast<const char*>(p.X) + (int64_t)bz * p.sXb * (p.x_bf16 ? 2 : 4);
    const float* arow = p.arow ? p.arow + (int64_t)bz * p.M : nullptr;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const bool y_vec = (p.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dY) & 15) == 0);
    const bool x_vec = (p.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const int kg = tid >> 5, c4 = tid & 31;                   // k-group (8 frames) x 4 columns

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_bias = (p.db != nullptr) && (blockIdx.y == 0);
    const int blk_kh = j / p.KW, blk_kw = j - blk_kh * p.KW;

    uint4 ra[4], rb[4];
    auto gload = [&](int mk) {
        float4 ya[8], xb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mk + kg * 8 + q;
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f), x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < mend) {
                const int n = n0 + 4 * c4;
                if (n < p.N) {
                    y = ld4_any(dY, p.y_bf16, (int64_t)m * p.ldy + n, p.N - n, y_vec);
                    if (arow) { const float s = arow[m]; y.x *= s; y.y *= s; y.z *= s; y.w *= s; }
                }
                const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
                const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
                const int c = c0 + 4 * c4;
                if (tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin && c < p.Cin)
                    x = ld4_any(X, p.x_bf16, (((int64_t)u * p.Hin + hh) * p.Tin + tt) * p.ldx + c, p.Cin - c, x_vec);
            }
            ya[q] = y; xb[q] = x;
            if (do_bias) { bsum.x += y.x; bsum.y += y.y; bsum.z += y.z; bsum.w += y.w; }
        }
        ra[0] = make_uint4(pk2(ya[0].x, ya[1].x), pk2(ya[2].x, ya[3].x), pk2(ya[4].x, ya[5].x), pk2(ya[6].x, ya[7].x));
        ra[1] = make_uint4(pk2(ya[0].y, ya[1].y), pk2(ya[2].y, ya[3].y), pk2(ya[4].y, ya[5].y), pk2(ya[6].y, ya[7].y));
        ra[2] = make_uint4(pk2(ya[0].z, ya[1].z), pk2(ya[2].z, ya[3].z), pk2(ya[4].z, ya[5].z), pk2(ya[6].z, ya[7].z));
        ra[3] = make_uint4(pk2(ya[0].w, ya[1].w), pk2(ya[2].w, ya[3].w), pk2(ya[4].w, ya[5].w), pk2(ya[6].w, ya[7].w));
        rb[0] = make_uint4(pk2(xb[0].x, xb[1].x), pk2(xb[2].x, xb[3].x), pk2(xb[4].x, xb[5].x), pk2(xb[6].x, xb[7].x));
        rb[1] = make_uint4(pk2(xb[0].y, xb[1].y), pk2(xb[2].y, xb[3].y), pk2(xb[4].y, xb[5].y), pk2(xb[6].y, xb[7].y));
        rb[2] = make_uint4(pk2(xb[0].z, xb[1].z), pk2(xb[2].z, xb[3].z), pk2(xb[4].z, xb[5].z), pk2(xb[6].z, xb[7].z));
        rb[3] = make_uint4(pk2(xb[0].w, xb[1].w), pk2(xb[2].w, xb[3].w), pk2(xb[4].w, xb[5].w), pk2(xb[6].w, xb[7].w));
    };
    auto sstore = [&](int buf) {
        unsigned short* as = As + buf * TBM * LDK;
        unsigned short* bs = Bs + buf * TBN * LDK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<uint4*>(as + (4 * c4 + q) * LDK + kg * 8) = ra[q];
            *reinterpret_cast<uint4*>(bs + (4 * c4 + q) * LDK + kg * 8) = rb[q];
        }
    };
    // ---- fast path: both operands bf16 with 16-byte rows.  Threads 0-127 stage the dY tile, 128-255 the X tile:
    // 8 frames x 8 channels per thread (eight 16-byte loads), 8x8 bf16 transpose in registers, eight ds_write_b128.
    constexpr bool fast = FAST;
    // lane -> (k-group, column-group): k-group fastest, so the 8 lanes of a ds_write_b128 group fill 128 contiguous
    // bytes of ONE LDS row (column-group fastest put all 8 lanes on the same banks: 8-way conflict)
    const int half = tid >> 7, ht = tid & 127, fkg = ht & 7, c8 = ht >> 3;
    uint4 r8[8];
    float bs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto gload_fast = [&](int mk) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mk + fkg * 8 + q;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < mend) {
                if (half == 0) {
                    const int n = n0 + 8 * c8;
                    if (n < p.N) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(dY) + (int64_t)m * p.ldy + n);
                } else {
                    const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
                    const int c = c0 + 8 * c8;
                    const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
                    if (tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin && c < p.Cin)
                        v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(X) + (((int64_t)u * p.Hin + hh) * p.Tin + tt) * p.ldx + c);
                }
            }
            r8[q] = v;
        }
        if (do_bias && half == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                bs8[0] += __uint_as_float(r8[q].x << 16); bs8[1] += __uint_as_float(r8[q].x & 0xffff0000u);
                bs8[2] += __uint_as_float(r8[q].y << 16); bs8[3] += __uint_as_float(r8[q].y & 0xffff0000u);
                bs8[4] += __uint_as_float(r8[q].z << 16); bs8[5] += __uint_as_float(r8[q].z & 0xffff0000u);
                bs8[6] += __uint_as_float(r8[q].w << 16); bs8[7] += __uint_as_float(r8[q].w & 0xffff0000u);
            }
        }
    };
    auto sstore_fast = [&](int buf) {
        unsigned short* dst = (half == 0 ? As + buf * TBM * LDK : Bs + buf * TBN * LDK) + (8 * c8) * LDK + fkg * 8;
        const unsigned* w = reinterpret_cast<const unsigned*>(r8);      // w[q*4 + d]: frame q, channel pair d
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint4 lo, hi;                                                // channels 2d and 2d+1, frames 0..7
            lo.x = (w[0 * 4 + d] & 0xffffu) | (w[1 * 4 + d] << 16);  hi.x = (w[0 * 4 + d] >> 16) | (w[1 * 4 + d] & 0xffff0000u);
            lo.y = (w[2 * 4 + d] & 0xffffu) | (w[3 * 4 + d] << 16);  hi.y = (w[2 * 4 + d] >> 16) | (w[3 * 4 + d] & 0xffff0000u);
            lo.z = (w[4 * 4 + d] & 0xffffu) | (w[5 * 4 + d] << 16);  hi.z = (w[4 * 4 + d] >> 16) | (w[5 * 4 + d] & 0xffff0000u);
            lo.w = (w[6 * 4 + d] & 0xffffu) | (w[7 * 4 + d] << 16);  hi.w = (w[6 * 4 + d] >> 16) | (w[7 * 4 + d] & 0xffff0000u);
            *reinterpret_cast<uint4*>(dst + (2 * d) * LDK) = lo;
            *reinterpret_cast<uint4*>(dst + (2 * d + 1) * LDK) = hi;
        }
    };
    const int niter = (mend - mbeg + TBK - 1) / TBK;
    if (niter > 0) {
        if constexpr (fast) { gload_fast(mbeg); sstore_fast(0); } else { gload(mbeg); sstore(0); }
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const int buf = it & 1;
            if (it + 1 < niter) { if constexpr (fast) gload_fast(mbeg + (it + 1) * TBK); else gload(mbeg + (it + 1) * TBK); }
            mma_tile_bf16<2, 2>(As + buf * TBM * LDK, Bs + buf * TBN * LDK, wm0, wn0, lane, acc);
            if (it + 1 < niter) { if constexpr (fast) sstore_fast(buf ^ 1); else sstore(buf ^ 1); }
            __syncthreads();
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
    float* dW = p.dW + (int64_t)bz * p.sWb;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int c = c0 + wn0 + 32 * jj + l31;
            if (c >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n >= p.N) continue;
                float* dst = dW + (int64_t)n * p.ldw + (int64_t)j * p.Cin + c;
                const float val = (p.oscale ? p.oscale[n] : 1.f) * acc[i][jj][r];
                if (p.splits == 1) *dst += val;            // this block owns the tile: no atomics
                else atomicAdd(dst, val);
            }
        }
    if (do_bias) {
        // reduce the per-thread column sums over the 8 k-groups (threads with equal columns) through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [8][128]
        if constexpr (fast) {
            if (half == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[fkg * 128 + 8 * c8 + e] = bs8[e];
            }
        } else {
            *reinterpret_cast<float4*>(red + kg * 128 + 4 * c4) = bsum;
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q * 128 + tid];
            atomicAdd(p.db + (int64_t)bz * p.sDb + n0 + tid, (p.oscale ? p.oscale[n0 + tid] : 1.f) * s);
        }
    }
}

// ---- transposed-read variant (both operands bf16 with 16-byte rows, N % 128 == 0, Cin % 128 == 0, no row scale).
// The reduction index (frames) is the SLOW index of both dY (M, N) and X (rows, Cin); the kernel above transposes 8x8
// blocks in registers while staging, which makes it VALU-bound (~500 VALU instructions per k-slab and wave, PMC).  Here
// the slabs are copied as they lie in HBM with global_load_lds (64 frames x 128 channels per operand, 256-byte rows), and
// the MFMA fragments (8 consecutive frames of one channel per lane) are produced by ds_read_b64_tr_b16, which transposes
// a 4 (frames) x 16 (channels) block per 16-lane group on the way out of LDS.
// Bank mapping: a 256-byte row covers all 64 banks, so the 4 frame rows of one transposed read would collide 4-way; the
// 16-byte slot index is XOR-ed with 4 * (row & 3) (applied to the global source address when staging and to the LDS
// address when reading), which puts the 8 (row, 16-channel group) segments of a 32-lane pass on 8 distinct bank ranges.
// T = 128: 4 waves of 64x64, 256-byte rows, slot ^= 4 * (row & 3).
// T = 64 (the 64-channel DiscriminatorR layers): 4 waves of 32x32, 128-byte rows (two rows per 64 banks), the 4 frame
// rows of a transposed read alternate bank halves and slot ^= 4 * ((row >> 1) & 1) separates the pairs.
typedef short s16x4 __attribute__((ext_vector_type(4)));
// F32 = true: f32 operands in HBM (the generator's activations); the slab goes global -> registers -> bf16 -> LDS (same
// LDS image as the DMA path, so the transposed reads are shared), with the optional per-frame scale `arow` applied to dY.
template <int T, bool F32 = false>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_tr_kernel(WgradB p) {
    constexpr int SK = 64;                                   // frames per slab
    constexpr int S = T / 8, RPI = 64 / S, NI = SK / RPI / 4, TI = T / 64;   // slots/row, rows/instruction, instr/wave/operand
    __shared__ __attribute__((aligned(1024))) unsigned short smem[2 * 2 * SK * T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (T / 2), wn0 = (wave & 1) * (T / 2);
    const int ctiles = p.Cin / T;
    const int j = blockIdx.y / ctiles, c0 = (blockIdx.y - j * ctiles) * T;
    const int n0 = blockIdx.x * T;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z - bz * p.splits;
    const unsigned short* dY = reinterpret_cast<const unsigned short*>(p.dY) + (int64_t)bz * p.sYb;
    const unsigned short* X = reinterpret_cast<const unsigned short*>(p.X) + (int64_t)bz * p.sXb;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const int blk_kh = j / p.KW, blk_kw = j - blk_kh * p.KW;
    const bool do_bias = (p.db != nullptr) && (blockIdx.y == 0);
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(osp_zero_page);
    const int64_t ldy = p.ldy, ldx = p.ldx;
    auto swz = [](int row) { return T == 128 ? 4 * (row & 3) : 4 * ((row >> 1) & 1); };

    f32x16 acc[TI][TI], accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int jj = 0; jj < TI; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    }
    // staging: wave w, instruction i covers slab rows RPI * (NI * w + i) + (lane / S); physical 16-byte slot = lane % S
    const int srow = lane / S, lslot = (lane % S) ^ swz(srow);
    // one (dY row, X row) pair of loads; `i` = instruction index 0..NI-1
    auto issue_pair = [&](int mk, int buf, int i) {
        unsigned short* ys = smem + buf * (2 * SK * T);
        unsigned short* xs = ys + SK * T;
        const int row0 = RPI * (NI * wave + i), m = mk + row0 + srow;
        const bool mv = m < mend;
        const unsigned short* src = mv ? dY + (int64_t)m * ldy + n0 + lslot * 8 : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ys + row0 * T), 16, 0, 0);
        const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
        const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
        const bool xv = mv && tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin;
        const unsigned short* xsrc = xv ? X + (((int64_t)u * p.Hin + hh) * p.Tin + tt) * ldx + c0 + lslot * 8 : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xsrc,
                                         (__attribute__((address_space(3))) void*)(xs + row0 * T), 16, 0, 0);
    };
    auto issue = [&](int mk, int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_pair(mk, buf, i);
    };
    // f32 operands: the same (row, slot) assignment, through registers
    const float* dYf = reinterpret_cast<const float*>(p.dY) + (int64_t)bz * p.sYb;
    const float* Xf = reinterpret_cast<const float*>(p.X) + (int64_t)bz * p.sXb;
    const float* arow = p.arow ? p.arow + (int64_t)bz * p.M : nullptr;
    float4 ry[NI][2], rx[NI][2];
    auto gload_f32 = [&](int mk) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row0 = RPI * (NI * wave + i), m = mk + row0 + srow;
            const bool mv = m < mend;
            const int mc = mv ? m : mbeg;                                         // index select: loads stay unconditional
            const float4* ys4 = reinterpret_cast<const float4*>(dYf + (int64_t)mc * ldy + n0 + lslot * 8);
            const float sc = mv ? (arow ? arow[mc] : 1.f) : 0.f;
            float4 a = ys4[0], b = ys4[1];
            ry[i][0] = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
            ry[i][1] = make_float4(b.x * sc, b.y * sc, b.z * sc, b.w * sc);
            const int u = fd_div(mc, p.fd_trows), t = mc - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
            const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
            const bool xv = mv && tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin;
            const int64_t xr = xv ? (((int64_t)u * p.Hin + hh) * p.Tin + tt) : 0;
            const float4* xs4 = reinterpret_cast<const float4*>(Xf + xr * ldx + c0 + lslot * 8);
            const float xsel = xv ? 1.f : 0.f;
            a = xs4[0]; b = xs4[1];
            rx[i][0] = make_float4(a.x * xsel, a.y * xsel, a.z * xsel, a.w * xsel);
            rx[i][1] = make_float4(b.x * xsel, b.y * xsel, b.z * xsel, b.w * xsel);
        }
    };
    auto sstore_f32 = [&](int buf) {
        unsigned short* ys = smem + buf * (2 * SK * T);
        unsigned short* xs = ys + SK * T;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int off = (RPI * (NI * wave + i) + srow) * T + (lane % S) * 8;
            *reinterpret_cast<uint4*>(ys + off) = make_uint4(pk2(ry[i][0].x, ry[i][0].y), pk2(ry[i][0].z, ry[i][0].w),
                                                             pk2(ry[i][1].x, ry[i][1].y), pk2(ry[i][1].z, ry[i][1].w));
            *reinterpret_cast<uint4*>(xs + off) = make_uint4(pk2(rx[i][0].x, rx[i][0].y), pk2(rx[i][0].z, rx[i][0].w),
                                                             pk2(rx[i][1].x, rx[i][1].y), pk2(rx[i][1].z, rx[i][1].w));
        }
    };
    // fragment of operand tile `base` ([SK][T]) for the 32 channels starting at `col0`, k-step ks: 8 consecutive frames
    const int r16 = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    auto frag = [&](const unsigned short* base, int col0, int ks) -> bf16x8 {
        const int col = col0 + 16 * g16 + 4 * (r16 & 3);                          // first of this lane's 4 source channels
        const int pslot = (col >> 3) ^ swz(r16 >> 2);
        const unsigned short* a0 = base + (16 * ks + 8 * kg + (r16 >> 2)) * T + pslot * 8 + (col & 7);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * T));
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo; u.s.h = hi;
        return u.v;
    };
    bf16x8 ones;
#pragma unroll
    for (int q = 0; q < 8; ++q) ones[q] = (__bf16)1.0f;
    // MFMA phase over slab `buf`; the next slab (frames from `mk_next`, < 0 = none) is staged one row pair per k-step
    auto mma = [&](int buf, int mk_next) {
        const unsigned short* ys = smem + buf * (2 * SK * T);
        const unsigned short* xs = ys + SK * T;
#pragma unroll
        for (int ks = 0; ks < SK / 16; ++ks) {
            bf16x8 a[TI], b[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = frag(ys, wm0 + 32 * i, ks);
#pragma unroll
            for (int jj = 0; jj < TI; ++jj) b[jj] = frag(xs, wn0 + 32 * jj, ks);
            if (mk_next >= 0 && ks < NI) issue_pair(mk_next, buf ^ 1, ks);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int jj = 0; jj < TI; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
            if (do_bias && wn0 == 0) {                                            // block-uniform x wave-uniform
#pragma unroll
                for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], ones, accb[i], 0, 0, 0);
            }
        }
    };
    const int niter = (mend - mbeg + SK - 1) / SK;
    if constexpr (F32) {
        if (niter > 0) {
            gload_f32(mbeg);
            sstore_f32(0);
            __syncthreads();
            for (int it = 0; it < niter; ++it) {
                const int buf = it & 1;
                if (it + 1 < niter) gload_f32(mbeg + (it + 1) * SK);
                mma(buf, -1);
                if (it + 1 < niter) sstore_f32(buf ^ 1);
                __syncthreads();
            }
        }
    } else if (niter > 0) {
        issue(mbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const int buf = it & 1;
            mma(buf, it + 1 < niter ? mbeg + (it + 1) * SK : -1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
    float* dW = p.dW + (int64_t)bz * p.sWb;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TI; ++jj) {
            const int c = c0 + wn0 + 32 * jj + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float* dst = dW + (int64_t)n * p.ldw + (int64_t)j * p.Cin + c;
                const float val = (p.oscale ? p.oscale[n] : 1.f) * acc[i][jj][r];
                if (p.splits == 1) *dst += val;            // this block owns the tile: no atomics
                else atomicAdd(dst, val);
            }
        }
    if (do_bias && wn0 == 0 && l31 == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                atomicAdd(p.db + (int64_t)bz * p.sDb + n, (p.oscale ? p.oscale[n] : 1.f) * accb[i][r]);
            }
    }
}

static int conv_wgrad_bf16_impl(const int64_t* d2, const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                   int64_t M, int64_t Trows, int64_t Tin, int64_t N, int64_t Cin, int64_t taps, int64_t pad,
                                   int64_t x_step, const float* arow, const float* oscale, float* dW, int64_t ldw,
                                   float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                   hipStream_t stream) {
    OSP_CHECK_ARG(dY && X && dW, "null operand");
    OSP_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && taps > 0 && Trows > 0 && M % Trows == 0 && batch > 0, "bad shape");
    WgradB p;
    p.dY = dY; p.y_bf16 = (int)y_bf16; p.ldy = ldy; p.X = X; p.x_bf16 = (int)x_bf16; p.ldx = ldx;
    p.M = (int)M; p.Trows = (int)Trows; p.Tin = (int)Tin; p.N = (int)N; p.Cin = (int)Cin; p.taps = (int)taps;
    p.pad = (int)pad; p.x_step = (int)x_step; p.arow = arow; p.oscale = oscale; p.dW = dW; p.ldw = ldw; p.db = db;
    p.sYb = sYb; p.sXb = sXb; p.sWb = sWb; p.sDb = sDb;
    p.Wrows = (int)d2[0]; p.Hin = (int)d2[1]; p.KW = (int)d2[2]; p.x_step_h = (int)d2[3]; p.pad_h = (int)d2[4];
    p.fd_trows = make_fastdiv((unsigned)Trows); p.fd_wrows = make_fastdiv((unsigned)d2[0]);
    const int64_t tiles = cdiv(N, TBM) * taps * cdiv(Cin, TBN) * batch;
    int64_t splits = tiles >= 192 ? 1 : cdiv(512, tiles);
    int64_t chunk = cdiv(cdiv(M, splits), TBK) * TBK;
    if (chunk < 2 * TBK) chunk = 2 * TBK;
    splits = cdiv(M, chunk);
    p.chunk = (int)chunk; p.splits = (int)splits;
    dim3 grid((unsigned)cdiv(N, TBM), (unsigned)(taps * cdiv(Cin, TBN)), (unsigned)(splits * batch));
    const bool fast = y_bf16 && x_bf16 && !arow && (ldy % 8 == 0) && (ldx % 8 == 0) && (N % 8 == 0) && (Cin % 8 == 0) &&
                      ((reinterpret_cast<uintptr_t>(dY) & 15) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                      (sYb % 8 == 0) && (sXb % 8 == 0);
    static int use_tr = -1;
    if (use_tr < 0) { const char* e = getenv("OSP_WGRAD_TR"); use_tr = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_tr && fast && N % 64 == 0 && Cin % 64 == 0) {
        // 128-tiles when both channel counts allow it, 64-tiles otherwise (DiscriminatorR).  The frames are split so that
        // the grid is close to a multiple of the resident workgroup count (2 / CU for T = 128, 4 / CU for T = 64);
        // partial sums meet in f32 atomics
        const int64_t T_ = (N % 128 == 0 && Cin % 128 == 0) ? 128 : 64;
        const int64_t tl = (N / T_) * taps * (Cin / T_) * batch, target = T_ == 128 ? 1024 : 2048;
        int64_t sp = tl >= target / 2 - 64 ? 1 : (target + tl / 2) / tl;
        int64_t ch = cdiv(cdiv(M, sp), TBK) * TBK;
        if (ch < 4 * TBK) ch = 4 * TBK;
        sp = cdiv(M, ch);
        p.chunk = (int)ch; p.splits = (int)sp;
        const dim3 g((unsigned)(N / T_), (unsigned)(taps * (Cin / T_)), (unsigned)(sp * batch));
        if (T_ == 128) hipLaunchKernelGGL(conv_wgrad_bf16_tr_kernel<128>, g, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv_wgrad_bf16_tr_kernel<64>, g, dim3(256), 0, stream, p);
    }
    else if (use_tr && !y_bf16 && !x_bf16 && N % 64 == 0 && Cin % 64 == 0 && (ldy % 4 == 0) && (ldx % 4 == 0) &&
             ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) == 0 && (sYb % 4 == 0) && (sXb % 4 == 0)) {
        // f32 operands (generator): 64-channel tiles through registers; small problems -> many splits
        const int64_t tl = (N / 64) * taps * (Cin / 64) * batch;
        int64_t sp = tl >= 960 ? 1 : (2048 + tl / 2) / tl;
        int64_t ch = cdiv(cdiv(M, sp), TBK) * TBK;
        if (ch < 2 * TBK) ch = 2 * TBK;
        sp = cdiv(M, ch);
        p.chunk = (int)ch; p.splits = (int)sp;
        const dim3 g((unsigned)(N / 64), (unsigned)(taps * (Cin / 64)), (unsigned)(sp * batch));
        hipLaunchKernelGGL((conv_wgrad_bf16_tr_kernel<64, true>), g, dim3(256), 0, stream, p);
    }
    else if (fast) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<false>), grid, dim3(256), 0, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_conv_wgrad_bf16(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                   int64_t M, int64_t Trows, int64_t Tin, int64_t N, int64_t Cin, int64_t taps, int64_t pad,
                                   int64_t x_step, const float* arow, const float* oscale, float* dW, int64_t ldw,
                                   float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                   hipStream_t stream) {
    const int64_t d2[5] = {Trows, 1, taps, 0, 0};
    return conv_wgrad_bf16_impl(d2, dY, y_bf16, ldy, X, x_bf16, ldx, M, Trows, Tin, N, Cin, taps, pad, x_step, arow, oscale, dW, ldw,
                                db, batch, sYb, sXb, sWb, sDb, stream);
}

// 2-D weight gradient: dW[n, kh, kw, c] += sum dY[u, th, tw, n] * X[u, th*x_step_h + kh - pad_h, tw*x_step + kw - pad, c]
extern "C" int osp_conv2d_wgrad_bf16(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                     int64_t M, int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t N, int64_t Cin,
                                     int64_t taps, int64_t KW, int64_t pad_h, int64_t pad, int64_t x_step_h, int64_t x_step,
                                     float* dW, float* db, hipStream_t stream) {
    const int64_t d2[5] = {Wrows, Hin, KW, x_step_h, pad_h};
    return conv_wgrad_bf16_impl(d2, dY, y_bf16, ldy, X, x_bf16, ldx, M, Trows, Win, N, Cin, taps, pad, x_step, nullptr, nullptr, dW,
                                taps * Cin, db, 1, 0, 0, 0, 0, stream);
}

// ------------------------------------------------------------------------------------------------ casts
__global__ void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<uint2*>(y)[i] = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = __builtin_bit_cast(unsigned short, (__bf16)x[i]);
}
extern "C" int osp_cast_bf16(const float* x, void* y, int64_t n, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && n > 0, "bad args");
    const int64_t blocks = cdiv(n, 1024);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, x, (unsigned short*)y, n);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ weight packing
// out[n][tap][k] (bf16, contiguous) = w[n*sN + tap*sT + k*sK] (f32; any strides, sT may be negative for a flipped kernel).
// The dgrad GEMMs of the generator address the weights transposed (k-strided); packing them once per call into the
// k-contiguous bf16 layout lets the GEMM use 16-byte operand loads instead of its transposing element loader, which is
// ~2x slower than the GEMM itself on these small shapes.  32x32 tiles through LDS, reads along the unit-stride axis.
// Algorithmic bytes: 4 read + 2 written per weight.
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int N, int taps,
                                                        int K, int64_t sN, int64_t sT, int64_t sK) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = w + (int64_t)tap * sT;
    const bool along_n = (sN < 0 ? -sN : sN) < (sK < 0 ? -sK : sK);
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = along_n ? n0 + tx : n0 + r, k = along_n ? k0 + r : k0 + tx;
        const float v = (n < N && k < K) ? src[(int64_t)n * sN + (int64_t)k * sK] : 0.f;
        if (along_n) tile[r][tx] = v; else tile[tx][r] = v;      // tile[k_local][n_local]
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) out[((int64_t)n * taps + tap) * K + k] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][r]);
    }
}
extern "C" int osp_pack_bf16(const float* w, void* out, int64_t N, int64_t taps, int64_t K, int64_t sN, int64_t sT, int64_t sK,
                             hipStream_t stream) {
    OSP_CHECK_ARG(w && out && N > 0 && taps > 0 && K > 0, "bad args");
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)cdiv(K, 32), (unsigned)cdiv(N, 32), (unsigned)taps), dim3(256), 0, stream, w,
                       (unsigned short*)out, (int)N, (int)taps, (int)K, sN, sT, sK);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
