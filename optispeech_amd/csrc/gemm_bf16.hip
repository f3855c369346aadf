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
#include "gemm_bf16_glds.h"
// phased 8-wave kernel (gemm_bf16_w8p.hip); variant 1: priority flips around the MFMA clusters, 2: none
int osp_launch_glds8p(const GemmB& p, dim3 grid, int variant, hipStream_t stream);
#ifndef OSP_W8P_DEFAULT
#define OSP_W8P_DEFAULT 2
#endif

// (the direct-to-LDS body: gemm_bf16_glds.h)

__global__ __launch_bounds__(256) void conv_gemm_bf16_glds_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2>(pp, glds_smem, grid_tile_ctx());
}
// narrow outputs (N <= 64: the DiscriminatorR stacks): 128x64 tiles, 48 KB of LDS -> 3 workgroups / CU
__global__ __launch_bounds__(256) void conv_gemm_bf16_glds_n64_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2, 64>(pp, glds_smem, grid_tile_ctx());
}
// (the 8-wave 256x256 kernels: gemm_bf16_w8.hip)

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
__device__ __forceinline__ void gemm_bf16_epi_elem_at(const GemmB& pp, float acc, int m, int n, int64_t bz, int u, int th, int tw);
__device__ __forceinline__ void gemm_bf16_epi_elem(const GemmB& pp, float acc, int m, int n, int64_t bz) {
    const int u = m / pp.Trows, t = m - u * pp.Trows, th = t / pp.Wrows, tw = t - th * pp.Wrows;
    gemm_bf16_epi_elem_at(pp, acc, m, n, bz, u, th, tw);
}
// ... with the row's (utterance, h, w) already known to the caller
__device__ __forceinline__ void gemm_bf16_epi_elem_at(const GemmB& pp, float acc, int m, int n, int64_t bz, int u, int th, int tw) {
    const float v = acc + (pp.bias ? pp.bias[n] : 0.f);
    const int64_t mr = bz * pp.M + m;
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

// The same with the tap window (KH x KW), the chunks per lane (CPL = Cin / 8 / L) and the rows per lane group and trip (R) fixed at
// compile time: every 16-byte load of a trip is requested before the first product, row maps use the multiply-high dividers.
// (The loop above walks the taps one L2 round trip at a time and spends ~30 VALU instructions per tap on 64-bit row arithmetic --
// at a 64-lane wave's 4 clocks per instruction that, not memory, set its time: 30 us per phase of the DiscriminatorR first-layer
// dgrad, 268 k rows x 64 channels x 6-12 taps.)
template <int L, int KH, int KW, int CPL, int R>
__global__ __launch_bounds__(256) void conv_rowdot_fixed_bf16_kernel(const GemmB pp) {
    extern __shared__ uint4 rd_w[];
    constexpr int TAPS = KH * KW, C8 = L * CPL, RPB = 256 / L;
    for (int idx = threadIdx.x; idx < TAPS * C8; idx += 256) {
        const int tap = idx / C8, c8 = idx - tap * C8, kh = tap / KW, kw = tap - kh * KW;
        rd_w[idx] = ld8_contig(pp.B, pp.b_bf16, kh * pp.sBtap_h + kw * pp.sBtap + c8 * 8, false);
    }
    __syncthreads();
    const unsigned short* __restrict__ A = reinterpret_cast<const unsigned short*>(pp.A);
    const int sub = threadIdx.x % L, rgrp = threadIdx.x / L;
    const int rstride = gridDim.x * RPB;
    const int HT = pp.Hin * pp.Tin;
    for (int m0 = blockIdx.x * RPB + rgrp; m0 < pp.M; m0 += rstride * R) {
        uint4 av[R][TAPS][CPL];
        bool ok[R][TAPS];
        int us[R], ths[R], tws[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int mm = m0 + r * rstride, m = mm < pp.M ? mm : m0;
            const int u = fd_div(m, pp.fd_trows), t = m - u * pp.Trows, th = fd_div(t, pp.fd_wrows), tw = t - th * pp.Wrows;
            us[r] = u; ths[r] = th; tws[r] = tw;
            const int at = tw * pp.a_step + pp.a_off, ah = th * pp.a_step_h + pp.a_off_h;
            const int base = u * HT;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const int hh = ah + kh * pp.a_tapstep_h, tt = at + kw * pp.a_tapstep;
                    const bool o = (unsigned)hh < (unsigned)pp.Hin && (unsigned)tt < (unsigned)pp.Tin;
                    const int row = o ? base + hh * pp.Tin + tt : 0;                         // index select: the load stays unconditional
                    const uint4* __restrict__ ar = reinterpret_cast<const uint4*>(A + (int64_t)row * pp.lda);
                    ok[r][kh * KW + kw] = o;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) av[r][kh * KW + kw][j] = ar[sub + j * L];
                }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) d = dot8_bf16(av[r][tap][j], rd_w[tap * C8 + sub + j * L], d);
                acc += ok[r][tap] ? d : 0.f;
            }
#pragma unroll
            for (int o = L >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            const int mm = m0 + r * rstride;
            if (sub == 0 && mm < pp.M) gemm_bf16_epi_elem_at(pp, acc, mm, 0, 0, us[r], ths[r], tws[r]);
        }
    }
}

// Cin == 1: y[m, n] = epi(bias[n] + sum_tap A[row(m, tap)] * B[n, tap]), taps <= 9.  One thread owns 8 consecutive n
// (weights in registers) and walks rows; the LRELU_BWD epilogue on bf16 operands is vectorised (16-byte aux / res / C).
// Algorithmic bytes per row: N * 2 written (+ N * 2 per bf16 epilogue operand read).
#define OUTER_MAXT 9
template <int MAXT>                      // 3: the (3, 1)-tap conv_post layers (24 weight registers instead of 72: twice the waves per CU)
__global__ __launch_bounds__(256) void conv_outer_bf16_kernel(const GemmB pp) {
    // the taps x N weights go through LDS: consecutive threads fetch consecutive n (stride sBn: a few cache lines per wave
    // instruction) and every thread then reads its 8 channels of a tap as two 16-byte LDS words.  (Until round 4 each thread
    // fetched its own 8 x taps weights with 2-byte global loads at stride sBn -- 72 instructions of ~24 cache lines per wave:
    // ~15 us of every workgroup round at N = 1024, the whole kernel at 1.2 TB/s.)
    extern __shared__ float ow_lds[];
    const int Nw = pp.N;
    for (int idx = threadIdx.x; idx < pp.taps * Nw; idx += 256) {
        const int tap = idx / Nw, nn = idx - tap * Nw, kh = tap / pp.KW, kw = tap - kh * pp.KW;
        ow_lds[idx] = ld_elem(pp.B, pp.b_bf16, (int64_t)nn * pp.sBn + kh * pp.sBtap_h + kw * pp.sBtap);
    }
    __syncthreads();
    const int N8 = pp.N >> 3, rpb = 256 / N8 > 0 ? 256 / N8 : 1;
    const int n8 = threadIdx.x % N8, rgrp = threadIdx.x / N8;
    if (rgrp >= rpb) return;
    const int n = n8 * 8;
    float w[MAXT][8], bias[8];
#pragma unroll
    for (int tap = 0; tap < MAXT; ++tap) {
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (tap < pp.taps) {
            lo = *reinterpret_cast<const float4*>(ow_lds + tap * Nw + n);
            hi = *reinterpret_cast<const float4*>(ow_lds + tap * Nw + n + 4);
        }
        w[tap][0] = lo.x; w[tap][1] = lo.y; w[tap][2] = lo.z; w[tap][3] = lo.w;
        w[tap][4] = hi.x; w[tap][5] = hi.y; w[tap][6] = hi.z; w[tap][7] = hi.w;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) bias[q] = pp.bias ? pp.bias[n + q] : 0.f;
    const bool vec_epi = pp.epi == BEPI_LRELU_BWD && pp.c_bf16 && pp.aux_bf16 && (!pp.res_any || pp.res_bf16) &&
                         (pp.ldc & 7) == 0 && (pp.ld_aux & 7) == 0 && (pp.ldr & 7) == 0 &&
                         ((reinterpret_cast<uintptr_t>(pp.C) | reinterpret_cast<uintptr_t>(pp.aux_in) |
                           reinterpret_cast<uintptr_t>(pp.res_any)) & 15) == 0;
    if constexpr (MAXT <= 3) {
      if (vec_epi) {
        // the conv_post dgrad (Cin = 1, 3 taps, LeakyReLU' on bf16 rows): 4 rows per trip with every load -- the taps' dy scalars,
        // the y and extra rows -- requested before the first use (round 2 walked one row at a time: a full memory latency per row
        // and 32 bytes in flight per thread, 61 us for the 1024-channel layer = 1.3 TB/s), multiply-high dividers for the row maps
        constexpr int UN = 4;
        const int64_t rstride = (int64_t)gridDim.x * rpb;
        for (int64_t m0 = (int64_t)blockIdx.x * rpb + rgrp; m0 < pp.M; m0 += rstride * UN) {
            float as[UN][3];
            uint4 yv[UN], ev[UN];
            int64_t crow[UN];
            bool live[UN];
#pragma unroll
            for (int i = 0; i < UN; ++i) {
                const int64_t mm = m0 + i * rstride;
                live[i] = mm < pp.M;
                const int m = live[i] ? (int)mm : (int)m0;
                const int u = fd_div(m, pp.fd_trows), t = m - u * pp.Trows, th = fd_div(t, pp.fd_wrows), tw = t - th * pp.Wrows;
                const int at = tw * pp.a_step + pp.a_off, ah = th * pp.a_step_h + pp.a_off_h;
                const int64_t base = (int64_t)u * pp.Hin * pp.Tin;
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    const int tp = tap < pp.taps ? tap : 0, kh = tp / pp.KW, kw = tp - kh * pp.KW;
                    const int hh = ah + kh * pp.a_tapstep_h, tt = at + kw * pp.a_tapstep;
                    const bool ok = tap < pp.taps && (unsigned)hh < (unsigned)pp.Hin && (unsigned)tt < (unsigned)pp.Tin;
                    const int64_t row = ok ? base + (int64_t)hh * pp.Tin + tt : 0;
                    const float a = ld_elem(pp.A, pp.a_bf16, row * pp.lda);
                    as[i][tap] = ok ? a : 0.f;
                }
                crow[i] = (int64_t)u * pp.Tc + (int64_t)(th * pp.c_step_h + pp.c_off_h) * pp.Wc + (int64_t)tw * pp.c_step + pp.c_off;
                yv[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(pp.aux_in) + crow[i] * pp.ld_aux + n);
                ev[i] = make_uint4(0, 0, 0, 0);
                if (pp.res_any) ev[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(pp.res_any) + crow[i] * pp.ldr + n);
            }
#pragma unroll
            for (int i = 0; i < UN; ++i) {
                if (!live[i]) continue;
                const unsigned yy[4] = {yv[i].x, yv[i].y, yv[i].z, yv[i].w}, ee[4] = {ev[i].x, ev[i].y, ev[i].z, ev[i].w};
                unsigned oo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a0 = bias[2 * q], a1 = bias[2 * q + 1];
#pragma unroll
                    for (int tap = 0; tap < 3; ++tap) { a0 = fmaf(as[i][tap], w[tap][2 * q], a0); a1 = fmaf(as[i][tap], w[tap][2 * q + 1], a1); }
                    const float y0 = __uint_as_float(yy[q] << 16), y1 = __uint_as_float(yy[q] & 0xffff0000u);
                    const float v0 = a0 + __uint_as_float(ee[q] << 16), v1 = a1 + __uint_as_float(ee[q] & 0xffff0000u);
                    oo[q] = pk2(y0 > 0.f ? v0 : v0 * pp.slope, y1 > 0.f ? v1 : v1 * pp.slope);
                }
                st_rows(reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(pp.C) + crow[i] * pp.ldc + n), make_uint4(oo[0], oo[1], oo[2], oo[3]), pp.nt_out);
            }
        }
        return;
      }
    }
    for (int m = blockIdx.x * rpb + rgrp; m < pp.M; m += gridDim.x * rpb) {
        const int u = m / pp.Trows, t = m - u * pp.Trows, th = t / pp.Wrows, tw = t - th * pp.Wrows;
        const int at = tw * pp.a_step + pp.a_off, ah = th * pp.a_step_h + pp.a_off_h;
        const int64_t base = (int64_t)u * pp.Hin * pp.Tin;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
        for (int tap = 0; tap < MAXT; ++tap) {
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
    static int nt_env = -1;
    if (nt_env < 0) { nt_env = 0; }
    p.nt_out = nt_env;
    const int64_t M = p.M, N = p.N, Cin = p.Cin, taps = p.taps, lda = p.lda, sBn = p.sBn, sBtap = p.sBtap, sBk = p.sBk,
                  sAb = p.sAb, sBb = p.sBb, a_bf16 = p.a_bf16, b_bf16 = p.b_bf16;
    const int64_t d2_9 = p.sBtap_h;
    const void *A = p.A, *B = p.B;
    const float* a_rowscale = p.a_rowscale;
    const int64_t batch = p.nphase > 0 ? p.nphase : batch_in;          // grid z
    {   // algorithmic flops of this launch (measurement aid, api.cpp)
        double fl = 0.0;
        if (p.nphase > 0) for (int i = 0; i < p.nphase; ++i) fl += 2.0 * p.ph[i].M * p.ph[i].taps * (double)Cin * N;
        else fl = 2.0 * M * taps * (double)Cin * N * batch_in;
        osp_note_flops(fl);
        // algorithmic bytes: unique input frames x Cin, the weights, the output (+ the epilogue's extra operands), each once
        const double ec = p.c_bf16 ? 2.0 : 4.0, nb = (double)batch_in, ea = a_bf16 ? 2.0 : 4.0, eb = b_bf16 ? 2.0 : 4.0;
        const double rows_in = (double)(M / (p.Trows > 0 ? p.Trows : 1)) * p.Hin * p.Tin;         // frames of the input tensor
        double rows_out = (double)M;
        if (p.nphase > 0) { rows_out = 0.0; for (int i = 0; i < p.nphase; ++i) rows_out += p.ph[i].M; }
        double by = nb * (rows_in * Cin * ea + (double)N * taps * Cin * eb + rows_out * N * ec);
        if (p.aux_in) by += nb * rows_out * N * (p.aux_bf16 ? 2.0 : 4.0);
        if (p.res_any) by += nb * rows_out * N * (p.res_bf16 ? 2.0 : 4.0);
        if (p.aux_out) by += nb * rows_out * N * (p.aux_bf16 ? 2.0 : 4.0);
        osp_note_bytes(by);
    }
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
        osp_note_symbol("conv_rowdot_bf16_kernel");
        {   // the hot shapes run with their tap window unrolled (same symbol note: one kernel family)
            const int kw_ = p.KW > 0 ? p.KW : 1, kh_ = (int)taps / kw_;
            static int use_fixed = -1;
            if (use_fixed < 0) { const char* e = getenv("OSP_ROWDOT_FIXED"); use_fixed = (e && atoi(e) == 0) ? 0 : 1; }
            bool done = false;
#define OSP_ROWDOT_FX(L_, KH_, KW_, CPL_, R_)                                                                                   \
            if (!done && use_fixed && M < (1 << 30) && c8 == L_ * CPL_ && kh_ == KH_ && kw_ == KW_ && kh_ * kw_ == taps) {          \
                const int64_t nbf = cdiv(M, (256 / L_) * R_);                                                                     \
                hipLaunchKernelGGL((conv_rowdot_fixed_bf16_kernel<L_, KH_, KW_, CPL_, R_>), dim3((unsigned)(nbf < 2048 ? nbf : 2048)), \
                                   dim3(256), lds, stream, p);                                                                    \
                done = true;                                                                                                      \
            }
            OSP_ROWDOT_FX(8, 3, 4, 1, 1) OSP_ROWDOT_FX(8, 3, 3, 1, 2) OSP_ROWDOT_FX(8, 2, 4, 1, 2) OSP_ROWDOT_FX(8, 2, 3, 1, 2)   // DiscriminatorR: first-layer dgrad phases, conv_post
            OSP_ROWDOT_FX(64, 1, 3, 2, 2)                                                                                         // DiscriminatorP conv_post (1024 channels, taps along the frame axis)
            OSP_ROWDOT_FX(4, 1, 2, 1, 4) OSP_ROWDOT_FX(4, 1, 1, 1, 4)                                                             // DiscriminatorP first-layer dgrad phases (32 channels)
#undef OSP_ROWDOT_FX
            if (done) { OSP_LAUNCH_CHECK(); return OSP_OK; }
        }
#define OSP_ROWDOT(L_) hipLaunchKernelGGL((conv_rowdot_bf16_kernel<L_>), grid, dim3(256), lds, stream, p)
        switch (L) { case 64: OSP_ROWDOT(64); break; case 32: OSP_ROWDOT(32); break; case 16: OSP_ROWDOT(16); break;
                     case 8: OSP_ROWDOT(8); break; case 4: OSP_ROWDOT(4); break; case 2: OSP_ROWDOT(2); break; default: OSP_ROWDOT(1); }
#undef OSP_ROWDOT
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    if (use_degen && Cin == 1 && single && !a_rowscale && taps <= OUTER_MAXT && N % 8 == 0 && N >= 8 && N <= 2048 && taps * N * 4 <= 65536) {
        // <= 3 taps: four rows per thread and trip, at most four workgroups per CU (the weights are staged once per workgroup:
        // 1624 -> 1024 workgroups = 24 -> 20 us at 12992 x 1024).  More taps: one row per thread and trip, so small problems
        // (the 64-channel conv_post of DiscriminatorR: 4896 rows) still spread over > 100 workgroups instead of 39.
        const int64_t rpb = 256 / (N / 8) > 0 ? 256 / (N / 8) : 1, nb = cdiv(M, rpb * (taps <= 3 ? 4 : 1));
        osp_note_symbol("conv_outer_bf16_kernel");
        if (taps <= 3) hipLaunchKernelGGL((conv_outer_bf16_kernel<3>), dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), (size_t)(taps * N * 4), stream, p);
        else hipLaunchKernelGGL((conv_outer_bf16_kernel<OUTER_MAXT>), dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), (size_t)(taps * N * 4), stream, p);
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    // small problems (the generator's own GEMMs: a few hundred 64x64 tiles, K <= ~1300) are latency-bound: deep LDS-DMA ring
    // on 64x64 tiles.  f32 A operands take it up to a larger size: their alternative is the register-staged kernel, which
    // converts right behind each load (no overlap at all).  OSP_GEMM_SMALL = 0 turns the path off (A/B runs).
    static int use_small = -1;
    static int64_t small_max = 160, small_max32 = 260;
    if (use_small < 0) {
        const char* e = getenv("OSP_GEMM_SMALL"); use_small = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (use_small && p.nphase == 0 && fast && sBk == 1 && b_bf16 && (Cin % TBK == 0) && N > 64 &&
        cdiv(M, 128) * cdiv(N, 128) * batch < (a_bf16 ? small_max : small_max32))
        return osp_launch_gemm_small(p, batch, stream);
    // tile shape: 128x128 by default; 128x64 for narrow outputs; 64x64 when the big tiles cannot fill the 256 CUs
    int bm = 128, bn = 128;
    if (N <= 64) bn = 64;
    else if (cdiv(M, 128) * cdiv(N, 128) * batch < 160) { bm = 64; bn = 64; }
    dim3 grid((unsigned)cdiv(N, bn), (unsigned)cdiv(M, bm), (unsigned)batch);
    static int use_glds = -1;
    if (use_glds < 0) { const char* e = getenv("OSP_GEMM_GLDS"); use_glds = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_glds && fast && sBk == 1 && a_bf16 && b_bf16 && Cin == 64 && N == 64 && bm == 128 && bn == 64 && taps > 1 &&
        osp_launch_conv2d_panel(p, batch_in, stream)) {           // conv2d_panel.hip: the 64 <- 64 DiscriminatorR layers and their dgrads
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    if (use_glds && fast && sBk == 1 && a_bf16 && b_bf16 && (Cin % TBK == 0) && bm == 128 && bn == 64 && N > 8) {
        static int attr64 = 0;
        if (!attr64) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds_n64_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 64) * TBK * 2);
            attr64 = 1;
        }
        osp_note_symbol("conv_gemm_bf16_glds_n64_kernel");
        hipLaunchKernelGGL(conv_gemm_bf16_glds_n64_kernel, grid, dim3(256), 2 * (128 + 64) * TBK * 2, stream, p);
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    if (use_glds && fast && sBk == 1 && a_bf16 && b_bf16 && (Cin % TBK == 0) && bm == 128 && bn == 128) {
        // 8-wave 256x256 tiles when they still give every CU work: OSP_GEMM_W8 = 0 (never) / 1 (whenever >= 1 tile per 2 CUs) /
        // unset: the measured crossover (tools/gemm_w8_probe.py (git history); tools/probes/w8p_probe.py)
        // (both read per call: tests and probes switch them in-process)
        // (round 6, with the phased kernel: >= 96 tiles and K >= 2048 -- the batch-32 1024 <- 1024 dgrad and the fused-phase 1024 -> 512
        // dgrads -- measured 14.34-14.37 vs 14.51-14.58 ms / step on one box, 14.51-14.58 vs 14.33-14.53 on another: no robust gain, and
        // those launches run at 40-60 % chip fill alone; fused-phase dgrads as balanced work lists (one 2-tap tile or two 1-tap tiles per
        // workgroup, 255 workgroups): 94.7 -> 86.1 us alone -- the second epilogue of the paired tiles is the long pole -- and 0.1 ms
        // SLOWER in the step; both dropped, profiles/r06_w8_threshold_ab.txt)
        int w8 = -1; int64_t w8_min = 160, w8_kmin = 2304;
        { const char* e = getenv("OSP_GEMM_W8"); if (e) w8 = atoi(e); const char* m = getenv("OSP_GEMM_W8_MIN"); if (m) w8_min = atoi(m);
          const char* k = getenv("OSP_GEMM_W8_KMIN"); if (k) w8_kmin = atoi(k); }
        const int64_t t256 = cdiv(M, 256) * cdiv(N, 256) * batch;
        // measured (round 2, profiles/r02_gemm_w8_probe_*.txt): +23..30 % where the reduction is long (K >= 2560: the
        // 512->1024 and 1024->1024 DiscriminatorP layers at M ~ 13k: 760 -> 950-990 TFLOP/s), -20 % on short-K / narrow layers
        // (K = 640, N = 512: the 256x256 prologue / epilogue is not amortised), neutral at half batch (too few tiles: not taken)
        if (w8 != 0 && N >= 256 && t256 >= (w8 == 2 ? 1 : w8 == 1 ? 128 : w8_min) && (w8 >= 1 || taps * Cin >= w8_kmin)) {    // OSP_GEMM_W8 = 2: any size (tests)
            const dim3 g8((unsigned)cdiv(N, 256), (unsigned)cdiv(M, 256), (unsigned)batch);
            // phased main loops: OSP_GEMM_W8P = 0 lock-step kernel (gemm_bf16_w8.hip) / 1 phased (gemm_bf16_w8p.hip) / 2, the default:
            // phased with the rotating unit schedule and 32-bit buffer addressing (gemm_bf16_w8q.hip; falls back to 1 for operands it
            // cannot address).  Same results bit for bit (tests/test_gpu_gemm_w8p.py); 148.9 -> 138.2 -> 125.0 us at 204 tiles.
            { const char* e = getenv("OSP_GEMM_W8P"); const int ph = e ? atoi(e) : OSP_W8P_DEFAULT; if (ph) return osp_launch_glds8p(p, g8, ph, stream); }
            static int early = -1;
            if (early < 0) { early = 1; }      // +1..3 % in A/B runs (tools/gemm_quick.py (git history))
            return osp_launch_glds8(p, g8, early != 0, stream);
        }
        static int attr_done = 0;
        if (!attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, GLDS_LDS);
            attr_done = 1;
        }
        // (a 256x128 three-stage variant at one wave per SIMD was measured in round 1 and did not beat this kernel at
        // 2 workgroups / CU -- 218 vs 209 us at M = 13056, N = 1024, K = 5120 -- and was removed in round 3)
        osp_note_symbol("conv_gemm_bf16_glds_kernel");
        hipLaunchKernelGGL(conv_gemm_bf16_glds_kernel, grid, dim3(256), GLDS_LDS, stream, p);
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    // narrow outputs (N <= 64, the DiscriminatorR stacks): short K loops are latency-bound at 2 workgroups / CU; the
    // BK = 32 instantiation halves the LDS footprint (5 workgroups / CU) -- pays off once there are many row tiles
    static int bk32_env = -2;
    if (bk32_env == -2) { bk32_env = -1; }
    const bool bk32 = bk32_env >= 0 ? bk32_env != 0 : (M >= 65536);
    return osp_launch_gemm_reg(p, grid, bm, bn, sBk == 1, fast, bk32, stream);
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
static int conv2d_dgrad_impl(const void* dy, int64_t dy_bf16, const void* wt, int64_t w_bf16, void* dx, int64_t dx_bf16,
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

extern "C" int osp_conv2d_dgrad_bf16(const void* dy, int64_t dy_bf16, const void* wt, int64_t w_bf16, void* dx, int64_t dx_bf16,
                                     int64_t U, int64_t H, int64_t W, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, int64_t KH,
                                     int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t epi, const void* aux_in,
                                     int64_t aux_bf16, const void* res, int64_t res_bf16, float slope, hipStream_t stream) {
    return conv2d_dgrad_impl(dy, dy_bf16, wt, w_bf16, dx, dx_bf16, U, H, W, Ho, Wo, Cin, Cout, KH, KW, sh, sw, ph, pw, epi, aux_in, aux_bf16,
                             res, res_bf16, slope, stream);
}


