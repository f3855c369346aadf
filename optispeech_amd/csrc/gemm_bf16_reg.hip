// Register-staged conv-GEMM kernel of the bf16 family (f32 or bf16 storage, converted while staging): its own translation
// unit because its 14 instantiations x 10 fused epilogues dominate the build time.  Dispatch lives in gemm_bf16.hip.
#include "gemm_bf16_common.h"

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
    // K order: when the channel count is a whole number of k-tiles, slab kt = (channel block kt / taps, tap kt % taps) -- the
    // channel-major order of the direct-to-LDS kernels (gemm_bf16.hip), so that a layer gives bit-identical results whichever
    // member of the family its size selects (tests/test_gpu_fullsize.py); otherwise the flat tap-major index k = tap * Cin + c.
    const bool cm = FAST && (p.Cin % BKT == 0);
    auto gload = [&](int kt) {
        const int k0 = kt * BKT + g * 8;
        const int j0k = cm ? kt % p.taps : k0 / p.Cin, c0k = cm ? (kt / p.taps) * BKT + g * 8 : k0 - j0k * p.Cin;
        const int kh0 = (p.KW == p.taps) ? 0 : j0k / p.KW, kw0 = j0k - kh0 * p.KW;
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
                    const int j = cm ? kt % p.taps : k / p.Cin, c = cm ? (kt / p.taps) * BKT + kg * 8 + q : k - j * p.Cin;
                    const int kh = j / p.KW, kw = j - kh * p.KW;
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
    gemm_bf16_epilogue<TM_, TN_>(pp, acc, m0, n0, wm0, wn0, lane, bz, smem + wave * (TM_ >= 2 ? 32 * TM_ : 64) * (32 * TN_ + 8));
}




int osp_launch_gemm_reg(const GemmB& p, dim3 grid, int bm, int bn, bool b_kcontig, bool fast, bool bk32, hipStream_t stream) {
    osp_note_symbol(bm == 128 && bn == 128 ? "conv_gemm_bf16_kernel<128x128>" : bm == 128 ? "conv_gemm_bf16_kernel<128x64>" : "conv_gemm_bf16_kernel<64x64>");
#define OSP_LAUNCH_TILE(KC, F)                                                                                              \
    do {                                                                                                                    \
        if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, 64, F, 128, 128>), grid, dim3(256), 0, stream, p); \
        else if (bm == 128 && KC && bk32) hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, (KC ? 32 : 64), F, 128, 64>), grid, dim3(256), 0, stream, p);  \
        else if (bm == 128) hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, 64, F, 128, 64>), grid, dim3(256), 0, stream, p);  \
        else hipLaunchKernelGGL((conv_gemm_bf16_kernel<KC, 64, F, 64, 64>), grid, dim3(256), 0, stream, p);                 \
    } while (0)
    if (!b_kcontig) { if (fast) OSP_LAUNCH_TILE(false, true); else OSP_LAUNCH_TILE(false, false); }
    else { if (fast) OSP_LAUNCH_TILE(true, true); else OSP_LAUNCH_TILE(true, false); }
#undef OSP_LAUNCH_TILE
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
