// 8-wave 256x256 direct-to-LDS conv-GEMM, PHASED form (round 6): the same tile, LDS image, XOR swizzle, row maps and epilogues as
// conv_gemm_bf16_glds8e_kernel (gemm_bf16_glds.h), with a different main loop.  The lock-step loop of that kernel lets both waves of
// a SIMD run the same code at the same time -- their fragment reads queue at the LDS together and their MFMAs queue at the matrix pipe
// together (profiles/r06_pmc_mfma_busy.txt: pipe busy 39 %, waves issue-stalled 42 %).  Here the two wave groups of the workgroup
// (waves 0-3 / 4-7 = the M halves of the tile; the hardware places wave w and w + 4 on the same SIMD) run HALF A PHASE APART:
//
//   a K-tile (64 deep) is four phases, one 64 x 32 quadrant of the wave's 128 x 64 output each (8 MFMAs = 256 matrix-pipe clocks):
//     phase   fragment reads (ds_read_b128)          quadrant        LDS-DMA issued (one 16 KB unit = 2 instructions per wave)
//       0     A rows 0-63 (8), B cols 0-31 (4)       (0, 0)          B cols 32-63 of tile t + 1
//       1     B cols 32-63 (4)                       (0, 1)          A rows 64-127 of tile t + 1
//       2     A rows 64-127 (8)                      (1, 1)          A rows 0-63 of tile t + 2
//       3     --                                     (1, 0)          B cols 0-31 of tile t + 2
//   every phase is   [reads + requests + counted vmcnt]  s_barrier  [8 MFMAs at priority 1]  s_barrier ,   and group 1 enters the loop
//   one barrier behind group 0: while one wave of a SIMD issues MFMAs its partner reads fragments and requests operands.
//
// Two LDS buffers hold 1.5 - 2 tiles in flight because a unit is re-requested as soon as it is free: two phases after its last
// fragment read (the reads of phase p are consumed by the MFMAs of phase p; the lagging group finishes those before the barrier that
// lets the leading group into phase p + 2).  A unit is waited for -- by every wave, for its own requests: vmcnt(8) = four younger
// units may stay in flight -- in the read part of the phase BEFORE the phase that reads it, so a barrier passed by all eight waves
// lies between the last wave's wait and the first wave's read.  No __syncthreads() in the loop: its fence would drain the LDS-DMA
// queue (vmcnt(0)) at every barrier.
#include "gemm_bf16_glds.h"

struct TapState { int kh, kw, cb; };

template <bool SETPRIO>
__device__ __forceinline__ void conv_gemm_bf16_glds8p_body(const GemmB& pin, unsigned short* smem, const TileCtx tc) {
    const GemmB pp = gemm_select_phase(pin, tc.z);
    constexpr int BM_ = 256, BN_ = 256;
    unsigned short* As = smem;                       // [2][256][64]
    unsigned short* Bs = smem + 2 * BM_ * TBK;       // [2][256][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: SGPR
    const int wm = wave >> 2, wn = wave & 3;
    const int wm0 = wm * 128, wn0 = wn * 64;
    int mb_, nb_;
    xcd_tile(tc, mb_, nb_);
    const int m0 = mb_ * BM_, n0 = nb_ * BN_;
    const int64_t bz = pp.nphase > 0 ? 0 : tc.z;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(pp.A) + bz * pp.sAb;
    const unsigned short* B = reinterpret_cast<const unsigned short*>(pp.B) + bz * pp.sBb;
    const int Cin = pp.Cin, Tin = pp.Tin, Hin = pp.Hin, KW = pp.KW, a_tapstep = pp.a_tapstep, a_tapstep_h = pp.a_tapstep_h;
    const int taps = pp.taps, KH = taps / KW;
    const int64_t lda = pp.lda, sBtap = pp.sBtap, sBtap_h = pp.sBtap_h;
    const int nk = (taps * Cin) / TBK;
    const int rsub = lane >> 3, pslot = lane & 7;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(osp_zero_page);

    // ---- staging maps.  Unit (A, mh): tile rows {0..63} + 64 mh and {128..191} + 64 mh (what the two wave groups read in one phase);
    // unit (B, nh): tile columns 64 q + 32 nh + {0..31}, q = 0..3.  A unit is 128 rows = 16 pieces of 8 rows; wave w requests pieces
    // 2 w and 2 w + 1.  Index [2 * half + piece].
    int a_t[4], a_h[4], a_r0[4], b_r0[4]; int64_t a_off0[4], b_row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mh = q >> 1, u0 = 8 * (2 * wave + (q & 1));
        const int r0 = (u0 & 63) + (u0 >> 6) * 128 + mh * 64, r = r0 + rsub;
        a_r0[q] = r0;
        const int m = m0 + r;
        if (m < pp.M) {
            const int u = fd_div(m, pp.fd_trows), t = m - u * pp.Trows, th = fd_div(t, pp.fd_wrows), tw = t - th * pp.Wrows;
            a_t[q] = tw * pp.a_step + pp.a_off;
            a_h[q] = th * pp.a_step_h + pp.a_off_h;
            a_off0[q] = ((int64_t)u * Hin * Tin + (int64_t)a_h[q] * Tin + a_t[q]) * lda + (pslot ^ ((r >> 1) & 7)) * 8;
        } else { a_t[q] = -0x40000000; a_h[q] = 0; a_off0[q] = 0; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nh = q >> 1, u0 = 8 * (2 * wave + (q & 1));
        const int r0 = (u0 >> 5) * 64 + nh * 32 + (u0 & 31), r = r0 + rsub;
        b_r0[q] = r0;
        const int n = n0 + r;
        b_row[q] = n < pp.N ? (int64_t)n * pp.sBn + (pslot ^ ((r >> 1) & 7)) * 8 : -1;
    }
    // K order: channel block outer, tap inner (gemm_bf16_glds.h: consecutive tiles re-read the same input rows one tap on -- L2 hits)
    auto tap_next = [&](TapState s) {
        if (++s.kw == KW) { s.kw = 0; if (++s.kh == KH) { s.kh = 0; s.cb += TBK; } }
        return s;
    };
    auto stage_a = [&](int buf, auto mhc, const TapState s) {
        constexpr int mh = decltype(mhc)::value;
        const int dt = s.kw * a_tapstep, dh = s.kh * a_tapstep_h;                                  // wave-uniform
        const int64_t offA = ((int64_t)dh * Tin + dt) * lda + s.cb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * mh + i;
            const bool ok = (unsigned)(a_t[q] + dt) < (unsigned)Tin && (unsigned)(a_h[q] + dh) < (unsigned)Hin;
            const unsigned short* src = ok ? A + a_off0[q] + offA : zero;
            unsigned short* dst = As + buf * BM_ * TBK + a_r0[q] * TBK;                          // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto stage_b = [&](int buf, auto nhc, const TapState s) {
        constexpr int nh = decltype(nhc)::value;
        const int64_t offB = (int64_t)s.kh * sBtap_h + (int64_t)s.kw * sBtap + s.cb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * nh + i;
            const unsigned short* src = b_row[q] >= 0 ? B + b_row[q] + offB : zero;
            unsigned short* dst = Bs + buf * BN_ * TBK + b_r0[q] * TBK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // accumulators as 64-row halves (the epilogue is instantiated per half: gemm_bf16_glds.h)
    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    // ---- fragment reads: every fragment row of this lane is l31 (mod 32), so the swizzle term is one per lane
    const int l31 = lane & 31, lh = lane >> 5, sw = (l31 >> 1) & 7;
    const unsigned short* a_frag = As + (wm0 + l31) * TBK;
    const unsigned short* b_frag = Bs + (wn0 + l31) * TBK;
    bf16x8 a[4][2], b0[4], b1[4];
    auto read_a = [&](int buf, int mh) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[ks][i] = *reinterpret_cast<const bf16x8*>(a_frag + buf * BM_ * TBK + (mh * 64 + 32 * i) * TBK + (((2 * ks + lh) ^ sw) << 3));
    };
    auto read_b = [&](int buf, int nh, bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            b[ks] = *reinterpret_cast<const bf16x8*>(b_frag + buf * BN_ * TBK + (nh * 32) * TBK + (((2 * ks + lh) ^ sw) << 3));
    };
    auto quad = [&](f32x16 (&acc)[2][2], int nh, const bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks], acc[i][nh], 0, 0, 0);
    };
#define W8P_MID()                                                                                      \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); \
    if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);
#define W8P_END()                                                                                      \
    if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);                                              \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: tile 0 whole, the first two units of tile 1 (the steady state's request order)
    TapState s1 = {0, 0, 0};                         // tap state of tile t + 1 (t = the tile being computed), s2: tile t + 2
    stage_a(0, I0{}, s1); stage_b(0, I0{}, s1); stage_b(0, I1{}, s1); stage_a(0, I1{}, s1);
    s1 = tap_next(s1);
    TapState s2 = tap_next(s1);
    if (nk > 1) {
        stage_a(1, I0{}, s1); stage_b(1, I0{}, s1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    auto tile = [&](auto bufc, int t) {
        constexpr int buf = decltype(bufc)::value;
        const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
        // phase 0
        read_a(buf, 0); read_b(buf, 0, b0);
        if (n1) { stage_b(buf ^ 1, I1{}, s1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        W8P_MID();
        quad(acc0, 0, b0);
        W8P_END();
        // phase 1
        read_b(buf, 1, b1);
        if (n1) { stage_a(buf ^ 1, I1{}, s1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W8P_MID();
        quad(acc0, 1, b1);
        W8P_END();
        // phase 2
        read_a(buf, 1);
        if (n2) stage_a(buf, I0{}, s2);
        W8P_MID();
        quad(acc1, 1, b1);
        W8P_END();
        // phase 3
        if (n2) { stage_b(buf, I0{}, s2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (n1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        W8P_MID();
        quad(acc1, 0, b0);
        W8P_END();
        s1 = s2; s2 = tap_next(s2);
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) { tile(I0{}, t); tile(I1{}, t + 1); }
    if (t < nk) tile(I0{}, t);
#undef W8P_MID
#undef W8P_END
    if (wm == 0) __builtin_amdgcn_s_barrier();      // group 0 catches the barrier count up
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                // every wave is done with the operand buffers: the epilogue stages through them
    constexpr int SP_ = 32 * 2 + 8;
    gemm_bf16_epilogue<2, 2>(pp, acc0, m0, n0, wm0, wn0, lane, bz, smem + wave * 128 * SP_);
    gemm_bf16_epilogue<2, 2>(pp, acc1, m0, n0, wm0 + 64, wn0, lane, bz, smem + wave * 128 * SP_ + 64 * SP_);
}

__global__ __launch_bounds__(512) void conv_gemm_bf16_glds8p_kernel(const GemmB pp) {
    conv_gemm_bf16_glds8p_body<true>(pp, glds_smem, grid_tile_ctx());
}
// (the same body without the priority flips, SETPRIO = false: measured equal, 130.0 vs 131.3 us at 204 tiles; not instantiated)

int osp_launch_glds8q(const GemmB& p, dim3 grid, hipStream_t stream);          // 1: taken, 0: declined
int osp_launch_glds8r(const GemmB& p, dim3 grid, hipStream_t stream);          // 1: taken, 0: declined

int osp_launch_glds8p(const GemmB& p, dim3 grid, int variant, hipStream_t stream) {
    static int done = 0;
    if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds8p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS8_LDS);
        done = 1;
    }
    if (variant == 3 && osp_launch_glds8r(p, grid, stream)) return OSP_OK;      // gemm_bf16_w8r.hip: tap reuse (stride-1, 5-tap problems; declines the rest)
    if (variant >= 2 && osp_launch_glds8q(p, grid, stream)) return OSP_OK;      // gemm_bf16_w8q.hip (declines what it cannot address)
    osp_note_symbol("conv_gemm_bf16_glds8p_kernel");
    hipLaunchKernelGGL(conv_gemm_bf16_glds8p_kernel, grid, dim3(512), GLDS8_LDS, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
