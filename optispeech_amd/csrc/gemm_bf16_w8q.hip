// 8-wave 256x256 direct-to-LDS conv-GEMM, phased form with a ROTATING unit schedule (round 6, second version of gemm_bf16_w8p.hip --
// read that file's header first: same tile, LDS image, wave groups half a phase apart, same epilogues).  What changed:
//
//  * fragment reads balanced over the phases (8 / 4 / 8 / 4 ds_read_b128 instead of 12 / 4 / 8 / 0): the B fragments of columns 0-31
//    of tile t + 1 are read in phase 3 of tile t into the registers tile t's columns 32-63 were in (dead after phase 2);
//  * one uniform rule for the LDS-DMA units.  Number the phases globally, g = 4 t + p; phase g READS one unit
//        p = 0: A rows 0-63 of tile t     1: B cols 32-63 of t     2: A rows 64-127 of t     3: B cols 0-31 of t + 1
//    REQUESTS the unit that phase g + 6 will read -- into the slot read by phase g - 2, free since every wave passed two barriers --
//    and WAITS (vmcnt(10): the five younger units stay in flight) for the unit phase g + 1 reads.  Every unit has five phases between
//    request and wait; six units (96 KB of the 128 KB of operand buffers) are in flight or landed-and-unread at any time;
//  * operand addresses are 32-bit byte offsets against buffer descriptors (raw_ptr_buffer_load_lds): the per-request address work is
//    one v_add + one select on a precomputed per-row tap-validity bit instead of three 64-bit adds, two range tests and two selects;
//    a request for a padding row / a row beyond M or N carries an out-of-range offset and the DMA writes zeros (no zero page).
//    The host launcher declines problems whose operands are not addressable that way (>= 2 GiB, more than 32 taps).
#include "gemm_bf16_glds.h"

struct TapStateQ { int kh, kw, cb, j; };

__device__ __forceinline__ void conv_gemm_bf16_glds8q_body(const GemmB& pin, unsigned short* smem, const TileCtx tc) {
    const GemmB pp = gemm_select_phase(pin, tc.z);
    constexpr int BM_ = 256, BN_ = 256;
    constexpr unsigned OOB = 0x80000000u;
    unsigned short* As = smem;                       // [2][256][64]
    unsigned short* Bs = smem + 2 * BM_ * TBK;       // [2][256][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int lda = (int)pp.lda, sBtap = (int)pp.sBtap, sBtap_h = (int)pp.sBtap_h;
    const int nk = (taps * Cin) / TBK;
    const int rsub = lane >> 3, pslot = lane & 7;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(B), 0, 0x7fffffff, 0x00020000);

    // ---- staging maps (units and pieces as in gemm_bf16_w8p.hip).  Per owned row: the BYTE offset of (tap 0, channel 0) -- modulo
    // 2^32, it may lie in front of the tensor -- and one validity bit per tap.
    unsigned a_off32[4], a_mask[4], b_off32[4]; int a_r0[4], b_r0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mh = q >> 1, u0 = 8 * (2 * wave + (q & 1));
        const int r0 = (u0 & 63) + (u0 >> 6) * 128 + mh * 64, r = r0 + rsub;
        a_r0[q] = r0;
        const int m = m0 + r;
        unsigned mask = 0, off = 0;
        if (m < pp.M) {
            const int u = fd_div(m, pp.fd_trows), t = m - u * pp.Trows, th = fd_div(t, pp.fd_wrows), tw = t - th * pp.Wrows;
            const int at = tw * pp.a_step + pp.a_off, ah = th * pp.a_step_h + pp.a_off_h;
            off = (unsigned)((((int64_t)u * Hin * Tin + (int64_t)ah * Tin + at) * lda + (pslot ^ ((r >> 1) & 7)) * 8) * 2);
            int kh = 0, kw = 0;
            for (int j = 0; j < taps; ++j) {
                const bool ok = (unsigned)(at + kw * a_tapstep) < (unsigned)Tin && (unsigned)(ah + kh * a_tapstep_h) < (unsigned)Hin;
                mask |= (ok ? 1u : 0u) << j;
                if (++kw == KW) { kw = 0; ++kh; }
            }
        }
        a_off32[q] = off; a_mask[q] = mask;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nh = q >> 1, u0 = 8 * (2 * wave + (q & 1));
        const int r0 = (u0 >> 5) * 64 + nh * 32 + (u0 & 31), r = r0 + rsub;
        b_r0[q] = r0;
        const int n = n0 + r;
        b_off32[q] = n < pp.N ? (unsigned)(((int64_t)n * pp.sBn + (pslot ^ ((r >> 1) & 7)) * 8) * 2) : OOB;   // OOB + (< 2^31) stays out of range
    }
    // K order: channel block outer, tap inner
    auto tap_next = [&](TapStateQ s) {
        ++s.j;
        if (++s.kw == KW) { s.kw = 0; if (++s.kh == KH) { s.kh = 0; s.j = 0; s.cb += TBK; } }
        return s;
    };
    auto stage_a = [&](int buf, auto mhc, const TapStateQ s) {
        constexpr int mh = decltype(mhc)::value;
        const unsigned offA = (unsigned)(((s.kh * a_tapstep_h * Tin + s.kw * a_tapstep) * lda + s.cb) * 2);   // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * mh + i;
            const unsigned voff = ((a_mask[q] >> s.j) & 1u) ? a_off32[q] + offA : OOB;
            unsigned short* dst = As + buf * BM_ * TBK + a_r0[q] * TBK;                          // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
        }
    };
    auto stage_b = [&](int buf, auto nhc, const TapStateQ s) {
        constexpr int nh = decltype(nhc)::value;
        const unsigned offB = (unsigned)((s.kh * sBtap_h + s.kw * sBtap + s.cb) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * nh + i;
            unsigned short* dst = Bs + buf * BN_ * TBK + b_r0[q] * TBK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)dst, 16, b_off32[q] + offB, 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int l31 = lane & 31, lh = lane >> 5, sw = (l31 >> 1) & 7;
    const unsigned short* a_frag = As + (wm0 + l31) * TBK;
    const unsigned short* b_frag = Bs + (wn0 + l31) * TBK;
    bf16x8 a[4][2], bx[4], by[4];
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
    // the wait of global phase g (after its request): the unit phase g + 1 reads must have landed; `rem` = how many younger units exist
    const int g_last = 4 * (nk - 1) + 2;             // the last unit (A rows 64-127 of the last tile) is read in this phase
    auto wait_for_next = [&](int g) {
        const int rem = g_last - (g + 1);
        if (rem >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (rem == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (rem == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#define W8Q_READS_DONE() __builtin_amdgcn_sched_barrier(0);
#define W8Q_MID()                                                                                      \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_setprio(1);
#define W8Q_END()                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                     \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: the units phases -1 .. 5 read, in that order
    TapStateQ s1 = {0, 0, 0, 0};                     // tap state of tile t + 1 while tile t is computed; s2: tile t + 2
    stage_b(0, I0{}, s1); stage_a(0, I0{}, s1); stage_b(0, I1{}, s1); stage_a(0, I1{}, s1);
    s1 = tap_next(s1);
    TapStateQ s2 = tap_next(s1);
    if (nk > 1) {
        stage_b(1, I0{}, s1); stage_a(1, I0{}, s1); stage_b(1, I1{}, s1);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_b(0, 0, bx);                               // "phase -1"
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    // one K-tile; X holds its B columns 0-31 on entry, Y receives columns 32-63 and then the next tile's columns 0-31
    auto tile = [&](auto bufc, int t, bf16x8 (&X)[4], bf16x8 (&Y)[4]) {
        constexpr int buf = decltype(bufc)::value;
        const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
        const int g = 4 * t;
        // phase 0
        read_a(buf, 0);
        W8Q_READS_DONE();
        if (n1) stage_a(buf ^ 1, I1{}, s1);
        wait_for_next(g);
        W8Q_MID();
        quad(acc0, 0, X);
        W8Q_END();
        // phase 1
        read_b(buf, 1, Y);
        W8Q_READS_DONE();
        if (n2) stage_b(buf, I0{}, s2);
        wait_for_next(g + 1);
        W8Q_MID();
        quad(acc0, 1, Y);
        W8Q_END();
        // phase 2
        read_a(buf, 1);
        W8Q_READS_DONE();
        if (n2) stage_a(buf, I0{}, s2);
        wait_for_next(g + 2);
        W8Q_MID();
        quad(acc1, 1, Y);
        W8Q_END();
        // phase 3
        if (n1) read_b(buf ^ 1, 0, Y);
        W8Q_READS_DONE();
        if (n2) stage_b(buf, I1{}, s2);
        wait_for_next(g + 3);
        W8Q_MID();
        quad(acc1, 0, X);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // phase 3's reads feed the NEXT phase's MFMAs: retire them inside their own phase (the two-phase re-request rule)
        W8Q_END();
        s1 = s2; s2 = tap_next(s2);
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) { tile(I0{}, t, bx, by); tile(I1{}, t + 1, by, bx); }
    if (t < nk) tile(I0{}, t, bx, by);
#undef W8Q_READS_DONE
#undef W8Q_MID
#undef W8Q_END
    if (wm == 0) __builtin_amdgcn_s_barrier();      // group 0 catches the barrier count up
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                // every wave is done with the operand buffers: the epilogue stages through them
    constexpr int SP_ = 32 * 2 + 8;
    gemm_bf16_epilogue<2, 2>(pp, acc0, m0, n0, wm0, wn0, lane, bz, smem + wave * 128 * SP_);
    gemm_bf16_epilogue<2, 2>(pp, acc1, m0, n0, wm0 + 64, wn0, lane, bz, smem + wave * 128 * SP_ + 64 * SP_);
}

__global__ __launch_bounds__(512) void conv_gemm_bf16_glds8q_kernel(const GemmB pp) {
    conv_gemm_bf16_glds8q_body(pp, glds_smem, grid_tile_ctx());
}

// 1: launched; 0: declined (the caller takes the 64-bit-pointer kernel)
int osp_launch_glds8q(const GemmB& p, dim3 grid, hipStream_t stream) {
    // every byte offset the kernel forms must stay below 2^31: the activation tensor, the weight tensor, and a tap-validity bit per tap
    const int np = p.nphase > 0 ? p.nphase : 1;
    int tmax = 0;
    for (int i = 0; i < np; ++i) { const int tp = p.nphase > 0 ? p.ph[i].taps : p.taps; tmax = tp > tmax ? tp : tmax; }
    if (tmax > 32) return 0;
    const int64_t rows_in = (int64_t)(p.M / (p.Trows > 0 ? p.Trows : 1) + 1) * p.Hin * p.Tin;
    const int64_t a_bytes = (rows_in + 1) * p.lda * 2;
    const int64_t kh = p.KW > 0 ? p.taps / p.KW : 1;
    int64_t b_max = 0;                               // (phases of a fused dgrad shift the weight base by b_off)
    for (int i = 0; i < np; ++i) { const int64_t bo = p.nphase > 0 ? p.ph[i].b_off : 0; b_max = bo > b_max ? bo : b_max; }
    const int64_t b_bytes = ((int64_t)p.N * p.sBn + kh * (p.sBtap_h > 0 ? p.sBtap_h : 0) + (int64_t)p.KW * p.sBtap + p.Cin + b_max) * 2;
    if (a_bytes >= (int64_t)0x7fff0000 || b_bytes >= (int64_t)0x7fff0000 || a_bytes <= 0 || b_bytes <= 0) return 0;
    static int done = 0;
    if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds8q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS8_LDS);
        done = 1;
    }
    osp_note_symbol("conv_gemm_bf16_glds8q_kernel");
    hipLaunchKernelGGL(conv_gemm_bf16_glds8q_kernel, grid, dim3(512), GLDS8_LDS, stream, p);
    OSP_LAUNCH_CHECK();
    return 1;
}
