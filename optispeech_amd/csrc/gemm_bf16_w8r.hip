// 8-wave 256x256 phased conv-GEMM with TAP REUSE of the activation panel (round 6, third member after gemm_bf16_w8p.hip / _w8q.hip: read
// those headers first -- same tile, wave groups half a phase apart, B units and epilogues as gemm_bf16_w8q.hip).
//
// A stride-1 conv reads, for tap j, the input rows of tap j - 1 shifted by one: the per-tap kernels fetch the 256-row activation tile of a
// channel block once per TAP (L2 hits, but 32 KB of LDS-DMA each).  Here the rows of one channel block are staged ONCE, as a PANEL
//     panel row  q = r + (taps - 1) * seg(r) + jj          r = tile row, seg(r) = which utterance of the tile r lies in, jj = tap
// -- consecutive tile rows of one utterance on consecutive panel rows, taps - 1 halo rows per utterance (zeros outside [0, Tin)), so tap jj
// of tile row r is panel row p_base(r) + jj -- and the fragment reads of a tap are the panel's rows at offset jj.  Per 5-tap channel block
// the LDS-DMA volume is 40 KB (panel) + 5 x 32 KB (weights) instead of 5 x 64 KB: -40 %.  profiles/r06_w8q_knockout.txt bounds what that
// can buy at 13 % per launch (a build of the per-tap kernel that requests the activation units for tap 0 only).
//
// Schedule of K-tile t = (channel block cb, tap j), B units exactly as in gemm_bf16_w8q.hip:
//     phase   fragment reads                                   LDS-DMA requested
//       0     A rows 0-63 from panel (cb & 1) at offset jj      piece j of panel cb + 1 (8 rows per wave: 40 pieces = 8 waves x 5 taps)
//       1     B cols 32-63 of t                                 B cols 0-31 of t + 2
//       2     A rows 64-127 from the panel                      --
//       3     B cols 0-31 of t + 1                              B cols 32-63 of t + 2
// Requests complete in order, so the counted waits are: phase 0 vmcnt(6) (B cols 32-63 of t landed; younger: two panel pieces, two B units),
// phase 2 vmcnt(5) (B cols 0-31 of t + 1), phase 3 of a channel block's LAST tap vmcnt(4) (the whole next panel); in the last channel block
// (no panel requests) 4 / 4, and vmcnt(0) once the B requests stop.  A panel buffer is rewritten two phases after its last read (phase 2 of
// the previous channel block's last tap -> phase 0 of this one's first).
// Taken for: 1-D row maps, input step 1, tap step +-1 (forward and the stride-1 dgrad), exactly 5 taps, utterances of >= 17 rows (at most 16
// utterances in a tile: 256 + 16 x 4 = 320 panel rows), operands addressable with 31-bit byte offsets.  Anything else: gemm_bf16_w8q.hip.
// Same k-slabs in the same order into the same accumulators: bit-identical to the other 8-wave kernels (tests/test_gpu_gemm_w8p.py).
#include "gemm_bf16_glds.h"

__device__ __forceinline__ void conv_gemm_bf16_glds8r_body(const GemmB& pin, unsigned short* smem, const TileCtx tc) {
    const GemmB pp = gemm_select_phase(pin, tc.z);
    constexpr int BM_ = 256, BN_ = 256, TAPS = 5, PROWS = 320;
    constexpr unsigned OOB = 0x80000000u;
    unsigned short* Ps = smem;                       // [2][PROWS][64]   (2 x 40 KB)
    unsigned short* Bs = smem + 2 * PROWS * TBK;     // [2][256][64]     (2 x 32 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int wm0 = wm * 128, wn0 = wn * 64;
    int mb_, nb_;
    xcd_tile(tc, mb_, nb_);
    const int m0 = mb_ * BM_, n0 = nb_ * BN_;
    const int64_t bz = pp.nphase > 0 ? 0 : tc.z;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(pp.A) + bz * pp.sAb;
    const unsigned short* B = reinterpret_cast<const unsigned short*>(pp.B) + bz * pp.sBb;
    const int Cin = pp.Cin, Tin = pp.Tin, Trows = pp.Trows, ts = pp.a_tapstep;
    const int lda = (int)pp.lda, sBtap = (int)pp.sBtap;
    const int ncb = Cin / TBK, nk = TAPS * ncb;
    const int rsub = lane >> 3, pslot = lane & 7;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(B), 0, 0x7fffffff, 0x00020000);
    // input position of (tile row r, tap j): t * 1 + a_off + j * ts = t + a_off_eff + jj with jj = j (ts = +1) or TAPS - 1 - j (ts = -1)
    const int a_off_eff = ts > 0 ? pp.a_off : pp.a_off - (TAPS - 1);
    const int u0 = fd_div(m0, pp.fd_trows), t0 = m0 - u0 * Trows, nutt = pp.M / Trows;
    const int L = Trows + (TAPS - 1), S1 = Trows - t0 + (TAPS - 1);      // panel rows per whole utterance; panel row where utterance 1 starts

    // ---- panel pieces of this lane: piece k = panel rows 8 (5 wave + k) + rsub, k = the tap of the tile that requests it.  With e = panel row
    // + t0, the row belongs to utterance u0 + e / L at input position a_off_eff + e % L: the state (pr = e % L, ps = e / L, prow = byte offset
    // of that input row) ADVANCES by 8 rows per piece and is reset at the end of a channel block -- a table of five offsets indexed by the
    // run-time tap went to scratch, and a scratch load among the LDS-DMA requests drains them (it is an ordinary VMEM load).
    const int pq0 = 8 * (TAPS * wave) + rsub;
    int pr0 = pq0 + t0, ps0 = 0;
    while (pr0 >= L) { pr0 -= L; ++ps0; }
    const unsigned prow0 = (unsigned)((((int64_t)(u0 + ps0) * Tin + (a_off_eff + pr0)) * lda) * 2);
    const unsigned p_adv = (unsigned)(8 * lda * 2), p_wrap = (unsigned)((Tin - L) * lda * 2);
    int pr = pr0, ps = ps0; unsigned prow = prow0;
    unsigned b_off32[4]; int b_r0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nh = q >> 1, u8 = 8 * (2 * wave + (q & 1));
        const int r0 = (u8 >> 5) * 64 + nh * 32 + (u8 & 31), r = r0 + rsub;
        b_r0[q] = r0;
        const int n = n0 + r;
        b_off32[q] = n < pp.N ? (unsigned)(((int64_t)n * pp.sBn + (pslot ^ ((r >> 1) & 7)) * 8) * 2) : OOB;
    }
    // piece k (run-time, wave-uniform: the requesting tile's tap) of the panel of channel offset cb (elements) into panel buffer pbuf;
    // then the state moves on to piece k + 1 (k = 4: back to piece 0)
    auto stage_panel_piece = [&](int pbuf, int k, int cb, bool issue) __attribute__((always_inline)) {
        if (issue) {
            unsigned short* dst = Ps + pbuf * PROWS * TBK + 8 * (TAPS * wave + k) * TBK;        // wave-uniform
            const bool ok = u0 + ps < nutt && (unsigned)(a_off_eff + pr) < (unsigned)Tin;
            const unsigned voff = ok ? prow + (unsigned)((pslot ^ (((pq0 >> 1) + 4 * k) & 7)) << 4) + (unsigned)(cb * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
        }
        if (k == TAPS - 1) { pr = pr0; ps = ps0; prow = prow0; }
        else {
            pr += 8; prow += p_adv;
            if (pr >= L) { pr -= L; ++ps; prow += p_wrap; }
        }
    };
    auto stage_b = [&](int buf, auto nhc, int j, int cb) __attribute__((always_inline)) {
        constexpr int nh = decltype(nhc)::value;
        const unsigned offB = (unsigned)((j * sBtap + cb) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * nh + i;
            unsigned short* dst = Bs + buf * BN_ * TBK + b_r0[q] * TBK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)dst, 16, b_off32[q] + offB, 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;

    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    // ---- fragment rows: panel row of (tile row, tap 0) for this lane's four 32-row blocks [2 mh + i]
    const int l31 = lane & 31, lh = lane >> 5, swb = (l31 >> 1) & 7;
    int p_base[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int r = wm0 + 32 * blk + l31;
        const int seg = fd_div(m0 + r, pp.fd_trows) - u0;
        p_base[blk] = r + (TAPS - 1) * seg;
    }
    const unsigned short* b_frag = Bs + (wn0 + l31) * TBK;
    bf16x8 a[4][2], bx[4], by[4];
    auto read_a = [&](int pbuf, int mh, int jj) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = p_base[2 * mh + i] + jj;
            const int x = ((q >> 1) & 7) ^ lh;
            const unsigned short* row = Ps + pbuf * PROWS * TBK + q * TBK;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks][i] = *reinterpret_cast<const bf16x8*>(row + ((x ^ (2 * ks)) << 3));
        }
    };
    auto read_b = [&](int buf, int nh, bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            b[ks] = *reinterpret_cast<const bf16x8*>(b_frag + buf * BN_ * TBK + (nh * 32) * TBK + (((2 * ks + lh) ^ swb) << 3));
    };
    auto quad = [&](f32x16 (&acc)[2][2], int nh, const bf16x8 (&b)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks], acc[i][nh], 0, 0, 0);
    };
#define W8R_READS_DONE() __builtin_amdgcn_sched_barrier(0);
#define W8R_MID()                                                                                      \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_setprio(1);
#define W8R_END()                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                     \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: panel 0 whole, the B units of tiles 0 and 1; everything landed before the loop starts
    for (int k = 0; k < TAPS; ++k) stage_panel_piece(0, k, 0, true);
    stage_b(0, I0{}, 0, 0); stage_b(0, I1{}, 0, 0);
    stage_b(1, I0{}, 1, 0); stage_b(1, I1{}, 1, 0);                          // tile 1 = (cb 0, tap 1): TAPS = 5 > 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_b(0, 0, bx);                               // "phase -1"
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    // one K-tile t = (cb, j); buf = t & 1 = its B buffer (compile time); X holds its B columns 0-31 on entry.  (cb, j) are run-time
    // scalars: ten unrolled tiles with compile-time taps spilled 79-623 registers (the fragment addresses of every copy were hoisted)
    auto tile = [&](auto bufc, int t, int cb, int j, bf16x8 (&X)[4], bf16x8 (&Y)[4]) {
        constexpr int buf = decltype(bufc)::value;
        const bool n1 = t + 1 < nk, n2 = t + 2 < nk, pnext = cb + 1 < ncb;
        const int jj = ts > 0 ? j : TAPS - 1 - j, pbuf = cb & 1;
        const int j2 = j + 2 >= TAPS ? j + 2 - TAPS : j + 2;                 // (tap, channel offset) of tile t + 2
        const int cb2 = (j + 2 >= TAPS ? cb + 1 : cb) * TBK;
        // phase 0
        read_a(pbuf, 0, jj);
        W8R_READS_DONE();
        stage_panel_piece(pbuf ^ 1, j, (cb + 1) * TBK, pnext);
        if (!n2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (pnext) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        W8R_MID();
        quad(acc0, 0, X);
        W8R_END();
        // phase 1
        read_b(buf, 1, Y);
        W8R_READS_DONE();
        if (n2) stage_b(buf, I0{}, j2, cb2);
        W8R_MID();
        quad(acc0, 1, Y);
        W8R_END();
        // phase 2
        read_a(pbuf, 1, jj);
        W8R_READS_DONE();
        if (!n2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (pnext) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        W8R_MID();
        quad(acc1, 1, Y);
        W8R_END();
        // phase 3
        if (n1) read_b(buf ^ 1, 0, Y);
        W8R_READS_DONE();
        if (n2) stage_b(buf, I1{}, j2, cb2);
        if (j == TAPS - 1) {                                                 // the next channel block's panel is read from the next phase on
            if (!n2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        W8R_MID();
        quad(acc1, 0, X);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // phase 3's reads feed the next phase's MFMAs: retire them inside their own phase
        W8R_END();
    };
    int cb = 0, jt = 0, t = 0;
#define W8R_ADV() do { ++t; if (++jt == TAPS) { jt = 0; ++cb; } } while (0)
    for (; t + 1 < nk;) {
        tile(I0{}, t, cb, jt, bx, by); W8R_ADV();
        tile(I1{}, t, cb, jt, by, bx); W8R_ADV();
    }
    if (t < nk) tile(I0{}, t, cb, jt, bx, by);
#undef W8R_ADV
#undef W8R_READS_DONE
#undef W8R_MID
#undef W8R_END
    if (wm == 0) __builtin_amdgcn_s_barrier();      // group 0 catches the barrier count up
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                // every wave is done with the operand buffers: the epilogue stages through them
    constexpr int SP_ = 32 * 2 + 8;
    gemm_bf16_epilogue<2, 2>(pp, acc0, m0, n0, wm0, wn0, lane, bz, smem + wave * 128 * SP_);
    gemm_bf16_epilogue<2, 2>(pp, acc1, m0, n0, wm0 + 64, wn0, lane, bz, smem + wave * 128 * SP_ + 64 * SP_);
}

__global__ __launch_bounds__(512) void conv_gemm_bf16_glds8r_kernel(const GemmB pp) {
    conv_gemm_bf16_glds8r_body(pp, glds_smem, grid_tile_ctx());
}

// 1: launched; 0: declined (the caller takes gemm_bf16_w8q.hip)
int osp_launch_glds8r(const GemmB& p, dim3 grid, hipStream_t stream) {
    if (p.nphase > 1) return 0;
    const int taps = p.nphase == 1 ? p.ph[0].taps : p.taps, KW = p.nphase == 1 ? p.ph[0].KW : p.KW;
    const int Trows = p.nphase == 1 ? p.ph[0].Trows : p.Trows, Wrows = p.nphase == 1 ? p.ph[0].Wrows : p.Wrows;
    const int M = p.nphase == 1 ? p.ph[0].M : p.M;
    if (!(taps == 5 && KW == 5 && p.Hin == 1 && Wrows == Trows && p.a_step == 1 && (p.a_tapstep == 1 || p.a_tapstep == -1))) return 0;
    if (Trows < 17 || M % Trows != 0 || p.Cin % TBK != 0) return 0;
    const int64_t rows_in = (int64_t)(M / Trows + 1) * p.Tin;
    const int64_t a_bytes = (rows_in + 1) * p.lda * 2;
    const int64_t b_off = p.nphase == 1 ? p.ph[0].b_off : 0;
    const int64_t b_bytes = ((int64_t)p.N * p.sBn + (int64_t)KW * p.sBtap + p.Cin + b_off) * 2;
    if (a_bytes >= (int64_t)0x7fff0000 || b_bytes >= (int64_t)0x7fff0000 || a_bytes <= 0 || b_bytes <= 0) return 0;
    static int done = 0;
    if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds8r_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS8_LDS);
        done = 1;
    }
    osp_note_symbol("conv_gemm_bf16_glds8r_kernel");
    hipLaunchKernelGGL(conv_gemm_bf16_glds8r_kernel, grid, dim3(512), GLDS8_LDS, stream, p);
    OSP_LAUNCH_CHECK();
    return 1;
}
