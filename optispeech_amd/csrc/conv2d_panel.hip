// Tap-reuse ("kw-panel") conv-GEMM for the 64 <- 64 channel DiscriminatorR layers.
//
// Reference op: the Conv2d(64, 64, (5, 3) / (3, 3), stride (2, 1) / (2, 2)) layers of DiscriminatorR and their input gradients
// (optispeech/model/vocoder/wavenext/disc/_discriminators.py:139-194); in this package's channels-last (U, frames, bins, 64)
// orientation the kernel is (KH = k_time, KW = k_freq) and the bin axis -- the fastest row index -- has stride 2.
//
// What conv_gemm_bf16_glds_n64_kernel (gemm_bf16_glds.h) does per TAP -- stage the 128 tap-shifted activation rows of the tile
// (16 KB) and the tap's weights (8 KB), wait, barrier, 8 MFMAs per wave -- this kernel does per KERNEL ROW kh:
//   * ONE panel of the input rows the tile touches for that kh is staged by LDS-DMA: for every (utterance, frame) line segment of
//     the tile the bins  tw_lo * SW + a_off .. tw_hi * SW + a_off + KW - 1,  i.e. 128 * SW + (KW - SW) * segments rows (<= 288)
//     instead of KW x 128; the KW taps of the kernel row then read the panel at row offsets 0 .. KW - 1.  Output row r of the tile
//     (segment s(r)) and tap kw use panel row  r * SW + s(r) * (KW - SW) + kw  -- a per-lane base computed once per tile plus a
//     wave-uniform tap offset; a line boundary inside the tile is the (KW - SW)-row gap of the verdict's sketch.
//   * SW = 2: a tap reads every second panel row, which would put all 16 lanes of a ds_read_b128 group on one half of the banks,
//     so the panel is stored de-interleaved (even panel rows in the first half, odd ones in the second: a tap's rows are then
//     consecutive physical rows) with the 16-byte-slot XOR swizzle of the glds kernels applied to the PHYSICAL row, on the source
//     address when staging and on the read address (both-sides rule).
//   * weights: 8 KB per tap, double-buffered, requested one whole tap ahead (L2-resident: 120 KB per layer).
// Per tile of 128 rows and per kh the LDS-DMA writes drop from KW x 16 KB to <= 36 KB (2.3x at KW = 5, 1.4x at KW = 3), the
// activation-row address arithmetic from KW x 4 to 9 rows per thread, and the panel wait happens KH times per tile instead of
// KH x KW.  LDS: 36 KB panel + 2 x 8 KB weights = 52 KB -> three workgroups per CU, as the kernel it replaces.
// Accumulator layout, tile shape (128 x 64, four waves of 64 x 32) and epilogues are those of the glds n64 kernel, so every
// epilogue (bias + LeakyReLU forward; LeakyReLU' + feature-matching addend in the fused-phase dgrad) is shared code.
#include "gemm_bf16_glds.h"
#include <vector>
#include <stdio.h>

#define PANEL_ROWS 288
#define PANEL_LDS ((PANEL_ROWS * TBK + 2 * 64 * TBK) * 2)

// TIMING (diagnostic instantiation, OSP_PANEL_TIMING=<file>): waves 0 and 3 of every 16th tile record s_memtime at the loop's wait points
__device__ unsigned long long* osp_panel_dbg = nullptr;
#define PANEL_DBG_EVENTS 72
#define PT_MARK() do { if constexpr (TIMING) { if (dbg && dbg_n < PANEL_DBG_EVENTS) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) dbg[dbg_n] = t_; ++dbg_n; } } } while (0)

// PIPE (second form, OSP_N64_PANEL=2): the loop software-pipelined inside each wave -- the fragments of tap t + 1 are read while tap t's
// MFMAs issue (two fragment register sets, the tap loop unrolled by two), on a ring of FOUR weight stages so that a tap's weights were
// published one barrier before the tap that prefetches them: LDS reads, MFMAs and the LDS-DMA wait of a wave overlap instead of
// following each other (section 10.1's timeline: 0.9 k + 0.45 k + 0.37 k + 0.45 k clocks per tap, serial).  ~200 VGPRs and 68 KB of
// LDS: two workgroups per CU.
#define PANEL_LDS_PIPE ((PANEL_ROWS * TBK + 4 * 64 * TBK) * 2)

template <int SW, bool TIMING = false, bool PIPE = false>
__global__ __launch_bounds__(256, PIPE ? 2 : 3) void conv2d_panel_n64_kernel(const GemmB pin) {
    const TileCtx tc = grid_tile_ctx();
    const GemmB pp = gemm_select_phase(pin, tc.z);
    constexpr int BM = 128, PR = PANEL_ROWS, HALF = PR / 2, NPR = PR / 32;      // NPR: panel rows staged per thread
    int mb_, nb_;
    xcd_tile(tc, mb_, nb_);
    const int m0 = mb_ * BM;
    if (m0 >= pp.M) return;                                     // (a fused-dgrad phase with fewer rows than the grid's maximum)
    unsigned short* Pn = glds_smem;                             // [PR][64]   physical rows
    unsigned short* Bs = glds_smem + PR * TBK;                  // [2 | 4][64][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
    const int l31 = lane & 31, lh = lane >> 5, rsub = lane >> 3, pslot = lane & 7;
    unsigned long long* dbg = nullptr; int dbg_n = 0;
    if constexpr (TIMING) {
        if (osp_panel_dbg && tc.z == 0 && (mb_ & 15) == 0 && (mb_ >> 4) < 64 && (wave == 0 || wave == 3))
            dbg = osp_panel_dbg + ((mb_ >> 4) * 2 + (wave == 3)) * PANEL_DBG_EVENTS;
    }
    PT_MARK();                                                  // 0: start
    const unsigned short* A = reinterpret_cast<const unsigned short*>(pp.A);
    const unsigned short* B = reinterpret_cast<const unsigned short*>(pp.B);
    const int Wo = pp.Wrows, KW = pp.KW, KH = pp.taps / pp.KW, Tin = pp.Tin, Hin = pp.Hin;
    const int tsw = pp.a_tapstep, tsh = pp.a_tapstep_h;         // +1 (forward) or -1 (dgrad: taps walk backwards over dy)
    const int G = KW - SW;                                      // panel rows between two line segments
    const int line0 = fd_div(m0, pp.fd_wrows), tw_first = m0 - line0 * Wo, r1 = Wo - tw_first;
    const int pb1 = r1 * SW + G, Lp = Wo * SW + G;
    const int nlines = pp.M / Wo;                               // (M = U * Ho * Wo)
    const int Ho = pp.Trows / Wo;
    const int w_base = pp.a_off + (tsw < 0 ? -(KW - 1) : 0);
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(osp_zero_page);

    // ---- panel staging map: thread owns physical rows q = 8 * (wave + 4 i) + rsub, i < NPR
    int a_off[NPR], a_h0[NPR];
#pragma unroll
    for (int i = 0; i < NPR; ++i) {
        const int q = 8 * (wave + 4 * i) + rsub;
        const int p = SW == 2 ? (q < HALF ? 2 * q : 2 * (q - HALF) + 1) : q;
        int s, j;
        if (p < pb1) { s = 0; j = p; }
        else { const int d = (int)((unsigned)(p - pb1) / (unsigned)Lp); s = 1 + d; j = p - pb1 - d * Lp; }     // (nine divisions per thread and TILE)
        const int line = line0 + s;
        const int w = (s == 0 ? tw_first : 0) * SW + w_base + j;
        const int u = fd_div(line * Wo, pp.fd_trows), th = line - u * Ho;
        const bool ok = line < nlines && (unsigned)w < (unsigned)Tin;
        const int h0 = th * pp.a_step_h + pp.a_off_h;
        a_h0[i] = ok ? h0 : -0x40000000;                         // (a row outside the tensor fails the frame test for every kh)
        a_off[i] = (int)(((int64_t)u * Hin * Tin + (int64_t)h0 * Tin + w) * pp.lda) + ((pslot ^ ((q >> 1) & 7)) << 3);
    }
    const int a_khstep = tsh * Tin * (int)pp.lda;
    auto stage_panel = [&](int kh) {
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const bool ok = (unsigned)(a_h0[i] + kh * tsh) < (unsigned)Hin;
            const unsigned short* src = ok ? A + a_off[i] + (int64_t)kh * a_khstep : zero;
            unsigned short* dst = Pn + 8 * (wave + 4 * i) * TBK;                         // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    // ---- weights: rows n = 8 * (2 wave + i) + rsub
    int64_t b_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (2 * wave + i) + rsub;
        b_row[i] = (int64_t)r * pp.sBn + ((pslot ^ ((r >> 1) & 7)) << 3);
    }
    auto stage_b = [&](int kh, int kw, int buf) {
        const int64_t off = (int64_t)kh * pp.sBtap_h + (int64_t)kw * pp.sBtap;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned short* dst = Bs + buf * 64 * TBK + 8 * (2 * wave + i) * TBK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + b_row[i] + off),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    // ---- fragment rows: output row r = wm0 + 32 i + l31 -> panel row base (tap 0)
    int p_base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wm0 + 32 * i + l31;
        const int s = r < r1 ? 0 : 1 + fd_div(r - r1, pp.fd_wrows);
        p_base[i] = r * SW + s * G + (tsw < 0 ? KW - 1 : 0);
    }
    const int b_frag = (wn0 + l31) * TBK, b_sw = ((wn0 + l31) >> 1) & 7;

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    // ---- main loop.  The first version of this loop (and the per-tap kernel's) compiled to  ds_read, ds_read, s_waitcnt lgkmcnt(0),
    // MFMA, ds_read, s_waitcnt lgkmcnt(0), MFMA ...  on ONE pair of fragment registers: eight exposed LDS latencies per tap and wave
    // (profiles/r06_panel_counters.txt: matrix pipe busy 16 %, waves parked 45 %, LDS array busy 17 % -- nothing was saturated).
    // Now all twelve fragment reads of a tap are in flight together (48 fragment registers), then its eight MFMAs issue back to back
    // under the weight / panel requests of the NEXT tap: one exposed LDS latency per tap, covered by the other waves of the SIMD.
    // At the last tap of a kernel row the panel is free as soon as every wave holds its A fragments (one barrier behind the reads):
    // the next row's panel is requested there and lands under that tap's MFMAs.
    const int taps = pp.taps;
    if constexpr (PIPE) {
        // tap t reads its weights from stage t & 3; stage (t + 3) & 3 is requested during tap t (it held tap t - 1's weights, whose
        // fragments every wave read during tap t - 2) and published by the barrier that ends tap t + 1.
        bf16x8 A0[4][2], A1[4][2], B0[4], B1[4];
        auto load_a = [&](bf16x8 (&a)[4][2], int kw_) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = p_base[i] + tsw * kw_;
                const int q = SW == 2 ? (p >> 1) + (p & 1) * HALF : p;
                const int row = q * TBK, sw_ = (q >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a[ks][i] = *reinterpret_cast<const bf16x8*>(Pn + row + (((2 * ks + lh) ^ sw_) << 3));
            }
        };
        auto load_b = [&](bf16x8 (&b)[4], int slot) {
            const unsigned short* bs = Bs + slot * 64 * TBK;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) b[ks] = *reinterpret_cast<const bf16x8*>(bs + b_frag + (((2 * ks + lh) ^ b_sw) << 3));
        };
        // loop state in plain scalars (a lambda capturing them by reference put them in scratch: VGPR conditions, exec-mask branches and a
        // `vmcnt(0)` for every scratch read -- i.e. no LDS-DMA in flight across anything)
        int kh = 0, kw = 0, t = 0;                               // (kh, kw) of tap t
        int kh3 = 0, kw3 = 0;                                    // (kh, kw) of tap t + 3: the next weight stage to request
#define PANEL_ADV(h, w) do { if (++(w) == KW) { (w) = 0; ++(h); } } while (0)
        stage_panel(0);
        stage_b(0, 0, 0);
        PANEL_ADV(kh3, kw3);
        if (taps > 1) stage_b(kh3, kw3, 1);
        PANEL_ADV(kh3, kw3);
        if (taps > 2) stage_b(kh3, kw3, 2);
        PANEL_ADV(kh3, kw3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        load_a(A0, 0);
        load_b(B0, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
#define PANEL_TAP(ca, cb, na, nb)                                                                                               \
        {                                                                                                                       \
            const bool has_next = t + 1 < taps, last_kw = kw + 1 == KW;                                                         \
            const bool new_panel = has_next && last_kw;                                                                         \
            if (t + 3 < taps) stage_b(kh3, kw3, (t + 3) & 3);                                                                   \
            PANEL_ADV(kh3, kw3);                                                                                                \
            if (new_panel) {                                                                                                    \
                /* this tap's A fragments were read one tap ago (KW >= 2) -- every wave passed a barrier since: the panel is free */ \
                if (KW == 1) __syncthreads();                   /* (KW = 1: they were read just behind the last barrier) */     \
                stage_panel(kh + 1);                                                                                            \
            }                                                                                                                   \
            if (has_next) load_b(nb, (t + 1) & 3);                                                                              \
            if (has_next && !last_kw) load_a(na, kw + 1);                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                    \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                   \
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[ks][i], cb[ks], acc[i][0], 0, 0, 0);                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            /* the stage requested THIS tap may stay in flight (two LDS-DMA instructions per wave); a panel must have landed */  \
            if (new_panel || t + 3 >= taps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                               \
            __syncthreads();                                                                                                    \
            if (new_panel) { load_a(na, 0); __builtin_amdgcn_s_waitcnt(0xc07f); }   /* first tap of the next kernel row */      \
            if (last_kw) { kw = 0; ++kh; } else ++kw;                                                                           \
            ++t;                                                                                                                \
        }
        while (t < taps) {
            PANEL_TAP(A0, B0, A1, B1)
            if (t < taps) PANEL_TAP(A1, B1, A0, B0)
        }
#undef PANEL_TAP
#undef PANEL_ADV
        constexpr int SPp = 32 * 1 + 8;
        gemm_bf16_epilogue<2, 1>(pp, acc, m0, 0, wm0, wn0, lane, 0, glds_smem + wave * 64 * SPp);
        return;
    }
    int kh = 0, kw = 0, buf = 0;
    PT_MARK();                                                  // 1: maps done
    stage_panel(0);
    stage_b(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PT_MARK();                                                  // 2: first panel + weights landed (this wave's)
    __syncthreads();
    PT_MARK();                                                  // 3: ... everybody's
    for (int t = 0; t < taps; ++t) {
        const bool has_next = t + 1 < taps, last_kw = kw + 1 == KW;
        const unsigned short* bs = Bs + buf * 64 * TBK;
        bf16x8 a[4][2], b[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b[ks] = *reinterpret_cast<const bf16x8*>(bs + b_frag + (((2 * ks + lh) ^ b_sw) << 3));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = p_base[i] + tsw * kw;
            const int q = SW == 2 ? (p >> 1) + (p & 1) * HALF : p;
            const int row = q * TBK, sw_ = (q >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks][i] = *reinterpret_cast<const bf16x8*>(Pn + row + (((2 * ks + lh) ^ sw_) << 3));
        }
        if (has_next) {
            if (!last_kw) stage_b(kh, kw + 1, buf ^ 1);
            else {
                stage_b(kh + 1, 0, buf ^ 1);
                __syncthreads();                                // every wave holds its fragments of this kernel row's last tap
                stage_panel(kh + 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TIMING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PT_MARK(); }      // 4 + 4t: fragments in registers
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks], acc[i][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                      // (the compiler sank the MFMAs below the barrier: they belong under the loads in flight)
        PT_MARK();                                              // 5 + 4t: MFMAs issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PT_MARK();                                              // 6 + 4t: next weights (panel) landed
        __syncthreads();                                        // next weights (and panel) landed; this tap's weights are free
        PT_MARK();                                              // 7 + 4t: barrier passed
        buf ^= 1;
        if (last_kw) { kw = 0; ++kh; } else ++kw;
    }
    constexpr int SP_ = 32 * 1 + 8;
    gemm_bf16_epilogue<2, 1>(pp, acc, m0, 0, wm0, wn0, lane, 0, glds_smem + wave * 64 * SP_);
    PT_MARK();                                                  // last: epilogue done
    if constexpr (TIMING) { if (dbg && lane == 0) dbg[PANEL_DBG_EVENTS - 1] = (unsigned long long)dbg_n; }
}

// host side: does the problem (every phase of a fused dgrad) fit the panel kernel?
static bool panel_fits(int Wo, int KW, int sw) {
    if (Wo <= 0 || KW < sw || KW > 8) return false;
    const int nseg = (128 + Wo - 2) / Wo + 1;
    return 128 * sw + nseg * (KW - sw) + KW <= PANEL_ROWS;
}

// Returns 1 when the kernel took the launch, 0 when it declines (the caller goes on to conv_gemm_bf16_glds_n64_kernel).
int osp_launch_conv2d_panel(const GemmB& p, int64_t batch_in, hipStream_t stream) {
    // OFF by default: measured equal to the per-tap kernel on the forward layers and 5-10 % slower on the fused-phase dgrads
    // (profiles/r06_panel_ab.txt; DESIGN.md section 13 says why: the loop is bound by what a tap costs a wave besides its operand
    // traffic).  OSP_N64_PANEL=1 takes it (read per call: tests/test_gpu_conv2d_panel.py switches it on in-process).
    static int attrs = 0;
    const char* e = getenv("OSP_N64_PANEL");
    const int on = e ? atoi(e) : 0;                              // 1: the barrier-per-tap form, 2: the software-pipelined form (PIPE)
    if (on && !attrs) {
        attrs = 1;
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_panel_n64_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_panel_n64_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_panel_n64_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_panel_n64_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_panel_n64_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS_PIPE);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_panel_n64_kernel<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS_PIPE);
    }
    if (!on || batch_in != 1) return 0;
    if (!(p.N == 64 && p.Cin == 64 && p.a_bf16 && p.b_bf16 && p.sBk == 1 && !p.a_rowscale && (p.a_step == 1 || p.a_step == 2))) return 0;
    if (!((p.a_tapstep == 1 || p.a_tapstep == -1) && (p.a_tapstep_h == 1 || p.a_tapstep_h == -1))) return 0;
    if (p.lda % 8 != 0 || p.sBn % 8 != 0 || p.sBtap % 8 != 0 || p.sBtap_h % 8 != 0) return 0;
    if (((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B)) & 15) != 0) return 0;
    const int np = p.nphase > 0 ? p.nphase : 1;
    int mmax = 0;
    for (int i = 0; i < np; ++i) {
        const int M = p.nphase > 0 ? p.ph[i].M : p.M, Wo = p.nphase > 0 ? p.ph[i].Wrows : p.Wrows;
        const int KW = p.nphase > 0 ? p.ph[i].KW : p.KW, taps = p.nphase > 0 ? p.ph[i].taps : p.taps;
        const int Trows = p.nphase > 0 ? p.ph[i].Trows : p.Trows;
        if (KW <= 0 || taps % KW != 0 || Trows % Wo != 0 || M % Trows != 0 || !panel_fits(Wo, KW, p.a_step)) return 0;
        mmax = M > mmax ? M : mmax;
    }
    // 31-bit element offsets into the activation tensor
    const int64_t rows_in = (int64_t)(p.M / (p.Trows > 0 ? p.Trows : 1) + 1) * p.Hin * p.Tin;
    if (rows_in * p.lda >= (int64_t)0x7fff0000) return 0;
    const dim3 grid(1, (unsigned)cdiv((int64_t)mmax, 128), (unsigned)np);
    osp_note_symbol("conv2d_panel_n64_kernel");
    static const char* timing = getenv("OSP_PANEL_TIMING");
    if (timing && timing[0]) {
        // diagnostic: one timed launch per call, timestamps of 64 tiles x 2 waves appended to the file (tools/probes/panel_timing.py reads it)
        static unsigned long long* buf = nullptr;
        const size_t nbytes = 64 * 2 * PANEL_DBG_EVENTS * sizeof(unsigned long long);
        if (!buf) { hipMalloc(&buf, nbytes); hipMemcpyToSymbol(HIP_SYMBOL(osp_panel_dbg), &buf, sizeof(buf)); }
        hipMemsetAsync(buf, 0, nbytes, stream);
        if (p.a_step == 2) hipLaunchKernelGGL((conv2d_panel_n64_kernel<2, true>), grid, dim3(256), PANEL_LDS, stream, p);
        else hipLaunchKernelGGL((conv2d_panel_n64_kernel<1, true>), grid, dim3(256), PANEL_LDS, stream, p);
        hipStreamSynchronize(stream);
        std::vector<unsigned long long> host(nbytes / 8);
        hipMemcpy(host.data(), buf, nbytes, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(timing, "ab")) {
            const long long hdr[4] = {(long long)grid.y, (long long)np, (long long)p.a_step, (long long)p.taps};
            fwrite(hdr, sizeof(hdr), 1, f); fwrite(host.data(), 1, nbytes, f); fclose(f);
        }
        return 1;
    }
    if (on == 2) {
        if (p.a_step == 2) hipLaunchKernelGGL((conv2d_panel_n64_kernel<2, false, true>), grid, dim3(256), PANEL_LDS_PIPE, stream, p);
        else hipLaunchKernelGGL((conv2d_panel_n64_kernel<1, false, true>), grid, dim3(256), PANEL_LDS_PIPE, stream, p);
        return 1;
    }
    if (p.a_step == 2) hipLaunchKernelGGL(conv2d_panel_n64_kernel<2>, grid, dim3(256), PANEL_LDS, stream, p);
    else hipLaunchKernelGGL(conv2d_panel_n64_kernel<1>, grid, dim3(256), PANEL_LDS, stream, p);
    return 1;
}
