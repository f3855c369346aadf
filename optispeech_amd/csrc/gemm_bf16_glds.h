// Direct-to-LDS body of the bf16 conv-GEMM family, shared by the translation units that instantiate it (gemm_bf16.hip: the 4-wave
// 128x128 / 128x64 / 256x128 kernels; gemm_bf16_w8.hip: the 8-wave 256x256 kernels).
// One instantiation of the body with its ten epilogues takes about a minute of compile time; one file per kernel class keeps
// the build parallel.  Each translation unit launches its own kernels through the host launchers declared at the end.
#pragma once
#include "gemm_bf16_common.h"

// ---- direct-to-LDS variant (bf16 operands in HBM, Cin % 64 == 0): the staging tiles are written by the LDS-DMA path
// (global_load_lds_dwordx4: 64 lanes x 16 B = 8 rows of 128 B per wave instruction), no VGPR round trip and no
// ds_write pass.  The LDS image is unpadded 128-byte rows; bank conflicts are avoided with an XOR swizzle of the
// 16-byte slot, applied on the SOURCE address when staging and on the read address (both-sides rule, guide section 5.4/21):
//   physical slot = logical k-group ^ ((row >> 1) & 7)
// Rows that fall into conv padding (or past M / N) read a zero page instead.
static __device__ __attribute__((aligned(256))) unsigned osp_zero_page[64];      // one copy per translation unit

// BM_ = 128: 4 waves of 64x64, two 32 KB stages, 2 workgroups / CU (prefetch distance 1; the second workgroup hides the wait).
// BM_ = 256: 4 waves of 128x64 (128 accumulator registers), three 48 KB stages, 1 workgroup / CU, prefetch distance 2.
//   Per k-slab a CU then reads (128 + 64) * 64 * 2 B * 4 waves = 96 KB of fragments for 2 * 256*128*64 flop, i.e. LDS
//   traffic per flop is 2/3 of the 128x128 tile's (which is LDS-bandwidth bound: 96 KB + 32 KB DMA per 512 MFMA clocks).
// NW = 8 (512 threads, two waves per SIMD): 256x256 tiles as 2 (M) x 4 (N) waves of 128x64 -- per k-step a wave reads
//   (128 + 64) rows x 32 B of fragments for 8 MFMAs (the 4-wave 128x128 tile: (64 + 64) x 32 B for 4), and a CU stages
//   (256 + 256) x 128 B per slab for 4x the flops of a 128x128 tile (2x fewer HBM / L2 bytes per flop); two 64 KB stages,
//   one barrier per slab, 1 workgroup / CU whose second wave per SIMD covers the other's LDS latency.
// F32 (round 4, gemm_f32_glds.hip): the operands are f32 and the products exact -- v_mfma_f32_32x32x2_f32.  Staging does not change
// at all: a 128-byte row is 32 floats instead of 64 bf16, and the caller passes every global stride / extent in 2-byte units (doubled)
// so that the address arithmetic below is the same; only the fragment reads and the MFMAs of a k-step differ.
// SPLIT (with F32; round 5, gemm_f32_split.hip): f32 operands staged exactly as in F32 mode, but every fragment element is split in
// registers into two bf16 numbers, x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest; x - hi is exact in f32,
// |r| <= 2^-18 |x|), and a product is three bf16 MFMAs accumulated in f32: lo_a hi_b + hi_a lo_b + hi_a hi_b.  What is dropped --
// lo_a lo_b and the two r terms -- is <= 3 x 2^-18 = 1.1e-5 of |a b| per product (random sign: ~4e-6 rms), against 6e-8 for the exact
// pipe and 4e-3 for plain bf16 operands.  The bf16 matrix pipe is 16x the f32 one, so three of its MFMAs cost a fifth of the exact
// product; the conversions (VALU) run beside them.  Used for the generator's GEMMs in the "mixed" parity mode (never on the
// index-critical path, whose discrete outputs must come from the same kernels as the f32 mode's).
__device__ __forceinline__ void glds_split_f32x8(const float4 v0, const float4 v1, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        lo[e] = (__bf16)(x[e] - (float)h);
    }
}

template <int BM_, int NST, int BN_ = TBN, int NW = 4, bool EARLY = false, bool F32 = false, bool SPLIT = false>
__device__ __forceinline__ void conv_gemm_bf16_glds_body(const GemmB& pin, unsigned short* smem, const TileCtx tc) {
    const GemmB pp = gemm_select_phase(pin, tc.z);
    constexpr int WN_ = NW == 8 ? 4 : 2, WM_ = NW / WN_;                                         // waves along N / M
    constexpr int RA = BM_ / (8 * NW), RB = BN_ / (8 * NW), TM_ = BM_ / (32 * WM_), TN_ = BN_ / (32 * WN_);   // rows staged per thread (A, B); 32x32 tiles per wave
    static_assert(TM_ == 2 || TM_ == 4 || (TM_ == 1 && F32), "wave tile is 64 or 128 rows (32: the 64 x 64 exact-f32 tiles only)");
    constexpr int TMA = TM_ < 2 ? 1 : 2;                                                          // 32-row blocks per accumulator array
    unsigned short* As = smem;                       // [NST][BM_][64]
    unsigned short* Bs = smem + NST * BM_ * TBK;     // [NST][BN_][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN_) * (BM_ / WM_), wn0 = (wave % WN_) * (BN_ / WN_);
    int mb_, nb_;
    xcd_tile(tc, mb_, nb_);
    const int m0 = mb_ * BM_, n0 = nb_ * BN_;
    const int64_t bz = pp.nphase > 0 ? 0 : tc.z;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(pp.A) + bz * pp.sAb;
    const unsigned short* B = reinterpret_cast<const unsigned short*>(pp.B) + bz * pp.sBb;
    const int Cin = pp.Cin, Tin = pp.Tin, Hin = pp.Hin, KW = pp.KW, a_tapstep = pp.a_tapstep, a_tapstep_h = pp.a_tapstep_h;
    const int taps = pp.taps;
    const int64_t lda = pp.lda, sBn = pp.sBn, sBtap = pp.sBtap, sBtap_h = pp.sBtap_h;
    const int K = taps * Cin;
    const int rsub = lane >> 3, pslot = lane & 7;
    // wave w stages rows 8 * (w * RA + i) + rsub of A (i < RA) and 8 * (w * RB + i) + rsub of B (i < RB)
    // K ORDER: channel block outer, TAP INNER.  Consecutive k-slabs of a tile then read the same 64 channels of input rows
    // shifted by one tap step -- 255 of 256 rows of a stride-1 conv were fetched one slab earlier and are L2 (TCP) hits.  In the
    // tap-major order of round 2 a tile came back to the same rows Cin / 64 slabs later, i.e. after the XCD's co-resident
    // tiles had streamed Cin / 64 x 32 KB x 32 tiles = 16 MB through its 4 MB L2: every tap re-fetched the activation panel
    // from MALL / HBM (TCC_EA traffic 2.7x the algorithmic bytes, profiles/r02b_pmc_glds.json).
    // Per row: the element offset of (tap 0, channel 0) -- may lie outside the tensor, only dereferenced when the tap's frame is
    // in range -- plus the frame coordinates for the range test; a tap adds a wave-uniform (SGPR) offset.
    int a_t[RA], a_h[RA]; int64_t a_off0[RA]; int64_t b_row[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int r = 8 * (wave * RA + i) + rsub;
        const int m = m0 + r;
        if (m < pp.M) {
            const int u = fd_div(m, pp.fd_trows), t = m - u * pp.Trows, th = fd_div(t, pp.fd_wrows), tw = t - th * pp.Wrows;
            a_t[i] = tw * pp.a_step + pp.a_off;
            a_h[i] = th * pp.a_step_h + pp.a_off_h;
            a_off0[i] = ((int64_t)u * Hin * Tin + (int64_t)a_h[i] * Tin + a_t[i]) * lda + (pslot ^ ((r >> 1) & 7)) * 8;
        } else { a_t[i] = -0x40000000; a_h[i] = 0; a_off0[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int r = 8 * (wave * RB + i) + rsub;
        const int n = n0 + r;
        b_row[i] = n < pp.N ? (int64_t)n * sBn + (pslot ^ ((r >> 1) & 7)) * 8 : -1;
    }
    // accumulators as 64-row halves: the epilogue is instantiated per half with compile-time indices only (one 512-byte
    // array indexed through the epilogue's nested loops stayed a stack object and was stored to scratch every iteration)
    f32x16 acc0[TMA][TN_], acc1[TMA][TN_];
#pragma unroll
    for (int i = 0; i < TMA; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(osp_zero_page);

    // Staging state: per owned row the source pointer of the slab being staged (or the zero page).
    const unsigned short* a_src[RA]; const unsigned short* b_src[RB];
    auto set_tap = [&](int j, int cb) {
        const int kh = (KW == taps) ? 0 : j / KW, kw = j - kh * KW;
        const int dt = kw * a_tapstep, dh = kh * a_tapstep_h;                                  // wave-uniform
        const int64_t offA = ((int64_t)dh * Tin + dt) * lda + cb;
        const int64_t offB = (int64_t)kh * sBtap_h + (int64_t)kw * sBtap + cb;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const bool ok = (unsigned)(a_t[i] + dt) < (unsigned)Tin && (unsigned)(a_h[i] + dh) < (unsigned)Hin;
            a_src[i] = ok ? A + a_off0[i] + offA : zero;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) b_src[i] = b_row[i] >= 0 ? B + b_row[i] + offB : zero;
    };
    int is_j = 0, is_cb = 0;                                   // (tap, channel offset) of the next k-slab to stage
    // one of the RA + RB row loads of a slab (compile-time index): the loads are spread over the 4 k-steps of the MFMA
    // phase -- issued back to back at the top of an iteration they queue behind each other in the texture-address unit
    // (4 waves x 12 x 1 KB at 64 B/clk) and the MFMA pipe idles until the last one has been accepted.
    auto issue_one = [&](int buf, auto idx) {
        constexpr int I = decltype(idx)::value;
        if constexpr (I < RA) {
            unsigned short* dst = As + buf * BM_ * TBK + (wave * RA + I) * 8 * TBK;      // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[I]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            constexpr int J = I - RA;
            unsigned short* dst = Bs + buf * BN_ * TBK + (wave * RB + J) * 8 * TBK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[J]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto issue_quarter = [&](int buf, auto qidx) {             // loads [Q * NL / 4, (Q + 1) * NL / 4)
        constexpr int Q = decltype(qidx)::value, NL = RA + RB, L0 = Q * NL / 4, L1 = (Q + 1) * NL / 4;
        if constexpr (L1 - L0 > 0) issue_one(buf, std::integral_constant<int, L0>{});
        if constexpr (L1 - L0 > 1) issue_one(buf, std::integral_constant<int, L0 + 1>{});
        if constexpr (L1 - L0 > 2) issue_one(buf, std::integral_constant<int, L0 + 2>{});
    };
    // The source pointers of a slab are computed right AFTER the previous slab's loads were issued (issue_end), i.e. in the
    // shadow of that iteration's remaining MFMAs -- computed at the top of the iteration they delayed its first loads and
    // first MFMA (8-wave kernel: 128 -> 139 us per launch when this order was introduced).
    auto issue_begin = [&]() {};
    // SPLIT kernels, taps == 1 (a plain GEMM -- every pointwise conv / linear of the generator): the rows of a slab are the rows of the
    // previous one 128 bytes further on, so the source pointers ADVANCE instead of being recomputed (a row on the zero page stays
    // there).  The split kernels are VALU-issue-bound (profiles/r05_split_pmc.txt) and the generic form costs ~10 VALU instructions per
    // staged row and slab -- as much as the (hi, lo) conversions of a 64 x 64 tile.
    int a_adv[SPLIT ? RA : 1], b_adv[SPLIT ? RB : 1];
    auto issue_end = [&]() {
        if constexpr (SPLIT) {
            if (taps == 1) {
                is_cb += TBK;
#pragma unroll
                for (int i = 0; i < RA; ++i) a_src[i] += a_adv[i];
#pragma unroll
                for (int i = 0; i < RB; ++i) b_src[i] += b_adv[i];
                return;
            }
        }
        ++is_j; if (is_j == taps) { is_j = 0; is_cb += TBK; } if (is_cb < Cin) set_tap(is_j, is_cb);
    };
    set_tap(0, 0);
    if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < RA; ++i) a_adv[i] = a_src[i] != zero ? TBK : 0;
#pragma unroll
        for (int i = 0; i < RB; ++i) b_adv[i] = b_src[i] != zero ? TBK : 0;
    }
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
            if constexpr (F32 && SPLIT) {
                // a 128-byte staged row is 32 floats = TWO 16-deep bf16 k-steps (h = 0, 1).  Lane (row l31, half lh) owns k = 16 h + 8 lh .. + 7
                // = the 16-byte slots 4 h + 2 lh and 4 h + 2 lh + 1 of its row (the operand layout of v_mfma_f32_32x32x16_bf16).
                // Everything happens in k-step 0 of the loop: ALL fragment reads of the slab first (no LDS-DMA request is in flight then --
                // the compiler puts a vmcnt(0) in front of every LDS read that follows a request), then the whole next slab is requested,
                // then conversions and MFMAs run while those loads are in flight: one exposed memory latency per slab.  (Requests dealt
                // over four k-steps in front of the reads -- the first version -- exposed up to four.)
                static_assert(TM_ <= 2, "f32 tiles: 32 or 64 rows per wave");
                if constexpr (ks == 0) {
                    float4 av[2][TM_][2], bv[2][TN_][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int i = 0; i < TM_; ++i) {
                            const int row = wm0 + 32 * i + l31, sw = (row >> 1) & 7;
                            av[h][i][0] = *reinterpret_cast<const float4*>(as + row * TBK + (((4 * h + 2 * lh) ^ sw) << 3));
                            av[h][i][1] = *reinterpret_cast<const float4*>(as + row * TBK + (((4 * h + 2 * lh + 1) ^ sw) << 3));
                        }
#pragma unroll
                        for (int j = 0; j < TN_; ++j) {
                            const int row = wn0 + 32 * j + l31, sw = (row >> 1) & 7;
                            bv[h][j][0] = *reinterpret_cast<const float4*>(bs + row * TBK + (((4 * h + 2 * lh) ^ sw) << 3));
                            bv[h][j][1] = *reinterpret_cast<const float4*>(bs + row * TBK + (((4 * h + 2 * lh + 1) ^ sw) << 3));
                        }
                    }
                    if (ld >= 0) {
                        issue_quarter(ld, std::integral_constant<int, 0>{}); issue_quarter(ld, std::integral_constant<int, 1>{});
                        issue_quarter(ld, std::integral_constant<int, 2>{}); issue_quarter(ld, std::integral_constant<int, 3>{});
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bf16x8 ah[TM_], al[TM_], bh[TN_], bl[TN_];
#pragma unroll
                        for (int i = 0; i < TM_; ++i) glds_split_f32x8(av[h][i][0], av[h][i][1], ah[i], al[i]);
#pragma unroll
                        for (int j = 0; j < TN_; ++j) glds_split_f32x8(bv[h][j][0], bv[h][j][1], bh[j], bl[j]);
#pragma unroll
                        for (int i = 0; i < TMA; ++i)
#pragma unroll
                            for (int j = 0; j < TN_; ++j) {
                                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc0[i][j], 0, 0, 0);
                                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc0[i][j], 0, 0, 0);
                                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                            }
                    }
                }
                return;
            }
            if constexpr (F32) {
                // the two 16-byte slots of this k-step hold 4 floats each = two MFMAs of K = 2.  Lane half lh reads the 8-byte half lh
                // of a slot, (k, k + 1) with k = 4 slot + 2 lh: the first MFMA contracts k = {4 slot, 4 slot + 2} (its k index IS the
                // lane half), the second {4 slot + 1, 4 slot + 3} -- A and B fragments follow the same assignment, every k once.
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                static_assert(TM_ <= 2, "f32 tiles: 32 or 64 rows per wave");
                if (ld >= 0) issue_quarter(ld, ksidx);
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    f32x2_ a[TM_], b[TN_];
#pragma unroll
                    for (int i = 0; i < TM_; ++i) {
                        const int row = wm0 + 32 * i + l31;
                        a[i] = *reinterpret_cast<const f32x2_*>(as + row * TBK + ((((2 * ks + g2) ^ ((row >> 1) & 7)) << 3) + 4 * lh));
                    }
#pragma unroll
                    for (int j = 0; j < TN_; ++j) {
                        const int row = wn0 + 32 * j + l31;
                        b[j] = *reinterpret_cast<const f32x2_*>(bs + row * TBK + ((((2 * ks + g2) ^ ((row >> 1) & 7)) << 3) + 4 * lh));
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int i = 0; i < TMA; ++i)
#pragma unroll
                            for (int j = 0; j < TN_; ++j)
                                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc0[i][j], 0, 0, 0);
                }
                return;
            }
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
            if constexpr (EARLY) {
                // the whole next slab is requested during the first two k-steps, so the last load has two k-steps of MFMA
                // time (>= 1000 cycles with two waves per SIMD) to land before the slab-closing vmcnt(0)
                if (ld >= 0 && ks < 2) {
                    issue_quarter(ld, std::integral_constant<int, 2 * (ks & 1)>{});
                    issue_quarter(ld, std::integral_constant<int, 2 * (ks & 1) + 1>{});
                }
            } else {
                if (ld >= 0) issue_quarter(ld, ksidx);
            }
#pragma unroll
            for (int i = 0; i < TMA; ++i)
#pragma unroll
                for (int j = 0; j < TN_; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc0[i][j], 0, 0, 0);
                    if constexpr (TM_ == 4) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 + i], b[j], acc1[i][j], 0, 0, 0);
                }
        };
        kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{});
        if constexpr (EARLY) { if (ld >= 0) issue_end(); }              // all loads of the slab are out: next pointers under k-steps 2-3
        kstep(std::integral_constant<int, 2>{}); kstep(std::integral_constant<int, 3>{});
        if constexpr (!EARLY) { if (ld >= 0) issue_end(); }
    };
    const int nk = K / TBK;
    // (s_setprio(1) for the second-dispatched half of an 8-wave workgroup -- MI355X_MICROARCH.md, 'two waves per SIMD' -- measured: no
    // gain on this loop, 136-141 us either way; not kept)
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
    // (wave-private staging: max(32 * TM_, 64) 16-bit rows = one 32-row block of f32)
    gemm_bf16_epilogue<TMA, TN_>(pp, acc0, m0, n0, wm0, wn0, lane, bz, smem + wave * (TM_ < 2 ? 64 : 32 * TM_) * SP_);
    if constexpr (TM_ == 4) gemm_bf16_epilogue<2, TN_>(pp, acc1, m0, n0, wm0 + 64, wn0, lane, bz, smem + wave * (32 * TM_) * SP_ + 64 * SP_);
}

extern __shared__ __attribute__((aligned(1024))) unsigned short glds_smem[];
// 8 waves, 256x256 tiles, two 64 KB stages (the epilogue's wave-private staging tiles need 144 KB: that is what is allocated)
#define GLDS8_LDS (8 * 128 * (32 * 2 + 8) * 2)
// 4 waves, 128x128 tiles: two 32 KB stages; the row-domain epilogues stage one 32-row f32 block per wave (4 x 9 KB)
#define GLDS_LDS (2 * (128 + TBN) * TBK * 2)

// host launchers of the kernels that live in other translation units
int osp_launch_glds8(const GemmB& p, dim3 grid, bool early, hipStream_t stream);              // gemm_bf16_w8.hip
int osp_launch_conv2d_panel(const GemmB& p, int64_t batch_in, hipStream_t stream);            // conv2d_panel.hip (1: taken, 0: declined)
