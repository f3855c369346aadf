// Small-problem member of the bf16 conv-GEMM family: 64x64 tiles, 4 waves (one 32x32 MFMA tile each), a deep LDS-DMA ring.
//
// The generator's own GEMMs (WaveNeXt blocks at M = B*segment = 2048 rows, the text-side ConvNeXt / predictor convs at
// M = B*T_text = 4096) have 50-600 output tiles of 64x64 and K = 256..1280: one or two workgroups per CU, so nothing but the
// workgroup's own prefetch depth hides the load latency.  The register-staged kernel (gemm_bf16_reg.hip) keeps ONE slab in
// flight and -- for f32 operands -- converts right behind the load, i.e. every 64-deep slab pays a full memory latency
// (measured 1.5-1.7 us per slab: 31-38 us for 2-5 GFLOP).  Here the slabs go HBM/L2 -> LDS through the DMA path
// (global_load_lds_dwordx4, no VGPR round trip), NST - 1 slabs ahead of the MFMAs, with a counted vmcnt wait:
// With one MFMA per wave and 16-deep k-step the ds_read -> MFMA chain would be the next latency wall (a wave is alone on its
// SIMD): the fragments of k-step s + 1 are read while the MFMA of k-step s runs (register double buffering).
//   bf16 A: 4 stages x (64 + 64) rows x 128 B = 64 KB  (2 workgroups / CU), prefetch distance 3
//   f32  A: 3 stages x (64 x 256 B + 64 x 128 B) = 72 KB (2 workgroups / CU), prefetch distance 2; the f32 rows are
//           DMA'd as they are and converted to bf16 (v_cvt_pk_bf16_f32, RNE: the same rounding the register-staged loader
//           applies) when the fragments are read from LDS -- activations saved in f32 need no bf16 copy in HBM.
// B (weights) is bf16, k-contiguous (the per-epoch packs of kernels.py).  Same GemmB contract and epilogues as the rest of the
// family (conv taps / dilation / strides / 2-D geometry, zero page for padding rows); no fused dgrad phases (nphase == 0).
//
// LDS images are unpadded rows with an XOR swizzle of the 16-byte slot, applied to the SOURCE address when staging and to the
// read address (both-sides rule, cdna_hip_programming.md section 5.4): bf16 rows have 8 slots, slot ^= (row >> 1) & 7 (as
// gemm_bf16.hip); f32 rows have 16 slots, slot ^= row & 15 -- the 16 lanes of one ds_read_b128 phase (16 consecutive rows,
// same logical slot) then hit 16 distinct slots.
#include "gemm_bf16_common.h"

__device__ __attribute__((aligned(256))) unsigned osp_zero_page_small[64];

#define SBM 64
#define SBN 64

template <bool A32, int NST>
__device__ __forceinline__ void gemm_small_body(const GemmB& pp, unsigned char* smem) {
    constexpr int NW = 4;
    constexpr int A_ROW_B = A32 ? 256 : 128;                       // bytes of one staged A row (64 k)
    constexpr int A_RPI = 1024 / A_ROW_B;                          // A rows one wave instruction stages (64 lanes x 16 B)
    constexpr int RA = SBM / (A_RPI * NW), RB = SBN / (8 * NW);    // DMA instructions per wave and slab
    constexpr int NL = RA + RB;
    constexpr int A_STAGE = SBM * A_ROW_B, B_STAGE = SBN * 128;
    unsigned char* As = smem;                                      // [NST][64][A_ROW_B]
    unsigned char* Bs = smem + NST * A_STAGE;                      // [NST][64][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    int mb_, nb_;
    xcd_tile(mb_, nb_);
    const int m0 = mb_ * SBM, n0 = nb_ * SBN;
    const int64_t bz = blockIdx.z;
    const char* A = reinterpret_cast<const char*>(pp.A) + bz * pp.sAb * (A32 ? 4 : 2);
    const char* B = reinterpret_cast<const char*>(pp.B) + bz * pp.sBb * 2;
    const int Cin = pp.Cin, Tin = pp.Tin, Hin = pp.Hin, KW = pp.KW, a_tapstep = pp.a_tapstep, a_tapstep_h = pp.a_tapstep_h;
    const int taps = pp.taps;
    const int64_t lda = pp.lda, sBn = pp.sBn, sBtap = pp.sBtap, sBtap_h = pp.sBtap_h;
    const int K = taps * Cin;
    // A: instruction i of this wave stages rows A_RPI * (wave * RA + i) + rsub, this lane the physical slot pslot of its row
    const int a_rsub = A32 ? (lane >> 4) : (lane >> 3), a_pslot = A32 ? (lane & 15) : (lane & 7);
    const int b_rsub = lane >> 3, b_pslot = lane & 7;
    int a_t[RA], a_h[RA]; int64_t a_base[RA]; int64_t b_row[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + A_RPI * (wave * RA + i) + a_rsub;
        if (m < pp.M) {
            const int u = fd_div(m, pp.fd_trows), t = m - u * pp.Trows, th = fd_div(t, pp.fd_wrows), tw = t - th * pp.Wrows;
            a_t[i] = tw * pp.a_step + pp.a_off;
            a_h[i] = th * pp.a_step_h + pp.a_off_h;
            a_base[i] = (int64_t)u * Hin * Tin;
        } else { a_t[i] = -0x40000000; a_h[i] = 0; a_base[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + 8 * (wave * RB + i) + b_rsub;
        b_row[i] = n < pp.N ? (int64_t)n * sBn : -1;
    }
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const char* zero = reinterpret_cast<const char*>(osp_zero_page_small);

    const char* a_src[RA]; const char* b_src[RB]; int a_inc[RA], b_inc[RB];
    int cur_tap = -1;
    auto set_tap = [&](int j) {
        const int kh = (KW == taps) ? 0 : j / KW, kw = j - kh * KW;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int r = A_RPI * (wave * RA + i) + a_rsub;
            const int q = A32 ? (a_pslot ^ (r & 15)) : (a_pslot ^ ((r >> 1) & 7));
            const int tt = a_t[i] + kw * a_tapstep, hh = a_h[i] + kh * a_tapstep_h;
            const bool ok = tt >= 0 && tt < Tin && hh >= 0 && hh < Hin;
            a_src[i] = ok ? A + ((a_base[i] + (int64_t)hh * Tin + tt) * lda) * (A32 ? 4 : 2) + q * 16 : zero;
            a_inc[i] = ok ? (A32 ? 4 : 2) : 0;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = 8 * (wave * RB + i) + b_rsub;
            const int q = b_pslot ^ ((r >> 1) & 7);
            b_src[i] = b_row[i] >= 0 ? B + (b_row[i] + (int64_t)kh * sBtap_h + (int64_t)kw * sBtap) * 2 + q * 16 : zero;
            b_inc[i] = b_row[i] >= 0 ? 2 : 0;
        }
        cur_tap = j;
    };
    int is_j = 0, is_cb = 0;                                       // (tap, channel offset) of the next slab to stage
    auto issue = [&](int buf) {
        if (is_j != cur_tap) set_tap(is_j);                        // wave-uniform
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            unsigned char* dst = As + buf * A_STAGE + (wave * RA + i) * 1024;          // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + is_cb * a_inc[i]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            unsigned char* dst = Bs + buf * B_STAGE + (wave * RB + i) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + is_cb * b_inc[i]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        // channel block outer, tap inner: the K order of the whole bf16 conv-GEMM family (gemm_bf16.hip explains why); a layer
        // must give bit-identical results whichever kernel its size selects
        ++is_j;
        if (is_j == taps) { is_j = 0; is_cb += TBK; }
    };
    const int l31 = lane & 31, lh = lane >> 5;
    const int arow = wm0 + l31, brow = wn0 + l31;
    auto frag_a = [&](const unsigned char* as, int ks) -> bf16x8 {
        if constexpr (A32) {
            const int s0 = 2 * (2 * ks + lh);
            const float4 lo = *reinterpret_cast<const float4*>(as + arow * 256 + ((s0 ^ (arow & 15)) << 4));
            const float4 hi = *reinterpret_cast<const float4*>(as + arow * 256 + (((s0 + 1) ^ (arow & 15)) << 4));
            const uint4 v = make_uint4(pk2(lo.x, lo.y), pk2(lo.z, lo.w), pk2(hi.x, hi.y), pk2(hi.z, hi.w));
            return __builtin_bit_cast(bf16x8, v);
        } else {
            return *reinterpret_cast<const bf16x8*>(as + arow * 128 + (((2 * ks + lh) ^ ((arow >> 1) & 7)) << 4));
        }
    };
    auto frag_b = [&](const unsigned char* bs, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(bs + brow * 128 + (((2 * ks + lh) ^ ((brow >> 1) & 7)) << 4));
    };
    auto mma = [&](int buf) {
        const unsigned char* as = As + buf * A_STAGE;
        const unsigned char* bs = Bs + buf * B_STAGE;
        // all four k-steps' fragments are requested up front (8-24 VGPRs each): the MFMA of step s only waits for its own pair
        bf16x8 a[4], b[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { a[ks] = frag_a(as, ks); b[ks] = frag_b(bs, ks); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[ks], acc[0][0], 0, 0, 0);
    };
    const int nk = K / TBK;
    // NST stages, prefetch distance NST - 1: slab kt + NST - 1 goes into the buffer slab kt - 1 was read from -- every wave has
    // passed this iteration's barrier, hence finished its MFMAs on kt - 1.  The wait leaves the younger slabs in flight.
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s);
    int buf = 0, nxt = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = nk - 1 - kt < NST - 2 ? nk - 1 - kt : NST - 2;      // slabs issued after slab kt and still wanted in flight
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // bare s_barrier (see gemm_bf16.hip): __syncthreads() would drain the younger slabs' DMA loads.  Every ds_read of the
        // previous slab has been consumed by an MFMA (the compiler's lgkmcnt waits), so its buffer may be overwritten.
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NST - 1 < nk) issue(nxt);
        mma(buf);
        buf = buf + 1 == NST ? 0 : buf + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    __syncthreads();
    gemm_bf16_epilogue<1, 1>(pp, acc, m0, n0, wm0, wn0, lane, bz, reinterpret_cast<unsigned short*>(smem) + wave * 64 * (32 + 8));
}

extern __shared__ __attribute__((aligned(1024))) unsigned char small_smem[];
__global__ __launch_bounds__(256) void conv_gemm_bf16_s64_kernel(const GemmB pp) { gemm_small_body<false, 4>(pp, small_smem); }
__global__ __launch_bounds__(256) void conv_gemm_bf16_s64_a32_kernel(const GemmB pp) { gemm_small_body<true, 3>(pp, small_smem); }

int osp_launch_gemm_small(const GemmB& p, int64_t batch, hipStream_t stream) {
    const dim3 grid((unsigned)cdiv((int64_t)p.N, SBN), (unsigned)cdiv((int64_t)p.M, SBM), (unsigned)batch);
    static int attr = 0;
    constexpr int LDS_B = 4 * (SBM * 128 + SBN * 128), LDS_F = 3 * (SBM * 256 + SBN * 128);
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_s64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_s64_a32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F);
        attr = 1;
    }
    osp_note_symbol(p.a_bf16 ? "conv_gemm_bf16_s64_kernel" : "conv_gemm_bf16_s64_a32_kernel");
    if (p.a_bf16) hipLaunchKernelGGL(conv_gemm_bf16_s64_kernel, grid, dim3(256), LDS_B, stream, p);
    else hipLaunchKernelGGL(conv_gemm_bf16_s64_a32_kernel, grid, dim3(256), LDS_F, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
