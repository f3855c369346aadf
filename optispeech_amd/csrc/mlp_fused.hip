// ConvNeXt block MLP in ONE kernel (no-grad / synthesise path): pwconv1 -> GELU -> pwconv2 -> layer scale + residual (+ mask),
//   y = (x + rowscale * gamma * (W2 gelu(W1 h + b1) + b2)) * rowmask     generator/modules/convnext.py:39-46 (+ DropPath :121-129, the backbone mask :99-101)
// The (rows x I) hidden activations never exist outside registers.  The unfused pair writes and re-reads them through HBM / L2
// (49k frames x 1152 x 2 B = 113 MB per vocoder block at the synthesise benchmark) and runs two launches whose 128x128 tiles are
// LDS-bandwidth bound (profiles/r04_synthesise_kernel_stats.csv: 24 x 121 us of the 7.0 ms synthesise call).
//
// Work split.  One workgroup = 4 waves = 128 rows; a wave owns 32 rows end to end, so nothing is exchanged between waves:
//   * its rows of h (bf16, 32 x C) live in registers as MFMA B-operand fragments for the whole kernel (C / 16 x 4 VGPRs);
//   * the hidden dimension is walked in chunks of 128 units.  Per chunk
//       phase 1   S^T (128 units x 32 rows) = W1[chunk] h^T          A = W1 rows from LDS, B = the h fragments; 4 accumulator tiles
//       GELU      on the accumulators (they start from b1: one extra MFMA per tile against a (hi, lo) bf16 table), rounded to bf16
//       phase 2   out (32 rows x C) += gelu(S) W2[:, chunk]^T        S^T's accumulator layout IS the A-operand layout of S when
//                 the 16 units of a k-step are taken in accumulator order (units 4h..4h+3 and 8+4h..8+4h+3 for lane half h, as in
//                 attention.hip).  W2 is PACKED with that order along K (kernels.param_bf16_kperm16), so its B fragments are
//                 the ordinary 16-byte reads.
//   * out (32 x C f32 = C / 2 accumulator registers) stays in registers over all chunks; epilogue: bias, layer scale, residual, mask.
// Registers: C = 384 -> 192 (out) + 64 (S^T) + 96 (h) + 32 (gelu(S) bf16) + weight fragments: one wave per SIMD (512-register budget).
//
// Weights stream through LDS by LDS-DMA (global_load_lds_dwordx4, rows of 128 B, slot XOR swizzle as in gemm_bf16_glds.h) in UNITS of
// 128 rows x 64 k = 16 KB: a W1 k-slab of the chunk is one unit, a W2 k-slab (C output rows x 64 units) is NP = C / 128 units; a chunk
// is KS1 = C / 64 phase-1 units followed by 2 NP phase-2 units, consumed in FOUR stages of UPS = NP units (two of phase 1, two of
// phase 2) out of a ring of R = 8 slots (128 KB).  Every stage opens with: wait until its units have landed (s_waitcnt vmcnt(n), loads
// complete in order), barrier (all waves are done with the previous stage, its slots are free); the UPS units that now fit are
// requested one LDS-DMA instruction per MFMA group during the stage (static schedule, MlpSched): a unit is requested two stages
// (>= 2000 MFMA clocks) before it is read.  Past the last chunk the schedule keeps requesting (the last chunk's units again, L2
// hits nobody reads) so that the vmcnt arithmetic stays the same.
//
// A stage is a static list of GROUPS = 4 fragment reads (one 16-byte ds_read per 32-row tile) + 4 MFMAs; the reads of group g + 1 are
// issued before the MFMAs of group g (two fragment buffers).  The reads are inline asm: behind the compiler's back for a reason --
// its wait-count pass treats every LDS read as possibly aliasing the LDS-DMA writes in flight and drains them (vmcnt(0)), which is
// the prefetch; ordering is by hand (lgkmcnt waits tied to the fragment registers, LDS returns in order).
// One wave per SIMD means nothing else hides latency: the GELU of half-tile k + 1 (VALU) is placed under the MFMAs of k-step k of
// phase 2 (mlp_gelu: a polynomial, 12 issue slots per element; the rcp + exp form of the two-launch path costs twice that).
// osp-flags: -fno-slp-vectorize
#include "gemm_bf16_common.h"
#include <utility>
#include <algorithm>

struct MlpP {
    const unsigned short* h; const unsigned short* w1; const unsigned short* w2p;
    const float *b1, *b2, *gamma, *x, *rowmask, *rowscale; float* y; int M, I;
    int nfull;                                                           // workgroups [0, nfull) take 128 rows each, the rest 64 (split mode)
    const int* rowperm;                                                  // position -> row (unmasked rows first), or NULL: position = row
};

template <int C> struct MlpSched {
    static constexpr int KS1 = C / 64, NP = C / 128, UPS = NP, UPC = 4 * UPS, R = 8, NSTG = 4, GPS = 4 * UPS;
    static_assert(KS1 == 2 * UPS, "two phase-1 stages");
    static constexpr int freed(int k) { return UPS * k; }                                             // units consumed before stage k
    static constexpr int lo(int k) { return k == 0 ? freed(NSTG - 1) - UPC + R : freed(k - 1) + R; }  // stage k issues units [lo, hi)
    static constexpr int hi(int k) { return freed(k) + R; }
    static constexpr int last_needed(int k) { return freed(k + 1) - 1; }
    static constexpr int vm(int k) { return 4 * (lo(k) - 1 - last_needed(k)); }                       // loads allowed in flight at its wait
    static_assert(hi(NSTG - 1) - UPC == lo(0), "schedule wraps");
    static_assert(hi(0) - lo(0) == UPS && hi(1) - lo(1) == UPS, "UPS units per stage");
    static_assert(hi(NSTG - 1) < 2 * UPC && vm(0) >= 0 && vm(1) >= 0 && vm(2) >= 0 && vm(3) >= 0, "schedule");
};

template <int LO, int... Is, class F>
__device__ __forceinline__ void mlp_sfor_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, LO + Is>{}), ...); }
template <int LO, int HI, class F>
__device__ __forceinline__ void mlp_sfor(F&& f) { mlp_sfor_impl<LO>(std::make_integer_sequence<int, HI - LO>{}, f); }

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ void mlp_lds_rd16(i32x4& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void mlp_lds_wait(i32x4 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N) : "memory");
}

template <int N> __device__ __forceinline__ void mlp_lds_wait2(i32x4 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N) : "memory");
}

// gelu(x) = x Phi(x), Phi(x) - 1/2 ~ xc Q(xc^2) with xc = x clamped to |x| <= 4.2 (minimax fit of degree 8 in xc^2: |Phi error| <= 7.4e-6
// inside, 1 - Phi(4.2) = 1.3e-5 outside; |gelu error| <= 5.3e-5 for |x| <= 7; the result is rounded to bf16, unit round-off 4e-3).
// 12 plain VALU instructions per element -- scalar on purpose (`// osp-flags: -fno-slp-vectorize` above): beside MFMAs a v_pk_fma_f32
// costs ~22 cycles more than the two v_fma_f32 it replaces (MI355X_MICROARCH.md, instruction timing table).
__device__ __forceinline__ float mlp_gelu(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.2f, 4.2f);
    const float w = xc * xc;
    float q = fmaf(w, 5.998066626711207e-11f, -5.633316924047449e-09f);
    q = fmaf(q, w, 2.3436740548277157e-07f);
    q = fmaf(q, w, -5.760768999607535e-06f);
    q = fmaf(q, w, 9.457439591642469e-05f);
    q = fmaf(q, w, -0.001114147948101163f);
    q = fmaf(q, w, 0.009830130264163017f);
    q = fmaf(q, w, -0.06636036932468414f);
    q = fmaf(q, w, 0.39890745282173157f);
    return x * fmaf(xc, q, 0.5f);
}

// Epilogue of both kernels.  Accumulator element i of lane (column 32 j + l31, half) is row 8 (i / 4) + 4 half + i % 4 of the wave's 32:
// stored from there a lane would touch 4 bytes per instruction (384 loads + stores per lane; measured ~15 us per workgroup,
// issue-bound).  Instead gamma * (acc + b2) goes through the wave's own LDS patch (TP tiles = CP columns at a time, row-major) and
// comes back as one float4 per lane: x in and y out as 16-byte accesses.  All x rows of a pass are REQUESTED before the first is
// used (the registers the pass's accumulator tiles just vacated hold them): as load / use / store per float4 the pass was a chain
// of NIT memory latencies, ~20 us per workgroup for the four passes.
template <int C, int NB, bool PERM, int Q0 = 0, int Q1 = -1>         // NB: float4s of x in flight per lane (registers); passes [Q0, Q1)
__device__ __forceinline__ void mlp_epilogue(const MlpP& p, f32x16 (&out)[C / 32], float* patch, int m0, int lane) {
    constexpr int NT = C / 32, TP = (NT % 3 == 0) ? 3 : 2, CP = 32 * TP, Q4 = CP / 4, NIT = 32 * Q4 / 64;
    static_assert(NT % TP == 0 && (32 * Q4) % 64 == 0, "epilogue tiling");
    const int l31 = lane & 31, half = lane >> 5;
    mlp_sfor<Q0, (Q1 < 0 ? NT / TP : Q1)>([&](auto qidx) {
        constexpr int q = decltype(qidx)::value;
#pragma unroll
        for (int tt = 0; tt < TP; ++tt) {
            const int n = 32 * (q * TP + tt) + l31;
            const float gm = p.gamma[n], bb = p.b2[n];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                patch[(8 * (i >> 2) + 4 * half + (i & 3)) * CP + 32 * tt + l31] = gm * (out[q * TP + tt][i] + bb);
        }
        __builtin_amdgcn_wave_barrier();
        static_assert(NIT % NB == 0, "batches");
#pragma unroll
        for (int b0 = 0; b0 < NIT; b0 += NB) {
            float4 xv[NB]; float rmk[NB], rs[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int idx = (b0 + k) * 64 + lane, r = idx / Q4, c4 = idx - r * Q4;
                const int mp = m0 + r < p.M ? m0 + r : p.M - 1;                  // (clamped: the loads are unconditional)
                const int m = PERM ? p.rowperm[mp] : mp;
                xv[k] = *reinterpret_cast<const float4*>(p.x + (int64_t)m * C + q * CP + 4 * c4);
                rmk[k] = p.rowmask ? p.rowmask[m] : 1.f;
                rs[k] = p.rowscale ? p.rowscale[m] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int idx = (b0 + k) * 64 + lane, r = idx / Q4, c4 = idx - r * Q4, mp = m0 + r;
                const int m = (PERM && mp < p.M) ? p.rowperm[mp] : mp;
                const float4 v = *reinterpret_cast<const float4*>(patch + r * CP + 4 * c4);
                float4 yv;
                yv.x = fmaf(rs[k], v.x, xv[k].x) * rmk[k]; yv.y = fmaf(rs[k], v.y, xv[k].y) * rmk[k];
                yv.z = fmaf(rs[k], v.z, xv[k].z) * rmk[k]; yv.w = fmaf(rs[k], v.w, xv[k].w) * rmk[k];
                if (mp < p.M) *reinterpret_cast<float4*>(p.y + (int64_t)m * C + q * CP + 4 * c4) = yv;
            }
        }
        __builtin_amdgcn_wave_barrier();
    });
}

extern __shared__ __attribute__((aligned(1024))) unsigned short mlp_smem[];
#define MLP_MAX_I 4096
#define MLP_LDS (8 * 128 * 64 * 2 + MLP_MAX_I * 4)                      // ring + (hi, lo) bias table

// SPLIT = false: the workgroup's 128 rows, a wave owns 32 rows and all 128 units of a chunk (the description above).
// SPLIT = true (round 6): 64 rows.  Waves 2 r and 2 r + 1 share row tile r; wave 2 r + uh owns S^T tiles uh and 2 + uh of every chunk
// (units 32 uh .. + 31 and 64 + 32 uh .. + 31): in phase 1 it runs 2 of the 4 MFMAs of a group, in phase 2 the two k-steps of each
// 64-unit slab its tiles cover -- HALF the MFMAs of every stage, the same LDS-DMA stream (so the ring schedule and its vmcnt
// arithmetic are those of the full mode).  Each wave of a pair ends with a partial 32 x C output tile; the pair swaps column halves
// through LDS and each finishes (bias, layer scale, residual, mask) its half of the columns.  What it is for: the LAST round of a
// launch.  One workgroup per CU means 384 row blocks on 256 CUs take two rounds for 1.5 rounds of work; as 256 full + 256 split
// workgroups the second round costs about half a round (osp_convnext_mlp_fused picks the mix; summation order of the two partials
// differs from the full mode's, so the two modes agree to f32 rounding, not bit for bit).
template <int C, bool SPLIT, bool PERM>
__device__ __forceinline__ void mlp_body(const MlpP& p, const int blk) {
    typedef MlpSched<C> S;
    constexpr int KS1 = S::KS1, NP = S::NP, UPS = S::UPS, UPC = S::UPC, R = S::R, GPS = S::GPS, NT = C / 32, KH = C / 16;
    constexpr int UNITB = 128 * 64 * 2;                              // bytes per ring slot
    constexpr int NTL = SPLIT ? 2 : 4;                               // S^T tiles of a chunk this wave owns
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int uh = SPLIT ? (wave & 1) : 0;
    const int I = p.I, nchunks = I / 128;
    const int m0 = SPLIT ? p.nfull * 128 + blk * 64 + (wave >> 1) * 32 : blk * 128 + wave * 32;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)mlp_smem;

    // ---- a row block whose rows are ALL masked (the padding behind an utterance's last frame: a quarter of the rows of the 64-sentence
    // synthesise benchmark) has y = 0 whatever the MLP gives: write the zeros and leave -- the CU takes the next block (round 6)
    if constexpr (PERM) {                                              // (the walk order is only given with a row mask)
        constexpr int ROWS = SPLIT ? 64 : 128;
        const int mb = SPLIT ? p.nfull * 128 + blk * 64 : blk * 128, mr = mb + (tid & (ROWS - 1));
        const int alive = mr < p.M && p.rowmask[p.rowperm[mr]] != 0.f;
        if (!__syncthreads_or(alive)) {
            const int nrow = p.M - mb < ROWS ? p.M - mb : ROWS;
            for (int i = tid; i < nrow * (C / 4); i += 256) {
                const int r = i / (C / 4), c4 = i - r * (C / 4), m = p.rowperm[mb + r];
                *reinterpret_cast<float4*>(p.y + (int64_t)m * C + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
    }

    // ---- b1 as (hi, lo) bf16 pairs in LDS behind the ring: phase 1 STARTS from the bias through one extra MFMA per tile and chunk,
    // A = [hi(b1[unit]), lo(b1[unit]), 0 ...] (k = 0, 1), B = ones in rows k = 0, 1: hi + lo carries b1 to 2^-17 relative, exact in
    // the f32 accumulate.  (Loading the bias into the accumulators instead -- 16 vector loads per chunk -- put compiler-visible VMEM
    // loads into the chunk loop: its wait for them is vmcnt(0), which drains every LDS-DMA request in flight once per chunk.)
    {
        unsigned* tab = reinterpret_cast<unsigned*>(mlp_smem) + R * (UNITB / 4);
        for (int i = tid; i < I; i += 256) {
            const float b = p.b1[i];
            const __bf16 hi = (__bf16)b, lo = (__bf16)(b - (float)hi);
            tab[i] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
        }
        __syncthreads();
    }
    const unsigned tab0 = lds0 + R * UNITB + 4 * l31 + 128 * uh;
    i32x4 onesf = {half ? 0 : 0x3F803F80, 0, 0, 0};

    // ---- staging: wave w writes rows 8 * (4 w + i) + rsub of a unit (i < 4), 16-byte slot pslot of each row
    const int rsub = lane >> 3, pslot = lane & 7;
    unsigned w1off[4], w2off[4];                                      // byte offsets
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * (4 * wave + i) + rsub, sw = (pslot ^ ((r >> 1) & 7)) * 8;
        w1off[i] = 2u * (unsigned)(r * C + sw);
        w2off[i] = 2u * (unsigned)(r * I + sw);
    }
    // one LDS-DMA instruction: part i of chunk-relative unit u of chunk cc
    auto issue_part = [&](auto uidx, auto iidx, int cq) {
        constexpr int u = decltype(uidx)::value, i = decltype(iidx)::value;
        const int slot = (cq * UPC + u) & (R - 1);
        // past the last chunk the schedule keeps requesting (the vmcnt arithmetic of the stages stays the same): the last chunk's
        // units again (L2 hits nobody reads) -- an index clamp, not a branch or a select of pointers: a stage stays one basic block
        const int cc = cq < nchunks ? cq : nchunks - 1;
        __attribute__((address_space(3))) unsigned short* dst =
            (__attribute__((address_space(3))) unsigned short*)mlp_smem + slot * (UNITB / 2) + wave * (4 * 8 * 64) + i * (8 * 64);
        // uniform 64-bit base (SALU) + 32-bit lane offset: the saddr form of global_load_lds, no 64-bit VALU add per request
        const char* src;
        if constexpr (u < KS1) src = reinterpret_cast<const char*>(p.w1 + (int64_t)cc * 128 * C + 64 * u) + w1off[i];
        else {
            constexpr int v = u - KS1, sl = v / NP, part = v % NP;
            src = reinterpret_cast<const char*>(p.w2p + (int64_t)(128 * part) * I + cc * 128 + 64 * sl) + w2off[i];
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // the n-th LDS-DMA instruction (n < 4 UPS) of the units stage K requests
    auto issue_nth = [&](auto kidx, auto nidx, int c) {
        constexpr int K = decltype(kidx)::value, n = decltype(nidx)::value, u = S::lo(K) + n / 4;
        if constexpr (u >= UPC) issue_part(std::integral_constant<int, u - UPC>{}, std::integral_constant<int, n % 4>{}, c + 1);
        else issue_part(std::integral_constant<int, u>{}, std::integral_constant<int, n % 4>{}, c);
    };

    // ---- this wave's rows of h as B-operand fragments: lane (row l31, half) holds k = 16 s + 8 half .. + 7 of step s
    bf16x8 hf[KH];
    {
        const int m = m0 + l31;
        const int mrow = m < p.M ? (PERM ? p.rowperm[m] : m) : 0;
        const unsigned short* hp = p.h + (int64_t)mrow * C + 8 * half;
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            uint4 v = *reinterpret_cast<const uint4*>(hp + 16 * s);
            if (m >= p.M) v = make_uint4(0u, 0u, 0u, 0u);
            hf[s] = __builtin_bit_cast(bf16x8, v);
        }
    }
    f32x16 out[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[j][i] = 0.f;
    // fragment read byte offsets inside a unit for k-step ks of the slab: row l31 of a 32-row tile (split mode: + this wave's first
    // tile in phase 1; fo2[] = the two k-steps of a phase-2 slab its tiles cover)
    unsigned fo[4], fo2[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = lds0 + 2 * (l31 * 64 + (((2 * ks + half) ^ ((l31 >> 1) & 7)) << 3));
    fo2[0] = uh ? fo[2] : fo[0]; fo2[1] = uh ? fo[3] : fo[1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fo[ks] += 4096u * (unsigned)uh;
    }

    f32x16 st[NTL];
    bf16x8 pf[NTL][2];
    i32x4 fb[2][4];
    // prologue: what stages 1 .. 3 of a chunk "-1" would have requested = units [0, lo(0)) of chunk 0
    mlp_sfor<0, S::lo(0)>([&](auto uidx) {
        mlp_sfor<0, 4>([&](auto iidx) { issue_part(uidx, iidx, 0); });
    });
    for (int c = 0; c < nchunks; ++c) {
        const int ring0 = (c * UPC) & (R - 1);
        // GELU of half-tile hh: k-step hh of phase 2 = accumulator elements 8 s .. 8 s + 7 of (local) tile t (hh = 2 t + s), rounded to bf16
        auto gelu_elems = [&](auto hidx, auto e0idx, auto e1idx) {         // elements [e0, e1) of half-tile hh
            constexpr int hh = decltype(hidx)::value, t = hh >> 1, s = hh & 1;
#pragma unroll
            for (int e = decltype(e0idx)::value; e < decltype(e1idx)::value; ++e)
                pf[t][s][e] = (__bf16)mlp_gelu(st[t][8 * s + e]);
        };
        // group gi of stage K: its unit (chunk-relative) and k-step.  Phase-1 stages walk unit-major, phase-2 stages k-step-major
        // (the GELU of the next k-step's operand then has NP groups of MFMAs to hide under)
        auto stage = [&](auto kidx) {
            constexpr int K = decltype(kidx)::value;
            constexpr bool P1 = K < 2;
            constexpr int NG = (SPLIT && !P1) ? GPS / 2 : GPS;           // groups of the stage (split phase 2: two of the four k-steps)
            constexpr int DPG = GPS / NG;                                // LDS-DMA instructions per group
            constexpr int RPG = (SPLIT && P1) ? 2 : 4;                   // fragment reads (= MFMAs) per group
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::vm(K)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            auto rd = [&](auto gidx) {
                constexpr int gi = decltype(gidx)::value;
                constexpr int ks = P1 ? gi % 4 : gi / NP;
                constexpr int u = P1 ? UPS * K + gi / 4 : KS1 + (K - 2) * NP + gi % NP;
                i32x4 (&f)[4] = fb[gi & 1];
                if constexpr (SPLIT && P1) {
                    const unsigned a = fo[ks] + ((ring0 + u) & (R - 1)) * UNITB;
                    mlp_lds_rd16<0>(f[0], a); mlp_lds_rd16<8192>(f[1], a);
                } else {
                    const unsigned a = (SPLIT ? fo2[ks] : fo[ks]) + ((ring0 + u) & (R - 1)) * UNITB;
                    mlp_lds_rd16<0>(f[0], a); mlp_lds_rd16<4096>(f[1], a); mlp_lds_rd16<8192>(f[2], a); mlp_lds_rd16<12288>(f[3], a);
                }
            };
            if constexpr (K == 0) {
                unsigned bt[NTL];
                const unsigned ta = tab0 + c * 512;
                const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (SPLIT) {
                    asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(bt[0]) : "v"(ta) : "memory");
                    asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(bt[1]) : "v"(ta) : "memory");
                    rd(std::integral_constant<int, 0>{});
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bt[0]), "+v"(bt[1]) : : "memory");
                } else {
                    asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(bt[0]) : "v"(ta) : "memory");
                    asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(bt[1]) : "v"(ta) : "memory");
                    asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(bt[2]) : "v"(ta) : "memory");
                    asm volatile("ds_read_b32 %0, %1 offset:384" : "=v"(bt[3]) : "v"(ta) : "memory");
                    rd(std::integral_constant<int, 0>{});
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bt[0]), "+v"(bt[1]), "+v"(bt[2]), "+v"(bt[3]) : : "memory");
                }
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    const i32x4 af = {(int)bt[t], 0, 0, 0};
                    st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, onesf), zacc, 0, 0, 0);
                }
            } else rd(std::integral_constant<int, 0>{});
            mlp_sfor<0, NG>([&](auto gidx) {
                constexpr int gi = decltype(gidx)::value;
                constexpr int ks = P1 ? gi % 4 : gi / NP;
                mlp_sfor<0, DPG>([&](auto didx) { issue_nth(kidx, std::integral_constant<int, DPG * gi + decltype(didx)::value>{}, c); });
                if constexpr (gi + 1 < NG) rd(std::integral_constant<int, gi + 1>{});
                i32x4 (&f)[4] = fb[gi & 1];
                if constexpr (RPG == 2) { if constexpr (gi + 1 < NG) mlp_lds_wait2<2>(f); else mlp_lds_wait2<0>(f); }
                else { if constexpr (gi + 1 < NG) mlp_lds_wait<4>(f); else mlp_lds_wait<0>(f); }
                // The GELU (VALU) is dealt out between the MFMAs it hides under, a few elements at a time: half-tile kk + 1 during the
                // NP groups of phase-2 k-step kk; half-tile 0 during the last phase-1 group, as its accumulators complete.  (One wave
                // per SIMD: whatever is not in an MFMA's shadow is serial time.  Emitted as one block after a group, the compiler kept
                // the 100 VALU instructions of a half-tile together and the next group's MFMAs behind them.)
                if constexpr (P1) {
                    constexpr int u = UPS * K + gi / 4;
                    constexpr bool LAST = K == 1 && gi == GPS - 1;
                    mlp_sfor<0, NTL>([&](auto tidx) {
                        constexpr int t = decltype(tidx)::value;
                        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[t]), hf[4 * u + ks], st[t], 0, 0, 0);
                        if constexpr (LAST && t >= 1)
                            gelu_elems(std::integral_constant<int, 0>{}, std::integral_constant<int, (t - 1) * 8 / (NTL - 1)>{},
                                       std::integral_constant<int, t * 8 / (NTL - 1)>{});
                    });
                } else {
                    // kk: the k-step's index among those this wave runs in the chunk = the half-tile of its OWN S^T tiles it consumes
                    constexpr int part = gi % NP, kk = (SPLIT ? 2 : 4) * (K - 2) + ks, NKK = 2 * NTL;
                    constexpr int E0 = part * 8 / NP, E1 = (part + 1) * 8 / NP;      // this group's share of half-tile kk + 1
                    mlp_sfor<0, 4>([&](auto iidx) {
                        constexpr int i = decltype(iidx)::value;
                        out[4 * part + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[kk >> 1][kk & 1], __builtin_bit_cast(bf16x8, f[i]),
                                                                                    out[4 * part + i], 0, 0, 0);
                        constexpr int a = E0 + (E1 - E0) * i / 4, b = E0 + (E1 - E0) * (i + 1) / 4;
                        if constexpr (kk + 1 < NKK && b > a)
                            gelu_elems(std::integral_constant<int, kk + 1>{}, std::integral_constant<int, a>{}, std::integral_constant<int, b>{});
                    });
                }
            });
        };
        stage(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the tail of the schedule (requests nobody reads)
    __syncthreads();                                                  // every wave is done with the ring: it becomes the epilogue's staging

    constexpr int PATCHF = 32 * ((C / 32) % 3 == 0 ? 96 : 64), NB = (C == 384 ? 12 : 8);
    float* patch = reinterpret_cast<float*>(mlp_smem) + wave * PATCHF;
    if constexpr (!SPLIT) mlp_epilogue<C, NB, PERM>(p, out, patch, m0, lane);
    else {
        // the pair swaps column halves of its partial output tiles: wave uh hands tiles of half 1 - uh over (16 bytes per lane and
        // instruction, lane-linear: no bank conflicts), adds what the partner left for its own half, finishes that half
        constexpr int HT = NT / 2, XF = HT * 16 * 64;                 // floats per wave in the exchange area (behind the patches)
        float4* xw = reinterpret_cast<float4*>(reinterpret_cast<float*>(mlp_smem) + 4 * PATCHF + wave * XF) + lane;
        const float4* xr = reinterpret_cast<const float4*>(reinterpret_cast<float*>(mlp_smem) + 4 * PATCHF + (wave ^ 1) * XF) + lane;
        static_assert((4 * PATCHF + 4 * XF) * 4 <= MLP_LDS, "exchange area");
        auto send = [&](auto sidx) {
            constexpr int SEND0 = decltype(sidx)::value ? 0 : HT;
#pragma unroll
            for (int j = 0; j < HT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xw[(j * 4 + q) * 64] = make_float4(out[SEND0 + j][4 * q], out[SEND0 + j][4 * q + 1], out[SEND0 + j][4 * q + 2], out[SEND0 + j][4 * q + 3]);
        };
        auto recv = [&](auto sidx) {
            constexpr int KEEP0 = decltype(sidx)::value ? HT : 0;
#pragma unroll
            for (int j = 0; j < HT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = xr[(j * 4 + q) * 64];
                    out[KEEP0 + j][4 * q] += v.x; out[KEEP0 + j][4 * q + 1] += v.y; out[KEEP0 + j][4 * q + 2] += v.z; out[KEEP0 + j][4 * q + 3] += v.w;
                }
        };
        constexpr int NQ = NT / ((NT % 3 == 0) ? 3 : 2);              // epilogue passes
        static_assert(NQ % 2 == 0, "passes split between the pair");
        if (uh == 0) send(std::integral_constant<int, 0>{}); else send(std::integral_constant<int, 1>{});
        __syncthreads();
        if (uh == 0) { recv(std::integral_constant<int, 0>{}); mlp_epilogue<C, NB, PERM, 0, NQ / 2>(p, out, patch, m0, lane); }
        else         { recv(std::integral_constant<int, 1>{}); mlp_epilogue<C, NB, PERM, NQ / 2, NQ>(p, out, patch, m0, lane); }
    }
}

// PERM: the rows are walked in the order p.rowperm gives (unmasked rows first) and a row block that is masked throughout leaves at once
// (osp_convnext_mlp_fused_live with a row order); without it the kernel is the one of the rounds before, instruction for instruction.
template <int C, bool PERM = false>
__global__ __launch_bounds__(256) void convnext_mlp_fused_kernel(const MlpP p) {
    if ((int)blockIdx.x < p.nfull) mlp_body<C, false, PERM>(p, (int)blockIdx.x);
    else mlp_body<C, true, PERM>(p, (int)blockIdx.x - p.nfull);
}

// (A producer / consumer variant of this kernel -- 8 waves, two per SIMD: four waves run phase 1 + GELU and hand the bf16 hidden tile
// to four phase-2 waves through LDS, GELU deferred by a chunk so that it runs under MFMAs -- was built and measured in round 4:
// correct on the first run, but the same 3.5 (C = 256) / 4.5 us (C = 384) per 128 hidden units as this kernel for a workgroup that
// has its CU to itself (tools/probes/mlp_fixed_cost.py (git history)), against 2.0 / 3.0 us of MFMA time.  Two designs with opposite issue
// structure and the same chunk time: the limiter is not the instruction stream of a wave.  Removed from the library; its source is in git history (commit 99fa51d, tools/probes/mlp_pc_kernel.hip).)

// f32 (N, K) -> bf16 (N, K) with every group of 16 along K stored as [0-3, 8-11, 4-7, 12-15] (the phase-2 operand order above)
__global__ __launch_bounds__(256) void pack_bf16_kperm16_kernel(const float* __restrict__ w, unsigned short* __restrict__ o, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;           // one float4 = 4 consecutive k of a row
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(w)[i];
    const int q = (int)(i & 3);                                          // quarter of the 16-group: 0 1 2 3 -> 0 2 1 3
    const int64_t d = (i & ~(int64_t)3) + ((q & 1) << 1 | (q >> 1));
    bf16x2 a, b;
    a[0] = (__bf16)v.x; a[1] = (__bf16)v.y; b[0] = (__bf16)v.z; b[1] = (__bf16)v.w;
    reinterpret_cast<uint2*>(o)[d] = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}

extern "C" int osp_pack_bf16_kperm16(const float* w, void* out, int64_t N, int64_t K, hipStream_t stream) {
    OSP_CHECK_ARG(w && out && N > 0 && K > 0 && K % 16 == 0, "bad args");
    const int64_t n4 = N * K / 4;
    hipLaunchKernelGGL(pack_bf16_kperm16_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, stream, w, reinterpret_cast<unsigned short*>(out), n4);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

static void mlp_attrs() {
    static int done = 0;
    if (done) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(convnext_mlp_fused_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(convnext_mlp_fused_kernel<384, false>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(convnext_mlp_fused_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(convnext_mlp_fused_kernel<384, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
    done = 1;
}

// h (M, C) bf16 = LayerNorm(dwconv7(x)) as osp_dwconv7_ln_fwd leaves it; w1 (I, C) bf16; w2_kperm (C, I) bf16 with every group of 16
// hidden units stored in the order [0-3, 8-11, 4-7, 12-15]; b1 (I), b2 (C), gamma (C), x / y (M, C) f32; rowmask / rowscale (M) f32 or NULL
// (rowscale = the DropPath factor of a row's utterance: the training step runs the decoder without a tape, DropPath on).
// C in {256, 384}, I % 128 == 0.
// rowperm (M int32, or NULL): the order the rows are WALKED in -- position k of the launch is row rowperm[k] of h / x / y / rowmask /
// rowscale -- with the unmasked rows first (synthesise: a stable partition of the padded batch's frames by the padding mask, made once per
// call).  Row blocks are then either live throughout or masked throughout (one mixed block), and a masked block writes its zeros and
// leaves at once: the launch costs what its LIVE rows cost.  live_rows: the number of unmasked rows where the caller knows it (the sum
// of the utterance lengths, which synthesise's one length sync already brings to the host), -1 otherwise; only the full / split
// workgroup mix is chosen from it -- any value gives the same output.
extern "C" int osp_convnext_mlp_fused_live(const void* h, const void* w1, const float* b1, const void* w2_kperm, const float* b2,
                                           const float* gamma, const float* x, const float* rowmask, const float* rowscale, float* y, int64_t M,
                                           int64_t C, int64_t I, const int* rowperm, int64_t live_rows, hipStream_t stream) {
    OSP_CHECK_ARG(h && w1 && b1 && w2_kperm && b2 && gamma && x && y, "null argument");
    OSP_CHECK_ARG(M > 0 && M < (1ll << 31) - 256, "row count out of range");
    OSP_CHECK_ARG(C == 256 || C == 384, "channel width must be 256 or 384");
    OSP_CHECK_ARG(I >= 128 && I % 128 == 0 && I <= MLP_MAX_I, "hidden width must be a multiple of 128, <= 4096");
    mlp_attrs();
    MlpP p;
    p.h = reinterpret_cast<const unsigned short*>(h); p.w1 = reinterpret_cast<const unsigned short*>(w1);
    p.w2p = reinterpret_cast<const unsigned short*>(w2_kperm);
    p.b1 = b1; p.b2 = b2; p.gamma = gamma; p.x = x; p.rowmask = rowmask; p.rowscale = rowscale; p.y = y; p.M = (int)M; p.I = (int)I;
    p.rowperm = rowmask ? rowperm : nullptr;
    // Row blocks -> workgroups.  One workgroup per CU (144 KB of LDS): nb blocks of 128 rows run in ceil(nb / CUs) rounds, and a last
    // round of r <= CUs / 2 blocks leaves half the chip idle for a whole round -- those r blocks go out as 2 r split-mode workgroups
    // of 64 rows (half the MFMAs per wave, mlp_body<C, true>).  OSP_MLP_SPLIT=0: full mode only (A/B runs, tests).
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const char* es = getenv("OSP_MLP_SPLIT");
    // With the live rows first (rowperm) and their count known, the rounds are counted in LIVE blocks: the first nb_live; the masked
    // blocks behind them cost nothing whichever mode they are launched in.
    const int64_t nb = cdiv(M, 128);
    const bool hinted = rowmask && rowperm && live_rows > 0 && live_rows < M;
    const int64_t nb_live = hinted ? cdiv(live_rows, 128) : nb, r_live = nb_live % ncu;
    int64_t nfull = nb, nsplit = 0;
    if (!(es && es[0] == '0') && r_live > 0 && 2 * r_live <= ncu) { nfull = nb_live - r_live; nsplit = cdiv(M - 128 * nfull, 64); }
    p.nfull = (int)nfull;
    const dim3 grid((unsigned)(nfull + nsplit));
    osp_note_symbol("convnext_mlp_fused_kernel");
    osp_note_flops(4.0 * (double)M * (double)C * (double)I);                                   // two GEMMs of 2 M C I
    osp_note_bytes((double)M * C * (2 + 4 + 4) + 4.0 * (double)C * I + 4.0 * (I + 2 * C));      // h in, x in, y out; both weight packs; biases, gamma
    if (p.rowperm) {
        if (C == 384) hipLaunchKernelGGL((convnext_mlp_fused_kernel<384, true>), grid, dim3(256), MLP_LDS, stream, p);
        else hipLaunchKernelGGL((convnext_mlp_fused_kernel<256, true>), grid, dim3(256), MLP_LDS, stream, p);
    } else if (C == 384) hipLaunchKernelGGL((convnext_mlp_fused_kernel<384, false>), grid, dim3(256), MLP_LDS, stream, p);
    else hipLaunchKernelGGL((convnext_mlp_fused_kernel<256, false>), grid, dim3(256), MLP_LDS, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_convnext_mlp_fused(const void* h, const void* w1, const float* b1, const void* w2_kperm, const float* b2,
                                      const float* gamma, const float* x, const float* rowmask, const float* rowscale, float* y, int64_t M, int64_t C,
                                      int64_t I, hipStream_t stream) {
    return osp_convnext_mlp_fused_live(h, w1, b1, w2_kperm, b2, gamma, x, rowmask, rowscale, y, M, C, I, nullptr, -1, stream);
}

// Row order of a padded batch for osp_convnext_mlp_fused_live: perm[k] = the k-th row of "unmasked rows first, each group in row order"
// (a stable partition of 0 .. M - 1 by rowmask != 0).  One workgroup: thread t owns rows [t c, (t + 1) c), c = ceil(M / 1024); counts its
// unmasked rows, an exclusive scan over the 1024 counts in LDS, then writes its rows' positions.  ~10 us at 49 k rows; runs on the
// stream, nothing read back (capturable), no temporary storage.
__global__ __launch_bounds__(1024) void row_order_kernel(const float* __restrict__ rowmask, int* __restrict__ perm, int M) {
    __shared__ int sc[1024];
    const int t = threadIdx.x, c = (M + 1023) / 1024;
    const int r0 = t * c < M ? t * c : M, r1 = r0 + c < M ? r0 + c : M;
    int cnt = 0;
    for (int r = r0; r < r1; ++r) cnt += rowmask[r] != 0.f;
    sc[t] = cnt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                                  // inclusive scan (Hillis-Steele)
        const int v = t >= d ? sc[t - d] : 0;
        __syncthreads();
        sc[t] += v;
        __syncthreads();
    }
    const int total = sc[1023];
    int lp = sc[t] - cnt, dp = total + (r0 - lp);                         // next unmasked / masked position of this thread's rows
    for (int r = r0; r < r1; ++r) {
        if (rowmask[r] != 0.f) perm[lp++] = r; else perm[dp++] = r;
    }
}

extern "C" int osp_row_order(const float* rowmask, int* perm, int64_t M, hipStream_t stream) {
    OSP_CHECK_ARG(rowmask && perm, "null argument");
    OSP_CHECK_ARG(M > 0 && M < (1ll << 31) - 1024, "row count out of range");
    hipLaunchKernelGGL(row_order_kernel, dim3(1), dim3(1024), 0, stream, rowmask, perm, (int)M);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
