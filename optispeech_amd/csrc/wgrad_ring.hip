// Weight gradients of the bf16 conv-GEMM family, round 5: ring-pipelined and atomic-free.
//
// What the tile-per-tap kernels of wgrad_bf16.hip were bound by (tools/probes/wgrad_shapes.py with the main loop or the
// epilogue compiled out, profiles/r05_wgrad_decomposition.txt):
//   * their slab loop keeps ONE 64-frame slab in flight and drains it (`s_waitcnt vmcnt(0)` + barrier) before the next MFMA
//     phase: 1.1-2.2 us per slab whatever the slab costs in MFMA time (0.06-0.25 us);
//   * the frame splits meet in f32 atomics, and the L2 atomic units retire ~0.1 T lanes / s: the ConvNeXt pointwise weight
//     gradients of the generator (8 splits x 442 k elements = 3.5 M atomics) spend 38 us of their 37-39 us there, the
//     DiscriminatorP 128 -> 512 layer 38 of 96 us; plain read-modify-write is no faster (the adds of one lane serialise on
//     possible aliasing).
// Here:
//   * a 1-D grid in XCD-aware order: the (tap x Cin-tile) workgroups of one (frame split, N tile) are neighbours on one XCD
//     and share their dY panel (and the tap-shifted X panels) through that XCD's L2 (TCC: 74-84 % hits instead of 25-28 %,
//     3-5x fewer reads from the fabric: profiles/r05_wgrad_pmc.txt);
//   * a split writes its partial tile with plain stores into a workspace the caller supplies, laid out like dW, and a
//     second small kernel adds the splits up in a fixed order (deterministic weight gradients) and applies oscale / `+=`.
//     With one split the workgroup owns its tile and adds straight into dW;
//   * without atomics more, smaller splits are free, and OCCUPANCY turned out to be what hides the load latency: a ring of
//     NST LDS stages with a counted vmcnt wait (the idiom of gemm_bf16_small.hip) was built for that, but two stages and
//     twice the resident workgroups beat four stages at every size measured (DiscriminatorP 128 -> 512: NST 4 / 256
//     workgroups 110 us, NST 2 / 512 workgroups 53 us, the tile-per-tap kernel 96 us; profiles/r05_wgrad_sweeps.txt);
//   * the fragments of k-step s + 1 are requested before the MFMAs of k-step s are issued.
// The LDS image, its XOR swizzle and the ds_read_b64_tr_b16 fragment reads are those of conv_wgrad_bf16_tr_kernel.
#include "wgrad_common.h"

typedef short s16x4r __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(1024))) unsigned short wgr_smem[];

// workspace block of one (batch, split): [N][taps][Cin] partial sums followed by [N] bias partial sums, padded to 4 floats
__host__ __device__ __forceinline__ int64_t wgr_block_elems(int64_t N, int64_t taps, int64_t Cin) {
    return ((N * taps * Cin + N + 3) / 4) * 4;
}

template <int T, int NST>
__global__ __launch_bounds__(256) void conv_wgrad_ring_kernel(WgradB p, float* __restrict__ ws) {
    constexpr int SK = 64;                                   // frames per slab
    constexpr int S = T / 8, RPI = 64 / S, NI = SK / RPI / 4, TI = T / 64;   // slots/row, rows/instruction, pairs/wave/slab
    constexpr int NL = 2 * NI;                               // DMA instructions per wave and slab
    constexpr int STAGE = 2 * SK * T;                        // 16-bit elements of one stage: dY slab, X slab
    unsigned short* smem = wgr_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (T / 2), wn0 = (wave & 1) * (T / 2);
    // ---- work item (see the header): XCD x owns the contiguous range of items starting at x * per
    const int ctiles = (p.Cin + T - 1) / T, ntiles = p.N / T, inner = p.taps * ctiles;
    const int total = gridDim.x, lin = blockIdx.x, xcd = lin & 7, local = lin >> 3;
    const int per = total >> 3, rem = total & 7;
    const int pid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + local;
    const int grp = pid / inner, within = pid - grp * inner;              // grp = (bz * splits + sp) * ntiles + nt
    const int zs = grp / ntiles, n0 = (grp - zs * ntiles) * T;
    const int j = within / ctiles, c0 = (within - j * ctiles) * T;
    const int bz = zs / p.splits, sp = zs - bz * p.splits;
    const unsigned short* dY = reinterpret_cast<const unsigned short*>(p.dY) + (int64_t)bz * p.sYb;
    const unsigned short* X = reinterpret_cast<const unsigned short*>(p.X) + (int64_t)bz * p.sXb;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const int blk_kh = j / p.KW, blk_kw = j - blk_kh * p.KW;
    const bool do_bias = (p.db != nullptr) && (within == 0);
    const bool bias_wave = __builtin_amdgcn_readfirstlane((int)(do_bias && wn0 == 0)) != 0;
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
    // staging: wave w, pair i covers slab rows RPI * (NI * w + i) + (lane / S); physical 16-byte slot = lane % S.  Buffer-resource
    // loads: 32-bit byte offsets against an SGPR descriptor, out-of-range offsets read as zero (padding rows, the frames past
    // the split's end, the upper half of a 32-channel X row).
    const int srow = lane / S, lslot = (lane % S) ^ swz(srow);
    const int ldy32 = (int)p.ldy, ldx32 = (int)p.ldx;
    const unsigned ycol2 = (unsigned)(n0 + lslot * 8) * 2u, xcol2 = (unsigned)(c0 + lslot * 8) * 2u;
    const bool xcol_ok = c0 + lslot * 8 < p.Cin;
    __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dY), 0, (int)p.y_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, (int)p.x_bytes, 0x00020000);
    auto issue_pair = [&](int mk, int buf, int i) {
        unsigned short* ys = smem + buf * STAGE;
        unsigned short* xs = ys + SK * T;
        const int row0 = RPI * (NI * wave + i), m = mk + row0 + srow;
        const bool mv = m < mend;
        const unsigned yo = mv ? (unsigned)(m * ldy32) * 2u + ycol2 : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (__attribute__((address_space(3))) void*)(ys + row0 * T), 16, yo, 0, 0, 0);
        const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
        const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
        const bool xv = mv && (unsigned)tt < (unsigned)p.Tin && (unsigned)hh < (unsigned)p.Hin && xcol_ok;
        const unsigned xo = xv ? (unsigned)(((u * p.Hin + hh) * p.Tin + tt) * ldx32) * 2u + xcol2 : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(xs + row0 * T), 16, xo, 0, 0, 0);
    };
    // fragment of operand tile `base` ([SK][T]) for the 32 channels starting at `col0`, k-step ks: 8 consecutive frames of one
    // channel per lane (inline asm: see conv_wgrad_bf16_tr_kernel -- the compiler would drain the DMA ring in front of a
    // builtin LDS read; the reads are waited for explicitly in frag_wait)
    const int r16 = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    auto frag = [&](const unsigned short* base, int col0, int ks) -> bf16x8 {
        const int col = col0 + 16 * g16 + 4 * (r16 & 3);
        const int pslot = (col >> 3) ^ swz(r16 >> 2);
        const unsigned short* a0 = base + (16 * ks + 8 * kg + (r16 >> 2)) * T + pslot * 8 + (col & 7);
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)a0;
        s16x4r lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * T * 2) : "memory");
        union { struct { s16x4r l, h; } s; bf16x8 v; } u;
        u.s.l = lo; u.s.h = hi;
        return u.v;
    };
    // wait until only the `LEFT` youngest LDS reads of this wave are outstanding; the fragments named become valid here
    auto frag_wait = [](bf16x8 (&a)[TI], bf16x8 (&b)[TI], auto left) {
        constexpr int LEFT = decltype(left)::value;
        if constexpr (TI == 2)
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(LEFT) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(LEFT) : "memory");
    };
    bf16x8 ones;
#pragma unroll
    for (int q = 0; q < 8; ++q) ones[q] = (__bf16)1.0f;
    // MFMA phase over stage `buf`; the slab NST - 1 ahead (frames from mk_next, < 0 = none) is requested one pair per k-step.
    // The fragments of k-step s + 1 are requested before the MFMAs of k-step s are issued (two register sets): with one wave
    // per SIMD (T = 128) nothing else covers the LDS latency, which otherwise sits between any two k-steps.
    auto mma = [&](int buf, int mk_next, int nxt) {
        const unsigned short* ys = smem + buf * STAGE;
        const unsigned short* xs = ys + SK * T;
        constexpr int KS = SK / 16, RD = 4 * TI;             // LDS read instructions of one k-step
        bf16x8 a[2][TI], b[2][TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) a[0][i] = frag(ys, wm0 + 32 * i, 0);
#pragma unroll
        for (int jj = 0; jj < TI; ++jj) b[0][jj] = frag(xs, wn0 + 32 * jj, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = ks & 1, oth = cur ^ 1;
            if (ks + 1 < KS) {
#pragma unroll
                for (int i = 0; i < TI; ++i) a[oth][i] = frag(ys, wm0 + 32 * i, ks + 1);
#pragma unroll
                for (int jj = 0; jj < TI; ++jj) b[oth][jj] = frag(xs, wn0 + 32 * jj, ks + 1);
            }
            if (mk_next >= 0 && ks < NI) issue_pair(mk_next, nxt, ks);
            if (ks + 1 < KS) frag_wait(a[cur], b[cur], std::integral_constant<int, RD>{});
            else frag_wait(a[cur], b[cur], std::integral_constant<int, 0>{});
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int jj = 0; jj < TI; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][jj], acc[i][jj], 0, 0, 0);
            if (bias_wave) {                                                      // scalar condition: no exec masking around the MFMAs
#pragma unroll
                for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], ones, accb[i], 0, 0, 0);
            }
        }
    };
    const int niter = (mend - mbeg + SK - 1) / SK;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < niter) {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue_pair(mbeg + s * SK, s, i);
        }
    int buf = 0, nxt = NST - 1;
    for (int it = 0; it < niter; ++it) {
        const int younger = niter - 1 - it < NST - 2 ? niter - 1 - it : NST - 2;      // slabs requested after slab `it`
        if (NST >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
        else if (NST >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // bare barrier: every wave's loads of slab `it` have landed, and every wave has finished the MFMA phase of slab it - 1,
        // whose stage the requests of this iteration overwrite (all its fragment reads were waited for in frag_wait)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        mma(buf, it + NST - 1 < niter ? mbeg + (it + NST - 1) * SK : -1, nxt);
        buf = buf + 1 == NST ? 0 : buf + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    // ---- epilogue
    const int l31 = lane & 31, lh = lane >> 5;
    if (p.splits > 1) {
        // partial tile -> workspace block (laid out like dW), plain stores; the reduction kernel applies oscale
        float* wb = ws + (int64_t)zs * wgr_block_elems(p.N, p.taps, p.Cin);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int jj = 0; jj < TI; ++jj) {
                const int c = c0 + wn0 + 32 * jj + l31;
                if (c < p.Cin) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        wb[((int64_t)n * p.taps + j) * p.Cin + c] = acc[i][jj][r];
                    }
                }
            }
        if (do_bias && wn0 == 0 && l31 == 0) {
            float* bb = wb + (int64_t)p.N * p.taps * p.Cin;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) bb[n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh] = accb[i][r];
        }
        return;
    }
    // one split: this workgroup owns its tile of dW.  All 16 loads of an accumulator block are requested before the first add
    // (as a chain of `*dst += v` every store orders itself behind the next load: 64 memory round trips per wave).
    float* dW = p.dW + (int64_t)bz * p.sWb;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TI; ++jj) {
            const int c = c0 + wn0 + 32 * jj + l31;
            if (c < p.Cin) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    old[r] = dW[(int64_t)n * p.ldw + (int64_t)j * p.Cin + c];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    dW[(int64_t)n * p.ldw + (int64_t)j * p.Cin + c] = old[r] + (p.oscale ? p.oscale[n] : 1.f) * acc[i][jj][r];
                }
            }
        }
    if (do_bias && wn0 == 0 && l31 == 0) {
        float* db = p.db + (int64_t)bz * p.sDb;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                db[n] += (p.oscale ? p.oscale[n] : 1.f) * accb[i][r];
            }
    }
}

// dW[bz][n][k] += oscale[n] * sum_sp ws[bz * splits + sp][n][k]  (k over taps x Cin), db[bz][n] likewise: one float4 per thread,
// the splits added in index order.  Algorithmic bytes: splits x block read + dW read + written.
__global__ __launch_bounds__(256) void wgrad_split_reduce_kernel(const float* __restrict__ ws, int splits, long long blk, long long E,
                                                                  int K, FastDiv fdK, const float* __restrict__ oscale,
                                                                  float* __restrict__ dW, long long ldw, float* __restrict__ db,
                                                                  int N, long long sWb, long long sDb) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int bz = blockIdx.y;
    const long long tot = db ? E + N : E;
    if (i4 >= tot) return;
    const float* src = ws + (long long)bz * splits * blk + i4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 8 <= splits; sp += 8) {                      // eight loads in flight, added in index order
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(src + (long long)(sp + q) * blk);
#pragma unroll
        for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
    for (; sp < splits; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long long)sp * blk);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (i4 < E) {                                             // K % 4 == 0: the four elements share their row n
        const int n = fd_div((int)(i4 >> 2), fdK), k = (int)i4 - n * K;      // fdK divides by K / 4
        const float sc = oscale ? oscale[n] : 1.f;
        float4* d = reinterpret_cast<float4*>(dW + (long long)bz * sWb + (long long)n * ldw + k);
        float4 o = *d;
        o.x += sc * s.x; o.y += sc * s.y; o.z += sc * s.z; o.w += sc * s.w;
        *d = o;
    } else {
        const int n = (int)(i4 - E);
        float* d = db + (long long)bz * sDb + n;
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n + q < N) d[q] += (oscale ? oscale[n + q] : 1.f) * v[q];
    }
}

// Launch plan.  Returns 1 when the ring kernel took the problem, 0 when it declines (the caller falls back to the kernels of
// wgrad_bf16.hip).  `p` arrives filled in except chunk / splits / y_bytes / x_bytes.  ws may be null: then only problems that
// need no split are taken.
int osp_launch_wgrad_ring(WgradB& p, int64_t batch, float* ws, int64_t ws_bytes, hipStream_t stream) {
    static int on = -1, tgt128 = 0, tgt64 = 0, nst128 = 2, nst64 = 2;
    if (on < 0) {
        { const char* e = getenv("OSP_WGRAD_RING_NST128"); nst128 = e ? atoi(e) : 2; }
        { const char* e = getenv("OSP_WGRAD_RING_NST64"); nst64 = e ? atoi(e) : 2; }
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_kernel<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 64 * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_kernel<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 64 * 64 * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_kernel<128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 128 * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_kernel<128, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 64 * 128 * 2);
        const char* e = getenv("OSP_WGRAD_RING"); on = (e && atoi(e) == 0) ? 0 : 1;
        e = getenv("OSP_WGRAD_RING_T128"); tgt128 = e ? atoi(e) : 512;
        e = getenv("OSP_WGRAD_RING_T64"); tgt64 = e ? atoi(e) : 1024;
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_kernel<128, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 64 * 128 * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_kernel<64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 64 * 64 * 2);
    }
    if (!on) return 0;
    const int64_t M = p.M, N = p.N, Cin = p.Cin, taps = p.taps;
    const bool cin32 = Cin == 32;
    if (!(p.y_bf16 && p.x_bf16 && !p.arow && N % 64 == 0 && (Cin % 64 == 0 || cin32) && p.ldy % 8 == 0 && p.ldx % 8 == 0 &&
          ((reinterpret_cast<uintptr_t>(p.dY) | reinterpret_cast<uintptr_t>(p.X)) & 15) == 0 && p.sYb % 8 == 0 && p.sXb % 8 == 0 &&
          p.ldw % 4 == 0 && p.sWb % 4 == 0 && (reinterpret_cast<uintptr_t>(p.dW) & 15) == 0))
        return 0;
    const int64_t rows_x = (M / p.Trows) * (int64_t)p.Hin * p.Tin;
    const int64_t yb = ((M - 1) * p.ldy + N) * 2, xb = ((rows_x - 1) * p.ldx + Cin) * 2;
    if (!(yb > 0 && xb > 0 && yb < (int64_t)0x7fffff00 && xb < (int64_t)0x7fffff00)) return 0;      // 31-bit buffer offsets
    const int64_t T_ = (N % 128 == 0 && Cin % 128 == 0) ? 128 : 64;
    const int64_t tl = (N / T_) * taps * cdiv(Cin, T_) * batch, target = T_ == 128 ? tgt128 : tgt64;
    const int64_t slabs = cdiv(M, 64);
    int64_t sp = tl >= target ? 1 : (target + tl / 2) / tl;
    if (sp > slabs / 4) sp = slabs / 4 > 0 ? slabs / 4 : 1;      // at least four slabs per split: below that the prologue and the partial tile dominate
    if (sp > 1 && !ws) return 0;                               // a split needs the workspace
    const int64_t blk = wgr_block_elems(N, taps, Cin);
    const int64_t cap = ws ? ws_bytes / (blk * 4 * batch) : 1;
    if (sp > cap) sp = cap;
    if (sp < 1) sp = 1;
    int64_t ch = cdiv(cdiv(M, sp), 64) * 64;
    sp = cdiv(M, ch);
    if (sp > 1 && (reinterpret_cast<uintptr_t>(ws) & 15) != 0) return 0;
    p.chunk = (int)ch; p.splits = (int)sp; p.y_bytes = (unsigned)yb; p.x_bytes = (unsigned)xb;
    const dim3 g((unsigned)(tl * sp));
    if (T_ == 128) {
        osp_note_symbol("conv_wgrad_ring_kernel<128>");
        if (nst128 == 2) hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 2>), g, dim3(256), 2 * 2 * 64 * 128 * 2, stream, p, ws);
        else if (nst128 == 3) hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 3>), g, dim3(256), 3 * 2 * 64 * 128 * 2, stream, p, ws);
        else hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 4>), g, dim3(256), 4 * 2 * 64 * 128 * 2, stream, p, ws);
    } else {
        osp_note_symbol("conv_wgrad_ring_kernel<64>");
        if (nst64 == 2) hipLaunchKernelGGL((conv_wgrad_ring_kernel<64, 2>), g, dim3(256), 2 * 2 * 64 * 64 * 2, stream, p, ws);
        else if (nst64 == 3) hipLaunchKernelGGL((conv_wgrad_ring_kernel<64, 3>), g, dim3(256), 3 * 2 * 64 * 64 * 2, stream, p, ws);
        else hipLaunchKernelGGL((conv_wgrad_ring_kernel<64, 4>), g, dim3(256), 4 * 2 * 64 * 64 * 2, stream, p, ws);
    }
    if (sp > 1) {
        const int64_t E = N * taps * Cin, K = taps * Cin, tot = p.db ? E + N : E;
        hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3((unsigned)cdiv(tot, 1024), (unsigned)batch), dim3(256), 0, stream, ws, (int)sp,
                           (long long)blk, (long long)E, (int)K, make_fastdiv((unsigned)(K / 4)), p.oscale, p.dW, (long long)p.ldw, p.db,
                           (int)N, (long long)p.sWb, (long long)p.sDb);
    }
    return 1;
}
