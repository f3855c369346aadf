// Direct (VALU) kernels for the Cin = 1 first layers of the discriminators: DiscriminatorP convs[0] (1 -> 32, k(5,1),
// _discriminators.py:53) and DiscriminatorR convs[0] (1 -> 64, k(7,5), _discriminators.py:155).  With one input channel
// the contraction depth is only `taps` (5 / 35): a GEMM tile would be > 90 % padding, and the layer is HBM-bound
// (it writes 32-64 channels per input sample).  One lane per output channel, the input sample is wave-uniform.
//   forward : y[m, n] = lrelu(b[n] + sum_j w[n, j] * x[in(m, j)])                 algorithmic bytes/row: Cout*2 (bf16 out)
//   wgrad   : dw[n, j] += sum_m dy[m, n] * x[in(m, j)],  db[n] += sum_m dy[m, n]   bytes/row: Cout*2 read
// Row geometry is the 2-D map of gemm_bf16.hip: m -> (u, th, tw); tap j -> (kh, kw);
//   in = ((u*Hin + th*sh + kh - ph) * Win + tw*sw + kw - pw), zero outside [0,Hin) x [0,Win).
#include "osp_common.h"
#include <cstdlib>

#define SC_MAXTAPS 40

struct SmallCin {
    const float* x; const void* y; int y_bf16; const float* w; const float* b; float* dw; float* db; float* ws;
    int M, Trows, Wrows, Hin, Win, Cout, taps, KW, sh, sw, ph, pw; float slope; int lrelu;
};

// Tap samples of a row: lane j (< taps) of each Cout-lane group fetches x for tap j once; the FMA loop then reads it
// with a lane broadcast (the sample is uniform over the group's output channels).
__device__ __forceinline__ float sc_tap_sample(const SmallCin& p, int64_t m, int jl, int dh, int dw) {
    if (m >= p.M || jl >= p.taps) return 0.f;
    const int u = (int)(m / p.Trows), t = (int)(m - (int64_t)u * p.Trows), th = t / p.Wrows, tw = t - th * p.Wrows;
    const int hh = th * p.sh + dh, ww = tw * p.sw + dw;
    if (hh < 0 || hh >= p.Hin || ww < 0 || ww >= p.Win) return 0.f;
    return p.x[((int64_t)u * p.Hin + hh) * p.Win + ww];
}

// Broadcast of tap sample j inside a Cout-lane group.  With one row per wave (Cout == 64) the source lane is a
// compile-time constant -> v_readlane_b32 (scalar broadcast, no LDS traffic); otherwise ds_bpermute via __shfl.
template <int J>
__device__ __forceinline__ float sc_bcast(float xs, int gbase, bool one_row) {
    if (one_row) return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xs), J));
    return __shfl(xs, gbase + J, 64);
}
template <int J, int JMAX>
struct ScFma {
    static __device__ __forceinline__ void fwd(const float (&w)[SC_MAXTAPS], float xs, int gbase, bool one_row, int taps, float& acc) {
        if (J < taps) acc = fmaf(w[J], sc_bcast<J>(xs, gbase, one_row), acc);
        ScFma<J + 1, JMAX>::fwd(w, xs, gbase, one_row, taps, acc);
    }
    static __device__ __forceinline__ void bwd(float (&acc)[SC_MAXTAPS], float g, float xs, int gbase, bool one_row, int taps) {
        if (J < taps) acc[J] = fmaf(g, sc_bcast<J>(xs, gbase, one_row), acc[J]);
        ScFma<J + 1, JMAX>::bwd(acc, g, xs, gbase, one_row, taps);
    }
};
template <int JMAX>
struct ScFma<JMAX, JMAX> {
    static __device__ __forceinline__ void fwd(const float (&)[SC_MAXTAPS], float, int, bool, int, float&) {}
    static __device__ __forceinline__ void bwd(float (&)[SC_MAXTAPS], float, float, int, bool, int) {}
};

// rows per wave iteration RPW = 64 / Cout (Cout in {16, 32, 64}); requires taps <= Cout
__global__ __launch_bounds__(256) void smallcin_fwd_kernel(SmallCin p) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / p.Cout, n = lane % p.Cout, sub = lane / p.Cout, gbase = sub * p.Cout;
    float w[SC_MAXTAPS];
#pragma unroll
    for (int j = 0; j < SC_MAXTAPS; ++j) w[j] = j < p.taps ? p.w[n * p.taps + j] : 0.f;
    const float bias = p.b ? p.b[n] : 0.f;
    const int kh = n / p.KW, dh = kh - p.ph, dw = (n - kh * p.KW) - p.pw;      // this lane's tap (n doubles as tap id)
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t m0 = wave_id * rpw; m0 < p.M; m0 += nwaves * rpw) {
        const int64_t m = m0 + sub;
        const float xs = sc_tap_sample(p, m, n, dh, dw);
        float acc = bias;
        ScFma<0, SC_MAXTAPS>::fwd(w, xs, gbase, rpw == 1, p.taps, acc);
        if (m >= p.M) continue;
        if (p.lrelu) acc = acc > 0.f ? acc : acc * p.slope;
        if (p.y_bf16) reinterpret_cast<__bf16*>(const_cast<void*>(p.y))[m * p.Cout + n] = (__bf16)acc;
        else reinterpret_cast<float*>(const_cast<void*>(p.y))[m * p.Cout + n] = acc;
    }
}

__global__ __launch_bounds__(256) void smallcin_wgrad_kernel(SmallCin p) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rpw = 64 / p.Cout, n = lane % p.Cout, sub = lane / p.Cout, gbase = sub * p.Cout;
    float acc[SC_MAXTAPS];
#pragma unroll
    for (int j = 0; j < SC_MAXTAPS; ++j) acc[j] = 0.f;
    float bsum = 0.f;
    const int kh = n / p.KW, dh = kh - p.ph, dw = (n - kh * p.KW) - p.pw;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t m0 = wave_id * rpw; m0 < p.M; m0 += nwaves * rpw) {
        const int64_t m = m0 + sub;
        const float xs = sc_tap_sample(p, m, n, dh, dw);
        float g = 0.f;
        if (m < p.M)
            g = p.y_bf16 ? __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(p.y)[m * p.Cout + n]) << 16)
                         : reinterpret_cast<const float*>(p.y)[m * p.Cout + n];
        bsum += g;
        ScFma<0, SC_MAXTAPS>::bwd(acc, g, xs, gbase, rpw == 1, p.taps);
    }
    // combine the `rpw` sub-rows of a wave (lanes n, n+Cout, ...), then the 4 waves, then one atomic per (n, j)
    for (int j = 0; j <= p.taps; ++j) {
        float v = 0.f;
        if (j < p.taps) {
#pragma unroll
            for (int q = 0; q < SC_MAXTAPS; ++q) v = (q == j) ? acc[q] : v;
        } else v = bsum;
        for (int o = p.Cout; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        red[wv][lane] = v;
        __syncthreads();
        if (threadIdx.x < p.Cout) {
            const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (j < p.taps) atomicAdd(p.dw + threadIdx.x * p.taps + j, s);
            else if (p.db) atomicAdd(p.db + threadIdx.x, s);
        }
    }
}


// ------------------------------------------------------------------------------------------------ MFMA variants (bf16 y / dy)
// The VALU kernels above are issue-bound (2 instructions per tap per row per wave, one dependent load per row).  With bf16
// activations the same contraction maps on v_mfma_f32_32x32x16_bf16 with the taps as the (zero-padded) K dimension:
//   forward : D[i = row][j = n]   = sum_tap X[row][tap] * W[tap][n]      (K = taps padded to 16 * KSTEPS)
//   wgrad   : D[i = tap][j = n]   = sum_row X[row][tap] * dY[row][n]      (K = rows, 16 per MFMA; tap == taps is a column of ones = db)
// X fragments are gathered straight from global memory (the (kh, kw) window of a row is a few cache lines that the taps of
// the same and of neighbouring rows re-read through L1/L2); no LDS staging is needed because the kernels are bound by the
// y / dy stream (Cout * 2 bytes per row), not by the gathers.
typedef float sc_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 sc_bf16x8 __attribute__((ext_vector_type(8)));

template <int KSTEPS, int NT>
__global__ __launch_bounds__(256) void smallcin_fwd_mfma_kernel(SmallCin p) {
    __shared__ __attribute__((aligned(16))) unsigned short stage[4][32 * (NT * 32 + 8)];
    const int lane = threadIdx.x & 63, li = lane & 31, kg = lane >> 5;
    sc_bf16x8 bw[KSTEPS][NT];
    int toff[KSTEPS][8], tdh[KSTEPS][8], tdw[KSTEPS][8];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tap = ks * 16 + kg * 8 + q, kh = tap / p.KW, kw = tap - kh * p.KW;
            tdh[ks][q] = tap < p.taps ? kh : (1 << 20);                       // padded taps fail the row bound check
            tdw[ks][q] = kw;
            toff[ks][q] = kh * p.Win + kw;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
            {
                const float wv = p.w[(nt * 32 + li) * p.taps + (tap < p.taps ? tap : 0)];   // unconditional load
                bw[ks][nt][q] = (__bf16)(tap < p.taps ? wv : 0.f);
            }
        }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = p.b ? p.b[nt * 32 + li] : 0.f;
    const int ntiles = (p.M + 31) >> 5;
    const int nwaves = gridDim.x * 4;
    __bf16* __restrict__ y = reinterpret_cast<__bf16*>(const_cast<void*>(p.y));
    for (int tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += nwaves) {
        const int m = tile * 32 + li;
        const bool valid = m < p.M;
        const int mm = valid ? m : p.M - 1;
        const int u = mm / p.Trows, t = mm - u * p.Trows, th = t / p.Wrows, tw = t - th * p.Wrows;
        const int h0 = th * p.sh - p.ph, w0 = tw * p.sw - p.pw;
        const int xbase = (u * p.Hin + h0) * p.Win + w0;                            // < 2^31 elements (checked on the host)
        sc_f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        // the f32 input sample (a spectrogram magnitude / wav sample with a wide dynamic range) enters as hi + lo bf16
        // halves, i.e. with ~16 mantissa bits; the weights are bf16 like in every other layer of the bf16 mode
        sc_bf16x8 a[KSTEPS], al[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int hh = h0 + tdh[ks][q], ww = w0 + tdw[ks][q];
                const bool ok = valid && (unsigned)hh < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win;
                const float xv = p.x[ok ? xbase + toff[ks][q] : 0];              // unconditional load, index select (no branch)
                const float xs = ok ? xv : 0.f;
                a[ks][q] = (__bf16)xs;
                al[ks][q] = (__bf16)(xs - (float)a[ks][q]);
            }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bw[ks][nt], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bw[ks][nt], acc[nt], 0, 0, 0);
            }
        // Output rows through a wave-private LDS tile: in the MFMA layout a lane owns one channel, i.e. 2-byte stores, 32 per lane
        // and tile, each instruction touching 64-byte pieces of two rows; row-major out of LDS a lane stores 16 bytes (8 channels)
        // and the tile leaves in 4 instructions (2 for 32 channels) of whole rows.
        constexpr int SP = NT * 32 + 8;                                          // bf16 elements per staged row
        unsigned short* st = stage[threadIdx.x >> 6];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = kg * 4 + (r >> 2) * 8 + (r & 3);
                float v = acc[nt][r] + bias[nt];
                if (p.lrelu) v = v > 0.f ? v : v * p.slope;
                st[lrow * SP + nt * 32 + li] = __builtin_bit_cast(unsigned short, (__bf16)v);
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int CPR = NT * 4, RPI = 64 / CPR;                              // 16-byte chunks per row, rows per wave instruction
        const int cc = lane % CPR, rr = lane / CPR;
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int lrow = it * RPI + rr, row = tile * 32 + lrow;
            if (row < p.M)
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(y) + (int64_t)row * p.Cout + cc * 8) =
                    *reinterpret_cast<const uint4*>(st + lrow * SP + cc * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// dw[n][tap] (tap < taps) and db[n] (the ones column tap == taps); TT = tap tiles of 32, NT = Cout / 32.
template <int TT, int NT>
__global__ __launch_bounds__(256) void smallcin_wgrad_mfma_kernel(SmallCin p) {
    __shared__ float red[TT * NT * 16 * 64];
    __shared__ __attribute__((aligned(16))) unsigned short dy_stage[4][16 * (NT * 32 + 8)];
    const int lane = threadIdx.x & 63, li = lane & 31, kg = lane >> 5, wv = threadIdx.x >> 6;
    int tdh[TT], tdw[TT], toff[TT];
    float tfill[TT];                                                          // value of a non-tap column: 1 for the bias column
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int tap = tt * 32 + li, kh = tap / p.KW, kw = tap - kh * p.KW;
        tfill[tt] = tap == p.taps ? 1.f : 0.f;
        tdh[tt] = tap < p.taps ? kh : (1 << 20); tdw[tt] = kw; toff[tt] = kh * p.Win + kw;
    }
    sc_f32x16 acc[TT][NT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][nt][r] = 0.f;
    const unsigned short* __restrict__ dy = reinterpret_cast<const unsigned short*>(p.y);
    const int Ho = p.Trows / p.Wrows;
    const int nsteps = (p.M + 15) >> 4, nwaves = gridDim.x * 4;
    // dY rows of a step (16 rows x Cout bf16) are fetched as 16-byte chunks (whole rows) into a wave-private LDS tile and read
    // back TRANSPOSED as the B operand (ds_read_b64_tr_b16, the addressing of attention_train.hip's value_product): in the MFMA
    // layout a lane owns one channel and 8 rows, i.e. eight 2-byte loads at a row stride -- 16 load instructions per step for
    // 2 KB.  The k index e of both operands is row 8 (e >> 2) + 4 kg + (e & 3) of the step.
    constexpr int LD = NT * 32 + 8, CPR = NT * 4;                              // staged row pitch (bf16), 16-byte chunks per row
    unsigned short* dyl = dy_stage[wv];
    const int r16 = lane & 15, g16 = (lane >> 4) & 1;
    // the dY rows of the wave's NEXT step are requested before this step's x gather and MFMAs (round 5: with one step in flight the
    // loop was a chain of HBM latency + L2 latency + compute per 16 rows: 0.8 TB/s of dY)
    constexpr int PER = 16 * CPR / 64;
    uint4 nv[PER];
    auto fetch = [&](int st) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i, rowl = c / CPR, cc = c - rowl * CPR, m = st * 16 + rowl;
            nv[i] = make_uint4(0, 0, 0, 0);
            if (st < nsteps && m < p.M) nv[i] = *reinterpret_cast<const uint4*>(dy + (int64_t)m * p.Cout + cc * 8);
        }
    };
    fetch(blockIdx.x * 4 + wv);
    for (int step = blockIdx.x * 4 + wv; step < nsteps; step += nwaves) {
        const int ms = step * 16;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i, rowl = c / CPR, cc = c - rowl * CPR;
            *reinterpret_cast<uint4*>(dyl + rowl * LD + cc * 8) = nv[i];
        }
        fetch(step + nwaves);
        sc_bf16x8 a[TT], al[TT], b[NT];                                        // x as hi + lo bf16 halves (see the forward kernel)
#pragma unroll
        for (int run = 0; run < 2; ++run) {
            const int m0 = ms + 8 * run + 4 * kg;
            int u = m0 / p.Trows, t = m0 - u * p.Trows, th = t / p.Wrows, tw = t - th * p.Wrows;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int q = 4 * run + qq, m = m0 + qq;
                const bool valid = m < p.M;
                const int h0 = th * p.sh - p.ph, w0 = tw * p.sw - p.pw;
                const int xbase = (u * p.Hin + h0) * p.Win + w0;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const int hh = h0 + tdh[tt], ww = w0 + tdw[tt];
                    const bool ok = valid && (unsigned)hh < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win;   // tdh = 2^20 for non-taps
                    const float xv = p.x[ok ? xbase + toff[tt] : 0];               // unconditional load, index select (no branch)
                    const float xs = ok ? xv : (valid ? tfill[tt] : 0.f);
                    a[tt][q] = (__bf16)xs;
                    al[tt][q] = (__bf16)(xs - (float)a[tt][q]);
                }
                if (++tw == p.Wrows) { tw = 0; if (++th == Ho) { th = 0; ++u; } }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 32 + 16 * g16 + 4 * (r16 & 3);
            const unsigned short* a0 = dyl + (4 * kg + (r16 >> 2)) * LD + col;
            const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)a0;
            typedef short s16x4_sc __attribute__((ext_vector_type(4)));
            s16x4_sc lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(8 * LD * 2) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
            union { struct { s16x4_sc l, h; } s2; sc_bf16x8 vv; } uu;
            uu.s2.l = lo; uu.s2.h = hi;
            b[nt] = uu.vv;
        }
        __builtin_amdgcn_wave_barrier();                                      // the tile is free for the next step's rows
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tt], b[nt], acc[tt][nt], 0, 0, 0);
                acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt], b[nt], acc[tt][nt], 0, 0, 0);
            }
    }
    // sum the 4 waves in LDS (wave 0 stores, the others add in turn), then one atomic per valid (tap, n)
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* s = red + ((tt * NT + nt) * 16 + r) * 64 + lane;
                        *s = (w == 0 ? 0.f : *s) + acc[tt][nt][r];
                    }
        }
        __syncthreads();
    }
    if (p.ws) {
        // two-stage reduction: every workgroup STORES its partial tile; smallcin_wgrad_reduce_kernel (next launch) sums the
        // workgroups.  With atomics the 256 workgroups added into the same 2 304 addresses (taps x Cout + Cout): same-address
        // atomics serialise at the memory side, and the kernel's time was proportional to the NUMBER OF WORKGROUPS
        // (256 / 512 / 1024 workgroups: 170 / 215 / 350 us for the 1 -> 64 (7, 5) layer).
        for (int e = threadIdx.x; e < TT * NT * 16 * 64; e += 256) p.ws[(int64_t)blockIdx.x * (TT * NT * 16 * 64) + e] = red[e];
        return;
    }
    for (int e = threadIdx.x; e < TT * NT * 16 * 64; e += 256) {
        const int l = e & 63, r = (e >> 6) & 15, tile = e >> 10, nt = tile % NT, tt = tile / NT;
        const int tap = tt * 32 + (r >> 2) * 8 + (l >> 5) * 4 + (r & 3), n = nt * 32 + (l & 31);
        if (tap < p.taps) atomicAdd(p.dw + n * p.taps + tap, red[e]);
        else if (tap == p.taps && p.db) atomicAdd(p.db + n, red[e]);
    }
}

// stage 2: a workgroup owns 64 consecutive accumulator entries (same (tap, n) map as above); its 4 waves each sum a quarter of the
// `nblocks` partial tiles with 8 loads in flight per lane (coalesced 256-byte rows), meet in LDS, and the first wave adds to
// dw / db -- a plain read-modify-write, every entry has one owner
__global__ __launch_bounds__(256) void smallcin_wgrad_reduce_kernel(const float* __restrict__ ws, int nblocks, int E, int NT, int taps,
                                                                    float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int per = (nblocks + 3) >> 2, b0 = wv * per, b1 = min(nblocks, b0 + per);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += ws[(int64_t)(b + k) * E + e];
    }
    for (; b < b1; ++b) s[0] += ws[(int64_t)b * E + e];
    part[wv][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (wv) return;
    const float tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int l = e & 63, r = (e >> 6) & 15, tile = e >> 10, nt = tile % NT, tt = tile / NT;
    const int tap = tt * 32 + (r >> 2) * 8 + (l >> 5) * 4 + (r & 3), n = nt * 32 + (l & 31);
    if (tap < taps) dw[n * taps + tap] += tot;
    else if (tap == taps && db) db[n] += tot;
}

static int smallcin_fill(SmallCin& p, int64_t M, int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t Cout,
                         int64_t taps, int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw) {
    if (!(Cout == 16 || Cout == 32 || Cout == 64) || taps < 1 || taps > 63 || taps % KW != 0 || M <= 0 ||
        M % Trows != 0 || Trows % Wrows != 0)
        return 0;
    p.M = (int)M; p.Trows = (int)Trows; p.Wrows = (int)Wrows; p.Hin = (int)Hin; p.Win = (int)Win; p.Cout = (int)Cout;
    p.taps = (int)taps; p.KW = (int)KW; p.sh = (int)sh; p.sw = (int)sw; p.ph = (int)ph; p.pw = (int)pw;
    return 1;
}

// x: (U, Hin, Win) f32 single-channel input; w: (Cout, taps) f32; y: (M, Cout) f32|bf16
extern "C" int osp_smallcin_conv_fwd(const float* x, const float* w, const float* b, void* y, int64_t y_bf16, int64_t M,
                                     int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t Cout, int64_t taps,
                                     int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t lrelu, float slope,
                                     hipStream_t stream) {
    OSP_CHECK_ARG(x && w && y, "null operand");
    SmallCin p;
    OSP_CHECK_ARG(smallcin_fill(p, M, Trows, Wrows, Hin, Win, Cout, taps, KW, sh, sw, ph, pw), "unsupported small-Cin geometry");
    OSP_CHECK_ARG(M * Cout < (1ll << 31) && (M / Trows) * Hin * Win < (1ll << 31), "small-Cin operand exceeds 32-bit indexing");
    p.x = x; p.w = w; p.b = b; p.y = y; p.y_bf16 = (int)y_bf16; p.dw = nullptr; p.db = nullptr; p.ws = nullptr; p.lrelu = (int)lrelu; p.slope = slope;
    if (y_bf16 && (Cout == 32 || Cout == 64) && taps <= 48 && !getenv("OSP_SMALLCIN_VALU")) {
        const int64_t tiles = cdiv(M, 32), nb = cdiv(tiles, 4);
        static int64_t fcap = 0;
        if (!fcap) { fcap = 512; }
        const dim3 grid((unsigned)(nb < fcap ? nb : fcap)), block(256);     // 2 blocks / CU: the per-wave weight prologue is amortised
        const int ks = (int)cdiv(taps, 16);
#define SC_FWD(KS, NT_) hipLaunchKernelGGL((smallcin_fwd_mfma_kernel<KS, NT_>), grid, block, 0, stream, p)
        if (Cout == 32) { if (ks == 1) SC_FWD(1, 1); else if (ks == 2) SC_FWD(2, 1); else SC_FWD(3, 1); }
        else            { if (ks == 1) SC_FWD(1, 2); else if (ks == 2) SC_FWD(2, 2); else SC_FWD(3, 2); }
#undef SC_FWD
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    OSP_CHECK_ARG(taps <= Cout && taps <= SC_MAXTAPS, "unsupported small-Cin geometry");
    const int64_t blocks = cdiv(M, 4 * (64 / Cout) * 8);
    hipLaunchKernelGGL(smallcin_fwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// dw (Cout, taps) and db (Cout) are accumulated.  ws (optional): ws_floats >= 256 * 4096 floats of scratch for the two-stage
// reduction of the MFMA variant (without it: f32 atomics from every workgroup).
extern "C" int osp_smallcin_conv_wgrad(const float* x, const void* dy, int64_t y_bf16, float* dw, float* db, int64_t M,
                                       int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t Cout, int64_t taps,
                                       int64_t KW, int64_t sh, int64_t sw, int64_t ph, int64_t pw, float* ws, int64_t ws_floats,
                                       hipStream_t stream) {
    OSP_CHECK_ARG(x && dy && dw, "null operand");
    SmallCin p;
    OSP_CHECK_ARG(smallcin_fill(p, M, Trows, Wrows, Hin, Win, Cout, taps, KW, sh, sw, ph, pw), "unsupported small-Cin geometry");
    OSP_CHECK_ARG(M * Cout < (1ll << 31) && (M / Trows) * Hin * Win < (1ll << 31), "small-Cin operand exceeds 32-bit indexing");
    p.x = x; p.w = nullptr; p.b = nullptr; p.y = dy; p.y_bf16 = (int)y_bf16; p.dw = dw; p.db = db; p.lrelu = 0; p.slope = 0.f; p.ws = nullptr;
    if (y_bf16 && (Cout == 32 || Cout == 64) && taps < 64 && !getenv("OSP_SMALLCIN_VALU")) {
        const int64_t steps = cdiv(M, 16), nb = cdiv(steps, 4 * 8);
        static int64_t wg_cap = 0;
        if (!wg_cap) { const char* e = getenv("OSP_SMALLCIN_WGS"); wg_cap = e ? atoi(e) : 512; }      // 256 / 512 / 1024: 87 / 61 / 69 us (profiles/r05_smallcin_wgs.txt)
        const dim3 grid((unsigned)(nb < wg_cap ? nb : wg_cap)), block(256);  // one atomic epilogue per block
        const int tt_ = taps < 32 ? 1 : 2, nt_ = Cout == 32 ? 1 : 2, E = tt_ * nt_ * 1024;
        if (ws && ws_floats >= (int64_t)grid.x * E) p.ws = ws;
#define SC_WG(TT_, NT_) hipLaunchKernelGGL((smallcin_wgrad_mfma_kernel<TT_, NT_>), grid, block, 0, stream, p)
        if (Cout == 32) { if (taps < 32) SC_WG(1, 1); else SC_WG(2, 1); }
        else            { if (taps < 32) SC_WG(1, 2); else SC_WG(2, 2); }
#undef SC_WG
        if (p.ws)
            hipLaunchKernelGGL(smallcin_wgrad_reduce_kernel, dim3((unsigned)(E / 64)), dim3(256), 0, stream, (const float*)ws, (int)grid.x, E,
                               nt_, (int)taps, dw, db);
        OSP_LAUNCH_CHECK();
        return OSP_OK;
    }
    OSP_CHECK_ARG(taps <= Cout && taps <= SC_MAXTAPS, "unsupported small-Cin geometry");
    const int64_t blocks = cdiv(M, 4 * (64 / Cout) * 64);
    hipLaunchKernelGGL(smallcin_wgrad_kernel, dim3((unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024)), dim3(256), 0, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
