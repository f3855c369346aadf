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

// ---- fused split reduction (round 6): the LAST workgroup to deliver a partial of an output tile adds the tile's splits up.
// Every workgroup publishes its partial (plain stores, agent-scope release fence), then bumps the tile's arrival counter; the one
// that reads splits - 1 resets the counter for the next launch, takes an acquire fence and sums the `splits` partial tiles in INDEX
// order into dW (`+=`, oscale) -- the arithmetic of wgrad_split_reduce_kernel element by element, so the result is bit-identical
// to the two-kernel form and independent of which workgroup happens to arrive last.  One launch per weight gradient instead of
// two (53-63 launches of 6 us per training step).  Counters: one u32 per output tile, zero between launches, in a small buffer
// the library keeps per launch stream (launches of a stream are ordered, so they can share it).
template <int T>
__device__ __forceinline__ void wgr_last_arriver_reduce(const WgradB& p, const float* __restrict__ ws, unsigned* __restrict__ cnt, int* lds_flag,
                                                        int tile_id, int bz, int n0, int j, int c0, bool do_bias) {
    const int tid = threadIdx.x;
    __threadfence();                                         // release: this workgroup's partial tile is visible device-wide
    __syncthreads();
    if (tid == 0) {
        const unsigned old = atomicAdd(cnt + tile_id, 1u);
        const int last = old == (unsigned)(p.splits - 1);
        if (last) cnt[tile_id] = 0u;                         // every split has arrived: clean for the next launch on this stream
        lds_flag[0] = last;
    }
    __syncthreads();
    if (!lds_flag[0]) return;
    __threadfence();                                         // acquire: the other splits' partial tiles
    const int64_t blk = wgr_block_elems(p.N, p.taps, p.Cin);
    const float* base = ws + (int64_t)bz * p.splits * blk;
    float* dW = p.dW + (int64_t)bz * p.sWb;
    constexpr int Q = T / 4;                                 // float4 per tile row
    for (int e = tid; e < T * Q; e += 256) {
        const int row = e / Q, c = c0 + (e - row * Q) * 4, n = n0 + row;
        if (c >= p.Cin) continue;
        const float* src = base + ((int64_t)n * p.taps + j) * p.Cin + c;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = 0;
        for (; sp + 8 <= p.splits; sp += 8) {                // eight loads in flight, added in index order
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(src + (int64_t)(sp + q) * blk);
#pragma unroll
            for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
        }
        for (; sp < p.splits; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)sp * blk);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float sc = p.oscale ? p.oscale[n] : 1.f;
        float4* d = reinterpret_cast<float4*>(dW + (int64_t)n * p.ldw + (int64_t)j * p.Cin + c);
        float4 o = *d;
        o.x += sc * s.x; o.y += sc * s.y; o.z += sc * s.z; o.w += sc * s.w;
        *d = o;
    }
    if (do_bias && tid < T) {
        const int n = n0 + tid;
        const float* src = base + (int64_t)p.N * p.taps * p.Cin + n;
        float sacc = 0.f;
        for (int sp = 0; sp < p.splits; ++sp) sacc += src[(int64_t)sp * blk];
        p.db[(int64_t)bz * p.sDb + n] += (p.oscale ? p.oscale[n] : 1.f) * sacc;
    }
}

#include <mutex>
#include <unordered_map>
#define WGR_MAX_TILES 4096
// arrival counters of the launch stream (nullptr: none could be made -- inside a capture, say -- the caller launches the reduce kernel)
static unsigned* wgr_counters(hipStream_t stream) {
    static std::mutex mu;
    static std::unordered_map<hipStream_t, unsigned*> pool;
    static int on = -1;
    // OFF by default -- measured: the agent-scope release / acquire fences (an L2 write-back + invalidate per workgroup on this
    // multi-XCD part) cost far more than the launch they save: 14.8 -> 19.6 ms per training step, the 128-tile ring kernel at 2 % of
    // peak.  OSP_WGRAD_FUSED_REDUCE=1 takes it; tests/test_gpu_wgrad_ring.py keeps it bit-identical.
    if (on < 0) { const char* e = getenv("OSP_WGRAD_FUSED_REDUCE"); on = (e && atoi(e) != 0) ? 1 : 0; }
    if (!on) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find(stream);
    if (it != pool.end()) return it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    unsigned* q = nullptr;
    if (hipMalloc(&q, WGR_MAX_TILES * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemset(q, 0, WGR_MAX_TILES * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(q); return nullptr; }
    pool[stream] = q;
    return q;
}

template <int T, int NST>
__global__ __launch_bounds__(256) void conv_wgrad_ring_kernel(WgradB p, float* __restrict__ ws, unsigned* __restrict__ cnt) {
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
        if (cnt) wgr_last_arriver_reduce<T>(p, ws, cnt, reinterpret_cast<int*>(smem), (bz * ntiles + n0 / T) * inner + within, bz, n0, j, c0, do_bias);
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

// ------------------------------------------------------------------------------------------------ exact f32 (round 5)
// The same plan for the exact-f32 weight gradients of the f32 / "mixed" parity modes (conv_wgrad_f32_kernel, gemm.hip: register-staged
// 16-frame slabs, one __syncthreads() per slab, f32 atomics for the frame splits: 41 launches x 75 us = 3.1 ms of the mixed step).
// f32 operands are staged as they lie by LDS-DMA (a row of 64 channels = 256 B, four rows per wave instruction, no swizzle: a
// fragment read is 32 consecutive floats of one frame), products on v_mfma_f32_32x32x2_f32 whose operands ARE one element per
// lane (A: dY[frame lh][channel l31], B: X[frame lh][channel l31]) -- no transposition anywhere.  64 x 64 tiles, four waves of
// 32 x 32, 64-frame slabs, two stages (66 KB: two workgroups per CU).  The row factor `arow` (padding mask / DropPath) is staged
// beside the slab and multiplied into the A operand; the bias gradient is a VALU sum of the same operand.
// SPLIT (late round 5; osp_conv_wgrad_f32_split_ws, the "mixed" parity mode): the same staging and the same LDS reads, but the products
// run on the bf16 matrix pipe -- every operand element as a (hi, lo) pair of bf16 numbers, hi = bf16(x), lo = bf16(x - hi), and
// lo_a hi_b + hi_a lo_b + hi_a hi_b per 16-frame block (three v_mfma_f32_32x32x16_bf16 where the exact form issues eight
// v_mfma_f32_32x32x2_f32 of twice the length each): <= 1.1e-5 of |a b| per product, f32 accumulation.  The k index of the bf16 MFMA is
// (lane half, element e) and may be ANY enumeration of the block's 16 frames as long as both operands use the same one: frame
// 2 e + lh -- exactly the eight floats lane half lh reads over two groups of four frame pairs below.  The bias gradient stays the
// exact f32 sum of the A operand.
template <bool AROW, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_wgrad_ring_f32_kernel(WgradB p, float* __restrict__ ws, unsigned* __restrict__ cnt) {
    constexpr int T = 64, SK = 64, NI = SK / 16;            // pairs of DMA instructions per wave and slab
    constexpr int STAGE = 2 * SK * T + SK;                   // floats of one stage: dY slab, X slab, row factors
    float* smem = reinterpret_cast<float*>(wgr_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int ctiles = (p.Cin + T - 1) / T, ntiles = p.N / T, inner = p.taps * ctiles;
    const int total = gridDim.x, lin = blockIdx.x, xcd = lin & 7, local = lin >> 3;
    const int per = total >> 3, rem = total & 7;
    const int pid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + local;
    const int grp = pid / inner, within = pid - grp * inner;
    const int zs = grp / ntiles, n0 = (grp - zs * ntiles) * T;
    const int j = within / ctiles, c0 = (within - j * ctiles) * T;
    const int bz = zs / p.splits, sp = zs - bz * p.splits;
    const float* dY = reinterpret_cast<const float*>(p.dY) + (int64_t)bz * p.sYb;
    const float* X = reinterpret_cast<const float*>(p.X) + (int64_t)bz * p.sXb;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const bool do_bias = (p.db != nullptr) && (within == 0);
    const bool bias_wave = __builtin_amdgcn_readfirstlane((int)(do_bias && wn0 == 0)) != 0;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    const int srow = lane >> 4, lslot = lane & 15;
    const int ldy32 = (int)p.ldy, ldx32 = (int)p.ldx;
    const unsigned ycol4 = (unsigned)(n0 + lslot * 4) * 4u, xcol4 = (unsigned)(c0 + lslot * 4) * 4u;
    const bool xcol_ok = c0 + lslot * 4 < p.Cin;
    __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, (int)p.y_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t asrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AROW ? p.arow + (int64_t)bz * p.M : dY), 0,
                                                                    AROW ? p.M * 4 : 0, 0x00020000);
    auto issue_pair = [&](int mk, int buf, int i) {
        float* ys = smem + buf * STAGE;
        float* xs = ys + SK * T;
        const int row0 = 4 * (NI * wave + i), m = mk + row0 + srow;
        const bool mv = m < mend;
        const unsigned yo = mv ? (unsigned)(m * ldy32) * 4u + ycol4 : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (__attribute__((address_space(3))) void*)(ys + row0 * T), 16, yo, 0, 0, 0);
        const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, tt = t + j - p.pad;
        const bool xv = mv && (unsigned)tt < (unsigned)p.Tin && xcol_ok;
        const unsigned xo = xv ? (unsigned)((u * p.Tin + tt) * ldx32) * 4u + xcol4 : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(xs + row0 * T), 16, xo, 0, 0, 0);
    };
    auto issue_arow = [&](int mk, int buf) {               // the slab's 64 row factors, 4 bytes per lane (every wave: same bytes)
        if constexpr (AROW) {
            float* ar = smem + buf * STAGE + 2 * SK * T;
            const int m = mk + lane;
            const unsigned ao = m < mend ? (unsigned)m * 4u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(asrd, (__attribute__((address_space(3))) void*)ar, 4, ao, 0, 0, 0);
        }
    };
    // operands of frame pair kk of stage `buf` (inline asm: a builtin LDS read would make the compiler drain the DMA queue first)
    auto lds_addr = [](const float* q) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)q; };
    auto rd4 = [&](unsigned ya, unsigned xa, unsigned ra, auto gidx, float (&a)[4], float (&b)[4], float (&r)[4]) {
        constexpr int G = decltype(gidx)::value;
#define WGF_RD(Q)                                                                                                                  \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[Q]) : "v"(ya), "n"((4 * G + Q) * 2 * T * 4) : "memory");             \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[Q]) : "v"(xa), "n"((4 * G + Q) * 2 * T * 4) : "memory");             \
        if constexpr (AROW) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[Q]) : "v"(ra), "n"((4 * G + Q) * 8) : "memory");
        WGF_RD(0) WGF_RD(1) WGF_RD(2) WGF_RD(3)
#undef WGF_RD
    };
    auto wait4 = [](float (&a)[4], float (&b)[4], float (&r)[4], auto left) {
        constexpr int LEFT = decltype(left)::value;
        asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                     "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(LEFT) : "memory");
    };
    constexpr int RD = AROW ? 12 : 8;                        // LDS reads of one group of four frame pairs
    auto mma = [&](int buf, int mk_next, int nxt) {
        const float* ys = smem + buf * STAGE;
        const unsigned ya = lds_addr(ys + lh * T + wm0 + l31), xa = lds_addr(ys + SK * T + lh * T + wn0 + l31);
        const unsigned ra = lds_addr(ys + 2 * SK * T + lh);
        float a[2][4], b[2][4], r[2][4];
        bf16x8 sah, sal, sbh, sbl;                           // SPLIT: the (hi, lo) operands of a 16-frame block, filled over two groups
#pragma unroll
        for (int q = 0; q < 4; ++q) { r[0][q] = 1.f; r[1][q] = 1.f; }
        auto group = [&](auto gidx) {
            constexpr int G = decltype(gidx)::value, cur = G & 1, oth = cur ^ 1;
            if constexpr (G + 1 < SK / 8) rd4(ya, xa, ra, std::integral_constant<int, G + 1>{}, a[oth], b[oth], r[oth]);
            if (mk_next >= 0) {
                if constexpr (G < NI) issue_pair(mk_next, nxt, G);
                if constexpr (G == NI) issue_arow(mk_next, nxt);
            }
            if constexpr (G + 1 < SK / 8) wait4(a[cur], b[cur], r[cur], std::integral_constant<int, RD>{});
            else wait4(a[cur], b[cur], r[cur], std::integral_constant<int, 0>{});
            if constexpr (SPLIT) {
                constexpr int E0 = 4 * (G & 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float av = AROW ? a[cur][q] * r[cur][q] : a[cur][q];
                    if (bias_wave) bsum += av;
                    const __bf16 ha = (__bf16)av;
                    sah[E0 + q] = ha; sal[E0 + q] = (__bf16)(av - (float)ha);
                    const float bv = b[cur][q];
                    const __bf16 hb = (__bf16)bv;
                    sbh[E0 + q] = hb; sbl[E0 + q] = (__bf16)(bv - (float)hb);
                }
                if constexpr ((G & 1) == 1) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sal, sbh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sah, sbl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sah, sbh, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float av = AROW ? a[cur][q] * r[cur][q] : a[cur][q];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[cur][q], acc, 0, 0, 0);
                    if (bias_wave) bsum += av;
                }
            }
        };
        rd4(ya, xa, ra, std::integral_constant<int, 0>{}, a[0], b[0], r[0]);
        group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{}); group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
        group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
    };
    const int niter = (mend - mbeg + SK - 1) / SK;
    if (niter > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_pair(mbeg, 0, i);
        issue_arow(mbeg, 0);
    }
    for (int it = 0; it < niter; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // slab `it` has landed for every wave; stage (it + 1) & 1 is free (its reads were waited for)
        asm volatile("" ::: "memory");
        mma(it & 1, it + 1 < niter ? mbeg + (it + 1) * SK : -1, (it + 1) & 1);
    }
    // ---- epilogue (layouts of conv_wgrad_ring_kernel)
    bsum += __shfl_xor(bsum, 32);
    if (p.splits > 1) {
        float* wb = ws + (int64_t)zs * wgr_block_elems(p.N, p.taps, p.Cin);
        const int c = c0 + wn0 + l31;
        if (c < p.Cin) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                wb[((int64_t)n * p.taps + j) * p.Cin + c] = acc[r];
            }
        }
        if (do_bias && wn0 == 0 && lh == 0) wb[(int64_t)p.N * p.taps * p.Cin + n0 + wm0 + l31] = bsum;
        if (cnt) wgr_last_arriver_reduce<64>(p, ws, cnt, reinterpret_cast<int*>(smem), (bz * ntiles + n0 / 64) * inner + within, bz, n0, j, c0, do_bias);
        return;
    }
    float* dW = p.dW + (int64_t)bz * p.sWb;
    const int c = c0 + wn0 + l31;
    if (c < p.Cin) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            old[r] = dW[(int64_t)n * p.ldw + (int64_t)j * p.Cin + c];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            dW[(int64_t)n * p.ldw + (int64_t)j * p.Cin + c] = old[r] + (p.oscale ? p.oscale[n] : 1.f) * acc[r];
        }
    }
    if (do_bias && wn0 == 0 && lh == 0) {
        const int n = n0 + wm0 + l31;
        p.db[(int64_t)bz * p.sDb + n] += (p.oscale ? p.oscale[n] : 1.f) * bsum;
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
    unsigned* cnt = (sp > 1 && tl <= WGR_MAX_TILES) ? wgr_counters(stream) : nullptr;
    if (T_ == 128) {
        osp_note_symbol("conv_wgrad_ring_kernel<128>");
        if (nst128 == 2) hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 2>), g, dim3(256), 2 * 2 * 64 * 128 * 2, stream, p, ws, cnt);
        else if (nst128 == 3) hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 3>), g, dim3(256), 3 * 2 * 64 * 128 * 2, stream, p, ws, cnt);
        else hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 4>), g, dim3(256), 4 * 2 * 64 * 128 * 2, stream, p, ws, cnt);
    } else {
        osp_note_symbol("conv_wgrad_ring_kernel<64>");
        if (nst64 == 2) hipLaunchKernelGGL((conv_wgrad_ring_kernel<64, 2>), g, dim3(256), 2 * 2 * 64 * 64 * 2, stream, p, ws, cnt);
        else if (nst64 == 3) hipLaunchKernelGGL((conv_wgrad_ring_kernel<64, 3>), g, dim3(256), 3 * 2 * 64 * 64 * 2, stream, p, ws, cnt);
        else hipLaunchKernelGGL((conv_wgrad_ring_kernel<64, 4>), g, dim3(256), 4 * 2 * 64 * 64 * 2, stream, p, ws, cnt);
    }
    if (sp > 1 && !cnt) {
        const int64_t E = N * taps * Cin, K = taps * Cin, tot = p.db ? E + N : E;
        hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3((unsigned)cdiv(tot, 1024), (unsigned)batch), dim3(256), 0, stream, ws, (int)sp,
                           (long long)blk, (long long)E, (int)K, make_fastdiv((unsigned)(K / 4)), p.oscale, p.dW, (long long)p.ldw, p.db,
                           (int)N, (long long)p.sWb, (long long)p.sDb);
    }
    return 1;
}

extern "C" int osp_conv_wgrad_f32(const float* dY, int64_t ldy, const float* X, int64_t ldx, int64_t M, int64_t T, int64_t N, int64_t Cin,
                                  int64_t taps, int64_t pad, const float* arow, const float* oscale, float* dW, int64_t ldw, float* db,
                                  int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb, hipStream_t stream);     // gemm.hip

// Exact-f32 weight gradient with a caller-supplied split workspace (see osp_conv_wgrad_bf16_ws): the ring kernel above where its
// conditions hold -- no atomics, bit-reproducible -- and osp_conv_wgrad_f32 (f32 atomics) otherwise.
//   dW[n, j, c] += oscale[n] * sum_m arow[m] * dY[m, n] * X[m + j - pad, c],  db[n] += oscale[n] * sum_m arow[m] * dY[m, n]
// Reference op: autograd of nn.Conv1d / nn.Linear (generator/modules/convnext.py:39-41, variance_predictor.py, alignments.py:55-64).
static int conv_wgrad_f32_ws_impl(int split, const float* dY, int64_t ldy, const float* X, int64_t ldx, int64_t M, int64_t T, int64_t N,
                                     int64_t Cin, int64_t taps, int64_t pad, const float* arow, const float* oscale, float* dW,
                                     int64_t ldw, float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                     float* ws, int64_t ws_bytes, hipStream_t stream) {
    OSP_CHECK_ARG(dY && X && dW, "null operand");
    OSP_CHECK_ARG(batch > 0, "batch");
    OSP_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && taps > 0 && T > 0 && M % T == 0, "bad shape");
    static int on = -1, tgt = 0;
    if (on < 0) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_f32_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 64 * 64 + 64) * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_f32_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 64 * 64 + 64) * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_f32_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 64 * 64 + 64) * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ring_f32_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 64 * 64 + 64) * 4);
        const char* e = getenv("OSP_WGRAD_RING_F32_TARGET"); tgt = e ? atoi(e) : 512;
        e = getenv("OSP_WGRAD_RING_F32"); on = (e && atoi(e) == 0) ? 0 : 1;
    }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int64_t yb = ((M - 1) * ldy + N) * 4, xb = ((M - 1) * ldx + Cin) * 4;
    bool take = on && N % 64 == 0 && Cin % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && al16(dY) && al16(X) && sYb % 4 == 0 && sXb % 4 == 0 &&
                ldw % 4 == 0 && sWb % 4 == 0 && al16(dW) && ldw == taps * Cin && yb > 0 && xb > 0 && yb < (int64_t)0x7fffff00 &&
                xb < (int64_t)0x7fffff00 && M * 4 < (int64_t)0x7fffff00;
    int64_t sp = 1, ch = 0;
    const int64_t tl = (N / 64) * taps * cdiv(Cin, 64) * batch;
    if (take) {
        const int64_t slabs = cdiv(M, 64);
        sp = tl >= tgt ? 1 : (tgt + tl / 2) / tl;
        if (sp > slabs / 4) sp = slabs / 4 > 0 ? slabs / 4 : 1;
        const int64_t blk = wgr_block_elems(N, taps, Cin);
        const int64_t cap = (ws && al16(ws)) ? ws_bytes / (blk * 4 * batch) : 1;
        if (sp > cap) sp = cap;
        if (sp < 1) sp = 1;
        ch = cdiv(cdiv(M, sp), 64) * 64;
        sp = cdiv(M, ch);
        // a single split with very few tiles would leave the chip idle: the tile-per-tap kernel (atomics) then has more parallelism
        if (sp == 1 && tl < 64 && M > 1024) take = false;
    }
    if (!take) return osp_conv_wgrad_f32(dY, ldy, X, ldx, M, T, N, Cin, taps, pad, arow, oscale, dW, ldw, db, batch, sYb, sXb, sWb, sDb, stream);
    WgradB p;
    p.dY = dY; p.y_bf16 = 0; p.ldy = ldy; p.X = X; p.x_bf16 = 0; p.ldx = ldx;
    p.M = (int)M; p.Trows = (int)T; p.Tin = (int)T; p.N = (int)N; p.Cin = (int)Cin; p.taps = (int)taps; p.pad = (int)pad; p.x_step = 1;
    p.Wrows = (int)T; p.Hin = 1; p.KW = (int)taps; p.x_step_h = 0; p.pad_h = 0;
    p.fd_trows = make_fastdiv((unsigned)T); p.fd_wrows = make_fastdiv((unsigned)T);
    p.arow = arow; p.oscale = oscale; p.dW = dW; p.ldw = ldw; p.db = db; p.chunk = (int)ch; p.splits = (int)sp;
    p.sYb = sYb; p.sXb = sXb; p.sWb = sWb; p.sDb = sDb; p.y_bytes = (unsigned)yb; p.x_bytes = (unsigned)xb;
    osp_note_symbol(split ? "conv_wgrad_ring_f32_split_kernel" : "conv_wgrad_ring_f32_kernel");
    osp_note_flops(2.0 * M * taps * (double)Cin * N * batch);
    osp_note_bytes(4.0 * batch * ((double)M * N + (double)M * Cin + (double)N * taps * Cin));
    const dim3 g((unsigned)(tl * sp));
    unsigned* cnt = (sp > 1 && tl <= WGR_MAX_TILES) ? wgr_counters(stream) : nullptr;
    constexpr int LDS = 2 * (2 * 64 * 64 + 64) * 4;
    if (split) {
        if (arow) hipLaunchKernelGGL((conv_wgrad_ring_f32_kernel<true, true>), g, dim3(256), LDS, stream, p, ws, cnt);
        else hipLaunchKernelGGL((conv_wgrad_ring_f32_kernel<false, true>), g, dim3(256), LDS, stream, p, ws, cnt);
    } else if (arow) hipLaunchKernelGGL((conv_wgrad_ring_f32_kernel<true>), g, dim3(256), LDS, stream, p, ws, cnt);
    else hipLaunchKernelGGL((conv_wgrad_ring_f32_kernel<false>), g, dim3(256), LDS, stream, p, ws, cnt);
    if (sp > 1 && !cnt) {
        const int64_t blk = wgr_block_elems(N, taps, Cin);
        const int64_t E = N * taps * Cin, K = taps * Cin, tot = db ? E + N : E;
        hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3((unsigned)cdiv(tot, 1024), (unsigned)batch), dim3(256), 0, stream, ws, (int)sp,
                           (long long)blk, (long long)E, (int)K, make_fastdiv((unsigned)(K / 4)), oscale, dW, (long long)ldw, db,
                           (int)N, (long long)sWb, (long long)sDb);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_conv_wgrad_f32_ws(const float* dY, int64_t ldy, const float* X, int64_t ldx, int64_t M, int64_t T, int64_t N,
                                     int64_t Cin, int64_t taps, int64_t pad, const float* arow, const float* oscale, float* dW,
                                     int64_t ldw, float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                     float* ws, int64_t ws_bytes, hipStream_t stream) {
    return conv_wgrad_f32_ws_impl(0, dY, ldy, X, ldx, M, T, N, Cin, taps, pad, arow, oscale, dW, ldw, db, batch, sYb, sXb, sWb, sDb, ws, ws_bytes, stream);
}

// The same weight gradient with SPLIT-bf16 products (kernel comment above): f32 operands in HBM, (hi, lo) bf16 pairs in registers, three
// bf16 MFMAs per 16-frame block, f32 accumulation and the same split workspace / ordered reduction -- bit-reproducible.  For the
// "mixed" parity mode's generator (precision.f32_split); problems the ring kernel declines run as osp_conv_wgrad_f32 (exact).
extern "C" int osp_conv_wgrad_f32_split_ws(const float* dY, int64_t ldy, const float* X, int64_t ldx, int64_t M, int64_t T, int64_t N,
                                     int64_t Cin, int64_t taps, int64_t pad, const float* arow, const float* oscale, float* dW,
                                     int64_t ldw, float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                     float* ws, int64_t ws_bytes, hipStream_t stream) {
    return conv_wgrad_f32_ws_impl(1, dY, ldy, X, ldx, M, T, N, Cin, taps, pad, arow, oscale, dW, ldw, db, batch, sYb, sXb, sWb, sDb, ws, ws_bytes, stream);
}
