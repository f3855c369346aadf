// 8-wave 256 (N) x 256 (Cin) transposed-read weight gradient, PHASED form (round 6): the tile, the LDS image ([y0, y1, x0, x1] sub-slabs
// of 64 frames x 128 channels, 4-slot XOR swizzle), the ds_read_b64_tr_b16 fragments, the work order and the epilogue of
// conv_wgrad_bf16_tr8_kernel (wgrad_bf16.hip), with the main loop of gemm_bf16_w8q.hip:
//
//  * a PHASE is one k-step: the 16 frames [16 ks, 16 ks + 16) of a 64-frame slab -- six fragments (twelve ds_read_b64_tr_b16) and the
//    8 MFMAs of the wave's 128 x 64 tile (every accumulator once: no dependent MFMAs inside a phase);
//  * the two wave groups (waves 0-3 / 4-7, one wave of each per SIMD) run half a phase apart:  [reads + requests + counted vmcnt]
//    s_barrier  [8 MFMAs]  s_barrier -- one group's matrix work beside the other's LDS / DMA work;
//  * the LDS-DMA UNIT is what a phase reads: rows [16 ks, 16 ks + 16) of the four sub-slabs = 16 KB = two 4-row pieces per wave
//    (wave w: sub-slab w & 1 of dY and of X, rows 4 (w >> 1) .. + 3).  Phase g reads unit g (slot g mod 8 of the two slab buffers),
//    requests unit g + 6 into the slot phase g - 2 read, and waits (vmcnt(10)) for unit g + 1: five phases between request and wait;
//  * the request addresses ADVANCE (16 frames per unit: one add for dY; for X one add plus the utterance wrap) instead of being rebuilt
//    from the frame index with two divisions per request (~40 VALU instructions in the lock-step kernel).  That form needs the 1-D
//    row map (Hin = 1: the DiscriminatorP layers this kernel serves) and 31-bit byte offsets; the launcher declines anything else.
// Measured (profiles/r06_tr8q.txt): 185 -> 173 us on the 1024 <- 1024 layer, 121 -> 115 us on 1024 <- 512 -- 5 %, far less than the same
// change bought the forward kernel.  A build with the epilogue or the loop compiled out says why: the loop alone is 146 / 66 us, the
// EPILOGUE alone 42 us on both layers -- 15.7 M agent-scope f32 atomics (3 / 6 frame splits of a 5.2 M / 2.6 M-element gradient), a
// third of the 1024 <- 512 launch, and on that layer it does not overlap the loop at all.
// Same frames in the same order into the same accumulators: results equal the lock-step kernel's bit for bit (tests/test_gpu_wgrad_tr8q.py).
#include "wgrad_common.h"

typedef short s16x4q __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(1024))) unsigned short wg8q_smem[];

__global__ __launch_bounds__(512) void conv_wgrad_bf16_tr8q_kernel(WgradB p) {
    constexpr int SK = 64, T = 128, SUB = SK * T;            // one sub-slab: 64 frames x 128 channels (16 KB)
    constexpr int S = 16;                                    // 16-byte slots per row
    constexpr unsigned OOB = 0x80000000u;
    unsigned short* smem = wg8q_smem;                        // [2 buffers][y0, y1, x0, x1][SUB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-aware work order: as conv_wgrad_bf16_tr8_kernel
    const int ctiles = p.Cin / 256, ntiles = p.N / 256, inner = p.taps * ctiles;
    const int total = gridDim.x, lin = blockIdx.x, xcd = lin & 7, local = lin >> 3;
    const int per = total >> 3, rem = total & 7;
    const int pid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + local;
    const int grp = pid / inner, within = pid - grp * inner;              // grp = (bz * splits + sp) * ntiles + nt
    const int j = within / ctiles, c0 = (within - j * ctiles) * 256;
    const int zs = grp / ntiles, n0 = (grp - zs * ntiles) * 256;
    const int bz = zs / p.splits, sp = zs - bz * p.splits;
    const unsigned short* dY = reinterpret_cast<const unsigned short*>(p.dY) + (int64_t)bz * p.sYb;
    const unsigned short* X = reinterpret_cast<const unsigned short*>(p.X) + (int64_t)bz * p.sXb;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const int kwp = j - p.pad;                               // (1-D: the tap index is the frame shift)
    const bool do_bias_wg = (p.db != nullptr) && (within == 0);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    float bsum = 0.f;

    // ---- request state of this lane: piece (sub-slab wave & 1, rows 4 (wave >> 1) + srow) of the unit to request next
    const int srow = lane / S, lslot = (lane % S) ^ (4 * (srow & 3));
    const int sub = wave & 1, prow = 4 * (wave >> 1);
    const int ldy32 = (int)p.ldy, ldx32 = (int)p.ldx, Trows = p.Trows, Tin = p.Tin, xs_ = p.x_step;
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dY), 0, (int)p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, (int)p.x_bytes, 0x00020000);
    int m_req = mbeg + prow + srow, t_req;
    unsigned yo, xo;
    {
        const int u = fd_div(m_req, p.fd_trows);
        t_req = m_req - u * Trows;
        yo = (unsigned)(m_req * ldy32 + n0 + sub * T + lslot * 8) * 2u;
        xo = (unsigned)((u * Tin + t_req * xs_ + kwp) * ldx32 + c0 + sub * T + lslot * 8) * 2u;
    }
    const unsigned y_adv = (unsigned)(16 * ldy32) * 2u, x_adv = (unsigned)(16 * xs_ * ldx32) * 2u;
    const unsigned x_wrap = (unsigned)((Tin - Trows * xs_) * ldx32) * 2u;
    // requests the current unit into `slot` (0 .. 7 = buffer * 4 + k-step), then advances the state by 16 frames
    auto request = [&](int slot) {
        unsigned short* base = smem + (slot >> 2) * (4 * SUB) + sub * SUB + (16 * (slot & 3) + prow) * T;     // wave-uniform
        const bool mv = m_req < mend;
        const bool xv = mv && (unsigned)(t_req * xs_ + kwp) < (unsigned)Tin;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (__attribute__((address_space(3))) void*)base, 16, mv ? yo : OOB, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(base + 2 * SUB), 16, xv ? xo : OOB, 0, 0, 0);
        m_req += 16; yo += y_adv; t_req += 16; xo += x_adv;
        while (t_req >= Trows) { t_req -= Trows; xo += x_wrap; }
    };

    // ---- fragment addresses (byte addresses in LDS, buffer 0, k-step 0): the XOR swizzle makes them non-additive in the 32-channel block
    const int r16 = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    auto frag_addr = [&](const unsigned short* base, int col0) -> unsigned {
        const int col = col0 + 16 * g16 + 4 * (r16 & 3);
        const int pslot = (col >> 3) ^ (4 * ((r16 >> 2) & 3));
        const unsigned short* a0 = base + (8 * kg + (r16 >> 2)) * T + pslot * 8 + (col & 7);
        return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)a0;
    };
    unsigned a_addr[4], b_addr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_addr[i] = frag_addr(smem + wm * SUB, 32 * i);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) b_addr[jj] = frag_addr(smem + (2 + (wn >> 1)) * SUB, (wn & 1) * 64 + 32 * jj);

    const int niter = (mend - mbeg + SK - 1) / SK, G = 4 * niter;      // phases
    // one phase; SLOT = g mod 8 (compile time): buffer SLOT >> 2 (64 KB apart: beyond the 16-bit offset field, added here), k-step SLOT & 3
    auto phase = [&](auto slotc, int g) {
        constexpr int SLOT = decltype(slotc)::value, KOFF = (SLOT & 3) * 16 * T * 2;
        const unsigned boff = (SLOT >> 2) * (4 * SUB * 2);
        s16x4q lo[6], hi[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[i]) : "v"(a_addr[i] + boff), "n"(KOFF) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[i]) : "v"(a_addr[i] + boff), "n"(KOFF + 4 * T * 2) : "memory");
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[4 + jj]) : "v"(b_addr[jj] + boff), "n"(KOFF) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[4 + jj]) : "v"(b_addr[jj] + boff), "n"(KOFF + 4 * T * 2) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g + 6 < G) request((SLOT + 6) & 7);
        {   // the unit phase g + 1 reads must have landed (this wave's pieces of it); `left` younger units stay in flight
            const int left = G - 1 - (g + 1);
            if (left >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (left == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (left == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (left == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]),
                       "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]) : : "memory");
        bf16x8 a[4], b[2];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            union { struct { s16x4q l, h; } s; bf16x8 v; } u;
            u.s.l = lo[i]; u.s.h = hi[i];
            if (i < 4) a[i] = u.v; else b[i - 4] = u.v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
        if (do_bias_wg) {                                     // workgroup-uniform: the wave with wn == i sums the 32-channel block i of its A fragments
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (wn == i) {
                    float sacc = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) sacc += (float)a[i][q];
                    bsum += sacc;
                }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
    };
    if (niter > 0) {
        // prologue: units 0 .. 5 (as many as exist), wait for unit 0
        const int npre = G < 6 ? G : 6;
        for (int q = 0; q < npre; ++q) request(q);
        {
            const int left = npre - 1;                        // younger than unit 0
            if (left >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (left == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // (G is a multiple of 4: 4 or >= 8 units)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind
        __builtin_amdgcn_sched_barrier(0);
        int it = 0;
        for (; it + 1 < niter; it += 2) {
            const int g = 4 * it;
            phase(std::integral_constant<int, 0>{}, g);     phase(std::integral_constant<int, 1>{}, g + 1);
            phase(std::integral_constant<int, 2>{}, g + 2); phase(std::integral_constant<int, 3>{}, g + 3);
            phase(std::integral_constant<int, 4>{}, g + 4); phase(std::integral_constant<int, 5>{}, g + 5);
            phase(std::integral_constant<int, 6>{}, g + 6); phase(std::integral_constant<int, 7>{}, g + 7);
        }
        if (it < niter) {
            const int g = 4 * it;
            phase(std::integral_constant<int, 0>{}, g);     phase(std::integral_constant<int, 1>{}, g + 1);
            phase(std::integral_constant<int, 2>{}, g + 2); phase(std::integral_constant<int, 3>{}, g + 3);
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();          // group 0 catches the barrier count up
        __builtin_amdgcn_sched_barrier(0);
    }
    const int l31 = lane & 31, lh = lane >> 5;
    float* dW = p.dW + (int64_t)bz * p.sWb;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int c = c0 + wn * 64 + 32 * jj + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 128 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float* dst = dW + (int64_t)n * p.ldw + (int64_t)j * p.Cin + c;
                const float val = (p.oscale ? p.oscale[n] : 1.f) * acc[i][jj][r];
                if (p.splits == 1) *dst += val;
                else atomicAdd(dst, val);
            }
        }
    if (do_bias_wg) {
        const float tot = bsum + __shfl_xor(bsum, 32, 64);                               // the two 8-frame halves of every k-step
        const int n = n0 + wm * 128 + 32 * wn + l31;
        if (lh == 0) atomicAdd(p.db + (int64_t)bz * p.sDb + n, (p.oscale ? p.oscale[n] : 1.f) * tot);
    }
}

// 1: launched, 0: declined (the caller launches conv_wgrad_bf16_tr8_kernel)
int osp_launch_wgrad_tr8q(const WgradB& p, dim3 grid, hipStream_t stream) {
    if (!(p.Hin == 1 && p.Wrows == p.Trows && p.KW == p.taps && p.y_bytes > 0 && p.x_bytes > 0 && p.Trows > 0)) return 0;
    static int done = 0;
    if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_tr8q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        done = 1;
    }
    osp_note_symbol("conv_wgrad_bf16_tr8q_kernel");
    hipLaunchKernelGGL(conv_wgrad_bf16_tr8q_kernel, grid, dim3(512), 131072, stream, p);
    return 1;
}
