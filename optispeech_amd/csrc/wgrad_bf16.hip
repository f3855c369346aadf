// Weight gradients of the bf16 conv-GEMM family + weight packing / casting (split from gemm_bf16.hip to halve the build time
// of the largest translation unit).  See gemm_bf16_common.h for the shared tile helpers.
#include "gemm_bf16_common.h"

// conv padding rows of the LDS-DMA staging read a zero page (one per translation unit)
__device__ __attribute__((aligned(256))) unsigned osp_zero_page_w[64];
#define osp_zero_page osp_zero_page_w

// ------------------------------------------------------------------------------------------------ wgrad
#include "wgrad_common.h"

__device__ __forceinline__ float4 ld4_any(const void* p, int is_bf16, int64_t off, int lim, bool vec) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lim >= 4 && vec) {
        if (is_bf16) {
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + off);
            x = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16),
                            __uint_as_float(h.y & 0xffff0000u));
        } else x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + off);
    } else {
        if (lim > 0) x.x = ld_elem(p, is_bf16, off);
        if (lim > 1) x.y = ld_elem(p, is_bf16, off + 1);
        if (lim > 2) x.z = ld_elem(p, is_bf16, off + 2);
        if (lim > 3) x.w = ld_elem(p, is_bf16, off + 3);
    }
    return x;
}

template <bool FAST>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(WgradB p) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (TBM + TBN) * LDK];
    unsigned short* As = smem;
    unsigned short* Bs = smem + 2 * TBM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int ctiles = (p.Cin + TBN - 1) / TBN;
    const int j = blockIdx.y / ctiles, c0 = (blockIdx.y - j * ctiles) * TBN;
    const int n0 = blockIdx.x * TBM;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z - bz * p.splits;
    const char* dY = reinterpret_cast<const char*>(p.dY) + (int64_t)bz * p.sYb * (p.y_bf16 ? 2 : 4);
    const char* X = reinterpret_cast<const char*>(p.X) + (int64_t)bz * p.sXb * (p.x_bf16 ? 2 : 4);
    const float* arow = p.arow ? p.arow + (int64_t)bz * p.M : nullptr;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const bool y_vec = (p.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dY) & 15) == 0);
    const bool x_vec = (p.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const int kg = tid >> 5, c4 = tid & 31;                   // k-group (8 frames) x 4 columns

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_bias = (p.db != nullptr) && (blockIdx.y == 0);
    const int blk_kh = j / p.KW, blk_kw = j - blk_kh * p.KW;

    uint4 ra[4], rb[4];
    auto gload = [&](int mk) {
        float4 ya[8], xb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mk + kg * 8 + q;
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f), x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < mend) {
                const int n = n0 + 4 * c4;
                if (n < p.N) {
                    y = ld4_any(dY, p.y_bf16, (int64_t)m * p.ldy + n, p.N - n, y_vec);
                    if (arow) { const float s = arow[m]; y.x *= s; y.y *= s; y.z *= s; y.w *= s; }
                }
                const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
                const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
                const int c = c0 + 4 * c4;
                if (tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin && c < p.Cin)
                    x = ld4_any(X, p.x_bf16, (((int64_t)u * p.Hin + hh) * p.Tin + tt) * p.ldx + c, p.Cin - c, x_vec);
            }
            ya[q] = y; xb[q] = x;
            if (do_bias) { bsum.x += y.x; bsum.y += y.y; bsum.z += y.z; bsum.w += y.w; }
        }
        ra[0] = make_uint4(pk2(ya[0].x, ya[1].x), pk2(ya[2].x, ya[3].x), pk2(ya[4].x, ya[5].x), pk2(ya[6].x, ya[7].x));
        ra[1] = make_uint4(pk2(ya[0].y, ya[1].y), pk2(ya[2].y, ya[3].y), pk2(ya[4].y, ya[5].y), pk2(ya[6].y, ya[7].y));
        ra[2] = make_uint4(pk2(ya[0].z, ya[1].z), pk2(ya[2].z, ya[3].z), pk2(ya[4].z, ya[5].z), pk2(ya[6].z, ya[7].z));
        ra[3] = make_uint4(pk2(ya[0].w, ya[1].w), pk2(ya[2].w, ya[3].w), pk2(ya[4].w, ya[5].w), pk2(ya[6].w, ya[7].w));
        rb[0] = make_uint4(pk2(xb[0].x, xb[1].x), pk2(xb[2].x, xb[3].x), pk2(xb[4].x, xb[5].x), pk2(xb[6].x, xb[7].x));
        rb[1] = make_uint4(pk2(xb[0].y, xb[1].y), pk2(xb[2].y, xb[3].y), pk2(xb[4].y, xb[5].y), pk2(xb[6].y, xb[7].y));
        rb[2] = make_uint4(pk2(xb[0].z, xb[1].z), pk2(xb[2].z, xb[3].z), pk2(xb[4].z, xb[5].z), pk2(xb[6].z, xb[7].z));
        rb[3] = make_uint4(pk2(xb[0].w, xb[1].w), pk2(xb[2].w, xb[3].w), pk2(xb[4].w, xb[5].w), pk2(xb[6].w, xb[7].w));
    };
    auto sstore = [&](int buf) {
        unsigned short* as = As + buf * TBM * LDK;
        unsigned short* bs = Bs + buf * TBN * LDK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<uint4*>(as + (4 * c4 + q) * LDK + kg * 8) = ra[q];
            *reinterpret_cast<uint4*>(bs + (4 * c4 + q) * LDK + kg * 8) = rb[q];
        }
    };
    // ---- fast path: both operands bf16 with 16-byte rows.  Threads 0-127 stage the dY tile, 128-255 the X tile:
    // 8 frames x 8 channels per thread (eight 16-byte loads), 8x8 bf16 transpose in registers, eight ds_write_b128.
    constexpr bool fast = FAST;
    // lane -> (k-group, column-group): k-group fastest, so the 8 lanes of a ds_write_b128 group fill 128 contiguous
    // bytes of ONE LDS row (column-group fastest put all 8 lanes on the same banks: 8-way conflict)
    const int half = tid >> 7, ht = tid & 127, fkg = ht & 7, c8 = ht >> 3;
    uint4 r8[8];
    float bs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto gload_fast = [&](int mk) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mk + fkg * 8 + q;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < mend) {
                if (half == 0) {
                    const int n = n0 + 8 * c8;
                    if (n < p.N) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(dY) + (int64_t)m * p.ldy + n);
                } else {
                    const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
                    const int c = c0 + 8 * c8;
                    const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
                    if (tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin && c < p.Cin)
                        v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(X) + (((int64_t)u * p.Hin + hh) * p.Tin + tt) * p.ldx + c);
                }
            }
            r8[q] = v;
        }
        if (do_bias && half == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                bs8[0] += __uint_as_float(r8[q].x << 16); bs8[1] += __uint_as_float(r8[q].x & 0xffff0000u);
                bs8[2] += __uint_as_float(r8[q].y << 16); bs8[3] += __uint_as_float(r8[q].y & 0xffff0000u);
                bs8[4] += __uint_as_float(r8[q].z << 16); bs8[5] += __uint_as_float(r8[q].z & 0xffff0000u);
                bs8[6] += __uint_as_float(r8[q].w << 16); bs8[7] += __uint_as_float(r8[q].w & 0xffff0000u);
            }
        }
    };
    auto sstore_fast = [&](int buf) {
        unsigned short* dst = (half == 0 ? As + buf * TBM * LDK : Bs + buf * TBN * LDK) + (8 * c8) * LDK + fkg * 8;
        const unsigned* w = reinterpret_cast<const unsigned*>(r8);      // w[q*4 + d]: frame q, channel pair d
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint4 lo, hi;                                                // channels 2d and 2d+1, frames 0..7
            lo.x = (w[0 * 4 + d] & 0xffffu) | (w[1 * 4 + d] << 16);  hi.x = (w[0 * 4 + d] >> 16) | (w[1 * 4 + d] & 0xffff0000u);
            lo.y = (w[2 * 4 + d] & 0xffffu) | (w[3 * 4 + d] << 16);  hi.y = (w[2 * 4 + d] >> 16) | (w[3 * 4 + d] & 0xffff0000u);
            lo.z = (w[4 * 4 + d] & 0xffffu) | (w[5 * 4 + d] << 16);  hi.z = (w[4 * 4 + d] >> 16) | (w[5 * 4 + d] & 0xffff0000u);
            lo.w = (w[6 * 4 + d] & 0xffffu) | (w[7 * 4 + d] << 16);  hi.w = (w[6 * 4 + d] >> 16) | (w[7 * 4 + d] & 0xffff0000u);
            *reinterpret_cast<uint4*>(dst + (2 * d) * LDK) = lo;
            *reinterpret_cast<uint4*>(dst + (2 * d + 1) * LDK) = hi;
        }
    };
    const int niter = (mend - mbeg + TBK - 1) / TBK;
    if (niter > 0) {
        if constexpr (fast) { gload_fast(mbeg); sstore_fast(0); } else { gload(mbeg); sstore(0); }
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const int buf = it & 1;
            if (it + 1 < niter) { if constexpr (fast) gload_fast(mbeg + (it + 1) * TBK); else gload(mbeg + (it + 1) * TBK); }
            mma_tile_bf16<2, 2>(As + buf * TBM * LDK, Bs + buf * TBN * LDK, wm0, wn0, lane, acc);
            if (it + 1 < niter) { if constexpr (fast) sstore_fast(buf ^ 1); else sstore(buf ^ 1); }
            __syncthreads();
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
    float* dW = p.dW + (int64_t)bz * p.sWb;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int c = c0 + wn0 + 32 * jj + l31;
            if (c >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n >= p.N) continue;
                float* dst = dW + (int64_t)n * p.ldw + (int64_t)j * p.Cin + c;
                const float val = (p.oscale ? p.oscale[n] : 1.f) * acc[i][jj][r];
                if (p.splits == 1) *dst += val;            // this block owns the tile: no atomics
                else atomicAdd(dst, val);
            }
        }
    if (do_bias) {
        // reduce the per-thread column sums over the 8 k-groups (threads with equal columns) through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [8][128]
        if constexpr (fast) {
            if (half == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[fkg * 128 + 8 * c8 + e] = bs8[e];
            }
        } else {
            *reinterpret_cast<float4*>(red + kg * 128 + 4 * c4) = bsum;
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q * 128 + tid];
            atomicAdd(p.db + (int64_t)bz * p.sDb + n0 + tid, (p.oscale ? p.oscale[n0 + tid] : 1.f) * s);
        }
    }
}

// ---- transposed-read variant (both operands bf16 with 16-byte rows, N % 128 == 0, Cin % 128 == 0, no row scale).
// The reduction index (frames) is the SLOW index of both dY (M, N) and X (rows, Cin); the kernel above transposes 8x8
// blocks in registers while staging, which makes it VALU-bound (~500 VALU instructions per k-slab and wave, PMC).  Here
// the slabs are copied as they lie in HBM with global_load_lds (64 frames x 128 channels per operand, 256-byte rows), and
// the MFMA fragments (8 consecutive frames of one channel per lane) are produced by ds_read_b64_tr_b16, which transposes
// a 4 (frames) x 16 (channels) block per 16-lane group on the way out of LDS.
// Bank mapping: a 256-byte row covers all 64 banks, so the 4 frame rows of one transposed read would collide 4-way; the
// 16-byte slot index is XOR-ed with 4 * (row & 3) (applied to the global source address when staging and to the LDS
// address when reading), which puts the 8 (row, 16-channel group) segments of a 32-lane pass on 8 distinct bank ranges.
// T = 128: 4 waves of 64x64, 256-byte rows, slot ^= 4 * (row & 3).
// T = 64 (the 64-channel DiscriminatorR layers): 4 waves of 32x32, 128-byte rows (two rows per 64 banks), the 4 frame
// rows of a transposed read alternate bank halves and slot ^= 4 * ((row >> 1) & 1) separates the pairs.
typedef short s16x4 __attribute__((ext_vector_type(4)));
// F32 = true: f32 operands in HBM (the generator's activations); the slab goes global -> registers -> bf16 -> LDS (same
// LDS image as the DMA path, so the transposed reads are shared), with the optional per-frame scale `arow` applied to dY.
template <int T, bool F32 = false, bool BUF = false>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_tr_kernel(WgradB p) {
    constexpr int SK = 64;                                   // frames per slab
    constexpr int S = T / 8, RPI = 64 / S, NI = SK / RPI / 4, TI = T / 64;   // slots/row, rows/instruction, instr/wave/operand
    __shared__ __attribute__((aligned(1024))) unsigned short smem[2 * 2 * SK * T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (T / 2), wn0 = (wave & 1) * (T / 2);
    const int ctiles = (p.Cin + T - 1) / T;                 // (Cin = 32 with T = 64, BUF form only: the upper half of the tile reads zeros)
    const int j = blockIdx.y / ctiles, c0 = (blockIdx.y - j * ctiles) * T;
    const int n0 = blockIdx.x * T;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z - bz * p.splits;
    const unsigned short* dY = reinterpret_cast<const unsigned short*>(p.dY) + (int64_t)bz * p.sYb;
    const unsigned short* X = reinterpret_cast<const unsigned short*>(p.X) + (int64_t)bz * p.sXb;
    const int mbeg = sp * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const int blk_kh = j / p.KW, blk_kw = j - blk_kh * p.KW;
    const bool do_bias = (p.db != nullptr) && (blockIdx.y == 0);
    const bool bias_wave = __builtin_amdgcn_readfirstlane((int)(do_bias && wn0 == 0)) != 0;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(osp_zero_page);
    const int64_t ldy = p.ldy, ldx = p.ldx;
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
    // staging: wave w, instruction i covers slab rows RPI * (NI * w + i) + (lane / S); physical 16-byte slot = lane % S
    const int srow = lane / S, lslot = (lane % S) ^ swz(srow);
    // one (dY row, X row) pair of loads; `i` = instruction index 0..NI-1
    const int ldy32 = (int)ldy, ldx32 = (int)ldx;
    const unsigned ycol2 = (unsigned)(n0 + lslot * 8) * 2u, xcol2 = (unsigned)(c0 + lslot * 8) * 2u;
    __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dY), 0, BUF ? (int)p.y_bytes : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, BUF ? (int)p.x_bytes : 0, 0x00020000);
    const unsigned short* dY_lane = dY + n0 + lslot * 8;
    const unsigned short* X_lane = X + c0 + lslot * 8;
    auto issue_pair = [&](int mk, int buf, int i) {
        unsigned short* ys = smem + buf * (2 * SK * T);
        unsigned short* xs = ys + SK * T;
        const int row0 = RPI * (NI * wave + i), m = mk + row0 + srow;
        const bool mv = m < mend;
        if constexpr (BUF) {
            // buffer-resource form: 32-bit byte offsets against an SGPR descriptor, out-of-range offsets read as zero (no
            // zero page, no 64-bit pointer arithmetic or pointer selects: the address VALU work was what bound this loop)
            const unsigned yo = mv ? (unsigned)(m * ldy32) * 2u + ycol2 : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (__attribute__((address_space(3))) void*)(ys + row0 * T), 16, yo, 0, 0, 0);
            const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
            const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
            const bool xv = mv && (unsigned)tt < (unsigned)p.Tin && (unsigned)hh < (unsigned)p.Hin && (c0 + lslot * 8 < p.Cin);
            const unsigned xo = xv ? (unsigned)(((u * p.Hin + hh) * p.Tin + tt) * ldx32) * 2u + xcol2 : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(xs + row0 * T), 16, xo, 0, 0, 0);
            return;
        }
        // row indices stay 32-bit (a tensor has < 2^31 rows); one 64-bit multiply-add per pointer, selects instead of branches
        const unsigned short* ysel = mv ? dY_lane : zero;
        const unsigned short* src = ysel + (int64_t)(mv ? m : 0) * ldy;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ys + row0 * T), 16, 0, 0);
        const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
        const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
        const bool xv = mv && tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin;
        const int xrow = xv ? (u * p.Hin + hh) * p.Tin + tt : 0;
        const unsigned short* xsel = xv ? X_lane : zero;
        const unsigned short* xsrc = xsel + (int64_t)xrow * ldx;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xsrc,
                                         (__attribute__((address_space(3))) void*)(xs + row0 * T), 16, 0, 0);
    };
    auto issue = [&](int mk, int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_pair(mk, buf, i);
    };
    // f32 operands: the same (row, slot) assignment, through registers
    const float* dYf = reinterpret_cast<const float*>(p.dY) + (int64_t)bz * p.sYb;
    const float* Xf = reinterpret_cast<const float*>(p.X) + (int64_t)bz * p.sXb;
    const float* arow = p.arow ? p.arow + (int64_t)bz * p.M : nullptr;
    float4 ry[NI][2], rx[NI][2];
    auto gload_f32 = [&](int mk) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row0 = RPI * (NI * wave + i), m = mk + row0 + srow;
            const bool mv = m < mend;
            const int mc = mv ? m : mbeg;                                         // index select: loads stay unconditional
            const float4* ys4 = reinterpret_cast<const float4*>(dYf + (int64_t)mc * ldy + n0 + lslot * 8);
            const float sc = mv ? (arow ? arow[mc] : 1.f) : 0.f;
            float4 a = ys4[0], b = ys4[1];
            ry[i][0] = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
            ry[i][1] = make_float4(b.x * sc, b.y * sc, b.z * sc, b.w * sc);
            const int u = fd_div(mc, p.fd_trows), t = mc - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
            const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
            const bool xv = mv && tt >= 0 && tt < p.Tin && hh >= 0 && hh < p.Hin;
            const int64_t xr = xv ? (((int64_t)u * p.Hin + hh) * p.Tin + tt) : 0;
            const float4* xs4 = reinterpret_cast<const float4*>(Xf + xr * ldx + c0 + lslot * 8);
            const float xsel = xv ? 1.f : 0.f;
            a = xs4[0]; b = xs4[1];
            rx[i][0] = make_float4(a.x * xsel, a.y * xsel, a.z * xsel, a.w * xsel);
            rx[i][1] = make_float4(b.x * xsel, b.y * xsel, b.z * xsel, b.w * xsel);
        }
    };
    auto sstore_f32 = [&](int buf) {
        unsigned short* ys = smem + buf * (2 * SK * T);
        unsigned short* xs = ys + SK * T;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int off = (RPI * (NI * wave + i) + srow) * T + (lane % S) * 8;
            *reinterpret_cast<uint4*>(ys + off) = make_uint4(pk2(ry[i][0].x, ry[i][0].y), pk2(ry[i][0].z, ry[i][0].w),
                                                             pk2(ry[i][1].x, ry[i][1].y), pk2(ry[i][1].z, ry[i][1].w));
            *reinterpret_cast<uint4*>(xs + off) = make_uint4(pk2(rx[i][0].x, rx[i][0].y), pk2(rx[i][0].z, rx[i][0].w),
                                                             pk2(rx[i][1].x, rx[i][1].y), pk2(rx[i][1].z, rx[i][1].w));
        }
    };
    // fragment of operand tile `base` ([SK][T]) for the 32 channels starting at `col0`, k-step ks: 8 consecutive frames
    const int r16 = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    auto frag = [&](const unsigned short* base, int col0, int ks) -> bf16x8 {
        const int col = col0 + 16 * g16 + 4 * (r16 & 3);                          // first of this lane's 4 source channels
        const int pslot = (col >> 3) ^ swz(r16 >> 2);
        const unsigned short* a0 = base + (16 * ks + 8 * kg + (r16 >> 2)) * T + pslot * 8 + (col & 7);
        // Issued through inline asm: for the builtin the compiler cannot tell that the LDS-DMA loads in flight (next slab,
        // other buffer) do not alias these reads and put `s_waitcnt vmcnt(0)` in front of every k-step's reads, i.e. each
        // k-step waited for the global loads issued one k-step earlier (the slab pipeline was serialised: 350 TFLOP/s on
        // the 1024x1024x5 layer vs 680 for the forward kernel).  The price: the compiler does not count these reads in
        // lgkmcnt either, so mma() waits for them explicitly (frag_wait) before the MFMAs consume the registers.
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)a0;
        s16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * T * 2) : "memory");
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo; u.s.h = hi;
        return u.v;
    };
    auto frag_wait = [](bf16x8 (&a)[TI], bf16x8 (&b)[TI]) {
        if constexpr (TI == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]) : : "memory");
    };
    bf16x8 ones;
#pragma unroll
    for (int q = 0; q < 8; ++q) ones[q] = (__bf16)1.0f;
    // MFMA phase over slab `buf`; the next slab (frames from `mk_next`, < 0 = none) is staged one row pair per k-step
    auto mma = [&](int buf, int mk_next) {
        const unsigned short* ys = smem + buf * (2 * SK * T);
        const unsigned short* xs = ys + SK * T;
#pragma unroll
        for (int ks = 0; ks < SK / 16; ++ks) {
            bf16x8 a[TI], b[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = frag(ys, wm0 + 32 * i, ks);
#pragma unroll
            for (int jj = 0; jj < TI; ++jj) b[jj] = frag(xs, wn0 + 32 * jj, ks);
            if (mk_next >= 0 && ks < NI) issue_pair(mk_next, buf ^ 1, ks);
            frag_wait(a, b);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int jj = 0; jj < TI; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
            if (bias_wave) {                                                      // scalar condition: no exec masking around the MFMAs
#pragma unroll
                for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], ones, accb[i], 0, 0, 0);
            }
        }
    };
    const int niter = (mend - mbeg + SK - 1) / SK;
    if constexpr (F32) {
        if (niter > 0) {
            gload_f32(mbeg);
            sstore_f32(0);
            __syncthreads();
            for (int it = 0; it < niter; ++it) {
                const int buf = it & 1;
                if (it + 1 < niter) gload_f32(mbeg + (it + 1) * SK);
                mma(buf, -1);
                if (it + 1 < niter) sstore_f32(buf ^ 1);
                __syncthreads();
            }
        }
    } else if (niter > 0) {
        issue(mbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const int buf = it & 1;
            mma(buf, it + 1 < niter ? mbeg + (it + 1) * SK : -1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
    float* dW = p.dW + (int64_t)bz * p.sWb;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TI; ++jj) {
            const int c = c0 + wn0 + 32 * jj + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float* dst = dW + (int64_t)n * p.ldw + (int64_t)j * p.Cin + c;
                const float val = (p.oscale ? p.oscale[n] : 1.f) * acc[i][jj][r];
                if (c < p.Cin) {
                    if (p.splits == 1) *dst += val;        // this block owns the tile: no atomics
                    else atomicAdd(dst, val);
                }
            }
        }
    if (do_bias && wn0 == 0 && l31 == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                atomicAdd(p.db + (int64_t)bz * p.sDb + n, (p.oscale ? p.oscale[n] : 1.f) * accb[i][r]);
            }
    }
}

// ---- 8-wave transposed-read variant (round 2): 256 (N) x 256 (Cin) tiles, 512 threads = 2 (N) x 4 (Cin) waves of 128 x 64.
// Same LDS image as the T = 128 kernel above -- each operand slab is kept as TWO [64 frames][128 channels] sub-slabs, so the
// staging addresses, the XOR swizzle and the ds_read_b64_tr_b16 fragment reads are unchanged -- but a workgroup stages
// 2 x (256 + 256) x 64 x 2 B for 4x the flops (half the L2 / HBM bytes per flop), and a wave reads (128 + 64) x 32 B of fragments
// per 8 MFMAs instead of 128 x 32 B per 4.  128 KB of LDS: one workgroup per CU, two waves per SIMD.
// The bias gradient does not get accumulator registers here (4 more 32x32 accumulators would not fit beside the 128): the
// waves that own Cin block 0 add up their own A fragments on the VALU (8 frames of one channel per lane).
extern __shared__ __attribute__((aligned(1024))) unsigned short wg8_smem[];
template <bool BUF>
__global__ __launch_bounds__(512) void conv_wgrad_bf16_tr8_kernel(WgradB p) {
    constexpr int SK = 64, T = 128, SUB = SK * T;            // one sub-slab: 64 frames x 128 channels (16 KB)
    constexpr int S = 16, RPI = 4;                           // 16-byte slots per row, rows per wave instruction
    unsigned short* smem = wg8_smem;                         // [2 stages][y0, y1, x0, x1][SUB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-aware work order (1-D grid).  The (taps x Cin-tiles) workgroups of one (batch, frame split, N tile) read the SAME dY
    // slabs, and the taps of one Cin tile read X slabs shifted by one frame: with one workgroup per tap and tile, the operands
    // were fetched 20x (1.07 GB through the fabric for 53 MB of unique data on the 1024x1024x5 layer: the kernel was bound by
    // that, not by the MFMAs).  Workgroups are dispatched round-robin over the 8 XCDs; the linear id is folded so that every
    // XCD owns a contiguous range of work items ordered (split, N tile) -> (tap, Cin tile): the ~32 workgroups resident on an
    // XCD walk the same frames of the same dY panel at the same time and share it (and the X panels) through that XCD's L2.
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
    const int blk_kh = j / p.KW, blk_kw = j - blk_kh * p.KW;
    const bool do_bias = (p.db != nullptr) && (within == 0);
    const bool do_bias_wg = __builtin_amdgcn_readfirstlane((int)do_bias) != 0;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(osp_zero_page);
    const int64_t ldy = p.ldy, ldx = p.ldx;
    auto swz = [](int row) { return 4 * (row & 3); };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    float bsum = 0.f;
    // staging: per slab and operand 2 sub-slabs x 16 row blocks of 4 rows; wave w, pair i: sub-slab i & 1, row block 2 w + (i >> 1)
    const int srow = lane / S, lslot = (lane % S) ^ swz(srow);
    const int ldy32 = (int)ldy, ldx32 = (int)ldx;
    __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(dY), 0, BUF ? (int)p.y_bytes : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, BUF ? (int)p.x_bytes : 0, 0x00020000);
    auto issue_pair = [&](int mk, int buf, int i) {
        const int sub = i & 1, row0 = RPI * (2 * wave + (i >> 1)), m = mk + row0 + srow;
        unsigned short* ys = smem + buf * (4 * SUB) + sub * SUB;
        unsigned short* xs = smem + buf * (4 * SUB) + (2 + sub) * SUB;
        const int ycol = n0 + sub * T + lslot * 8, xcol = c0 + sub * T + lslot * 8;
        const bool mv = m < mend;
        const int u = fd_div(m, p.fd_trows), t = m - u * p.Trows, th = fd_div(t, p.fd_wrows), tw = t - th * p.Wrows;
        const int tt = tw * p.x_step + blk_kw - p.pad, hh = th * p.x_step_h + blk_kh - p.pad_h;
        const bool xv = mv && (unsigned)tt < (unsigned)p.Tin && (unsigned)hh < (unsigned)p.Hin;
        if constexpr (BUF) {
            const unsigned yo = mv ? (unsigned)(m * ldy32 + ycol) * 2u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (__attribute__((address_space(3))) void*)(ys + row0 * T), 16, yo, 0, 0, 0);
            const unsigned xo = xv ? (unsigned)(((u * p.Hin + hh) * p.Tin + tt) * ldx32 + xcol) * 2u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(xs + row0 * T), 16, xo, 0, 0, 0);
        } else {
            const unsigned short* src = mv ? dY + (int64_t)m * ldy + ycol : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ys + row0 * T), 16, 0, 0);
            const unsigned short* xsrc = xv ? X + (int64_t)((u * p.Hin + hh) * p.Tin + tt) * ldx + xcol : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xsrc,
                                             (__attribute__((address_space(3))) void*)(xs + row0 * T), 16, 0, 0);
        }
    };
    const int r16 = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    auto frag = [&](const unsigned short* base, int col0, int ks) -> bf16x8 {
        const int col = col0 + 16 * g16 + 4 * (r16 & 3);
        const int pslot = (col >> 3) ^ swz(r16 >> 2);
        const unsigned short* a0 = base + (16 * ks + 8 * kg + (r16 >> 2)) * T + pslot * 8 + (col & 7);
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)a0;
        s16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * T * 2) : "memory");
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo; u.s.h = hi;
        return u.v;
    };
    auto mma = [&](int buf, int mk_next) {
        const unsigned short* ys = smem + buf * (4 * SUB) + wm * SUB;
        const unsigned short* xs = smem + buf * (4 * SUB) + (2 + (wn >> 1)) * SUB;
        const int xc0 = (wn & 1) * 64;
        // (requesting the fragments of k-step s + 1 before the MFMAs of k-step s -- what helps the one-wave-per-SIMD ring kernels --
        // measured 184 vs 187 us here: two waves per SIMD already cover each other's LDS latency; not kept)
#pragma unroll
        for (int ks = 0; ks < SK / 16; ++ks) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag(ys, 32 * i, ks);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) b[jj] = frag(xs, xc0 + 32 * jj, ks);
            if (mk_next >= 0) issue_pair(mk_next, buf ^ 1, ks);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]) : : "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
            if (do_bias_wg) {                                 // workgroup-uniform; lane (l31, lh) holds channel l31, frames 8 lh .. 8 lh + 7
                // the four waves that share these A fragments (wn = 0..3) take ONE 32-channel block each: summing all four in the
                // wn = 0 waves (32 cvt + 32 add per k-step on 2 of 8 waves) made them the stragglers of every slab barrier --
                // 200 -> 185 us on the 1024 <- 1024 layer, 131 -> 121 us on 1024 <- 512 (round 5)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (wn == i) {                            // wave-uniform
                        float sacc = 0.f;
#pragma unroll
                        for (int q = 0; q < 8; ++q) sacc += (float)a[i][q];
                        bsum += sacc;
                    }
            }
        }
    };
    const int niter = (mend - mbeg + SK - 1) / SK;
    if (niter > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_pair(mbeg, 0, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const int buf = it & 1;
            mma(buf, it + 1 < niter ? mbeg + (it + 1) * SK : -1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
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

int osp_launch_wgrad_n1(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t ldx, int64_t M, int64_t Trows, int64_t Wrows, int64_t Hin,
                        int64_t Tin, int64_t Cin, int64_t taps, int64_t KW, int64_t pad, int64_t pad_h, int64_t x_step, int64_t x_step_h,
                        const float* arow, const float* oscale, float* dW, float* db, hipStream_t stream);      // wgrad_n1.hip

int osp_launch_wgrad_ring(WgradB& p, int64_t batch, float* ws, int64_t ws_bytes, hipStream_t stream);     // wgrad_ring.hip
int osp_launch_wgrad_tr8q(const WgradB& p, dim3 grid, hipStream_t stream);                                // wgrad_bf16_tr8q.hip (1: taken)
#ifndef OSP_WGRAD_W8Q_DEFAULT
#define OSP_WGRAD_W8Q_DEFAULT 1
#endif

static int conv_wgrad_bf16_impl(const int64_t* d2, const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                   int64_t M, int64_t Trows, int64_t Tin, int64_t N, int64_t Cin, int64_t taps, int64_t pad,
                                   int64_t x_step, const float* arow, const float* oscale, float* dW, int64_t ldw,
                                   float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                   float* ws, int64_t ws_bytes, hipStream_t stream) {
    OSP_CHECK_ARG(dY && X && dW, "null operand");
    OSP_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && taps > 0 && Trows > 0 && M % Trows == 0 && batch > 0, "bad shape");
    WgradB p;
    p.dY = dY; p.y_bf16 = (int)y_bf16; p.ldy = ldy; p.X = X; p.x_bf16 = (int)x_bf16; p.ldx = ldx;
    p.M = (int)M; p.Trows = (int)Trows; p.Tin = (int)Tin; p.N = (int)N; p.Cin = (int)Cin; p.taps = (int)taps;
    p.pad = (int)pad; p.x_step = (int)x_step; p.arow = arow; p.oscale = oscale; p.dW = dW; p.ldw = ldw; p.db = db;
    p.sYb = sYb; p.sXb = sXb; p.sWb = sWb; p.sDb = sDb;
    p.y_bytes = 0; p.x_bytes = 0;
    p.Wrows = (int)d2[0]; p.Hin = (int)d2[1]; p.KW = (int)d2[2]; p.x_step_h = (int)d2[3]; p.pad_h = (int)d2[4];
    p.fd_trows = make_fastdiv((unsigned)Trows); p.fd_wrows = make_fastdiv((unsigned)d2[0]);
    osp_note_flops(2.0 * M * taps * (double)Cin * N * batch);          // measurement aid (api.cpp)
    osp_note_bytes((double)batch * ((double)M * N * (y_bf16 ? 2.0 : 4.0) + (double)(M / Trows) * d2[1] * Tin * Cin * (x_bf16 ? 2.0 : 4.0) +
                                    (double)N * taps * Cin * 4.0));
    // one output channel (the discriminators' conv_post): a dY-weighted column sum, not a GEMM (wgrad_n1.hip)
    static int use_n1 = -1;
    if (use_n1 < 0) { const char* e = getenv("OSP_WGRAD_N1"); use_n1 = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_n1 && N == 1 && x_bf16 && batch == 1 && Cin % 64 == 0 && taps <= 9 && ldx % 8 == 0 &&
        (reinterpret_cast<uintptr_t>(X) & 15) == 0)
        return osp_launch_wgrad_n1(dY, y_bf16, ldy, X, ldx, M, Trows, d2[0], d2[1], Tin, Cin, taps, d2[2], pad, d2[4], x_step, d2[3], arow, oscale,
                                   dW, db, stream);
    // ring-pipelined, atomic-free kernels (wgrad_ring.hip) for every shape they take; the 8-wave 256 x 256 kernel below keeps the
    // widest DiscriminatorP layers
    static int w8first = -1;
    if (w8first < 0) { const char* e = getenv("OSP_WGRAD_W8"); w8first = (e && atoi(e) == 0) ? 0 : 1; }
    // (the 8-wave kernel only where its 256 x 256 tiles are many: N x Cin >= 512 x 512.  A 256 <- 256 layer at M = 25 600 -- the alignment
    // module's feature convs -- is ONE tile per tap there, i.e. 85 frame splits and 16.7 M atomics: 76 us, on the tail of the generator's backward)
    const bool wide8 = N % 256 == 0 && Cin % 256 == 0 && M >= 8192 && N * Cin >= 512 * 512;
    if (!(w8first && wide8 && batch == 1)) {
        const int took = osp_launch_wgrad_ring(p, batch, ws, ws_bytes, stream);
        if (took) { OSP_LAUNCH_CHECK(); return OSP_OK; }
    }
    const int64_t tiles = cdiv(N, TBM) * taps * cdiv(Cin, TBN) * batch;
    static int64_t tgt_gen = 0;
    if (!tgt_gen) { tgt_gen = 512; }
    int64_t splits = tiles >= 192 ? 1 : cdiv(tgt_gen, tiles);
    int64_t chunk = cdiv(cdiv(M, splits), TBK) * TBK;
    if (chunk < 2 * TBK) chunk = 2 * TBK;
    splits = cdiv(M, chunk);
    p.chunk = (int)chunk; p.splits = (int)splits;
    dim3 grid((unsigned)cdiv(N, TBM), (unsigned)(taps * cdiv(Cin, TBN)), (unsigned)(splits * batch));
    const bool fast = y_bf16 && x_bf16 && !arow && (ldy % 8 == 0) && (ldx % 8 == 0) && (N % 8 == 0) && (Cin % 8 == 0) &&
                      ((reinterpret_cast<uintptr_t>(dY) & 15) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                      (sYb % 8 == 0) && (sXb % 8 == 0);
    static int use_tr = -1;
    if (use_tr < 0) { const char* e = getenv("OSP_WGRAD_TR"); use_tr = (e && atoi(e) == 0) ? 0 : 1; }
    // 32-channel inputs (DiscriminatorP 32 -> 128): the 64-tile kernel with the upper half of every X row reading zeros through the
    // buffer descriptor -- half the MFMA work is padding, but the generic tile kernel below (no DMA, no double buffering) took
    // 84 us for the same layer
    const bool cin32 = Cin == 32 && N % 64 == 0 && ((int64_t)(M / Trows) * d2[1] * Tin * ldx) * 2 < (int64_t)0x7fffff00 &&
                       (M * ldy) * 2 < (int64_t)0x7fffff00;      // needs the descriptor's bounds check
    if (use_tr && fast && N % 64 == 0 && (Cin % 64 == 0 || cin32)) {
        // 128-tiles when both channel counts allow it, 64-tiles otherwise (DiscriminatorR).  The frames are split until the grid
        // has about `target` workgroups; partial sums meet in f32 atomics, and every split adds a tile's worth of them: round 3
        // lowered the targets from 1024 / 2048 (two to four rounds of resident workgroups) to 256 / 512 (ONE workgroup per CU and
        // four times fewer atomics): DiscriminatorR 64 -> 64 (3, 9) 95 -> 73 us, DiscriminatorP 128 -> 512 89 -> 60 us
        // (tools/mrd_bench.py, tools/mpd_bench.py with OSP_WGRAD_TARGET = 128 ... 2048)
        const int64_t T_ = (N % 128 == 0 && Cin % 128 == 0) ? 128 : 64;
        static int64_t tgt_env = -1;
        static int64_t tgt64_env = -1;
        if (tgt_env < 0) { tgt_env = 0; }
        if (tgt64_env < 0) { tgt64_env = 0; }
        const int64_t tl = (N / T_) * taps * cdiv(Cin, T_) * batch, target = T_ == 128 ? (tgt_env > 0 ? tgt_env : 256) : (tgt64_env > 0 ? tgt64_env : (tgt_env > 0 ? 2 * tgt_env : 512));
        int64_t sp = tl >= target / 2 - 64 ? 1 : (target + tl / 2) / tl;
        int64_t ch = cdiv(cdiv(M, sp), TBK) * TBK;
        if (ch < 4 * TBK) ch = 4 * TBK;
        sp = cdiv(M, ch);
        p.chunk = (int)ch; p.splits = (int)sp;
        const dim3 g((unsigned)(N / T_), (unsigned)(taps * cdiv(Cin, T_)), (unsigned)(sp * batch));
        // buffer-resource loads when one batch slice of each operand is addressable with 31-bit byte offsets
        const int64_t rows_x = (M / Trows) * (int64_t)p.Hin * Tin;
        const int64_t yb = ((M - 1) * ldy + N) * 2, xb = ((rows_x - 1) * ldx + Cin) * 2;
        static int use_buf = -1;
        if (use_buf < 0) { use_buf = 1; }
        const bool buf = use_buf && yb > 0 && xb > 0 && yb < (int64_t)0x7fffff00 && xb < (int64_t)0x7fffff00;
        p.y_bytes = buf ? (unsigned)yb : 0; p.x_bytes = buf ? (unsigned)xb : 0;
        // 8-wave 256 x 256 tiles for the wide layers (DiscriminatorP 512->1024 and 1024->1024): OSP_WGRAD_W8 = 0 switches them off
        static int w8 = -1;
        if (w8 < 0) {
            const char* e = getenv("OSP_WGRAD_W8"); w8 = (e && atoi(e) == 0) ? 0 : 1;
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_tr8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_tr8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        }
        if (w8 && wide8) {
            const int64_t tl8 = (N / 256) * taps * (Cin / 256) * batch;
            static int64_t tgt8 = -1;
            if (tgt8 < 0) { tgt8 = 256; }
            int64_t sp8 = tl8 >= tgt8 / 2 ? 1 : tgt8 / tl8;                              // workgroups over the launch (1 resident per CU)
            if (sp8 < 1) sp8 = 1;
            int64_t ch8 = cdiv(cdiv(M, sp8), TBK) * TBK;
            if (ch8 < 4 * TBK) ch8 = 4 * TBK;
            sp8 = cdiv(M, ch8);
            p.chunk = (int)ch8; p.splits = (int)sp8;
            const dim3 g8((unsigned)((N / 256) * taps * (Cin / 256) * sp8 * batch));
            // phased main loop (wgrad_bf16_tr8q.hip; declines 2-D row maps / operands without a buffer descriptor): OSP_WGRAD_W8Q = 0
            // keeps the lock-step kernel (read per call: tests and probes switch it in-process)
            { const char* e = getenv("OSP_WGRAD_W8Q"); const int q = e ? atoi(e) : OSP_WGRAD_W8Q_DEFAULT;
              if (q && buf && osp_launch_wgrad_tr8q(p, g8, stream)) { OSP_LAUNCH_CHECK(); return OSP_OK; } }
            osp_note_symbol("conv_wgrad_bf16_tr8_kernel");
            if (buf) hipLaunchKernelGGL((conv_wgrad_bf16_tr8_kernel<true>), g8, dim3(512), 131072, stream, p);
            else hipLaunchKernelGGL((conv_wgrad_bf16_tr8_kernel<false>), g8, dim3(512), 131072, stream, p);
        } else if (T_ == 128) {
            osp_note_symbol("conv_wgrad_bf16_tr_kernel<128>");
            if (buf) hipLaunchKernelGGL((conv_wgrad_bf16_tr_kernel<128, false, true>), g, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL(conv_wgrad_bf16_tr_kernel<128>, g, dim3(256), 0, stream, p);
        } else {
            osp_note_symbol("conv_wgrad_bf16_tr_kernel<64>");
            if (buf) hipLaunchKernelGGL((conv_wgrad_bf16_tr_kernel<64, false, true>), g, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL(conv_wgrad_bf16_tr_kernel<64>, g, dim3(256), 0, stream, p);
        }
    }
    else if (use_tr && !y_bf16 && !x_bf16 && N % 64 == 0 && Cin % 64 == 0 && (ldy % 4 == 0) && (ldx % 4 == 0) &&
             ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) == 0 && (sYb % 4 == 0) && (sXb % 4 == 0)) {
        // f32 operands (generator): 64-channel tiles through registers; small problems -> many splits
        const int64_t tl = (N / 64) * taps * (Cin / 64) * batch;
        static int64_t tgt32 = 0;
        if (!tgt32) { tgt32 = 512; }    // 256 / 512 / 1024 / 2048 workgroups: 78 / 34 / 39 / 52 us per launch (16 per step; 2048 was the round-2 value)
        int64_t sp = tl >= tgt32 / 2 - 64 ? 1 : (tgt32 + tl / 2) / tl;
        int64_t ch = cdiv(cdiv(M, sp), TBK) * TBK;
        if (ch < 2 * TBK) ch = 2 * TBK;
        sp = cdiv(M, ch);
        p.chunk = (int)ch; p.splits = (int)sp;
        const dim3 g((unsigned)(N / 64), (unsigned)(taps * (Cin / 64)), (unsigned)(sp * batch));
        osp_note_symbol("conv_wgrad_bf16_tr_kernel<64,f32>");
        hipLaunchKernelGGL((conv_wgrad_bf16_tr_kernel<64, true>), g, dim3(256), 0, stream, p);
    }
    else if (fast) { osp_note_symbol("conv_wgrad_bf16_kernel<fast>"); hipLaunchKernelGGL((conv_wgrad_bf16_kernel<true>), grid, dim3(256), 0, stream, p); }
    else { osp_note_symbol("conv_wgrad_bf16_kernel"); hipLaunchKernelGGL((conv_wgrad_bf16_kernel<false>), grid, dim3(256), 0, stream, p); }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

extern "C" int osp_conv_wgrad_bf16(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                   int64_t M, int64_t Trows, int64_t Tin, int64_t N, int64_t Cin, int64_t taps, int64_t pad,
                                   int64_t x_step, const float* arow, const float* oscale, float* dW, int64_t ldw,
                                   float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                   hipStream_t stream) {
    const int64_t d2[5] = {Trows, 1, taps, 0, 0};
    return conv_wgrad_bf16_impl(d2, dY, y_bf16, ldy, X, x_bf16, ldx, M, Trows, Tin, N, Cin, taps, pad, x_step, arow, oscale, dW, ldw,
                                db, batch, sYb, sXb, sWb, sDb, nullptr, 0, stream);
}

// The same with a caller-supplied workspace (ws_bytes of device memory, 16-byte aligned, private to this call until the stream
// has passed it): the frame splits leave their partial sums there and a second launch adds them up in a fixed order --
// no f32 atomics, bit-reproducible weight gradients (csrc/wgrad_ring.hip).  The workspace bounds the number of splits.
extern "C" int osp_conv_wgrad_bf16_ws(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                      int64_t M, int64_t Trows, int64_t Tin, int64_t N, int64_t Cin, int64_t taps, int64_t pad,
                                      int64_t x_step, const float* arow, const float* oscale, float* dW, int64_t ldw,
                                      float* db, int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb,
                                      float* ws, int64_t ws_bytes, hipStream_t stream) {
    const int64_t d2[5] = {Trows, 1, taps, 0, 0};
    return conv_wgrad_bf16_impl(d2, dY, y_bf16, ldy, X, x_bf16, ldx, M, Trows, Tin, N, Cin, taps, pad, x_step, arow, oscale, dW, ldw,
                                db, batch, sYb, sXb, sWb, sDb, ws, ws_bytes, stream);
}

// 2-D weight gradient: dW[n, kh, kw, c] += sum dY[u, th, tw, n] * X[u, th*x_step_h + kh - pad_h, tw*x_step + kw - pad, c]
extern "C" int osp_conv2d_wgrad_bf16(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                     int64_t M, int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t N, int64_t Cin,
                                     int64_t taps, int64_t KW, int64_t pad_h, int64_t pad, int64_t x_step_h, int64_t x_step,
                                     float* dW, float* db, hipStream_t stream) {
    const int64_t d2[5] = {Wrows, Hin, KW, x_step_h, pad_h};
    return conv_wgrad_bf16_impl(d2, dY, y_bf16, ldy, X, x_bf16, ldx, M, Trows, Win, N, Cin, taps, pad, x_step, nullptr, nullptr, dW,
                                taps * Cin, db, 1, 0, 0, 0, 0, nullptr, 0, stream);
}

extern "C" int osp_conv2d_wgrad_bf16_ws(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                        int64_t M, int64_t Trows, int64_t Wrows, int64_t Hin, int64_t Win, int64_t N, int64_t Cin,
                                        int64_t taps, int64_t KW, int64_t pad_h, int64_t pad, int64_t x_step_h, int64_t x_step,
                                        float* dW, float* db, float* ws, int64_t ws_bytes, hipStream_t stream) {
    const int64_t d2[5] = {Wrows, Hin, KW, x_step_h, pad_h};
    return conv_wgrad_bf16_impl(d2, dY, y_bf16, ldy, X, x_bf16, ldx, M, Trows, Win, N, Cin, taps, pad, x_step, nullptr, nullptr, dW,
                                taps * Cin, db, 1, 0, 0, 0, 0, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ casts
__global__ void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<uint2*>(y)[i] = make_uint2(pk2(v.x, v.y), pk2(v.z, v.w));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = __builtin_bit_cast(unsigned short, (__bf16)x[i]);
}
extern "C" int osp_cast_bf16(const float* x, void* y, int64_t n, hipStream_t stream) {
    OSP_CHECK_ARG(x && y && n > 0, "bad args");
    const int64_t blocks = cdiv(n, 1024);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, x, (unsigned short*)y, n);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}

// ------------------------------------------------------------------------------------------------ weight packing
// out[n][tap][k] (bf16, contiguous) = w[n*sN + tap*sT + k*sK] (f32; any strides, sT may be negative for a flipped kernel).
// The dgrad GEMMs of the generator address the weights transposed (k-strided); packing them once per call into the
// k-contiguous bf16 layout lets the GEMM use 16-byte operand loads instead of its transposing element loader, which is
// ~2x slower than the GEMM itself on these small shapes.  32x32 tiles through LDS, reads along the unit-stride axis.
// Algorithmic bytes: 4 read + 2 written per weight.
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ w, const float* __restrict__ kscale,
                                                        unsigned short* __restrict__ out, int N, int taps,
                                                        int K, int64_t sN, int64_t sT, int64_t sK) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = w + (int64_t)tap * sT;
    const bool along_n = (sN < 0 ? -sN : sN) < (sK < 0 ? -sK : sK);
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = along_n ? n0 + tx : n0 + r, k = along_n ? k0 + r : k0 + tx;
        const float v = (n < N && k < K) ? src[(int64_t)n * sN + (int64_t)k * sK] * (kscale ? kscale[k] : 1.f) : 0.f;
        if (along_n) tile[r][tx] = v; else tile[tx][r] = v;      // tile[k_local][n_local]
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) out[((int64_t)n * taps + tap) * K + k] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][r]);
    }
}
// kscale (optional, K floats): every element is multiplied by kscale[k] first (layer-scale gamma folded into the dgrad weights).
extern "C" int osp_pack_bf16(const float* w, const float* kscale, void* out, int64_t N, int64_t taps, int64_t K, int64_t sN, int64_t sT,
                             int64_t sK, hipStream_t stream) {
    OSP_CHECK_ARG(w && out && N > 0 && taps > 0 && K > 0, "bad args");
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)cdiv(K, 32), (unsigned)cdiv(N, 32), (unsigned)taps), dim3(256), 0, stream, w, kscale,
                       (unsigned short*)out, (int)N, (int)taps, (int)K, sN, sT, sK);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}


// Many weight packs in one launch (the ConvNeXt blocks of a backbone need 3-4 bf16 copies each per optimiser step: W1, W2,
// W1^T and gamma * W2^T; one osp_pack_bf16 launch apiece was 48-64 launches per step).  desc_host: count rows of 10 int64
// {w, kscale (0 = none), out, N, taps, K, sN, sT, sK, out_f32}; a workgroup finds its item in a prefix table of 32x32 tiles.
// out_f32 = 1: the pack stays f32 (the k-contiguous weight copies of the exact-f32 input-gradient GEMMs, round 5: made by a torch gather
// before, which a call tape cannot hold -- the taped segments therefore kept the k-strided views and the register-staged kernel).
#define PACK_MULTI_MAX 32
struct PackMulti { const float* w[PACK_MULTI_MAX]; const float* ks[PACK_MULTI_MAX]; unsigned short* out[PACK_MULTI_MAX];
                   int N[PACK_MULTI_MAX], taps[PACK_MULTI_MAX], K[PACK_MULTI_MAX], tk[PACK_MULTI_MAX], tn[PACK_MULTI_MAX];
                   long long sN[PACK_MULTI_MAX], sT[PACK_MULTI_MAX], sK[PACK_MULTI_MAX]; int first[PACK_MULTI_MAX + 1]; int f32[PACK_MULTI_MAX];
                   int count; };
__global__ __launch_bounds__(256) void pack_bf16_multi_kernel(PackMulti d) {
    __shared__ float tile[32][33];
    int it = 0;
    for (int i = 1; i < d.count; ++i) it = (int)blockIdx.x >= d.first[i] ? i : it;
    int b = blockIdx.x - d.first[it];
    const int kt = b % d.tk[it]; b /= d.tk[it];
    const int nt = b % d.tn[it], tap = b / d.tn[it];
    const int N = d.N[it], K = d.K[it], taps = d.taps[it];
    const long long sN = d.sN[it], sK = d.sK[it];
    const float* src = d.w[it] + (long long)tap * d.sT[it];
    const float* kscale = d.ks[it];
    const int n0 = nt * 32, k0 = kt * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool along_n = (sN < 0 ? -sN : sN) < (sK < 0 ? -sK : sK);
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = along_n ? n0 + tx : n0 + r, k = along_n ? k0 + r : k0 + tx;
        const float v = (n < N && k < K) ? src[(long long)n * sN + (long long)k * sK] * (kscale ? kscale[k] : 1.f) : 0.f;
        if (along_n) tile[r][tx] = v; else tile[tx][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) {
            const long long o = ((long long)n * taps + tap) * K + k;
            if (d.f32[it]) reinterpret_cast<float*>(d.out[it])[o] = tile[tx][r];
            else d.out[it][o] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][r]);
        }
    }
}
extern "C" int osp_pack_bf16_multi(const int64_t* desc_host, int64_t count, hipStream_t stream) {
    OSP_CHECK_ARG(desc_host && count > 0, "bad args");
    for (int64_t lo = 0; lo < count; lo += PACK_MULTI_MAX) {
        PackMulti d;
        const int64_t hi = lo + PACK_MULTI_MAX < count ? lo + PACK_MULTI_MAX : count;
        int blocks = 0;
        d.count = (int)(hi - lo);
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t* r = desc_host + 10 * i;
            const int k = (int)(i - lo);
            d.w[k] = (const float*)(intptr_t)r[0]; d.ks[k] = (const float*)(intptr_t)r[1]; d.out[k] = (unsigned short*)(intptr_t)r[2];
            d.N[k] = (int)r[3]; d.taps[k] = (int)r[4]; d.K[k] = (int)r[5]; d.sN[k] = r[6]; d.sT[k] = r[7]; d.sK[k] = r[8]; d.f32[k] = r[9] != 0;
            OSP_CHECK_ARG(d.w[k] && d.out[k] && d.N[k] > 0 && d.taps[k] > 0 && d.K[k] > 0, "bad descriptor");
            d.tk[k] = (d.K[k] + 31) / 32; d.tn[k] = (d.N[k] + 31) / 32;
            d.first[k] = blocks;
            blocks += d.tk[k] * d.tn[k] * d.taps[k];
        }
        d.first[d.count] = blocks;
        hipLaunchKernelGGL(pack_bf16_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
