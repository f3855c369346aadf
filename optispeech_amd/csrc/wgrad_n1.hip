// Weight gradient of a convolution with ONE output channel: the `conv_post` layers of DiscriminatorP / DiscriminatorR
// (vocoder/wavenext/disc/_discriminators.py:60, :160: Conv2d(1024, 1, (3, 1)) / Conv2d(C, 1, (3, 3))).
//   dW[j, c] += oscale * sum_m arow[m] * dY[m] * X[pos(m, j), c],   db += oscale * sum_m arow[m] * dY[m]
// This is not a GEMM -- a 128-wide tile is 99 % padding (the tile kernel took 80 us for the 1024-channel layer, 0.3 TB/s) -- but
// a dY-weighted column sum over the activation: HBM-bound, Cin * 2 bytes per row (every X row is wanted by `taps` output rows;
// neighbouring rows are in flight in the same workgroup at the same time, so the repeats are L1 / L2 hits).
//
// A workgroup owns a SLICE of 64 channels and a contiguous share of the rows: 8 lanes x 16 bytes cover the slice (one 128-byte
// line per row and tap), 32 row groups walk the share, 4 rows per trip with all their loads requested before the first FMA.
// The 32 row groups meet in LDS; one atomic per (tap, channel) and workgroup (49 k atomics for the 1024-channel layer).
// NOT bit-reproducible run to run: the row shares of a channel slice meet in f32 atomics, whose order the hardware decides (as in
// every split-K weight-gradient kernel of the library except smallcin_wgrad, which reduces owner-writes); the sums agree to f32
// round-off (1e-7 relative), which is what the determinism tests of the data-parallel path allow for.
#include "osp_common.h"

#define WN1_MAXT 9
struct WgradN1 {
    const void* dY; int y_bf16; int64_t ldy; const unsigned short* X; int64_t ldx;
    int M, Trows, Wrows, Hin, Tin, Cin, taps, KW, pad, pad_h, x_step, x_step_h;
    const float* arow; const float* oscale; float* dW; float* db; int nslices, rows_per_split;
};

template <int TAPS, int UN>
__global__ __launch_bounds__(256) void conv_wgrad_n1_kernel(const WgradN1 p) {
    constexpr int TCH = TAPS < 4 ? TAPS : 4;                     // taps reduced per LDS pass (9 taps x 32 row groups would not fit)
    __shared__ float red[32][TCH * 64 + 1];                      // [row group][tap][lane-in-slice][8 channels] (+ the bias column)
    const int tid = threadIdx.x, c8 = tid & 7, rgrp = tid >> 3;
    const int slice = blockIdx.x % p.nslices, split = blockIdx.x / p.nslices;
    const int c = slice * 64 + c8 * 8;
    const int mbeg = split * p.rows_per_split, mend = min(p.M, mbeg + p.rows_per_split);
    float acc[TAPS][8];
#pragma unroll
    for (int j = 0; j < TAPS; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[j][q] = 0.f;
    float bsum = 0.f;
    for (int m0 = mbeg + rgrp; m0 < mend; m0 += 32 * UN) {
        uint4 xv[UN][TAPS];
        float dy[UN];
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            const int m = m0 + 32 * i;
            const bool live = m < mend;
            const int ms = live ? m : mbeg;
            const int u = ms / p.Trows, t = ms - u * p.Trows, th = t / p.Wrows, tw = t - th * p.Wrows;
            float d = p.y_bf16 ? __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(p.dY)[(int64_t)ms * p.ldy]) << 16)
                               : reinterpret_cast<const float*>(p.dY)[(int64_t)ms * p.ldy];
            if (p.arow) d *= p.arow[ms];
            dy[i] = live ? d : 0.f;
            const int64_t base = (int64_t)u * p.Hin * p.Tin;
#pragma unroll
            for (int j = 0; j < TAPS; ++j) {
                const int kh = j / p.KW, kw = j - kh * p.KW;
                const int hh = th * p.x_step_h + kh - p.pad_h, tt = tw * p.x_step + kw - p.pad;
                const bool ok = live && j < p.taps && (unsigned)hh < (unsigned)p.Hin && (unsigned)tt < (unsigned)p.Tin;
                const int64_t row = ok ? base + (int64_t)hh * p.Tin + tt : 0;
                const uint4 x = *reinterpret_cast<const uint4*>(p.X + row * p.ldx + c);      // unconditional load, row 0 when out of range
                xv[i][j] = ok ? x : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            bsum += dy[i];
#pragma unroll
            for (int j = 0; j < TAPS; ++j) {
                const unsigned w[4] = {xv[i][j].x, xv[i][j].y, xv[i][j].z, xv[i][j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[j][2 * q] = fmaf(dy[i], __uint_as_float(w[q] << 16), acc[j][2 * q]);
                    acc[j][2 * q + 1] = fmaf(dy[i], __uint_as_float(w[q] & 0xffff0000u), acc[j][2 * q + 1]);
                }
            }
        }
    }
    const float os = p.oscale ? p.oscale[0] : 1.f;
#pragma unroll
    for (int j0 = 0; j0 < TAPS; j0 += TCH) {
        if (j0) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < TCH; ++jj)
            if (j0 + jj < TAPS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) red[rgrp][(jj * 8 + c8) * 8 + q] = acc[j0 + jj][q];
            }
        if (j0 == 0 && c8 == 0) red[rgrp][TCH * 64] = bsum;
        __syncthreads();
        for (int e = tid; e < TCH * 64 + 1; e += 256) {
            float s = 0.f;
#pragma unroll 8
            for (int g = 0; g < 32; ++g) s += red[g][e];
            if (e < TCH * 64) {
                const int j = j0 + (e >> 6), cc = e & 63;
                if (j < p.taps) atomicAdd(p.dW + (int64_t)j * p.Cin + slice * 64 + cc, os * s);
            } else if (j0 == 0 && p.db && slice == 0) {
                atomicAdd(p.db, os * s);
            }
        }
    }
}

// Called by the weight-gradient dispatcher (wgrad_bf16.hip) for N == 1, bf16 X (dY bf16 or f32: the score gradient), Cin % 64 == 0, taps <= 9, one problem.
int osp_launch_wgrad_n1(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t ldx, int64_t M, int64_t Trows, int64_t Wrows, int64_t Hin,
                        int64_t Tin, int64_t Cin, int64_t taps, int64_t KW, int64_t pad, int64_t pad_h, int64_t x_step, int64_t x_step_h,
                        const float* arow, const float* oscale, float* dW, float* db, hipStream_t stream) {
    WgradN1 p;
    p.dY = dY; p.y_bf16 = (int)y_bf16; p.ldy = ldy; p.X = reinterpret_cast<const unsigned short*>(X); p.ldx = ldx;
    p.M = (int)M; p.Trows = (int)Trows; p.Wrows = (int)Wrows; p.Hin = (int)Hin; p.Tin = (int)Tin; p.Cin = (int)Cin; p.taps = (int)taps;
    p.KW = (int)KW; p.pad = (int)pad; p.pad_h = (int)pad_h; p.x_step = (int)x_step; p.x_step_h = (int)x_step_h;
    p.arow = arow; p.oscale = oscale; p.dW = dW; p.db = db;
    p.nslices = (int)(Cin / 64);
    // ~512 workgroups over the chip (2 per CU), each at least one 128-row trip
    int64_t splits = 512 / p.nslices;
    if (splits < 1) splits = 1;
    int64_t rps = cdiv(cdiv(M, splits), 32) * 32;
    if (rps < 128) rps = 128;
    splits = cdiv(M, rps);
    p.rows_per_split = (int)rps;
    const dim3 grid((unsigned)(splits * p.nslices));
    osp_note_symbol("conv_wgrad_n1_kernel");
    if (taps <= 3) hipLaunchKernelGGL((conv_wgrad_n1_kernel<3, 4>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_n1_kernel<WN1_MAXT, 2>), grid, dim3(256), 0, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
