// Shared by the weight-gradient translation units (wgrad_bf16.hip: tile-per-tap kernels with f32 atomics; wgrad_ring.hip:
// ring-pipelined, atomic-free kernels with a split workspace).
#pragma once
#include "gemm_bf16_common.h"

// dW[n, j, c] += oscale[n] * sum_{u,t} arow * dY[u, t, n] * X[u, t*x_step + j - pad, c];  db[n] likewise.
// Both operands are reduction-major -> transposing loader for both.  Split over the frame dimension, f32 atomics.
struct WgradB {
    const void* dY; int y_bf16; int64_t ldy; const void* X; int x_bf16; int64_t ldx;
    int M, Trows, Tin, N, Cin, taps, pad, x_step;
    int Wrows, Hin, KW, x_step_h, pad_h;              // 2-D extension (1-D: Wrows = Trows, Hin = 1, KW = taps)
    FastDiv fd_trows, fd_wrows;
    const float *arow, *oscale; float* dW; int64_t ldw; float* db; int chunk, splits;
    int64_t sYb, sXb, sWb, sDb;
    unsigned y_bytes, x_bytes;                        // extent of one batch slice of dY / X (buffer-resource variant)
};

