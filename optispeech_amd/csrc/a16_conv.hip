// Dilated and transposed 1-D convolutions of the HiFi-GAN-style vocoder (reference:
// vocoder/streaming_hifigan/modules/conv_layer.py:118-200 -- CausalConv1d = left zero pad (k-1)*dilation + conv1d(dilation),
// CausalConvTranspose1d = conv_transpose1d(stride s, kernel k) behind a replication pad; residual_block.py:24-106).
//
// Thin entries onto the conv-GEMM family (gemm_bf16.hip / wgrad_bf16.hip): on channels-last (B, T, C) frames
//   * a dilated k-tap conv is the GEMM over K = k*Cin whose operand row for output frame t and tap j is frame t + j*dil - pad
//     (`a_tapstep` = dilation; rows outside [0, T) read as zero = the causal / "same" zero padding, no padded copy);
//   * a transposed conv of stride s splits into s output phases r = t_out mod s; phase r is a dense conv over the input with the
//     sub-sampled taps j = r + i*s, written to output rows q*s + r (`c_step` = s, `c_off` = r): no zero-stuffed input, no
//     multiply-by-zero work;
//   * its input gradient is a stride-s conv over dy (`a_step` = s), its weight gradient the weight-gradient GEMM with the roles of
//     the operands exchanged (x is the dense side, dy the strided one), which lands in the (Cin, k, Cout) layout the transposed
//     conv keeps its weights in.
// Operands are f32 or bf16 in HBM (flags), products are bf16 x bf16 with f32 accumulation; the exact-f32 parity mode composes
// three calls on hi / lo operand halves (optispeech_amd/ops.py: _split3), as the discriminator stacks do.
#include "osp_common.h"

extern "C" int osp_conv_gemm_bf16(const void* A, int64_t a_bf16, int64_t lda, int64_t M, int64_t Trows, int64_t Tin,
                                  int64_t Cin, int64_t taps, int64_t a_step, int64_t a_tapstep, int64_t a_off,
                                  const float* a_rowscale, const void* B, int64_t b_bf16, int64_t sBn, int64_t sBtap,
                                  int64_t sBk, int64_t N, void* C, int64_t c_bf16, int64_t ldc, int64_t Tc,
                                  int64_t c_step, int64_t c_off, int64_t epi, const float* bias, const float* gamma,
                                  const void* res, int64_t res_bf16, int64_t ldr, const float* rowmask, const float* rowscale,
                                  void* aux_out, const void* aux_in, int64_t aux_bf16, int64_t ld_aux, float slope,
                                  int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate,
                                  hipStream_t stream);
extern "C" int osp_conv_wgrad_bf16(const void* dY, int64_t y_bf16, int64_t ldy, const void* X, int64_t x_bf16, int64_t ldx,
                                   int64_t M, int64_t Trows, int64_t Tin, int64_t N, int64_t Cin, int64_t taps, int64_t pad,
                                   int64_t x_step, const float* arow, const float* oscale, float* dW, int64_t ldw, float* db,
                                   int64_t batch, int64_t sYb, int64_t sXb, int64_t sWb, int64_t sDb, hipStream_t stream);

static inline const void* off_elems(const void* p, int64_t elems, int64_t is_bf16) {
    return reinterpret_cast<const char*>(p) + elems * (is_bf16 ? 2 : 4);
}

// y[b, t, n] (+)= bias[n] + sum_{j < k} sum_c x[b, t + j*dil - pad_left, c] * w[n, j, c]          0 <= t < T
//   x (B, T, Cin), w (Cout, k, Cin) kernel-native, y (B, T, Cout); pad_left = (k-1)*dil: causal, (k-1)*dil/2: "same".
//   accumulate != 0 adds to an f32 y (used by the split-operand parity mode).
extern "C" int osp_conv1d_dilated_fwd(const void* x, int64_t x_bf16, const void* w, int64_t w_bf16, const float* bias, void* y,
                                      int64_t y_bf16, int64_t B, int64_t T, int64_t Cin, int64_t Cout, int64_t k, int64_t dil,
                                      int64_t pad_left, int64_t accumulate, hipStream_t stream) {
    OSP_CHECK_ARG(x && w && y && B > 0 && T > 0 && Cin > 0 && Cout > 0 && k > 0 && dil > 0 && pad_left >= 0, "bad args");
    return osp_conv_gemm_bf16(x, x_bf16, Cin, B * T, T, T, Cin, k, 1, dil, -pad_left, nullptr, w, w_bf16, k * Cin, Cin, 1, Cout, y, y_bf16,
                              Cout, T, 1, 0, 0, bias, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.f, 1, 0, 0, 0, 0,
                              accumulate, stream);
}

// Backward of the above.  dx (optional): dx[b, t, c] (+)= sum_{j, n} dy[b, t - j*dil + pad_left, n] * w[n, j, c]
//                         dw (optional, f32, accumulated): dw[n, j, c] += sum_{b, t} dy[b, t, n] * x[b, t + j*dil - pad_left, c]
//                         db (optional, f32, accumulated): db[n] += sum_{b, t} dy[b, t, n]
extern "C" int osp_conv1d_dilated_bwd(const void* dy, int64_t dy_bf16, const void* x, int64_t x_bf16, const void* w, int64_t w_bf16,
                                      void* dx, int64_t dx_bf16, float* dw, float* db, int64_t B, int64_t T, int64_t Cin, int64_t Cout,
                                      int64_t k, int64_t dil, int64_t pad_left, int64_t accumulate_dx, hipStream_t stream) {
    OSP_CHECK_ARG(dy && B > 0 && T > 0 && Cin > 0 && Cout > 0 && k > 0 && dil > 0 && pad_left >= 0, "bad args");
    OSP_CHECK_ARG((!dx || w) && (!dw || x) && (!db || dw), "dx needs w, dw needs x, db comes with dw");
    int rc = OSP_OK;
    if (dx) {
        // taps reversed: j' = k-1-j reads dy row t + j'*dil - ((k-1)*dil - pad_left); weight element (c, j', n) = w[n, k-1-j', c]
        const void* wl = off_elems(w, (k - 1) * Cin, w_bf16);
        rc = osp_conv_gemm_bf16(dy, dy_bf16, Cout, B * T, T, T, Cout, k, 1, dil, -((k - 1) * dil - pad_left), nullptr, wl, w_bf16, 1, -Cin,
                                k * Cin, Cin, dx, dx_bf16, Cin, T, 1, 0, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr,
                                0, 0, 0.f, 1, 0, 0, 0, 0, accumulate_dx, stream);
        if (rc != OSP_OK) return rc;
    }
    if (dw) {
        // one tap per launch: tap j of a dilated conv is the 1-tap weight gradient with pad = pad_left - j*dil
        for (int64_t j = 0; j < k && rc == OSP_OK; ++j)
            rc = osp_conv_wgrad_bf16(dy, dy_bf16, Cout, x, x_bf16, Cin, B * T, T, T, Cout, Cin, 1, pad_left - j * dil, 1, nullptr, nullptr,
                                     dw + j * Cin, k * Cin, j == 0 ? db : nullptr, 1, 0, 0, 0, 0, stream);
    }
    return rc;
}

// y[b, to, n] = bias[n] + sum_{ti, j : ti*s + j == to} sum_c x[b, ti, c] * wn[n, j, c]        0 <= to < Tout = (T-1)*s + k
//   = torch.nn.functional.conv_transpose1d(stride s, padding 0) with wn[n, j, c] = weight[c, n, j]  (wn: the (Cout, k, Cin) pack).
extern "C" int osp_conv_transpose1d_fwd(const void* x, int64_t x_bf16, const void* wn, int64_t w_bf16, const float* bias, void* y,
                                        int64_t y_bf16, int64_t B, int64_t T, int64_t Cin, int64_t Cout, int64_t k, int64_t s,
                                        int64_t accumulate, hipStream_t stream) {
    OSP_CHECK_ARG(x && wn && y && B > 0 && T > 0 && Cin > 0 && Cout > 0 && k > 0 && s > 0 && k >= s, "bad args (kernel >= stride)");
    const int64_t Tout = (T - 1) * s + k;
    int rc = OSP_OK;
    for (int64_t r = 0; r < s && rc == OSP_OK; ++r) {
        const int64_t Q = (Tout - r + s - 1) / s, taps = (k - r + s - 1) / s;
        if (Q <= 0 || taps <= 0) continue;
        rc = osp_conv_gemm_bf16(x, x_bf16, Cin, B * Q, Q, T, Cin, taps, 1, -1, 0, nullptr, off_elems(wn, r * Cin, w_bf16), w_bf16, k * Cin,
                                s * Cin, 1, Cout, y, y_bf16, Cout, Tout, s, r, 0, bias, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr,
                                0, 0, 0.f, 1, 0, 0, 0, 0, accumulate, stream);
    }
    return rc;
}

// Backward of the above, weights in the transposed conv's own layout wt (Cin, k, Cout) = weight[c, n, j] -> wt[c, j, n]:
//   dx (optional): dx[b, ti, c] (+)= sum_{j, n} dy[b, ti*s + j, n] * wt[c, j, n]
//   dwt (optional, f32, accumulated, (Cin, k, Cout)): dwt[c, j, n] += sum_{b, ti} x[b, ti, c] * dy[b, ti*s + j, n]
// (the bias gradient is a column sum of dy: osp_colsum_prod with b = null)
extern "C" int osp_conv_transpose1d_bwd(const void* dy, int64_t dy_bf16, const void* x, int64_t x_bf16, const void* wt, int64_t w_bf16,
                                        void* dx, int64_t dx_bf16, float* dwt, int64_t B, int64_t T, int64_t Cin, int64_t Cout, int64_t k,
                                        int64_t s, int64_t accumulate_dx, hipStream_t stream) {
    OSP_CHECK_ARG(dy && B > 0 && T > 0 && Cin > 0 && Cout > 0 && k > 0 && s > 0 && k >= s, "bad args (kernel >= stride)");
    OSP_CHECK_ARG((!dx || wt) && (!dwt || x), "dx needs wt, dwt needs x");
    const int64_t Tout = (T - 1) * s + k;
    int rc = OSP_OK;
    if (dx)
        rc = osp_conv_gemm_bf16(dy, dy_bf16, Cout, B * T, T, Tout, Cout, k, s, 1, 0, nullptr, wt, w_bf16, k * Cout, Cout, 1, Cin, dx, dx_bf16, Cin,
                                T, 1, 0, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.f, 1, 0, 0, 0, 0,
                                accumulate_dx, stream);
    if (rc == OSP_OK && dwt)
        rc = osp_conv_wgrad_bf16(x, x_bf16, Cin, dy, dy_bf16, Cout, B * T, T, Tout, Cin, Cout, k, 0, s, nullptr, nullptr, dwt, k * Cout, nullptr,
                                 1, 0, 0, 0, 0, stream);
    return rc;
}
