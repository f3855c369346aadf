// Exact-f32 conv-GEMM on the direct-to-LDS body (gemm_bf16_glds.h, F32 mode): f32 operands staged as they lie by LDS-DMA, products on
// v_mfma_f32_32x32x2_f32, the row-domain epilogues of the bf16 family with an f32 destination (exact erf GELU).  Serves
// osp_conv_gemm_f32 for the shapes the old register-staged kernel (gemm.hip: BK = 16, four 4-byte LDS stores per loaded float4, one
// __syncthreads() -- i.e. a vmcnt(0) -- per 16-deep slab) was slowest on: the index-critical forward of the generator (text encoder,
// duration predictor: exact f32 in every precision mode) and the generator's forward in the f32 / "mixed" parity modes.
//   8 192 x 1 024 x 256 (encoder pwconv1 at 64 sentences): 106 us (40 TFLOP/s) with the old kernel.
// Not taken (the old kernel stays): k-strided weights (the dgrad views), a per-row A scale, Cin % 32 != 0, unaligned operands.
#include "gemm_bf16_glds.h"

__global__ __launch_bounds__(256) void conv_gemm_f32_glds_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2, TBN, 4, false, true>(pp, glds_smem, grid_tile_ctx());
}
// 128 x 64 tiles: twice the tiles where 128 x 128 ones would leave CUs idle (N = 256 at 8 k rows: 128 -> 256 tiles)
__global__ __launch_bounds__(256) void conv_gemm_f32_glds_n64_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2, 64, 4, false, true>(pp, glds_smem, grid_tile_ctx());
}

// 64 x 64 tiles (four waves of 32 x 32, two 16 KB stages: five workgroups per CU), round 5: the generator's GEMMs at 2 048 - 4 096 rows are
// 96 - 128 tiles of 128 x 64 -- a third of the CUs -- and the exact-f32 matrix pipe is 16x slower than the bf16 one, so what such a launch
// costs is (tiles in flight) x (k-slabs), not bytes: four times the tiles at twice the operand traffic per flop
__global__ __launch_bounds__(256) void conv_gemm_f32_glds_s64_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<64, 2, 64, 4, false, true>(pp, glds_smem, grid_tile_ctx());
}

// the split-bf16 instantiations of the same three tile shapes live in gemm_f32_split.hip (a translation unit of their own: compile time)
int osp_launch_f32_split(const GemmB& p, int shape, dim3 grid, int lds, hipStream_t stream);

// returns 1 when the launch was taken, 0 when the caller should use its own kernel, < 0 on a launch error
static int try_gemm_f32_glds_impl(int split, const float* A, int64_t lda, int64_t M, int64_t T, int64_t Cin, int64_t taps, int64_t pad,
                          const float* a_rowscale, const float* B, int64_t sBn, int64_t sBtap, int64_t sBk, int64_t N, float* C,
                          int64_t ldc, int64_t epi, const float* bias, const float* gamma, const float* res, int64_t ldr,
                          const float* rowmask, const float* rowscale, float* aux_out, const float* aux_in, int64_t ld_aux,
                          int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate, hipStream_t stream) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("OSP_GEMM_F32_DMA"); on = (e && atoi(e) == 0) ? 0 : 1; }       // 0: the old kernel (A/B runs)
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!on || sBk != 1 || a_rowscale || Cin % 32 != 0 || lda % 4 != 0 || sBn % 4 != 0 || sBtap % 4 != 0 || sAb % 4 != 0 || sBb % 4 != 0 ||
        !al16(A) || !al16(B) || N % 8 != 0 || ldc % 4 != 0 || !al16(C))
        return 0;
    // tile shape: 128 x 128 once that fills the chip (>= 192 tiles), else 128 x 64; fewer than 64 of those: the old kernel with its 64 x 64 tiles
    static int64_t tmin = -1;
    if (tmin < 0) { const char* e = getenv("OSP_GEMM_F32_DMA_MIN"); tmin = e ? atoi(e) : 64; }           // (measured: 10-18 % over the old kernel down to ~60 tiles, profiles/r04_gemm_f32_probe.txt)
    const bool wide = cdiv(M, 128) * cdiv(N, 128) * batch >= 192;
    // 128 x 64 while those tiles fill the chip; below that 64 x 64 tiles (down to 32 of them)
    static int64_t s64max = -1;
    if (s64max < 0) { const char* e = getenv("OSP_GEMM_F32_S64_BELOW"); s64max = e ? atoi(e) : 384; }
    const int64_t t64 = cdiv(M, 128) * cdiv(N, 64) * batch;
    const bool small = !wide && t64 < s64max && cdiv(M, 64) * cdiv(N, 64) * batch >= 32;
    if (!wide && !small && t64 < tmin) return 0;
    GemmB p;
    // global strides and extents of the two operands in 2-BYTE units (see the body): an f32 element is two of them
    p.A = A; p.a_bf16 = 0; p.lda = 2 * lda; p.M = (int)M; p.Trows = (int)T; p.Tin = (int)T; p.Cin = (int)(2 * Cin);
    p.taps = (int)taps; p.a_step = 1; p.a_tapstep = 1; p.a_off = (int)-pad; p.a_rowscale = nullptr;
    p.B = B; p.b_bf16 = 0; p.sBn = 2 * sBn; p.sBtap = 2 * sBtap; p.sBk = 1; p.N = (int)N;
    p.C = C; p.c_bf16 = 0; p.ldc = ldc; p.Tc = (int)T; p.c_step = 1; p.c_off = 0;
    p.epi = (int)epi; p.bias = bias; p.gamma = gamma; p.res = res; p.ldr = ldr; p.res_any = res; p.res_bf16 = 0;
    p.rowmask = rowmask; p.rowscale = rowscale; p.aux_out = aux_out; p.aux_in = aux_in; p.aux_bf16 = 0; p.ld_aux = ld_aux; p.slope = 0.f;
    p.sAb = 2 * sAb; p.sBb = 2 * sBb; p.sCb = sCb; p.sXb = sXb; p.accumulate = (int)accumulate;
    p.nphase = 0; p.nt_out = 0;
    p.fd_trows = make_fastdiv((unsigned)T); p.fd_wrows = make_fastdiv((unsigned)T);
    p.Wrows = (int)T; p.Hin = 1; p.KW = (int)taps; p.a_step_h = 0; p.a_tapstep_h = 0; p.a_off_h = 0; p.Wc = (int)T; p.c_step_h = 0;
    p.c_off_h = 0; p.sBtap_h = 0;
    constexpr int LDS64 = 2 * (128 + 64) * TBK * 2;
    constexpr int LDSS = 2 * (64 + 64) * TBK * 2;
    static int attr = 0;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_glds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_glds_n64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS64);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_glds_s64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDSS);
        attr = 1;
    }
    osp_note_symbol(split ? "conv_gemm_f32_split_kernel" : "conv_gemm_f32_glds_kernel");
    osp_note_flops(2.0 * M * taps * (double)Cin * N * batch);
    osp_note_bytes(4.0 * batch * ((double)M * Cin + (double)N * taps * Cin + (double)M * N));
    if (split) {
        const int shape = small ? 2 : wide ? 0 : 1;
        const dim3 grid((unsigned)cdiv(N, shape == 0 ? 128 : 64), (unsigned)cdiv(M, shape == 2 ? 64 : 128), (unsigned)batch);
        return osp_launch_f32_split(p, shape, grid, shape == 0 ? GLDS_LDS : shape == 1 ? LDS64 : LDSS, stream);
    }
    if (small) hipLaunchKernelGGL(conv_gemm_f32_glds_s64_kernel, dim3((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64), (unsigned)batch), dim3(256), LDSS, stream, p);
    else if (wide) hipLaunchKernelGGL(conv_gemm_f32_glds_kernel, dim3((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128), (unsigned)batch), dim3(256), GLDS_LDS, stream, p);
    else hipLaunchKernelGGL(conv_gemm_f32_glds_n64_kernel, dim3((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 128), (unsigned)batch), dim3(256), LDS64, stream, p);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}

int osp_try_gemm_f32_glds(const float* A, int64_t lda, int64_t M, int64_t T, int64_t Cin, int64_t taps, int64_t pad,
                          const float* a_rowscale, const float* B, int64_t sBn, int64_t sBtap, int64_t sBk, int64_t N, float* C,
                          int64_t ldc, int64_t epi, const float* bias, const float* gamma, const float* res, int64_t ldr,
                          const float* rowmask, const float* rowscale, float* aux_out, const float* aux_in, int64_t ld_aux,
                          int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate, hipStream_t stream) {
    return try_gemm_f32_glds_impl(0, A, lda, M, T, Cin, taps, pad, a_rowscale, B, sBn, sBtap, sBk, N, C, ldc, epi, bias, gamma, res, ldr, rowmask,
                                  rowscale, aux_out, aux_in, ld_aux, batch, sAb, sBb, sCb, sXb, accumulate, stream);
}

extern "C" int osp_conv_gemm_f32(const float* A, int64_t lda, int64_t M, int64_t T, int64_t Cin, int64_t taps, int64_t pad,
                                 const float* a_rowscale, const float* B, int64_t sBn, int64_t sBtap, int64_t sBk, int64_t N, float* C,
                                 int64_t ldc, int64_t epi, const float* bias, const float* gamma, const float* res, int64_t ldr,
                                 const float* rowmask, const float* rowscale, float* aux_out, const float* aux_in, int64_t ld_aux,
                                 int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate, hipStream_t stream);

// osp_conv_gemm_f32 with f32 operands in HBM and SPLIT-bf16 products (gemm_bf16_glds.h, SPLIT): every operand element enters the matrix
// pipe as hi + lo (two bf16 numbers, 16 significand bits), three bf16 MFMAs per product, f32 accumulate -- <= 1.1e-5 of |a b| per product
// where the exact pipe gives 6e-8 and plain bf16 operands 4e-3.  For GEMMs whose result feeds continuous quantities only (the "mixed"
// parity mode's generator outside the index-critical path).  Same arguments, same epilogues; shapes the direct-to-LDS kernel does not take
// run as osp_conv_gemm_f32 (exact).
extern "C" int osp_conv_gemm_f32_split(const float* A, int64_t lda, int64_t M, int64_t T, int64_t Cin, int64_t taps, int64_t pad,
                                       const float* a_rowscale, const float* B, int64_t sBn, int64_t sBtap, int64_t sBk, int64_t N, float* C,
                                       int64_t ldc, int64_t epi, const float* bias, const float* gamma, const float* res, int64_t ldr,
                                       const float* rowmask, const float* rowscale, float* aux_out, const float* aux_in, int64_t ld_aux,
                                       int64_t batch, int64_t sAb, int64_t sBb, int64_t sCb, int64_t sXb, int64_t accumulate, hipStream_t stream) {
    OSP_CHECK_ARG(A && B && C, "null operand");
    OSP_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && taps > 0 && T > 0 && batch > 0, "bad shape");
    OSP_CHECK_ARG(M % T == 0, "M must be a whole number of utterances of T frames");
    OSP_CHECK_ARG(epi >= 0 && epi <= BEPI_MASK, "unknown epilogue");
    OSP_CHECK_ARG(epi != BEPI_SCALE_RES_MASK || res, "epilogue needs res");
    OSP_CHECK_ARG((epi != BEPI_GELU_BWD && epi != BEPI_RELU_BWD && epi != BEPI_AXMY) || aux_in, "epilogue needs aux_in");
    const int r = try_gemm_f32_glds_impl(1, A, lda, M, T, Cin, taps, pad, a_rowscale, B, sBn, sBtap, sBk, N, C, ldc, epi, bias, gamma, res, ldr, rowmask,
                                         rowscale, aux_out, aux_in, ld_aux, batch, sAb, sBb, sCb, sXb, accumulate, stream);
    if (r < 0) { osp_set_error("osp_conv_gemm_f32_split: launch failed"); return OSP_ERR_HIP; }
    if (r > 0) return OSP_OK;
    return osp_conv_gemm_f32(A, lda, M, T, Cin, taps, pad, a_rowscale, B, sBn, sBtap, sBk, N, C, ldc, epi, bias, gamma, res, ldr, rowmask, rowscale,
                             aux_out, aux_in, ld_aux, batch, sAb, sBb, sCb, sXb, accumulate, stream);
}
