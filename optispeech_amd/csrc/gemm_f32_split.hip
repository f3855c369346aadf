// Split-bf16 instantiations of the exact-f32 conv-GEMM's three tile shapes (gemm_f32_glds.hip launches them; the body and what
// "split" means: gemm_bf16_glds.h, SPLIT).  f32 operands in HBM and in LDS, (hi, lo) bf16 pairs made in registers, three
// v_mfma_f32_32x32x16_bf16 per 16-deep product block, f32 accumulators and the f32 row-domain epilogues.
#include "gemm_bf16_glds.h"

__global__ __launch_bounds__(256) void conv_gemm_f32_split_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2, TBN, 4, false, true, true>(pp, glds_smem, grid_tile_ctx());
}
__global__ __launch_bounds__(256) void conv_gemm_f32_split_n64_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<128, 2, 64, 4, false, true, true>(pp, glds_smem, grid_tile_ctx());
}
__global__ __launch_bounds__(256) void conv_gemm_f32_split_s64_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<64, 2, 64, 4, false, true, true>(pp, glds_smem, grid_tile_ctx());
}

// shape 0: 128 x 128 tiles, 1: 128 x 64, 2: 64 x 64 (grid and LDS bytes chosen by the caller exactly as for the exact kernels)
int osp_launch_f32_split(const GemmB& p, int shape, dim3 grid, int lds, hipStream_t stream) {
    static int attr = 0;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_split_n64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 64) * TBK * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_f32_split_s64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 + 64) * TBK * 2);
        attr = 1;
    }
    if (shape == 0) hipLaunchKernelGGL(conv_gemm_f32_split_kernel, grid, dim3(256), lds, stream, p);
    else if (shape == 1) hipLaunchKernelGGL(conv_gemm_f32_split_n64_kernel, grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(conv_gemm_f32_split_s64_kernel, grid, dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}
