// Grouped launches of the 4-wave direct-to-LDS conv-GEMM kernels (body: gemm_bf16_glds.h): the same layer of the stacks of one
// discriminator family in ONE grid (opt-in OSP_DISC_GROUPED=1, disc_ops.MultiConvStackFn).
#include "gemm_bf16_glds.h"

__global__ __launch_bounds__(256) void conv_gemm_bf16_glds_grp_kernel(const GemmGroup g) {
    TileCtx tc; const int k = group_pick(g, tc);
    conv_gemm_bf16_glds_body<128, 2>(g.p[k], glds_smem, tc);
}
__global__ __launch_bounds__(256) void conv_gemm_bf16_glds_n64_grp_kernel(const GemmGroup g) {
    TileCtx tc; const int k = group_pick(g, tc);
    conv_gemm_bf16_glds_body<128, 2, 64>(g.p[k], glds_smem, tc);
}

int osp_launch_glds_grp(const GemmGroup& g, int tiles, bool n64, hipStream_t stream) {
    static int attr = 0;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds_grp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds_n64_grp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 64) * TBK * 2);
        attr = 1;
    }
    if (n64) {
        osp_note_symbol("conv_gemm_bf16_glds_n64_grp_kernel");
        hipLaunchKernelGGL(conv_gemm_bf16_glds_n64_grp_kernel, dim3((unsigned)tiles), dim3(256), 2 * (128 + 64) * TBK * 2, stream, g);
    } else {
        osp_note_symbol("conv_gemm_bf16_glds_grp_kernel");
        hipLaunchKernelGGL(conv_gemm_bf16_glds_grp_kernel, dim3((unsigned)tiles), dim3(256), GLDS_LDS, stream, g);
    }
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
