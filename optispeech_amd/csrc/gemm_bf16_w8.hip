// 8-wave 256x256 direct-to-LDS conv-GEMM kernels (body: gemm_bf16_glds.h): the DiscriminatorP 512 -> 1024 / 1024 -> 1024 layers
// (vocoder/wavenext/disc/_discriminators.py:51-60) forward + fused-phase dgrad.
#include "gemm_bf16_glds.h"

__global__ __launch_bounds__(512) void conv_gemm_bf16_glds8_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<256, 2, 256, 8>(pp, glds_smem, grid_tile_ctx());
}
__global__ __launch_bounds__(512) void conv_gemm_bf16_glds8e_kernel(const GemmB pp) {
    conv_gemm_bf16_glds_body<256, 2, 256, 8, true>(pp, glds_smem, grid_tile_ctx());
}

static void w8_attrs() {
    static int done = 0;
    if (done) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS8_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_bf16_glds8e_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GLDS8_LDS);
    done = 1;
}

int osp_launch_glds8(const GemmB& p, dim3 grid, bool early, hipStream_t stream) {
    w8_attrs();
    osp_note_symbol(early ? "conv_gemm_bf16_glds8e_kernel" : "conv_gemm_bf16_glds8_kernel");
    if (early) hipLaunchKernelGGL(conv_gemm_bf16_glds8e_kernel, grid, dim3(512), GLDS8_LDS, stream, p);
    else hipLaunchKernelGGL(conv_gemm_bf16_glds8_kernel, grid, dim3(512), GLDS8_LDS, stream, p);
    OSP_LAUNCH_CHECK();
    return OSP_OK;
}
