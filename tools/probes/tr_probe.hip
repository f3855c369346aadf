// What does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i (u16).  Each lane supplies an address; print the 4 u16 per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k2(unsigned* out, int stride_bytes, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = lane * stride_bytes;
    else addr = (lane & 15) * stride_bytes + (lane >> 4) * 8;   // 16 rows of `stride_bytes`, 4 column-chunks of 8 B
    addr += (unsigned)(size_t)(&lds[0]);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[lane * 2] = v[0]; out[lane * 2 + 1] = v[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 2 * 4);
    unsigned h[128];
    for (int mode = 0; mode < 2; ++mode)
    for (int stride : {8, 32, 128}) {
        hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, stride, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d stride %d B (values are element indices = byte_addr/2):\n", mode, stride);
        for (int l = 0; l < 64; ++l) {
            printf(" l%02d:[%4u %4u %4u %4u]", l, h[2*l] & 0xffff, h[2*l] >> 16, h[2*l+1] & 0xffff, h[2*l+1] >> 16);
            if ((l & 3) == 3) printf("\n");
        }
    }
    return 0;
}
