// Stand-alone reproducer (no torch, no Python): does a kernel return different results for the SAME input when a second
// PROCESS computes on the same MI355X?  Run one instance alone, then two side by side (tools/probes/stft_shared_gpu.sh).
//
//   variant "stft"  : the shipped STFT magnitude kernel (csrc/stft.hip, N = 512 / 1024 / 2048: 2 / 4 / 8 waves per workgroup,
//                     LDS ping-pong with one __syncthreads per stage)
//   variant "wave"  : the same kernel at N = 256 (ONE wave per workgroup: barriers are trivial)
//   variant "ldsmix": a control with no FFT in it -- every workgroup (256 threads) writes its slice to LDS, barrier, reads it
//                     back permuted, 16 rounds: only LDS + s_barrier + global loads / stores
//   variant "nolds" : a control without LDS / barriers: out[i] = sum of 64 strided reads (global memory only)
// Each iteration recomputes the output from the unchanged input and compares it BIT FOR BIT with the first iteration's output
// on the device (a mismatch counter kernel); the number of deviating iterations and elements is printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <unistd.h>
#include "../../optispeech_amd/csrc/stft.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void mismatch_kernel(const unsigned* a, const unsigned* b, size_t n, unsigned long long* cnt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}

__global__ __launch_bounds__(256) void ldsmix_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
    __shared__ float s[2][1024];
    const int base = blockIdx.x * 1024, j = threadIdx.x;
    for (int r = 0; r < 4; ++r) s[0][j + 256 * r] = x[(base + j + 256 * r) % n];
    int cur = 0;
    for (int round = 0; round < 16; ++round) {
        __syncthreads();
        for (int r = 0; r < 4; ++r) {
            const int i = j + 256 * r;
            s[cur ^ 1][i] = s[cur][(i * 37 + round) & 1023] * 0.5f + s[cur][(i + 512) & 1023] * 0.25f;
        }
        cur ^= 1;
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) out[base + j + 256 * r] = s[cur][j + 256 * r];
}

__global__ void nolds_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc += x[(i + k * 4099) % n] * (1.0f / (k + 1));
    out[i] = acc;
}

// "hog": the OTHER process of a pair -- long kernels (~2 ms each) on every CU with 64 KB of LDS per workgroup, back to back for
// `iters` milliseconds, so that the victim's workgroups can only run by pre-empting / time-slicing against them.
// (dynamic LDS: OSP_PROBE_HOG_LDS bytes, default 64 KB; 147456 = the 144 KB of the 8-wave conv-GEMM -- more than the 64 KB a
// kernel gets without hipFuncAttributeMaxDynamicSharedMemorySize)
extern __shared__ float hog_s[];
__global__ __launch_bounds__(256) void hog_kernel(float* out, int spins) {
    float* s = hog_s;
    for (int i = threadIdx.x; i < 16384; i += 256) s[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < spins; ++k) {
        acc += s[(threadIdx.x * 33 + k * 7) & 16383];
        if ((k & 1023) == 0) __syncthreads();
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

static size_t hog_lds() {
    static size_t v = 0;
    if (!v) {
        v = getenv("OSP_PROBE_HOG_LDS") ? (size_t)atoi(getenv("OSP_PROBE_HOG_LDS")) : 65536;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)v));
    }
    return v;
}

int main(int argc, char** argv) {
    const char* variant = argc > 1 ? argv[1] : "stft";
    if (!strcmp(variant, "hog")) {
        const int ms = argc > 2 ? atoi(argv[2]) : 5000;
        float* o; CK(hipMalloc(&o, 1 << 20));
        hipStream_t hs; CK(hipStreamCreate(&hs));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float total = 0.f; int n = 0;
        while (total < (float)ms) {
            CK(hipEventRecord(e0, hs));
            for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(hog_kernel, dim3(1024), dim3(256), hog_lds(), hs, o, 400000);
            CK(hipEventRecord(e1, hs)); CK(hipStreamSynchronize(hs));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); total += t; n += 8;
        }
        printf("RESULT hog pid %d: %d kernels, %.2f ms each\n", (int)getpid(), n, total / n);
        return 0;
    }
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int B = 8, T = 64 * 256 * 4;
    std::vector<float> hx((size_t)B * T);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.6f; }
    float *x, *out, *ref, *tw;
    unsigned long long* cnt;
    const size_t out_elems = (size_t)B * (1 + T / 128) * 1025 + (size_t)B * T;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&out, out_elems * 4)); CK(hipMalloc(&ref, out_elems * 4));
    CK(hipMalloc(&tw, 4096 * 8)); CK(hipMalloc(&cnt, 8));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Cfg { int N, hop; };
    std::vector<Cfg> cfgs;
    if (!strcmp(variant, "stft")) cfgs = {{1024, 256}, {2048, 512}, {512, 128}};
    else if (!strcmp(variant, "wave")) cfgs = {{256, 64}};
    else cfgs = {{0, 0}};
    long long bad_iters = 0, bad_elems = 0;
    const int mix_spins = getenv("OSP_PROBE_MIX") ? atoi(getenv("OSP_PROBE_MIX")) : 0;
    float* hog_out = nullptr;
    if (mix_spins > 0) CK(hipMalloc(&hog_out, 1 << 20));
    for (size_t c = 0; c < cfgs.size(); ++c) {
        const int N = cfgs[c].N, hop = cfgs[c].hop;
        size_t n_out;
        if (N) { if (osp_fft_twiddles(tw, N, st) != 0) { printf("twiddles failed\n"); return 2; } n_out = (size_t)B * (1 + T / hop) * (N / 2 + 1); }
        else n_out = (size_t)B * T;
        long long bi = 0, be = 0;
        for (int it = 0; it <= iters; ++it) {
            float* dst = it == 0 ? ref : out;
            // OSP_PROBE_MIX=<spins>: this process is HEAVY too -- a long LDS-resident kernel on every CU before each victim launch
            // (profiles/r03_shared_gpu_race_matrix.txt: the deviation needs BOTH processes to be heavy)
            if (mix_spins > 0 && it > 0) hipLaunchKernelGGL(hog_kernel, dim3(1024), dim3(256), hog_lds(), st, hog_out, mix_spins);
            if (N) { if (osp_stft_mag_fwd(x, nullptr, tw, -1.f, dst, B, T, N, hop, st) != 0) { printf("launch failed: %s\n", osp_last_error()); return 2; } }
            else if (!strcmp(variant, "ldsmix")) hipLaunchKernelGGL(ldsmix_kernel, dim3(B * T / 1024), dim3(256), 0, st, x, dst, B * T);
            else hipLaunchKernelGGL(nolds_kernel, dim3(B * T / 256), dim3(256), 0, st, x, dst, B * T);
            if (it == 0) continue;
            CK(hipMemsetAsync(cnt, 0, 8, st));
            hipLaunchKernelGGL(mismatch_kernel, dim3(512), dim3(256), 0, st, (const unsigned*)out, (const unsigned*)ref, n_out, cnt);
            if (it % 8 == 0 || it == iters) {                 // keep a few launches queued (like the training step does)
                unsigned long long h = 0;
                CK(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                if (h) { ++bi; be += (long long)h; if (bi <= 3) printf("  %s N=%d iteration %d: %llu of %zu elements differ from the first run\n", variant, N, it, h, n_out); }
            }
        }
        printf("%s N=%d: %lld of %d checked iterations deviate (%lld elements)\n", variant, N, bi, iters / 8, be);
        bad_iters += bi; bad_elems += be;
    }
    printf("RESULT %s pid %d: deviating iterations %lld, elements %lld\n", variant, (int)getpid(), bad_iters, bad_elems);
    return 0;
}
