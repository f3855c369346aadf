// What a non-Python host links against for the data-parallel gradient exchange: the four osp_comm_* / osp_allreduce_bucket entry
// points of include/osp.h, nothing from torch.  tests/test_abi.py compiles and links this file against libosp_hip.so on every
// run; tests/test_gpu_dp.py executes it on the GPU box (one rank: the all-reduce must return the buffer unchanged; with N ranks
// each process passes its own rank / the shared id and gets the sum).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "osp.h"

int main(int argc, char** argv) {
    const int64_t rank = argc > 1 ? atoll(argv[1]) : 0, world = argc > 2 ? atoll(argv[2]) : 1;
    char id[128];
    if (rank == 0 && osp_comm_unique_id(id) != OSP_OK) { fprintf(stderr, "unique_id: %s\n", osp_last_error()); return 2; }
    // (N ranks: ship `id` from rank 0 to the others here -- a file, a socket, MPI_Bcast ...)
    if (osp_comm_init(rank, world, id) != OSP_OK) { fprintf(stderr, "init: %s\n", osp_last_error()); return 3; }
    const int64_t n = 1 << 20;
    std::vector<float> h(n);
    for (int64_t i = 0; i < n; ++i) h[i] = (float)(i % 97) * 0.25f;
    float* d = nullptr;
    hipStream_t s;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess || hipStreamCreate(&s) != hipSuccess) return 4;
    hipMemcpyAsync(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice, s);
    if (osp_allreduce_bucket(d, n, s) != OSP_OK) { fprintf(stderr, "allreduce: %s\n", osp_last_error()); return 5; }
    std::vector<float> out(n);
    hipMemcpyAsync(out.data(), d, n * sizeof(float), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    int bad = 0;
    for (int64_t i = 0; i < n; ++i) bad += out[i] != h[i] * (float)world;      // identical inputs on every rank -> world * x
    osp_comm_destroy();
    hipFree(d);
    printf("osp_comm example: world %lld, %d mismatches\n", (long long)world, bad);
    return bad ? 1 : 0;
}
