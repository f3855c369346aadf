// libosp_hip: error reporting + version.  All entry points are extern "C", return int (0 = ok),
// never throw, never allocate, never synchronise: work is enqueued on the caller's hipStream_t.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void osp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* osp_last_error() { return g_err; }

// ---- measurement aid (bench.py's roofline block): which matrix-core kernel did the last entry-point call of this thread
// launch, and how many algorithmic flops (2 * M * taps * Cin * N, summed over the launches of the call) did it stand for?  The
// dispatchers know both; mirroring their selection rules in Python went stale once already (the bench bracketed
// 35 of the 47 launches of a symbol).  Two thread-local words, written by the launch helpers, cost nothing on the hot path.
#include <stdint.h>
#include <string.h>
static thread_local const char* g_note_sym = nullptr;
static thread_local double g_note_flops = 0.0;
static thread_local double g_note_bytes = 0.0;
void osp_note_symbol(const char* sym) { g_note_sym = sym; }
void osp_note_flops(double flops) { g_note_flops += flops; }
void osp_note_bytes(double bytes) { g_note_bytes += bytes; }
// name_host: buffer of `cap` bytes for the symbol ("" when the call launched no noted kernel); flops_host: one double.
// The note is cleared by the read, so call it right after the entry point it asks about.
extern "C" int osp_kernel_note_host(char* name_host, int64_t cap, double* flops_host) {
    if (name_host && cap > 0) {
        const char* sname = g_note_sym ? g_note_sym : "";
        strncpy(name_host, sname, (size_t)cap - 1);
        name_host[cap - 1] = 0;
    }
    if (flops_host) *flops_host = g_note_flops;
    g_note_sym = nullptr;
    g_note_flops = 0.0;
    return 0;
}
// ALGORITHMIC HBM bytes of the same launches (every operand and the output once: unique input rows x channels, the weights, the
// output tile, the epilogue's extra operands) -- what decides whether a layer is bound by the matrix pipe or by HBM: a 5-tap
// 32 -> 128 convolution (K = 160) moves ~100 KB per 5 MFLOP tile, i.e. cannot exceed ~420 TFLOP/s at 8 TB/s whatever the kernel does.
// Read (and cleared) separately so that older callers of osp_kernel_note_host keep working.
extern "C" int osp_kernel_note_bytes_host(double* bytes_host) {
    if (bytes_host) *bytes_host = g_note_bytes;
    g_note_bytes = 0.0;
    return 0;
}
// ABI history: 1 = rounds 1-2; 2 = round 3 added trailing workspace parameters to osp_layernorm_bwd / osp_ln_dwconv7_bwd /
// osp_smallcin_conv_wgrad (a consumer built against the version-1 header would pass its stream in the workspace slot) and round 4
// added osp_memset / osp_copy / osp_store_i64 / the fused block entry points.  Bump whenever an EXISTING signature changes.
extern "C" int osp_abi_version() { return 2; }

// Content hash of the sources (and flags) this library was compiled from: optispeech_amd/build.py passes it on the command line
// of this file and compares it with the sources next to a shipped library (the marker prefix makes it findable without dlopen).
#ifndef OSP_SOURCE_HASH
#define OSP_SOURCE_HASH "0000000000000000000000000000000000000000"
#endif
static const char g_source_hash[] = "OSP_SOURCE_HASH=" OSP_SOURCE_HASH;
extern "C" const char* osp_source_hash() { return g_source_hash + 16; }

// Stream hand-over in one call: record `event_host_handle` (a hipEvent_t) on `stream` and make `dst_stream_host_handle` (a
// hipStream_t) wait for it.  The step does this ~45 times (weight-gradient side streams, ops.side_wgrad); through torch it is two
// Python calls + two runtime calls each.
extern "C" int osp_stream_handover(void* event_host_handle, void* dst_stream_host_handle, hipStream_t stream) {
    hipError_t e = hipEventRecord(reinterpret_cast<hipEvent_t>(event_host_handle), stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(reinterpret_cast<hipStream_t>(dst_stream_host_handle), reinterpret_cast<hipEvent_t>(event_host_handle), 0);
    if (e != hipSuccess) { osp_set_error("osp_stream_handover: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

// ---- plumbing entry points of the call tapes (optispeech_amd/tape.py): a recorded region of the step may contain nothing but
// C-ABI calls, so the fills and copies torch would launch in between have their own entry points.
extern "C" int osp_memset(void* dst, int64_t value, int64_t nbytes, hipStream_t stream) {
    if (!dst || nbytes < 0) { osp_set_error("osp_memset: bad args"); return -1; }
    if (nbytes == 0) return 0;
    hipError_t e = hipMemsetAsync(dst, (int)(value & 0xff), (size_t)nbytes, stream);
    if (e != hipSuccess) { osp_set_error("osp_memset: %s", hipGetErrorString(e)); return -2; }
    return 0;
}
extern "C" int osp_copy(void* dst, const void* src, int64_t nbytes, hipStream_t stream) {
    if (!dst || !src || nbytes < 0) { osp_set_error("osp_copy: bad args"); return -1; }
    if (nbytes == 0) return 0;
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) { osp_set_error("osp_copy: %s", hipGetErrorString(e)); return -2; }
    return 0;
}
// Test aid of the tape machinery: adds `add` into *counter ON THE HOST, synchronously (no device work, no GPU needed) -- lets the
// CPU test suite record, patch and replay tapes.  `counter` is a host address in this one entry point.
extern "C" int osp_tape_selftest(int64_t* counter, int64_t add, hipStream_t stream) {
    if (!counter) { osp_set_error("osp_tape_selftest: null counter"); return -1; }
    *counter += add + (int64_t)(intptr_t)stream;
    return 0;
}
// Second test aid: vals_host[i] are HOST addresses of int64 values; their sum is added into *counter (host).  Exercises the tape's
// copy of a host descriptor table and the patching of addresses inside it.
extern "C" int osp_tape_selftest_table(const int64_t* vals_host, int64_t count, int64_t* counter, hipStream_t stream) {
    if (!vals_host || !counter || count < 0) { osp_set_error("osp_tape_selftest_table: bad args"); return -1; }
    int64_t acc = 0;
    for (int64_t i = 0; i < count; ++i) acc += *reinterpret_cast<const int64_t*>(static_cast<intptr_t>(vals_host[i]));
    *counter += acc;
    return 0;
}
