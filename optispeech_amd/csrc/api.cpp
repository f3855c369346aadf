// libosp_hip: error reporting + version.  All entry points are extern "C", return int (0 = ok),
// never throw, never allocate, never synchronise: work is enqueued on the caller's hipStream_t.
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
extern "C" int osp_abi_version() { return 1; }

// Content hash of the sources (and flags) this library was compiled from: optispeech_amd/build.py passes it on the command line
// of this file and compares it with the sources next to a shipped library (the marker prefix makes it findable without dlopen).
#ifndef OSP_SOURCE_HASH
#define OSP_SOURCE_HASH "0000000000000000000000000000000000000000"
#endif
static const char g_source_hash[] = "OSP_SOURCE_HASH=" OSP_SOURCE_HASH;
extern "C" const char* osp_source_hash() { return g_source_hash + 16; }
