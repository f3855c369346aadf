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
