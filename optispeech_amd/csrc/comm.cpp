// Data-parallel gradient exchange behind the C ABI (SURVEY.md section 8b "DP" entries; reference: Lightning's DDP strategy,
// configs/trainer/ddp.yaml:4-9, gradient reducers at base_lightning_module.py:99,119): one RCCL communicator per process
// (one process per GPU), sum-all-reduce of flat f32 gradient-arena slices over xGMI on the caller's stream.
//
//   osp_comm_unique_id(out_host[128])         rank 0: the id every rank passes to osp_comm_init (ship it any way you like)
//   osp_comm_init(rank, world, id_host)       collective; binds the communicator to the CURRENT device
//   osp_allreduce_bucket(ptr, n, stream)      in-place f32 sum over ranks of n elements, enqueued on `stream` (no host sync)
//   osp_comm_destroy()
//
// RCCL is bound at run time (dlopen "librccl.so"), so libosp_hip.so itself has no link-time dependency on it and single-GPU
// users never load it.  The only global state of the library is this communicator handle.
#include "osp_common.h"
#include <dlfcn.h>

namespace {
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;                               // ncclSuccess = 0
enum { kNcclFloat32 = 7, kNcclSum = 0 };                // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0;
int g_device = -1;                                      // the device the communicator was bound to (osp_comm_init)

int load_rccl() {
    if (g_rccl.lib) return OSP_OK;
    // resolve into a local table and publish it only when every symbol is there: a half-filled g_rccl with lib != null would
    // let the next osp_comm_* call through this guard and into a null function pointer
    Rccl t;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    char tried[512] = "";
    for (const char* n : names) {
        t.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (t.lib) break;
        const char* e = dlerror();
        size_t used = strlen(tried);
        snprintf(tried + used, sizeof(tried) - used, "%s%s: %s", used ? "; " : "", n, e ? e : "?");
    }
    if (!t.lib) { osp_set_error("osp_comm: cannot load librccl (%s)", tried); return OSP_ERR_UNSUPPORTED; }
#define SYM(field, name)                                                                    \
    t.field = reinterpret_cast<decltype(t.field)>(dlsym(t.lib, name));                      \
    if (!t.field) { osp_set_error("osp_comm: librccl.so lacks %s", name); dlclose(t.lib); return OSP_ERR_UNSUPPORTED; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(AllReduce, "ncclAllReduce")
    SYM(CommDestroy, "ncclCommDestroy") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = t;
    return OSP_OK;
}
int check(ncclResult_t r, const char* what) {
    if (r == 0) return OSP_OK;
    osp_set_error("osp_comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return OSP_ERR_HIP;
}
}  // namespace

extern "C" int osp_comm_unique_id(void* id_host) {
    OSP_CHECK_ARG(id_host, "null id buffer (128 bytes of host memory)");
    int rc = load_rccl();
    if (rc != OSP_OK) return rc;
    return check(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_host)), "ncclGetUniqueId");
}

extern "C" int osp_comm_init(int64_t rank, int64_t world, const void* id_host) {
    OSP_CHECK_ARG(id_host && world > 0 && rank >= 0 && rank < world, "bad rank / world / id");
    OSP_CHECK_ARG(!g_comm, "communicator already initialised (osp_comm_destroy first)");
    int rc = load_rccl();
    if (rc != OSP_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    rc = check(g_rccl.CommInitRank(&g_comm, (int)world, id, (int)rank), "ncclCommInitRank");
    if (rc == OSP_OK) { g_world = (int)world; (void)hipGetDevice(&g_device); } else g_comm = nullptr;
    return rc;
}

extern "C" int64_t osp_comm_world() { return g_comm ? g_world : 0; }

extern "C" int osp_allreduce_bucket(float* ptr, int64_t n, hipStream_t stream) {
    OSP_CHECK_ARG(ptr && n > 0, "bad bucket");
    OSP_CHECK_ARG(g_comm, "osp_comm_init has not been called");
    int dev = -1;
    (void)hipGetDevice(&dev);
    OSP_CHECK_ARG(dev == g_device, "the current device is not the one osp_comm_init bound the communicator to");
    return check(g_rccl.AllReduce(ptr, ptr, (size_t)n, kNcclFloat32, kNcclSum, g_comm, stream), "ncclAllReduce");
}

extern "C" int osp_comm_destroy() {
    if (!g_comm) return OSP_OK;
    const int rc = check(g_rccl.CommDestroy(g_comm), "ncclCommDestroy");
    g_comm = nullptr; g_world = 0; g_device = -1;
    return rc;
}
