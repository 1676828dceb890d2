// hb_comm.hip — the multi-GPU collective of the marker-sharded sweep (SURVEY.md §8 e), inside the library: one
// ncclAllReduce (RCCL over xGMI) of the residual delta and the sweep's scalar sums, enqueued on the sweep's own HIP stream —
// no host synchronisation, no callback into the host language. The reference has no distributed path; from the R shim this
// is what makes a sharded run possible at all.
// librccl is opened lazily (dlopen) the first time a communicator is asked for, so single-GPU users never load it.
#include "hb_internal.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>

namespace {
struct rccl_api {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
rccl_api g_rccl;

int load_rccl()
{
    if (g_rccl.h) return HB_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return hb_fail(HB_ERR_COMM, std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"));
    auto sym = [&](const char *n) { return dlsym(h, n); };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(sym("ncclCommCount"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return hb_fail(HB_ERR_COMM, "librccl lacks the NCCL entry points");
    g_rccl.h = h;
    return HB_OK;
}

std::string nccl_msg(const char *what, ncclResult_t r)
{
    return std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}
} // namespace

struct hb_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

int hb_comm_allreduce_f64(hb_comm *c, double *buf, size_t count, hipStream_t st);

extern "C" {

int hb_comm_unique_id(void *id128)
{
    if (!id128) return hb_fail(HB_ERR_INVALID, "hb_comm_unique_id: null argument");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return hb_fail(HB_ERR_COMM, nccl_msg("ncclGetUniqueId", r));
    static_assert(sizeof(ncclUniqueId) == HB_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id128, &id, sizeof(id));
    return HB_OK;
}

int hb_comm_init(hb_comm **out, const void *id128, int32_t rank, int32_t world, int32_t device)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return hb_fail(HB_ERR_INVALID, "hb_comm_init: bad argument");
    *out = nullptr;
    int rc = load_rccl();
    if (rc) return rc;
    HB_HIP(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    hb_comm *c = new hb_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return hb_fail(HB_ERR_COMM, nccl_msg("ncclCommInitRank", r));
    }
    *out = c;
    return HB_OK;
}

int hb_comm_world(const hb_comm *c)
{
    if (!c) return 0;
    int n = c->world;
    if (g_rccl.CommCount) (void)g_rccl.CommCount(c->comm, &n); // what RCCL itself says
    return n;
}

int hb_comm_rank(const hb_comm *c) { return c ? c->rank : -1; }

// One small all-reduce with a known answer (every rank contributes rank + 1 in 16 doubles), synchronised: a caller can
// run it under its own deadline right after hb_comm_init() and fall back to another collective if the fabric does not
// deliver, instead of finding out inside the first sweep.
int hb_comm_selftest(hb_comm *c)
{
    if (!c || !c->comm) return hb_fail(HB_ERR_COMM, "hb_comm_selftest: no communicator");
    HB_HIP(hipSetDevice(c->device));
    double h[16], *d = nullptr;
    for (double &v : h) v = (double)(c->rank + 1);
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)));
    hipStream_t st = nullptr;
    HB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HB_HIP(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, st));
    int rc = hb_comm_allreduce_f64(c, d, 16, st);
    if (!rc) {
        HB_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
        HB_HIP(hipStreamSynchronize(st));
        const double want = 0.5 * c->world * (c->world + 1);
        for (double v : h)
            if (v != want) rc = hb_fail(HB_ERR_COMM, "hb_comm_selftest: the all-reduce returned a wrong sum");
    }
    (void)hipStreamDestroy(st);
    (void)hipFree(d);
    return rc;
}

void hb_comm_destroy(hb_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

} // extern "C"

// in-place sum over ranks of `count` doubles in device memory, enqueued on `st` (no host synchronisation)
int hb_comm_allreduce_f64(hb_comm *c, double *buf, size_t count, hipStream_t st)
{
    if (!c || !c->comm) return hb_fail(HB_ERR_COMM, "hb_comm_allreduce_f64: no communicator");
    ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, st);
    if (r != ncclSuccess) return hb_fail(HB_ERR_COMM, nccl_msg("ncclAllReduce", r));
    return HB_OK;
}
