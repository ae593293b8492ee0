// comm.hip -- the one exchange step of the sharded swarm, inside the C-ABI: RCCL over xGMI.
//
// The reference's only parallel mode hands the particles of a generation to
// multiprocessing.Pool.map (nmrfit/utils.py:182, `processes=self.processes` -> pyswarm).  Here
// the swarm axis is sharded over one process per GPU and the only thing that crosses ranks is
// the (D+1)-double candidate record [f_best, x_best[D]] of each rank, once per generation:
// one ncclAllGather on the context's stream, followed by the deterministic fold kernel
// (pso.hip).  616 B per rank at D = 76: latency-bound, the xGMI links are idle.
//
// librccl is opened lazily (dlopen) the first time a communicator is asked for, so single-GPU
// users never load it and the library has no link-time dependency on it.  No PyTorch anywhere:
// the 128-byte unique id travels between the ranks however the caller likes (the Python side
// uses stdlib sockets, nmrfit_amd/rendezvous.py).
#include "nmrfit_internal.h"
#include "nmrfit_amd_diag.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>

struct nmrfit_comm {
    nmrfit_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int32_t rank = 0, nranks = 1;
    double *d_scratch = nullptr;       // host-value collectives: [kScratch] send + [kScratch * nranks] recv
    int64_t scratch_cap = 0;           // doubles
    double *d_gather = nullptr;        // candidate all-gather: nranks x (D+1), grown on demand
    int64_t gather_cap = 0;
    std::atomic<int> attached{0};      // swarms that point at this communicator (nmrfit_pso_set_comm): at most ONE --
                                       // the all-gathers of two swarms, issued from two host threads on two streams in
                                       // whatever order the threads run, would pair up differently on different ranks
};

namespace nmrfit {
namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

Rccl g_rccl;

int load_rccl()
{
    // (contexts may be driven from different host threads: the first communicators of two of them may race here)
    static std::mutex load_lock;
    std::lock_guard<std::mutex> guard(load_lock);
    if (g_rccl.handle) return NMRFIT_OK;
    // NMRFIT_RCCL_LIB: a specific RCCL build (or, in tests, a missing one) instead of the usual names
    const char *forced = getenv("NMRFIT_RCCL_LIB");
    const char *names[] = {forced ? forced : "librccl.so.1", forced ? nullptr : "librccl.so",
                           forced ? nullptr : "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    std::string tried;
    for (const char *n : names) {
        if (!n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        const char *e = dlerror();
        tried += std::string(n) + ": " + (e ? e : "?") + "; ";
    }
    if (!h) {
        set_error("cannot load RCCL (" + tried + ")");
        return NMRFIT_E_UNSUPPORTED;
    }
    Rccl r;
    r.handle = h;
#define SYM(field, name)                                                          \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));               \
    if (!r.field) {                                                               \
        set_error(std::string("RCCL symbol missing: ") + name);                  \
        dlclose(h);                                                               \
        return NMRFIT_E_UNSUPPORTED;                                              \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommAbort, "ncclCommAbort")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
    SYM(Broadcast, "ncclBroadcast")
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(GetVersion, "ncclGetVersion")
#undef SYM
    g_rccl = r;
    return NMRFIT_OK;
}

int rccl_fail(ncclResult_t r, const char *what, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "RCCL error %d (%s) in `%s` at comm.hip:%d", (int)r,
             g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what, line);
    set_error(buf);
    return NMRFIT_E_COMM;
}

#define NMRFIT_RCCL(call)                                                   \
    do {                                                                    \
        ncclResult_t _r = (call);                                           \
        if (_r != ncclSuccess) return rccl_fail(_r, #call, __LINE__);       \
    } while (0)

constexpr int64_t kScratch = 64;   // doubles per rank for the host-value collectives

int bind_comm(const nmrfit_comm *c)
{
    if (!c || !c->ctx || !c->comm) {
        set_error("null communicator");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipSetDevice(c->ctx->device));
    return NMRFIT_OK;
}

}  // namespace

nmrfit_ctx *comm_ctx(const nmrfit_comm *c) { return c ? c->ctx : nullptr; }
bool comm_attach(nmrfit_comm *c)
{
    int none = 0;
    return c && c->attached.compare_exchange_strong(none, 1);   // (fit_many's host threads may race for one communicator)
}
void comm_detach(nmrfit_comm *c)
{
    if (c) c->attached.store(0);
}

// used by pso.hip: gather every rank's (D+1)-double record on `stream` -- the stream of the swarm's context, which
// need not be the context the communicator was created on (same device: one communicator serves fit after fit, each
// with a context of its own, without another ncclCommInitRank).  One swarm at a time per communicator.
// d_send may be any device buffer; *d_all receives a pointer to nranks x n doubles in rank order.
int comm_all_gather(nmrfit_comm *c, hipStream_t stream, const double *d_send, int64_t n, const double **d_all)
{
    int rc = bind_comm(c);
    if (rc != NMRFIT_OK) return rc;
    if (c->gather_cap < n * c->nranks) {
        NMRFIT_HIP(hipDeviceSynchronize());   // (rare: whoever used the old buffer, on whichever stream, is done)
        if (c->d_gather) NMRFIT_HIP(hipFree(c->d_gather));
        c->d_gather = nullptr;
        c->gather_cap = 0;
        NMRFIT_HIP(hipMalloc((void **)&c->d_gather, (size_t)(n * c->nranks) * sizeof(double)));
        c->gather_cap = n * c->nranks;
    }
    NMRFIT_RCCL(g_rccl.AllGather(d_send, c->d_gather, (size_t)n, ncclDouble, c->comm, stream));
    *d_all = c->d_gather;
    return NMRFIT_OK;
}

}  // namespace nmrfit

using namespace nmrfit;

#pragma GCC visibility push(default)   // the C-ABI: the only symbols the library exports (build.sh: -fvisibility=hidden)
extern "C" {

int nmrfit_comm_available(void)
{
    // dlopen + every symbol the library uses; no communicator, no GPU work.  Lets every rank find
    // out -- and tell the others -- BEFORE anyone enters the collective ncclCommInitRank.
    return load_rccl();
}

int nmrfit_comm_unique_id(void *out128)
{
    if (!out128) {
        set_error("null id buffer");
        return NMRFIT_E_INVALID;
    }
    int rc = load_rccl();
    if (rc != NMRFIT_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == NMRFIT_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    NMRFIT_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
    return NMRFIT_OK;
}

int nmrfit_comm_create(nmrfit_ctx *ctx, int32_t rank, int32_t nranks, const void *unique_id128, nmrfit_comm **out)
{
    if (!out) {
        set_error("null out pointer");
        return NMRFIT_E_INVALID;
    }
    *out = nullptr;
    if (!ctx || !unique_id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("nmrfit_comm_create: bad arguments");
        return NMRFIT_E_INVALID;
    }
    int rc = load_rccl();
    if (rc != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipSetDevice(ctx->device));
    nmrfit_comm *c = new (std::nothrow) nmrfit_comm();
    if (!c) {
        set_error("out of host memory");
        return NMRFIT_E_INVALID;
    }
    c->ctx = ctx;
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);   // collective: every rank calls it
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail(r, "ncclCommInitRank", __LINE__);
    }
    c->scratch_cap = kScratch * (1 + nranks);
    hipError_t e = hipMalloc((void **)&c->d_scratch, (size_t)c->scratch_cap * sizeof(double));
    if (e != hipSuccess) {
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return hip_fail(e, "hipMalloc(comm scratch)", __FILE__, __LINE__);
    }
    *out = c;
    return NMRFIT_OK;
}

int nmrfit_comm_destroy(nmrfit_comm *c)
{
    if (!c) return NMRFIT_OK;
    if (c->attached.load() > 0) {
        set_error("nmrfit_comm_destroy: a swarm still uses this communicator (nmrfit_pso_set_comm(pso, NULL) or "
                  "nmrfit_pso_destroy first)");
        return NMRFIT_E_STATE;
    }
    if (c->ctx) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
    }
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_gather) (void)hipFree(c->d_gather);
    delete c;
    return NMRFIT_OK;
}

int nmrfit_comm_info(const nmrfit_comm *c, int32_t *rank, int32_t *nranks, int32_t *rccl_version)
{
    if (!c) {
        set_error("null communicator");
        return NMRFIT_E_INVALID;
    }
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (rccl_version) {
        int v = 0;
        if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
        *rccl_version = v;
    }
    return NMRFIT_OK;
}

int nmrfit_comm_describe(const nmrfit_comm *c, char *buf, int len)
{
    if (!c || !c->ctx || !buf || len < 1) {
        set_error("nmrfit_comm_describe: null communicator or buffer");
        return NMRFIT_E_INVALID;
    }
    char pci[64] = "?";
    (void)hipDeviceGetPCIBusId(pci, (int)sizeof pci, c->ctx->device);
    int v = 0;
    if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
    snprintf(buf, (size_t)len, "rank %d of %d, HIP device %d, PCI %s, RCCL %d", (int)c->rank, (int)c->nranks,
             c->ctx->device, pci, v);
    return NMRFIT_OK;
}

int nmrfit_comm_all_gather_dev(nmrfit_comm *c, const double *d_send, double *d_recv, int64_t n)
{
    int rc = bind_comm(c);
    if (rc != NMRFIT_OK) return rc;
    if (n < 0 || (n > 0 && (!d_send || !d_recv))) {
        set_error("nmrfit_comm_all_gather_dev: bad arguments");
        return NMRFIT_E_INVALID;
    }
    if (n == 0) return NMRFIT_OK;
    NMRFIT_RCCL(g_rccl.AllGather(d_send, d_recv, (size_t)n, ncclDouble, c->comm, c->ctx->stream));
    return NMRFIT_OK;
}

// Host-value collectives for the bookkeeping around a run (timing maxima, the seed, barriers):
// staged through a small device buffer on the context's stream; synchronous.
int nmrfit_comm_all_reduce_host(nmrfit_comm *c, double *inout, int32_t n, int32_t op)
{
    int rc = bind_comm(c);
    if (rc != NMRFIT_OK) return rc;
    if (!inout || n < 1 || n > kScratch || op < 0 || op > 2) {
        set_error("nmrfit_comm_all_reduce_host: n must be 1..64 and op 0 (sum), 1 (max) or 2 (min)");
        return NMRFIT_E_INVALID;
    }
    hipStream_t st = c->ctx->stream;
    const ncclRedOp_t ops[] = {ncclSum, ncclMax, ncclMin};
    NMRFIT_HIP(hipMemcpyAsync(c->d_scratch, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    NMRFIT_RCCL(g_rccl.AllReduce(c->d_scratch, c->d_scratch + kScratch, (size_t)n, ncclDouble, ops[op], c->comm, st));
    NMRFIT_HIP(hipMemcpyAsync(inout, c->d_scratch + kScratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    NMRFIT_HIP(hipStreamSynchronize(st));
    return NMRFIT_OK;
}

int nmrfit_comm_broadcast_host(nmrfit_comm *c, void *buf, int64_t bytes, int32_t root)
{
    int rc = bind_comm(c);
    if (rc != NMRFIT_OK) return rc;
    if (!buf || bytes < 1 || bytes > kScratch * (int64_t)sizeof(double) || root < 0 || root >= c->nranks) {
        set_error("nmrfit_comm_broadcast_host: 1..512 bytes, root within the communicator");
        return NMRFIT_E_INVALID;
    }
    hipStream_t st = c->ctx->stream;
    NMRFIT_HIP(hipMemcpyAsync(c->d_scratch, buf, (size_t)bytes, hipMemcpyHostToDevice, st));
    NMRFIT_RCCL(g_rccl.Broadcast(c->d_scratch, c->d_scratch, (size_t)bytes, ncclChar, root, c->comm, st));
    NMRFIT_HIP(hipMemcpyAsync(buf, c->d_scratch, (size_t)bytes, hipMemcpyDeviceToHost, st));
    NMRFIT_HIP(hipStreamSynchronize(st));
    return NMRFIT_OK;
}

int nmrfit_comm_barrier(nmrfit_comm *c)
{
    double one = 1.0;
    return nmrfit_comm_all_reduce_host(c, &one, 1, 0);
}

}  // extern "C"
#pragma GCC visibility pop
