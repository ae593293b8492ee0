// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY.  A stand-in for librccl that lets SEVERAL ranks share ONE GPU.
//
// RCCL refuses two ranks on the same device, and the boxes this project is built on have one GPU, so the
// multi-rank branch of the product's own code (csrc/comm.hip, nmrfit_pso_step with a communicator attached,
// bench.py's N > 1 RCCL branch, fit(options={"exchange": "rccl"}) with world > 1) could never run with more
// than one rank on hardware.  This library implements the nine RCCL entry points libnmrfit_amd.so dlopens
// (csrc/comm.hip load_rccl) with the same signatures and the same semantics as seen from a HIP stream --
// the collective is complete when the call's work on the stream is complete -- but moves the data through
// host memory shared between the processes (a file in /dev/shm named by the unique id):
//     hipStreamSynchronize -> hipMemcpy D2H into this rank's slot -> barrier -> hipMemcpy H2D -> barrier.
// It is selected ONLY by the tests, through NMRFIT_RCCL_LIB (tests/test_gpu_fake_rccl.py); it reports
// version 99999 so that nothing measured with it can be mistaken for RCCL, and bench.py prints the library
// override in its `rccl` object.  It exercises OUR code on the multi-rank path; it says nothing about RCCL.
//
// Build: hipcc -shared -fPIC -O2 tests/fake_rccl/fake_rccl.cpp -o <dir>/libfake_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr size_t kSlot = 1 << 16;     // bytes per rank per collective (the product sends <= 24 KiB)
constexpr int kMaxRanks = 64;
constexpr double kTimeoutS = 60.0;

struct Header {
    std::atomic<int> ready;           // rank 0 has initialised the segment
    std::atomic<int> arrived;         // barrier: ranks that have arrived in the current phase
    std::atomic<int> phase;           // barrier: phase counter
    std::atomic<int> attached;        // ranks that have mapped the segment (the last one unlinks it)
    int nranks;
};

struct FakeComm {
    int rank = 0, nranks = 1;
    Header *hdr = nullptr;
    unsigned char *slots = nullptr;   // nranks x kSlot
    size_t bytes = 0;
    char name[160];
};

double now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

bool barrier(FakeComm *c)
{
    Header *h = c->hdr;
    const int phase = h->phase.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) == c->nranks - 1) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->phase.store(phase + 1, std::memory_order_release);
        return true;
    }
    const double t0 = now();
    while (h->phase.load(std::memory_order_acquire) == phase) {
        if (now() - t0 > kTimeoutS) return false;     // a rank never arrived: an error, never a hang
        usleep(20);
    }
    return true;
}

size_t dtype_size(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

// this rank's contribution -> its slot; every slot is complete when this returns true
bool publish(FakeComm *c, const void *dev, size_t bytes, hipStream_t st)
{
    if (bytes > kSlot) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;     // everything the stream produced is there
    if (bytes && hipMemcpy(c->slots + (size_t)c->rank * kSlot, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    return barrier(c);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version)
{
    if (version) *version = 99999;     // not an RCCL version: anything measured with this is labelled
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "fake_rccl_%d_%lld", (int)getpid(), (long long)(now() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    // test hook: FAKE_RCCL_HANG_IN_INIT=<rank> makes that rank never come back from the collective creation
    // (what a wedged ncclCommInitRank looks like from outside): the callers' watchdogs have to deal with it
    if (const char *h = getenv("FAKE_RCCL_HANG_IN_INIT"))
        if (atoi(h) == rank)
            for (;;) sleep(1);
    FakeComm *c = new (std::nothrow) FakeComm();
    if (!c) return ncclSystemError;
    c->rank = rank;
    c->nranks = nranks;
    c->bytes = sizeof(Header) + (size_t)nranks * kSlot;
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(c->name, sizeof c->name, "/%s", id.internal);
    int fd = -1;
    const double t0 = now();
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
            delete c;
            return ncclSystemError;
        }
    } else {
        while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0) {      // rank 0 has not created it yet
            if (now() - t0 > kTimeoutS) {
                delete c;
                return ncclSystemError;
            }
            usleep(200);
        }
        struct stat sb;
        while (fstat(fd, &sb) == 0 && (size_t)sb.st_size < c->bytes) {
            if (now() - t0 > kTimeoutS) {
                close(fd);
                delete c;
                return ncclSystemError;
            }
            usleep(200);
        }
    }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->hdr = static_cast<Header *>(p);
    c->slots = static_cast<unsigned char *>(p) + sizeof(Header);
    if (rank == 0) {
        c->hdr->arrived.store(0);
        c->hdr->phase.store(0);
        c->hdr->attached.store(0);
        c->hdr->nranks = nranks;
        c->hdr->ready.store(1, std::memory_order_release);
    } else {
        while (c->hdr->ready.load(std::memory_order_acquire) != 1) {
            if (now() - t0 > kTimeoutS) {
                munmap(p, c->bytes);
                delete c;
                return ncclSystemError;
            }
            usleep(200);
        }
    }
    c->hdr->attached.fetch_add(1);
    if (!barrier(c)) {       // collective, like the real one: returns when every rank is in
        munmap(p, c->bytes);
        delete c;
        return ncclSystemError;
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c) return ncclSuccess;
    if (c->hdr->attached.fetch_sub(1) == 1) shm_unlink(c->name);     // the last rank out removes the segment
    munmap(c->hdr, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t bytes = count * dtype_size(dt);
    if (!c || !bytes) return ncclInvalidArgument;
    if (!publish(c, send, bytes, st)) return ncclSystemError;
    for (int r = 0; r < c->nranks; ++r)
        if (hipMemcpy((unsigned char *)recv + (size_t)r * bytes, c->slots + (size_t)r * kSlot, bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;     // nobody's slot is overwritten before all have read it
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t st)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c || dt != ncclFloat64 || count * 8 > kSlot) return ncclInvalidArgument;     // (the product reduces doubles only)
    if (!publish(c, send, count * 8, st)) return ncclSystemError;
    double out[kSlot / 8];
    for (size_t i = 0; i < count; ++i) {
        double a = reinterpret_cast<const double *>(c->slots)[i];
        for (int r = 1; r < c->nranks; ++r) {      // rank order: the same answer on every rank
            const double b = reinterpret_cast<const double *>(c->slots + (size_t)r * kSlot)[i];
            a = (op == ncclSum) ? a + b : (op == ncclMax) ? (b > a ? b : a) : (op == ncclMin) ? (b < a ? b : a) : a * b;
        }
        out[i] = a;
    }
    if (hipMemcpy(recv, out, count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t st)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t bytes = count * dtype_size(dt);
    if (!c || !bytes || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (!publish(c, send, bytes, st)) return ncclSystemError;
    if (hipMemcpy(recv, c->slots + (size_t)root * kSlot, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;
}

}  // extern "C"
