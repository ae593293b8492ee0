// cabi.hip -- the extern "C" surface of libnmrfit_amd.so (include/nmrfit_amd.h): context
// life-cycle, host-pointer and device-pointer forms of the objective / residual calls,
// device memory and HIP-event timing helpers.  No exception leaves this file.
#include "nmrfit_internal.h"
#include "nmrfit_amd_diag.h"
#include "result_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

namespace nmrfit {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_last_error = buf;
    return NMRFIT_E_HIP;
}

int ensure(nmrfit_ctx *ctx, double **buf, int64_t *cap, int64_t need)
{
    if (need <= *cap) return NMRFIT_OK;
    // the old buffer may still be in use by work enqueued on the stream
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    if (*buf) NMRFIT_HIP(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    int64_t n = need + need / 4 + 64;
    NMRFIT_HIP(hipMalloc((void **)buf, (size_t)n * sizeof(double)));
    *cap = n;
    return NMRFIT_OK;
}

// A context's own stream, recycled: a new HIP stream costs about 4 ms on this stack when it is first used (the hardware
// queue behind it is made then; measured, tools/archive/hip_call_costs.hip) and a whole default fit is 25 ms -- a fit
// that pyswarm's rule stops after a few hundred generations 3 ms.  Contexts that come and go (one per fitted spectrum)
// hand their idle stream to the next one on the same device instead.  NMRFIT_NO_STREAM_CACHE=1 turns it off.
namespace {
struct StreamPool {
    std::mutex lock;
    std::vector<std::pair<int, hipStream_t>> idle;
};
StreamPool &stream_pool()
{
    static StreamPool *pool = new StreamPool;   // (never destroyed: no order-of-destruction trouble at process exit)
    return *pool;
}
constexpr size_t kMaxIdleStreams = 16;
bool stream_cache_on()
{
    static const bool on = getenv("NMRFIT_NO_STREAM_CACHE") == nullptr;
    return on;
}
}  // namespace

// What the kernels need to know about a grid besides its values: the centring offset, the span, and -- for the Gaussian
// recurrence -- whether it is uniformly spaced (np.linspace grids, ascending or descending) and how exactly.
void analyse_grid(const double *w, int64_t N, double *w0_out, double *wspan_out, double *lane_step_out, double *grid_dev_out)
{
    const double w0 = w[N / 2];
    double wspan = 0.0, lane_step = 0.0, grid_dev = 0.0;
    for (int64_t j = 0; j < N; ++j) wspan = std::fmax(wspan, std::fabs(w[j] - w0));
    // Measured on the centred values the kernel sees: deviation of every point from the straight line through the ends.
    if (N >= 2 * kChunk) {
        const double first = w[0] - w0, step = ((w[N - 1] - w0) - first) / (double)(N - 1);
        double dev = 0.0;
        for (int64_t j = 0; j < N; ++j) dev = std::fmax(dev, std::fabs((w[j] - w0) - (first + (double)j * step)));
        if (step != 0.0 && dev <= 1.0e-6 * std::fabs(step)) {   // NaN fails the test
            lane_step = step * kWave;
            grid_dev = 2.0 * dev;
        }
    }
    *w0_out = w0;
    *wspan_out = wspan;
    *lane_step_out = lane_step;
    *grid_dev_out = grid_dev;
}

hipError_t take_stream(int device, hipStream_t *out)
{
    if (stream_cache_on()) {
        StreamPool &pool = stream_pool();
        std::lock_guard<std::mutex> guard(pool.lock);
        for (size_t i = 0; i < pool.idle.size(); ++i)
            if (pool.idle[i].first == device) {
                *out = pool.idle[i].second;
                pool.idle.erase(pool.idle.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// (the caller has synchronised the stream: nothing is queued on it)
void give_stream(int device, hipStream_t s)
{
    if (stream_cache_on()) {
        StreamPool &pool = stream_pool();
        std::lock_guard<std::mutex> guard(pool.lock);
        if (pool.idle.size() < kMaxIdleStreams) {
            pool.idle.emplace_back(device, s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

// Large results to PAGEABLE host memory (numpy arrays the caller has just made): a plain hipMemcpy pins the destination's
// pages on the fly, 13-26 ms for the 30 MB a batch of 50 reconstructed fits returns, 1 ms when the runtime happens to
// know the pages (tools/generate_breakdown.py; profiles/r06/generate_breakdown.txt).  Here the copy goes through two
// pinned buffers the process keeps per device: the DMA engine fills one while the CPU copies the other out, so the call
// costs what the CPU copy into the caller's pages costs (~3 ms for 30 MB) whatever the runtime's pinning cache holds.
// Synchronous: the data is in `dst` on return.  Small copies take the plain path.
namespace {
struct HostStage {
    std::mutex lock;                 // one staged copy at a time per device (the buffers are the resource)
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
constexpr size_t kStageChunk = (size_t)4 << 20;
constexpr size_t kStageMin = (size_t)256 << 10;
HostStage *host_stage(int device)
{
    static std::mutex table_lock;
    static std::vector<HostStage *> table;   // (never destroyed: pinned memory outlives every context, freed at process exit)
    std::lock_guard<std::mutex> guard(table_lock);
    if ((int)table.size() <= device) table.resize((size_t)device + 1, nullptr);
    if (!table[(size_t)device]) table[(size_t)device] = new HostStage;
    return table[(size_t)device];
}
}  // namespace

int staged_d2h(int device, hipStream_t st, void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return NMRFIT_OK;
    static const bool off = getenv("NMRFIT_NO_STAGED_COPIES") != nullptr;   // A/B knob
    if (bytes < kStageMin || off) {
        NMRFIT_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        NMRFIT_HIP(hipStreamSynchronize(st));
        return NMRFIT_OK;
    }
    HostStage *hs = host_stage(device);
    std::lock_guard<std::mutex> guard(hs->lock);
    for (int i = 0; i < 2; ++i)
        if (!hs->buf[i]) {
            NMRFIT_HIP(hipHostMalloc(&hs->buf[i], kStageChunk, hipHostMallocDefault));
            NMRFIT_HIP(hipEventCreateWithFlags(&hs->ev[i], hipEventDisableTiming));
        }
    const unsigned char *src = static_cast<const unsigned char *>(src_dev);
    unsigned char *dst = static_cast<unsigned char *>(dst_host);
    const size_t n_chunks = (bytes + kStageChunk - 1) / kStageChunk;
    for (size_t i = 0; i <= n_chunks; ++i) {
        if (i < n_chunks) {   // (buffer i % 2 was copied out two rounds ago, before chunk i - 1 was waited for)
            const size_t off_i = i * kStageChunk, n = std::min(kStageChunk, bytes - off_i);
            NMRFIT_HIP(hipMemcpyAsync(hs->buf[i & 1], src + off_i, n, hipMemcpyDeviceToHost, st));
            NMRFIT_HIP(hipEventRecord(hs->ev[i & 1], st));
        }
        if (i > 0) {
            const size_t off_p = (i - 1) * kStageChunk, n = std::min(kStageChunk, bytes - off_p);
            NMRFIT_HIP(hipEventSynchronize(hs->ev[(i - 1) & 1]));
            memcpy(dst + off_p, hs->buf[(i - 1) & 1], n);
        }
    }
    return NMRFIT_OK;
}

// (hipGetDeviceProperties costs about a millisecond: once per device and PROCESS -- a default fit is 30 ms, and
// fit_many's worker threads come and go)
int device_info_cached(int device, DeviceInfo *out)
{
    static std::mutex dev_lock;
    static std::vector<DeviceInfo> dev_cache;
    std::lock_guard<std::mutex> guard(dev_lock);
    if ((int)dev_cache.size() <= device) dev_cache.resize((size_t)device + 1);
    if (!dev_cache[(size_t)device].known) {
        hipDeviceProp_t hp;
        NMRFIT_HIP(hipGetDeviceProperties(&hp, device));
        dev_cache[(size_t)device].cus = hp.multiProcessorCount;
        strncpy(dev_cache[(size_t)device].arch, hp.gcnArchName, sizeof(dev_cache[0].arch) - 1);
        dev_cache[(size_t)device].known = true;
    }
    *out = dev_cache[(size_t)device];
    return NMRFIT_OK;
}

static int bind(const nmrfit_ctx *ctx)
{
    if (!ctx) {
        set_error("null context");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipSetDevice(ctx->device));
    return NMRFIT_OK;
}

static int check_batch(const nmrfit_ctx *ctx, int64_t S, int32_t P, const void *X, const void *out)
{
    if (S < 0 || P < 0) {
        set_error("negative batch size or peak count");
        return NMRFIT_E_INVALID;
    }
    if (P > kMaxPeaks) {
        set_error("P exceeds the supported maximum of " + std::to_string(kMaxPeaks) + " peaks (per-peak records of a workgroup live in a CU's 160 KiB of LDS)");
        return NMRFIT_E_INVALID;
    }
    if (S > 0 && (!X || !out)) {
        set_error("null parameter/output pointer");
        return NMRFIT_E_INVALID;
    }
    (void)ctx;
    return NMRFIT_OK;
}

}  // namespace nmrfit

using namespace nmrfit;

#pragma GCC visibility push(default)   // the C-ABI: the only symbols the library exports (build.sh: -fvisibility=hidden)
extern "C" {

int nmrfit_abi_version(void) { return NMRFIT_ABI_VERSION; }

int nmrfit_diag_ab_build(void)
{
#ifdef NMRFIT_AB_BUILD
    return 1;
#else
    return 0;
#endif
}

const char *nmrfit_last_error(void) { return g_last_error.c_str(); }

int nmrfit_device_count(int *count)
{
    if (!count) {
        set_error("null count pointer");
        return NMRFIT_E_INVALID;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        set_error(std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
        return NMRFIT_E_NO_DEVICE;
    }
    *count = n;
    return NMRFIT_OK;
}

int nmrfit_device_info(int device, char *name, int name_len, int *compute_units, char *arch, int arch_len)
{
    int n = 0;
    int rc = nmrfit_device_count(&n);
    if (rc != NMRFIT_OK) return rc;
    if (device < 0 || device >= n) {
        set_error("device index out of range");
        return NMRFIT_E_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    NMRFIT_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        strncpy(name, prop.name, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return NMRFIT_OK;
}

int nmrfit_device_pci_bus_id(int device, char *buf, int len)
{
    int n = 0;
    int rc = nmrfit_device_count(&n);
    if (rc != NMRFIT_OK) return rc;
    if (device < 0 || device >= n) {
        set_error("device index out of range");
        return NMRFIT_E_NO_DEVICE;
    }
    if (!buf || len < 16) {
        set_error("nmrfit_device_pci_bus_id: buffer of at least 16 bytes");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipDeviceGetPCIBusId(buf, len, device));
    return NMRFIT_OK;
}

int nmrfit_ctx_create(int device, int64_t N, const double *w, const double *u, const double *v,
                      const double *weights, nmrfit_ctx **out)
{
    if (!out) {
        set_error("null out pointer");
        return NMRFIT_E_INVALID;
    }
    *out = nullptr;
    if (N <= 0 || !w || !u || !v || !weights) {
        set_error("nmrfit_ctx_create: N must be > 0 and w, u, v, weights non-null");
        return NMRFIT_E_INVALID;
    }
    int n = 0;
    int rc = nmrfit_device_count(&n);
    if (rc != NMRFIT_OK) return rc;
    if (n == 0) {
        set_error("no HIP device visible: libnmrfit_amd has no CPU fallback");
        return NMRFIT_E_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device index out of range");
        return NMRFIT_E_NO_DEVICE;
    }
    NMRFIT_HIP(hipSetDevice(device));
    DeviceInfo prop;
    if ((rc = device_info_cached(device, &prop)) != NMRFIT_OK) return rc;
    if (strncmp(prop.arch, "gfx950", 6) != 0) {
        set_error(std::string("device is ") + prop.arch + ", this library is built for gfx950 only");
        return NMRFIT_E_NO_DEVICE;
    }
    nmrfit_ctx *ctx = new (std::nothrow) nmrfit_ctx();
    if (!ctx) {
        set_error("out of host memory");
        return NMRFIT_E_INVALID;
    }
    ctx->device = device;
    ctx->compute_units = prop.cus;
    ctx->N = N;
    ctx->n_chunks = (N + kChunk - 1) / kChunk;
    if (const char *tw = getenv("NMRFIT_TARGET_WAVES")) ctx->target_waves = atoll(tw);   // tuning knob
    if (getenv("NMRFIT_NO_WIDE_WORKGROUPS")) ctx->wide_workgroups = false;               // A/B knob
    // test knob: run a whole test suite with another kernel variant as every context's default
    if (const char *dv = getenv("NMRFIT_DEFAULT_VARIANT")) {
        const int vnum = atoi(dv);
#ifdef NMRFIT_AB_BUILD
        if (vnum >= 0 && vnum <= NMRFIT_VARIANT_FARFIELD32) ctx->variant = vnum;
#else
        if (vnum == NMRFIT_VARIANT_DEFAULT || vnum == NMRFIT_VARIANT_FARFIELD || vnum == NMRFIT_VARIANT_NOREC ||
            vnum == NMRFIT_VARIANT_FARFIELD32)
            ctx->variant = vnum;
#endif
    }
    analyse_grid(w, N, &ctx->w0, &ctx->wspan, &ctx->lane_step, &ctx->grid_dev);
    const size_t bytes = (size_t)N * sizeof(double);
    const size_t padded = (size_t)ctx->n_chunks * kChunk * sizeof(double);   // whole chunks, grid_slot order
    double *d_w_raw = nullptr;
#define CTX_HIP(call)                                                              \
    do {                                                                           \
        hipError_t _e = (call);                                                    \
        if (_e != hipSuccess) {                                                    \
            int _rc = hip_fail(_e, #call, __FILE__, __LINE__);                     \
            nmrfit_ctx_destroy(ctx);                                               \
            return _rc;                                                            \
        }                                                                          \
    } while (0)
    CTX_HIP(take_stream(device, &ctx->own_stream));
    ctx->stream = ctx->own_stream;
    CTX_HIP(hipEventCreate(&ctx->ev0));
    CTX_HIP(hipEventCreate(&ctx->ev1));
    // ONE allocation for the four padded grid arrays, the chunk table and the landing buffer (which first holds the raw w)
    const size_t chunk_bytes = ((size_t)ctx->n_chunks * sizeof(double2) + 255) & ~(size_t)255;
    const size_t padded_al = (padded + 255) & ~(size_t)255;
    CTX_HIP(hipMalloc((void **)&ctx->d_block, 4 * padded_al + chunk_bytes + bytes));
    CTX_HIP(hipMemsetAsync(ctx->d_block, 0, 4 * padded_al, ctx->stream));
    {
        unsigned char *base = reinterpret_cast<unsigned char *>(ctx->d_block);
        ctx->d_wc = reinterpret_cast<double *>(base);
        ctx->d_u = reinterpret_cast<double *>(base + padded_al);
        ctx->d_v = reinterpret_cast<double *>(base + 2 * padded_al);
        ctx->d_wt = reinterpret_cast<double *>(base + 3 * padded_al);
        ctx->d_chunk = reinterpret_cast<double2 *>(base + 4 * padded_al);
        ctx->d_stage = reinterpret_cast<double *>(base + 4 * padded_al + chunk_bytes);
    }
    d_w_raw = ctx->d_stage;
    CTX_HIP(hipMemcpyAsync(d_w_raw, w, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = prepare_grid(ctx, d_w_raw);
    // u, v, weights: land in plain order, then into the pair-interleaved order the kernels read (nmrfit_internal.h,
    // grid_slot); stream order lets the one landing buffer serve all three
    const double *host_arrays[] = {u, v, weights};
    double *dev_arrays[] = {ctx->d_u, ctx->d_v, ctx->d_wt};
    for (int a = 0; a < 3 && rc == NMRFIT_OK; ++a) {
        CTX_HIP(hipMemcpyAsync(ctx->d_stage, host_arrays[a], bytes, hipMemcpyHostToDevice, ctx->stream));
        rc = scatter_grid(ctx, ctx->d_stage, dev_arrays[a]);
    }
    if (rc != NMRFIT_OK) {
        nmrfit_ctx_destroy(ctx);
        return rc;
    }
    CTX_HIP(hipStreamSynchronize(ctx->stream));
#undef CTX_HIP
    *out = ctx;
    return NMRFIT_OK;
}

int nmrfit_ctx_destroy(nmrfit_ctx *ctx)
{
    if (!ctx) return NMRFIT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    }
    void *bufs[] = {ctx->d_block /* wc, u, v, weights, chunk table, landing buffer */, ctx->d_X, ctx->d_f, ctx->d_partial, ctx->d_R};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    (void)nmrfit_prof_enable(ctx, 0);
    if (ctx->own_stream) give_stream(ctx->device, ctx->own_stream);   // (synchronised above)
    delete ctx;
    return NMRFIT_OK;
}

int nmrfit_ctx_set_weights(nmrfit_ctx *ctx, const double *weights)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (!weights) {
        set_error("null weights");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipMemcpyAsync(ctx->d_stage, weights, (size_t)ctx->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = scatter_grid(ctx, ctx->d_stage, ctx->d_wt);
    if (rc != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    return NMRFIT_OK;
}

int nmrfit_ctx_synchronize(nmrfit_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    return NMRFIT_OK;
}

int nmrfit_ctx_set_stream(nmrfit_ctx *ctx, void *hip_stream)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));   // drain work queued on the old stream
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return NMRFIT_OK;
}

int nmrfit_ctx_set_variant(nmrfit_ctx *ctx, int variant)
{
    if (!ctx || variant < 0 || variant > NMRFIT_VARIANT_FARFIELD32) {
        set_error("bad context or variant");
        return NMRFIT_E_INVALID;
    }
#ifndef NMRFIT_AB_BUILD
    if (variant != NMRFIT_VARIANT_DEFAULT && variant != NMRFIT_VARIANT_FARFIELD && variant != NMRFIT_VARIANT_NOREC &&
        variant != NMRFIT_VARIANT_FARFIELD32) {
        set_error("this kernel variant is an A/B form: it exists in libnmrfit_amd_ab.so (nmrfit_amd/csrc/build.sh --ab) only");
        return NMRFIT_E_UNSUPPORTED;
    }
#endif
    ctx->variant = variant;
    return NMRFIT_OK;
}

int nmrfit_ctx_n(const nmrfit_ctx *ctx, int64_t *N)
{
    if (!ctx || !N) {
        set_error("null argument");
        return NMRFIT_E_INVALID;
    }
    *N = ctx->N;
    return NMRFIT_OK;
}

int nmrfit_objective_batch_dev(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *dX, double *df_out)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    rc = check_batch(ctx, S, P, dX, df_out);
    if (rc != NMRFIT_OK) return rc;
    return launch_objective(ctx, S, P, dX, df_out, nullptr);
}

int nmrfit_residual_batch_dev(nmrfit_ctx *ctx, int64_t B, int32_t P, const double *dX, double *dR_out, double *df_out)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    rc = check_batch(ctx, B, P, dX, dR_out);
    if (rc != NMRFIT_OK) return rc;
    if (B == 0) return NMRFIT_OK;
    if (!df_out) {
        rc = ensure(ctx, &ctx->d_f, &ctx->cap_f, B);
        if (rc != NMRFIT_OK) return rc;
        df_out = ctx->d_f;
    }
    return launch_objective(ctx, B, P, dX, df_out, dR_out);
}

int nmrfit_ctx_set_fit_im(nmrfit_ctx *ctx, int fit_im)
{
    if (!ctx || fit_im < 0 || fit_im > NMRFIT_FIT_IM_SUM) {
        set_error("fit_im must be 0 (real part), 1 (reference fit_im=True) or 2 (all-peak imaginary model)");
        return NMRFIT_E_INVALID;
    }
    ctx->fit_im = fit_im;
    return NMRFIT_OK;
}

int nmrfit_objective_batch(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *X, int fit_im, double *f_out)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (fit_im < 0 || fit_im > NMRFIT_FIT_IM_SUM) {
        set_error("fit_im must be 0 (real part), 1 (reference fit_im=True) or 2 (all-peak imaginary model)");
        return NMRFIT_E_INVALID;
    }
    rc = check_batch(ctx, S, P, X, f_out);
    if (rc != NMRFIT_OK) return rc;
    if (S == 0) return NMRFIT_OK;
    const int64_t D = 4 + 3 * (int64_t)P;
    if ((rc = ensure(ctx, &ctx->d_X, &ctx->cap_X, S * D)) != NMRFIT_OK) return rc;
    if ((rc = ensure(ctx, &ctx->d_f, &ctx->cap_f, S)) != NMRFIT_OK) return rc;
    const int saved = ctx->fit_im;
    ctx->fit_im = fit_im;
    // (Round 6, measured and rejected: the upload cut into slices through pinned memory, the kernel of a slice starting as
    // soon as its rows have landed -- 1.52 ms per C3 call against 1.34 ms for this plain form, resident launch 1.20: four
    // kernels each drain on their own, which costs more than the 0.1 ms of upload they hide;
    // profiles/r06/host_pointer_pipelined_ab.txt.)
    const int64_t x_bytes = S * D * (int64_t)sizeof(double);
    {
        hipError_t e = hipMemcpyAsync(ctx->d_X, X, (size_t)x_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            ctx->fit_im = saved;
            return hip_fail(e, "hipMemcpyAsync(X)", __FILE__, __LINE__);
        }
    }
    rc = launch_objective(ctx, S, P, ctx->d_X, ctx->d_f, nullptr);
    ctx->fit_im = saved;
    if (rc != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipMemcpyAsync(f_out, ctx->d_f, (size_t)S * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    return NMRFIT_OK;
}

// The reconstruction of ONE fit through the kernel of result.hip: scratch device memory for the parameter vector, the
// optional output grid and the outputs, one launch, the copies back.  (A device batch does the same for all its fits in
// one launch from its resident state: nmrfit_batch_contributions, batch.hip.)
static int generate_one(nmrfit_ctx *ctx, int32_t P, const double *x, int64_t Nout, const double *w_out, double *real_out,
                        double *imag_out, double *fit_out, double *data_out, const char *who)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (!w_out) Nout = ctx->N;
    if (P < 0 || P > kMaxPeaks || !x || Nout < 0 || (P > 0 && Nout > 0 && (!real_out != !imag_out))) {
        set_error(std::string(who) + ": bad arguments");
        return NMRFIT_E_INVALID;
    }
    const int64_t N = ctx->N;
    const int64_t n_contrib = real_out ? (int64_t)P * Nout : 0;
    const int64_t n_fit = fit_out ? 4 * Nout : 0, n_data = data_out ? 2 * N : 0;
    const int64_t n_out = 2 * n_contrib + n_fit + n_data;
    if (n_out == 0) return NMRFIT_OK;
    const int64_t D = 4 + 3 * (int64_t)P;
    double *d_in = nullptr, *d_out = nullptr;
    hipStream_t st = ctx->stream;
#define CB_HIP(call)                                              \
    do {                                                          \
        hipError_t _e = (call);                                   \
        if (_e != hipSuccess) {                                   \
            rc = hip_fail(_e, #call, __FILE__, __LINE__);         \
            goto done;                                            \
        }                                                         \
    } while (0)
    CB_HIP(hipMalloc((void **)&d_in, (size_t)(D + (w_out ? Nout : 0)) * sizeof(double)));
    CB_HIP(hipMalloc((void **)&d_out, (size_t)n_out * sizeof(double)));
    CB_HIP(hipMemcpyAsync(d_in, x, (size_t)D * sizeof(double), hipMemcpyHostToDevice, st));
    if (w_out) CB_HIP(hipMemcpyAsync(d_in + D, w_out, (size_t)Nout * sizeof(double), hipMemcpyHostToDevice, st));
    {
        ResultJob job{};
        job.wc = ctx->d_wc;
        job.w_plain = w_out ? d_in + D : nullptr;   // (centred in the kernel with the context's offset: it works on w - w0)
        job.x = d_in;
        job.u = ctx->d_u;
        job.v = ctx->d_v;
        job.w0 = ctx->w0;
        job.wspan = ctx->wspan;
        job.Nout = Nout;
        job.N = N;
        job.P = P;
        job.real = real_out ? d_out : nullptr;
        job.imag = real_out ? d_out + n_contrib : nullptr;
        job.fit = fit_out ? d_out + 2 * n_contrib : nullptr;
        job.data = data_out ? d_out + 2 * n_contrib + n_fit : nullptr;
        if ((rc = launch_result_one(st, job)) != NMRFIT_OK) goto done;
        if (n_contrib) {
            CB_HIP(hipMemcpyAsync(real_out, job.real, (size_t)n_contrib * sizeof(double), hipMemcpyDeviceToHost, st));
            CB_HIP(hipMemcpyAsync(imag_out, job.imag, (size_t)n_contrib * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        if (n_fit) CB_HIP(hipMemcpyAsync(fit_out, job.fit, (size_t)n_fit * sizeof(double), hipMemcpyDeviceToHost, st));
        if (n_data) CB_HIP(hipMemcpyAsync(data_out, job.data, (size_t)n_data * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    CB_HIP(hipStreamSynchronize(st));
#undef CB_HIP
done:
    if (rc != NMRFIT_OK) (void)hipStreamSynchronize(st);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

int nmrfit_contributions(nmrfit_ctx *ctx, int32_t P, const double *x, int64_t Nout, const double *w_out,
                         double *real_out, double *imag_out)
{
    if (P > 0 && (w_out ? Nout > 0 : true) && (!real_out || !imag_out)) {
        set_error("nmrfit_contributions: bad arguments");
        return NMRFIT_E_INVALID;
    }
    return generate_one(ctx, P, x, Nout, w_out, real_out, imag_out, nullptr, nullptr, "nmrfit_contributions");
}

int nmrfit_generate_result(nmrfit_ctx *ctx, int32_t P, const double *x, int64_t Nout, const double *w_out,
                           double *real_out, double *imag_out, double *fit_out, double *data_out)
{
    return generate_one(ctx, P, x, Nout, w_out, real_out, imag_out, fit_out, data_out, "nmrfit_generate_result");
}

int nmrfit_residual_batch(nmrfit_ctx *ctx, int64_t B, int32_t P, const double *X, double *R_out, double *f_out)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    rc = check_batch(ctx, B, P, X, R_out);
    if (rc != NMRFIT_OK) return rc;
    if (B == 0) return NMRFIT_OK;
    const int64_t D = 4 + 3 * (int64_t)P;
    if ((rc = ensure(ctx, &ctx->d_X, &ctx->cap_X, B * D)) != NMRFIT_OK) return rc;
    if ((rc = ensure(ctx, &ctx->d_f, &ctx->cap_f, B)) != NMRFIT_OK) return rc;
    if ((rc = ensure(ctx, &ctx->d_R, &ctx->cap_R, B * ctx->N)) != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipMemcpyAsync(ctx->d_X, X, (size_t)(B * D) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_objective(ctx, B, P, ctx->d_X, ctx->d_f, ctx->d_R)) != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipMemcpyAsync(R_out, ctx->d_R, (size_t)(B * ctx->N) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (f_out)
        NMRFIT_HIP(hipMemcpyAsync(f_out, ctx->d_f, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    return NMRFIT_OK;
}

int nmrfit_dev_alloc(nmrfit_ctx *ctx, int64_t bytes, void **dptr)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (!dptr || bytes < 0) {
        set_error("bad arguments to nmrfit_dev_alloc");
        return NMRFIT_E_INVALID;
    }
    *dptr = nullptr;
    if (bytes == 0) return NMRFIT_OK;
    NMRFIT_HIP(hipMalloc(dptr, (size_t)bytes));
    return NMRFIT_OK;
}

int nmrfit_dev_free(nmrfit_ctx *ctx, void *dptr)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (!dptr) return NMRFIT_OK;
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    NMRFIT_HIP(hipFree(dptr));
    return NMRFIT_OK;
}

int nmrfit_memcpy_h2d(nmrfit_ctx *ctx, void *dst_dev, const void *src_host, int64_t bytes)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (bytes < 0 || (bytes > 0 && (!dst_dev || !src_host))) {
        set_error("bad arguments to nmrfit_memcpy_h2d");
        return NMRFIT_E_INVALID;
    }
    if (bytes == 0) return NMRFIT_OK;
    NMRFIT_HIP(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    return NMRFIT_OK;
}

int nmrfit_memcpy_d2h(nmrfit_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (bytes < 0 || (bytes > 0 && (!dst_host || !src_dev))) {
        set_error("bad arguments to nmrfit_memcpy_d2h");
        return NMRFIT_E_INVALID;
    }
    if (bytes == 0) return NMRFIT_OK;
    NMRFIT_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    return NMRFIT_OK;
}

int nmrfit_timer_begin(nmrfit_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return NMRFIT_OK;
}

int nmrfit_timer_end(nmrfit_ctx *ctx, double *elapsed_ms)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (!elapsed_ms) {
        set_error("null elapsed_ms");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NMRFIT_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    NMRFIT_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return NMRFIT_OK;
}

int nmrfit_prof_enable(nmrfit_ctx *ctx, int64_t capacity)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (capacity < 0 || capacity > (1 << 20)) {
        set_error("nmrfit_prof_enable: capacity must be 0..2^20");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    for (auto *vec : {&ctx->prof_k0, &ctx->prof_k1, &ctx->prof_marks}) {
        for (hipEvent_t e : *vec) (void)hipEventDestroy(e);
        vec->clear();
    }
    ctx->prof_cap = 0;
    ctx->prof_nk = ctx->prof_nm = 0;
    if (capacity == 0) {
        if (ctx->d_clk) (void)hipFree(ctx->d_clk);
        ctx->d_clk = nullptr;
        return NMRFIT_OK;
    }
    if (!ctx->d_clk) {
#ifdef NMRFIT_DIAG_STAMPS   // diagnostic builds: room for 16 phase stamps of up to 1024 workgroups behind the clock ticks
        constexpr size_t kClkWords = 4 + 16 * 1024;
#else
        constexpr size_t kClkWords = 4;
#endif
        NMRFIT_HIP(hipMalloc((void **)&ctx->d_clk, kClkWords * sizeof(unsigned long long)));
        NMRFIT_HIP(hipMemsetAsync(ctx->d_clk, 0, kClkWords * sizeof(unsigned long long), ctx->stream));
    }
    for (int64_t i = 0; i < capacity; ++i) {
        hipEvent_t a = nullptr, b = nullptr, c = nullptr;
        NMRFIT_HIP(hipEventCreate(&a));
        ctx->prof_k0.push_back(a);
        NMRFIT_HIP(hipEventCreate(&b));
        ctx->prof_k1.push_back(b);
        NMRFIT_HIP(hipEventCreate(&c));
        ctx->prof_marks.push_back(c);
    }
    // one more mark than steps: n steps are bracketed by n + 1 marks
    hipEvent_t last = nullptr;
    NMRFIT_HIP(hipEventCreate(&last));
    ctx->prof_marks.push_back(last);
    ctx->prof_cap = capacity;
    return NMRFIT_OK;
}

int nmrfit_prof_mark(nmrfit_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (ctx->prof_cap == 0) {
        set_error("nmrfit_prof_mark before nmrfit_prof_enable");
        return NMRFIT_E_STATE;
    }
    if (ctx->prof_nm > ctx->prof_cap) return NMRFIT_OK;   // full: later marks are dropped
    NMRFIT_HIP(hipEventRecord(ctx->prof_marks[(size_t)ctx->prof_nm], ctx->stream));
    ++ctx->prof_nm;
    return NMRFIT_OK;
}

int nmrfit_prof_read(nmrfit_ctx *ctx, double *kernel_ms, int64_t kernel_cap, int64_t *n_kernel, double *step_ms,
                     int64_t step_cap, int64_t *n_step, double *clock_mhz)
{
    int rc = bind(ctx);
    if (rc != NMRFIT_OK) return rc;
    if (kernel_cap < 0 || step_cap < 0 || (kernel_cap > 0 && !kernel_ms) || (step_cap > 0 && !step_ms)) {
        set_error("nmrfit_prof_read: bad arguments");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t nk = std::min<int64_t>(ctx->prof_nk, kernel_cap);
    for (int64_t i = 0; i < nk; ++i) {
        float ms = 0.f;
        NMRFIT_HIP(hipEventElapsedTime(&ms, ctx->prof_k0[(size_t)i], ctx->prof_k1[(size_t)i]));
        kernel_ms[i] = (double)ms;
    }
    const int64_t ns = std::min<int64_t>(std::max<int64_t>(ctx->prof_nm - 1, 0), step_cap);
    for (int64_t i = 0; i < ns; ++i) {
        float ms = 0.f;
        NMRFIT_HIP(hipEventElapsedTime(&ms, ctx->prof_marks[(size_t)i], ctx->prof_marks[(size_t)i + 1]));
        step_ms[i] = (double)ms;
    }
    if (n_kernel) *n_kernel = nk;
    if (n_step) *n_step = ns;
    if (clock_mhz) {
        *clock_mhz = 0.0;
        if (ctx->d_clk && ctx->prof_nk > 0) {
            unsigned long long t[4] = {0, 0, 0, 0};
            NMRFIT_HIP(hipMemcpy(t, ctx->d_clk, sizeof t, hipMemcpyDeviceToHost));
            if (t[3] > t[1] && t[2] > t[0]) *clock_mhz = 100.0 * (double)(t[2] - t[0]) / (double)(t[3] - t[1]);
        }
    }
    ctx->prof_nk = ctx->prof_nm = 0;   // reading rewinds
    return NMRFIT_OK;
}

int nmrfit_last_launch(const nmrfit_ctx *ctx, int64_t *waves, int32_t *segments, int64_t *segment_len)
{
    if (!ctx) {
        set_error("null context");
        return NMRFIT_E_INVALID;
    }
    if (waves) *waves = ctx->last.waves;
    if (segments) *segments = ctx->last.nseg;
    if (segment_len) *segment_len = ctx->last.seg_len;
    return NMRFIT_OK;
}

int nmrfit_last_launch_workgroup(const nmrfit_ctx *ctx, int32_t *waves_per_workgroup)
{
    if (!ctx) {
        set_error("null context");
        return NMRFIT_E_INVALID;
    }
    if (waves_per_workgroup) *waves_per_workgroup = ctx->last.waves_per_workgroup;
    return NMRFIT_OK;
}

#ifdef NMRFIT_DIAG_STAMPS
// diagnostic builds only (not in the header): the phase stamps of the last profiled launch, [workgroup][16]
int nmrfit_diag_read_stamps(nmrfit_ctx *ctx, unsigned long long *out, int64_t workgroups)
{
    if (!ctx || !ctx->d_clk || workgroups > 1024) return NMRFIT_E_INVALID;
    NMRFIT_HIP(hipStreamSynchronize(ctx->stream));
    NMRFIT_HIP(hipMemcpy(out, ctx->d_clk + 4, (size_t)workgroups * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return NMRFIT_OK;
}
#endif

}  // extern "C"
#pragma GCC visibility pop
