// batch.hip -- device-batched fits: K independent nmrfit fits advanced together, ONE kernel launch per swarm generation
// for all of them (the nmrfit_batch_* entry points of include/nmrfit_amd.h).
//
// The reference's users call `nmrfit.fit` once per spectrum (nmrfit/core.py:64, README.md:64-66), each with a
// 204-particle swarm (nmrfit/utils.py:177).  A lone swarm of that size occupies a fraction of an MI355X and its
// generation is one wave's critical path (11.7 us for ~0.9 us of throughput-bound work); host threads driving one
// context each saturate at ~95 fits/s.  Here the K spectra are prepared into ONE device allocation, the K swarms live
// next to them, and a generation of every swarm is one launch of the objective kernel whose workgroups find their fit's
// record (BatchFit) from blockIdx (objective_batch.hip).  Each fit keeps its own (g, fg), stop flags and random stream
// (Philox keyed by its seed), runs exactly the operations a lone `nmrfit_amd.fit` runs in the same order, and stops by
// its own pyswarm rule: the K results are bit-identical to K lone fits.
//
// State machine (the deferred fold of pso.hip, for K swarms in lock step): after generation 0 every launch moves, evaluates
// and personal-bests every particle and leaves its generation to be folded by the NEXT launch's prologue; x / v and
// (p, fp) ping-pong every launch, the (g, fg | flags) blocks whenever a launch folded.  The kernel reads which buffer is
// which from one of 8 + 1 descriptor tables built once at creation (phase of x / p, phase of the state block, fold
// pending or not; + plain evaluation), so a generation costs the host one launch and no copies.
#include "batch_internal.h"
#include "nmrfit_amd_diag.h"
#include "result_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

struct BatchPart {
    int device = -1;
    int compute_units = 0;
    hipStream_t stream = nullptr;
    int32_t K = 0;
    int64_t N = 0;                       // the fits' common grid length, or 0 when they differ (ragged: wave = particle form only)
    int64_t S = 0, n_chunks = 0;         // particles per fit, or 0 when the swarms differ in size (wave = particle form only); chunks of the LONGEST grid
    std::vector<int64_t> Sk;             // per fit: swarm size
    int64_t Smax = 0, Ssum = 0;
    std::vector<int64_t> Nk, noff;       // per fit: grid length; offset of its first point in the concatenated arrays (+ total)
    int64_t Nmax = 0;
    int variant = NMRFIT_VARIANT_DEFAULT;
    int fit_im = NMRFIT_FIT_IM_OFF;
    int32_t Pmax = 0;
    std::vector<int32_t> P;
    std::vector<int64_t> D, boff;        // per fit: 4 + 3P, offset of its bounds / best row in the concatenated arrays
    int64_t Dsum = 0;
    void *d_block = nullptr;             // the one allocation behind everything below
    nmrfit::BatchFit *d_tables = nullptr;   // [9][K]: t = xp + 2 b + 4 pending (fused generations), 8 = plain evaluation
    double *d_summary = nullptr;         // [K][4]: generations, stop code, fg, best_f   (written by batch_tail_kernel)
    double *d_bestx = nullptr;           // [Dsum]: best_x rows, concatenated
    std::vector<nmrfit::BatchFit> h_fits;   // host copy of table 0: every fit's array pointers and grid constants
    int64_t Psum = 0;
    // scratch of a reconstruction call in flight (nmrfit_batch_contributions): device block, and what goes where on the host
    void *d_result = nullptr;
    struct ResultCopy {
        void *host;
        const void *dev;
        size_t bytes;
    };
    std::vector<ResultCopy> result_copies;
    // launch geometry: [0] workgroup = particle, [1] wave = particle
    nmrfit::BatchLaunch geom[2];
    bool geom_ok[2] = {false, false};
    int mode = 0;                        // 0 workgroup form, 1 wave form (chosen at creation; nmrfit_batch_set_geometry)
    // phases
    int xp = 0, b = 0;
    bool fold_pending = false;
    bool initialized = false;
    int64_t launches = 0;
};

namespace nmrfit {
namespace {

#pragma clang fp contract(off)

struct PrepareArgs {
    int64_t plane;           // doubles per uploaded array (the sum of the fits' lengths)
    const double *raw;       // [4][plane]: w, u, v, weights as uploaded, fit after fit
    const BatchFit *fits;    // any table: wc, u, v, wt, chunk, w0, N, raw_off
};

// centred grid + scatter of the four arrays into the pair-interleaved order the kernels read (nmrfit_internal.h,
// grid_slot), for every fit of the batch at once (blockIdx.y = the fit); the padding up to whole chunks was zeroed before
__global__ void batch_prepare_kernel(PrepareArgs a)
{
    const BatchFit &f = a.fits[blockIdx.y];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= f.N) return;
    const int64_t idx = f.raw_off + j;
    const int64_t slot = grid_slot(j);
    const_cast<double *>(f.wc)[slot] = a.raw[idx] - f.w0;
    const_cast<double *>(f.u)[slot] = a.raw[a.plane + idx];
    const_cast<double *>(f.v)[slot] = a.raw[2 * a.plane + idx];
    const_cast<double *>(f.wt)[slot] = a.raw[3 * a.plane + idx];
}

// per-chunk (min, max) of every fit's centred grid: one wave per (fit, chunk)
__global__ void batch_chunk_minmax_kernel(PrepareArgs a)
{
    const BatchFit &f = a.fits[blockIdx.y];
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t c = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    if (c * kChunk >= f.N) return;
    double lo = INFINITY, hi = -INFINITY;
    for (int q = 0; q < kPointsPerLane; ++q) {
        const int64_t j = c * kChunk + q * kWave + lane;
        if (j < f.N) {
            const double x = f.wc[grid_slot(j)];
            lo = fmin(lo, x);
            hi = fmax(hi, x);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off, kWave));
        hi = fmax(hi, __shfl_down(hi, off, kWave));
    }
    if (lane == 0) const_cast<double2 *>(f.chunk)[c] = make_double2(lo, hi);
}

// generation 0 of every swarm: x ~ U(lb, ub), v ~ U(-|ub - lb|, |ub - lb|), p = 0, fp = +inf (pso.hip, pso_init_kernel)
__global__ void batch_init_kernel(const BatchFit *__restrict__ fits, int64_t Dmax)
{
    const BatchFit &f = fits[blockIdx.y];
    const int64_t S = f.S;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= S * Dmax) return;
    const int64_t i = r / Dmax;
    const int d = (int)(r - i * Dmax);
    const int64_t D = 4 + 3 * (int64_t)f.P;
    if (d >= D) return;
    const PsoFused &u = f.upd;
    double r0, r1;
    uniform2(u.seed, 0u, (uint32_t)d, (uint64_t)i, &r0, &r1);
    const double lo = u.lb[d], hi = u.ub[d];
    const double vhigh = fabs(hi - lo), vlow = -vhigh;
    const int64_t e = i * D + d;
    const_cast<double *>(u.x_in)[e] = lo + r0 * (hi - lo);
    const_cast<double *>(u.v_in)[e] = vlow + r1 * (vhigh - vlow);
    double *p = const_cast<double *>(u.p);
    p[e] = 0.0;
    if (d == 0) p[S * D + i] = INFINITY;
    if (r == 0) {
        const_cast<long long *>(u.flags)[0] = 0;
        const_cast<long long *>(u.flags)[1] = 0;
    }
}

enum { kBatchPbest = 1, kBatchArgmin = 2, kBatchApply = 4 };

// One workgroup per fit: what follows an objective launch that did not finish the generation itself -- personal bests
// (generation 0), argmin over fp -> candidate record, fold with pyswarm's rule -- and the fit's line of the summary
// the host reads (generations, stop code, fg, best value, best position).  Same device bodies as pso_tail_kernel.
__global__ __launch_bounds__(1024) void batch_tail_kernel(const BatchFit *__restrict__ fits, int phases, int is_init,
                                                          double *__restrict__ summary, double *__restrict__ bestx,
                                                          const int64_t *__restrict__ boff)
{
    const BatchFit &f = fits[blockIdx.x];
    const PsoFused &u = f.upd;
    const int64_t D = 4 + 3 * (int64_t)f.P, S = f.S;
    long long *flags = const_cast<long long *>(u.flags);
    double *best = const_cast<double *>(u.best);
    double *p = const_cast<double *>(u.p), *fp = p + S * D;
    __shared__ double s_val[16];
    __shared__ long long s_idx[16];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nw = blockDim.x / kWave;
    if (flags[1] == 0) {   // (after a stop nothing moves: pso_tail_kernel returns at once)
        if (phases & kBatchPbest) {
            for (int64_t i = wave; i < S; i += nw) pbest_particle(i, lane, D, u.x_in, f.fx, p, fp);
            __syncthreads();
        }
        if (phases & kBatchArgmin) {
            argmin_block(S, D, fp, p, u.x_in, u.cand, s_val, s_idx);
            __syncthreads();
        }
        if (phases & kBatchApply) {
            if (wave == 0) apply_wave(lane, D, 1, is_init, u.minstep, u.minfunc, u.cand, flags, best);
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        double *s = summary + 4 * (int64_t)blockIdx.x;
        s[0] = (double)flags[0];
        s[1] = (double)flags[1];
        s[2] = best[0];
        s[3] = best[1];
    }
    for (int64_t d = threadIdx.x; d < D; d += blockDim.x) bestx[boff[blockIdx.x] + d] = best[2 + D + d];
}

int bind_batch(const BatchPart *b)
{
    if (!b) {
        set_error("null batch handle");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipSetDevice(b->device));
    return NMRFIT_OK;
}

int launch_generation(const BatchLaunch &g) { return g.fit_im ? launch_objective_batch_im(g) : launch_objective_batch(g); }

const BatchFit *table(const BatchPart *b, int t) { return b->d_tables + (size_t)t * (size_t)b->K; }

int launch_tail(BatchPart *b, int phases, int is_init)
{
    // the table of the CURRENT phases: x_in / p are the buffers the last launch wrote, best / flags the current block
    const BatchFit *t = table(b, b->xp + 2 * b->b);
    hipLaunchKernelGGL(batch_tail_kernel, dim3((unsigned)b->K), dim3(1024), 0, b->stream, t, phases, is_init,
                       b->d_summary, b->d_bestx, reinterpret_cast<const int64_t *>(b->d_bestx + b->Dsum));
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

// fold the generation whose personal bests are still waiting, in a launch of its own (pso.hip, flush_fold)
int flush_fold(BatchPart *b)
{
    if (!b->fold_pending) return NMRFIT_OK;
    const int rc = launch_tail(b, kBatchArgmin | kBatchApply, 0);
    if (rc == NMRFIT_OK) b->fold_pending = false;
    return rc;
}

// the two launch geometries of a batch (objective.hip's launch_objective picks among the same forms for a lone swarm)
void plan_geometry(BatchPart *b)
{
    // (ragged batches: the wave = particle form takes every fit's grid from its record; the launch record then carries
    // the longest grid's figures, which that kernel does not read)
    const int64_t N = b->N ? b->N : b->Nmax;
    const int64_t n_chunks = (N + kChunk - 1) / kChunk;
    const int blk_chunks = (int)((n_chunks + kMaxBlocks - 1) / kMaxBlocks);
    const int64_t n_blocks = (n_chunks + blk_chunks - 1) / blk_chunks;
    const int64_t blk_len = (int64_t)blk_chunks * kChunk;
    const int64_t Dmax = 4 + 3 * (int64_t)b->Pmax;
    for (int m = 0; m < 2; ++m) {
        BatchLaunch &g = b->geom[m];
        g = BatchLaunch{};
        g.stream = b->stream;
        g.K = b->K;
        g.S = b->Smax;
        g.N = N;
        g.blk_chunks = blk_chunks;
        g.n_blocks = (int)n_blocks;
        g.variant = b->variant;
        g.fit_im = b->fit_im;
        b->geom_ok[m] = false;
        int slices, rows;
        size_t row_bytes;
        if (m == 0) {
            // workgroup = particle: eight segments (an eight-wave workgroup) when the grid cuts into exactly eight, else
            // four; the deferred fold's argmin covers kDeferredPerLane x 64 entries per wave (pso_update.h)
            int wpb = 0;
            int64_t seg_len = 0;
            for (int w : {kWideWaves, kWavesPerBlock}) {
                if (b->fit_im != NMRFIT_FIT_IM_OFF) continue;   // (the imaginary channel: wave = particle only)
                if (b->N == 0 || b->S == 0) continue;           // (fits of different lengths or swarm sizes: wave = particle only)
                if (n_blocks < w) continue;
                const int64_t sl = ((n_blocks + w - 1) / w) * blk_len;
                if ((N + sl - 1) / sl != w) continue;
                if (b->S > (int64_t)kDeferredPerLane * kWave * w) continue;
                wpb = w;
                seg_len = sl;
                break;
            }
            if (!wpb) continue;
            g.wpb = wpb;
            g.nseg = wpb;
            g.seg_len = seg_len;
            g.seg_blocks = (int)(seg_len / blk_len);
            g.wave_swarm = false;
            g.blocks_per_fit = b->S;
            slices = 1;
            rows = 3;
            row_bytes = 3 * (size_t)Dmax * sizeof(double);
        } else {
            // wave = particle: one segment, four particles per workgroup (the last workgroup of a fit padded with idle waves,
            // so that a workgroup never spans two fits)
            g.wpb = kWavesPerBlock;
            g.nseg = 1;
            g.seg_len = n_blocks * blk_len;
            g.seg_blocks = (int)n_blocks;
            g.wave_swarm = true;
            g.blocks_per_fit = (b->Smax + kWavesPerBlock - 1) / kWavesPerBlock;
            slices = kWavesPerBlock;
            rows = 14;   // >= 4 x (3 D + 2) doubles for every D >= 4
            row_bytes = (size_t)kWavesPerBlock * (3 * (size_t)Dmax + 2) * sizeof(double);
        }
        int v = b->variant;
        unsigned aux = 0;
        size_t lds = objective_lds(b->variant, b->Pmax, false, b->fit_im, &v, &aux, g.wpb, slices, rows);
        if (v != b->variant) continue;   // (would run another kernel than a lone fit: not bit-identical)
        lds = (lds + 15) & ~(size_t)15;   // the row copies come last (their offset, xrow_offset below, travels in the
        lds += row_bytes;                 // descriptor tables: PsoFused::xrow_off, re-stamped when the geometry changes)
        if (lds + kObjectiveStaticLds + 16 > 160 * 1024) continue;
        g.lds = lds;
        g.aux_off = aux;
        b->geom_ok[m] = true;
    }
}

unsigned xrow_offset(const BatchPart *b, int m)
{
    const BatchLaunch &g = b->geom[m];
    const int64_t Dmax = 4 + 3 * (int64_t)b->Pmax;
    const size_t row_bytes = g.wave_swarm ? (size_t)kWavesPerBlock * (3 * (size_t)Dmax + 2) * sizeof(double)
                                          : 3 * (size_t)Dmax * sizeof(double);
    return (unsigned)(g.lds - row_bytes);
}

}  // namespace
}  // namespace nmrfit

using namespace nmrfit;

namespace {

// per-fit device memory, carved out of the one allocation
struct FitMem {
    double *wc, *u, *v, *wt;
    double2 *chunk;
    double *lb, *ub, *x, *vel, *x2, *vel2, *p, *p_alt, *fx, *cand, *best, *best_alt;
};

struct Carver {
    size_t total = 0;
    size_t take(size_t bytes)
    {
        const size_t at = total;
        total += (bytes + 255) & ~(size_t)255;
        return at;
    }
};

}  // namespace

// ---- one part of a batch: a set of fits advanced by one launch per generation on one stream ----------------

static int part_destroy(BatchPart *b);

// (w, u, v, weights: the part's fits one after the other, fit k's Nk[k] points at offset sum_{i<k} Nk[i])
static int part_create(int device, int32_t K, const int64_t *Nk, const double *w, const double *u, const double *v,
                       const double *weights, const int32_t *P, const double *lower, const double *upper,
                       const int64_t *swarm, const nmrfit_pso_params *params, int variant, int fit_im, BatchPart **out)
{
    if (!out) {
        set_error("null out pointer");
        return NMRFIT_E_INVALID;
    }
    *out = nullptr;
    if (K <= 0 || !Nk || !swarm || !w || !u || !v || !weights || !P || !lower || !upper || !params) {
        set_error("nmrfit_batch_create: K, N, swarmsize must be > 0 and every array non-null");
        return NMRFIT_E_INVALID;
    }
    for (int32_t k = 0; k < K; ++k)
        if (Nk[k] <= 0 || swarm[k] <= 0 || swarm[k] > 0x7fffffffLL / 8) {
            set_error("nmrfit_batch_create: every grid length and swarm size must be > 0 (and a swarm below 2^28 particles)");
            return NMRFIT_E_INVALID;
        }
    if (variant != NMRFIT_VARIANT_DEFAULT && variant != NMRFIT_VARIANT_FARFIELD) {
        set_error("nmrfit_batch_create: device-batched fits run the DEFAULT and FARFIELD kernels");
        return NMRFIT_E_UNSUPPORTED;
    }
    if (fit_im < 0 || fit_im > NMRFIT_FIT_IM_SUM) {
        set_error("fit_im must be 0 (real part), 1 (reference fit_im=True) or 2 (all-peak imaginary model)");
        return NMRFIT_E_INVALID;
    }
    if (K > 65535) {
        set_error("nmrfit_batch_create: more than 65535 fits in one part");
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
    BatchPart *b = new (std::nothrow) BatchPart();
    if (!b) {
        set_error("out of host memory");
        return NMRFIT_E_INVALID;
    }
    b->device = device;
    b->compute_units = prop.cus;
    b->K = K;
    b->Nk.assign(Nk, Nk + K);
    b->noff.resize((size_t)K + 1);
    b->noff[0] = 0;
    b->N = Nk[0];
    for (int32_t k = 0; k < K; ++k) {
        b->noff[(size_t)k + 1] = b->noff[(size_t)k] + Nk[k];
        b->Nmax = std::max(b->Nmax, Nk[k]);
        if (Nk[k] != Nk[0]) b->N = 0;   // ragged
    }
    b->Sk.assign(swarm, swarm + K);
    b->S = swarm[0];
    for (int32_t k = 0; k < K; ++k) {
        b->Smax = std::max(b->Smax, swarm[k]);
        b->Ssum += swarm[k];
        if (swarm[k] != swarm[0]) b->S = 0;   // swarms of different sizes
    }
    b->n_chunks = (b->Nmax + kChunk - 1) / kChunk;
    b->variant = variant;
    b->fit_im = fit_im;
    b->P.assign(P, P + K);
    b->D.resize((size_t)K);
    b->boff.resize((size_t)K);
    for (int32_t k = 0; k < K; ++k) {
        if (P[k] < 0 || 4 + 3 * (int64_t)P[k] > kFusedMaxD) {
            set_error("nmrfit_batch_create: peak counts must be 0 <= P and 4 + 3 P <= " + std::to_string(kFusedMaxD));
            delete b;
            return NMRFIT_E_INVALID;
        }
        b->D[(size_t)k] = 4 + 3 * (int64_t)P[k];
        b->boff[(size_t)k] = b->Dsum;
        b->Dsum += b->D[(size_t)k];
        b->Pmax = std::max(b->Pmax, P[k]);
        b->Psum += P[k];
    }
    for (int64_t d = 0; d < b->Dsum; ++d)
        if (!(upper[d] > lower[d])) {   // pyswarm: assert np.all(ub > lb)
            set_error("All upper-bound values must be greater than lower-bound values");
            delete b;
            return NMRFIT_E_INVALID;
        }
#define BATCH_HIP(call)                                                \
    do {                                                               \
        hipError_t _e = (call);                                        \
        if (_e != hipSuccess) {                                        \
            int _rc = hip_fail(_e, #call, __FILE__, __LINE__);         \
            part_destroy(b);                                   \
            return _rc;                                                \
        }                                                              \
    } while (0)
    BATCH_HIP(take_stream(device, &b->stream));
    plan_geometry(b);
    if (!b->geom_ok[0] && !b->geom_ok[1]) {
        set_error(b->N == 0 && b->fit_im == NMRFIT_FIT_IM_OFF
                      ? "nmrfit_batch_create: fits of different grid lengths run in the wave = particle geometry, and these peak "
                        "counts leave its LDS records no room"
                      : "nmrfit_batch_create: too many peaks for the kernel's LDS records in a batched launch");
        part_destroy(b);
        return NMRFIT_E_UNSUPPORTED;
    }
    // ---- one allocation: per fit the four padded grid arrays + chunk table + swarm state; then the summary, the best
    // rows (+ their offsets), the descriptor tables, and the landing buffer of the upload
    const int64_t Nsum = b->noff[(size_t)K];
    auto chunks_of = [&](int32_t k) { return (b->Nk[(size_t)k] + kChunk - 1) / kChunk; };
    auto padded_of = [&](int32_t k) { return (((size_t)chunks_of(k) * kChunk * sizeof(double)) + 255) & ~(size_t)255; };
    Carver c;
    std::vector<size_t> o_grid((size_t)K), o_chunk((size_t)K), o_x((size_t)K), o_p((size_t)K), o_fx((size_t)K),
        o_cand((size_t)K), o_state((size_t)K);
    std::vector<size_t> state_bytes((size_t)K);
    for (int32_t k = 0; k < K; ++k) {
        const size_t D = (size_t)b->D[(size_t)k];
        const int64_t S = b->Sk[(size_t)k];
        o_grid[(size_t)k] = c.take(4 * padded_of(k));
        o_chunk[(size_t)k] = c.take((size_t)chunks_of(k) * sizeof(double2));
        o_x[(size_t)k] = c.take(4 * (((size_t)S * D * sizeof(double) + 255) & ~(size_t)255));
        o_p[(size_t)k] = c.take(2 * ((((size_t)S * D + (size_t)S) * sizeof(double) + 255) & ~(size_t)255));
        o_fx[(size_t)k] = c.take((size_t)S * sizeof(double));
        o_cand[(size_t)k] = c.take((D + 1) * sizeof(double));
        // (fg, best_f, g[D], best_x[D] | generations, stop code): two copies, the flags right behind the doubles (pso.hip)
        state_bytes[(size_t)k] = (((2 + 2 * D) * sizeof(double) + 2 * sizeof(long long)) + 255) & ~(size_t)255;
        o_state[(size_t)k] = c.take(2 * state_bytes[(size_t)k]);
    }
    // the K boxes, concatenated like the caller's arrays: two uploads for the whole part (they were 2 K small ones, ~8 us each)
    const size_t o_lball = c.take((size_t)b->Dsum * sizeof(double)), o_uball = c.take((size_t)b->Dsum * sizeof(double));
    const size_t grid_state_end = c.total;
    const size_t o_summary = c.take((size_t)K * 4 * sizeof(double));
    const size_t o_bestx = c.take((size_t)b->Dsum * sizeof(double) + (size_t)K * sizeof(int64_t));
    const size_t o_tables = c.take((size_t)9 * (size_t)K * sizeof(BatchFit));
    const size_t o_raw = c.take((size_t)4 * (size_t)Nsum * sizeof(double));
    BATCH_HIP(hipMalloc(&b->d_block, c.total));
    unsigned char *base = reinterpret_cast<unsigned char *>(b->d_block);
    BATCH_HIP(hipMemsetAsync(base, 0, grid_state_end, b->stream));   // padding of the grid arrays (weight 0), state blocks
    b->d_summary = reinterpret_cast<double *>(base + o_summary);
    b->d_bestx = reinterpret_cast<double *>(base + o_bestx);
    b->d_tables = reinterpret_cast<BatchFit *>(base + o_tables);
    double *d_raw = reinterpret_cast<double *>(base + o_raw);
    // ---- descriptor tables (host copy, uploaded once)
    std::vector<FitMem> mem((size_t)K);
    std::vector<BatchFit> tabs((size_t)9 * (size_t)K);
    for (int32_t k = 0; k < K; ++k) {
        const size_t D = (size_t)b->D[(size_t)k];
        const int64_t S = b->Sk[(size_t)k];
        FitMem &m = mem[(size_t)k];
        const size_t pad_al = padded_of(k);
        m.wc = reinterpret_cast<double *>(base + o_grid[(size_t)k]);
        m.u = reinterpret_cast<double *>(base + o_grid[(size_t)k] + pad_al);
        m.v = reinterpret_cast<double *>(base + o_grid[(size_t)k] + 2 * pad_al);
        m.wt = reinterpret_cast<double *>(base + o_grid[(size_t)k] + 3 * pad_al);
        m.chunk = reinterpret_cast<double2 *>(base + o_chunk[(size_t)k]);
        m.lb = reinterpret_cast<double *>(base + o_lball) + b->boff[(size_t)k];
        m.ub = reinterpret_cast<double *>(base + o_uball) + b->boff[(size_t)k];
        const size_t sd_al = ((size_t)S * D * sizeof(double) + 255) & ~(size_t)255;
        m.x = reinterpret_cast<double *>(base + o_x[(size_t)k]);
        m.vel = reinterpret_cast<double *>(base + o_x[(size_t)k] + sd_al);
        m.x2 = reinterpret_cast<double *>(base + o_x[(size_t)k] + 2 * sd_al);
        m.vel2 = reinterpret_cast<double *>(base + o_x[(size_t)k] + 3 * sd_al);
        const size_t p_al = ((((size_t)S * D + (size_t)S) * sizeof(double)) + 255) & ~(size_t)255;
        m.p = reinterpret_cast<double *>(base + o_p[(size_t)k]);
        m.p_alt = reinterpret_cast<double *>(base + o_p[(size_t)k] + p_al);
        m.fx = reinterpret_cast<double *>(base + o_fx[(size_t)k]);
        m.cand = reinterpret_cast<double *>(base + o_cand[(size_t)k]);
        m.best = reinterpret_cast<double *>(base + o_state[(size_t)k]);
        m.best_alt = reinterpret_cast<double *>(base + o_state[(size_t)k] + state_bytes[(size_t)k]);
        BatchFit f{};
        f.wc = m.wc;
        f.u = m.u;
        f.v = m.v;
        f.wt = m.wt;
        f.chunk = m.chunk;
        double grid_dev = 0.0;
        const int64_t N = b->Nk[(size_t)k];
        analyse_grid(w + b->noff[(size_t)k], N, &f.w0, &f.wspan, &f.lane_step, &grid_dev);
        f.rec_devk = grid_dev * 11.0e10;   // (as launch_variant passes it: objective_kernel.h)
        {
            // the fit's own block structure: a function of its N only (objective.hip, launch_objective), one segment
            const int64_t n_chunks = chunks_of(k);
            const int blk_chunks = (int)((n_chunks + kMaxBlocks - 1) / kMaxBlocks);
            const int64_t n_blocks = (n_chunks + blk_chunks - 1) / blk_chunks;
            f.N = N;
            f.blk_chunks = blk_chunks;
            f.n_blocks = (int32_t)n_blocks;
            f.seg_len = n_blocks * (int64_t)blk_chunks * kChunk;
            f.raw_off = b->noff[(size_t)k];
            f.S = S;
        }
        f.fx = m.fx;
        f.P = b->P[(size_t)k];
        const nmrfit_pso_params &prm = params[k];
        for (int t = 0; t < 9; ++t) {
            BatchFit e = f;
            PsoFused &q = e.upd;
            if (t == 8) {
                e.X = m.x;   // plain evaluation of generation 0's positions
                // (what batch_init_kernel and the generation-0 tail need travels in table 0)
            } else {
                const int xp = t & 1, bb = (t >> 1) & 1, pending = (t >> 2) & 1;
                q.x_in = xp ? m.x2 : m.x;
                q.v_in = xp ? m.vel2 : m.vel;
                q.x_out = xp ? m.x : m.x2;
                q.v_out = xp ? m.vel : m.vel2;
                q.p = xp ? m.p_alt : m.p;
                q.pflip = (int)((xp ? m.p : m.p_alt) - (xp ? m.p_alt : m.p));
                q.best = bb ? m.best_alt : m.best;
                q.flags = reinterpret_cast<const long long *>(q.best + 2 + 2 * D);
                q.flip = (int)((bb ? m.best : m.best_alt) - (bb ? m.best_alt : m.best));
                q.lb = m.lb;
                q.ub = m.ub;
                q.seed = prm.seed;
                q.offset = 0;
                q.omega = prm.omega;
                q.phip = prm.phip;
                q.phig = prm.phig;
                q.minstep = prm.minstep;
                q.minfunc = prm.minfunc;
                q.cand = m.cand;
                q.pbest = 1u;
                q.tail = 1u;
                q.pending = (unsigned)pending;
                q.xrow_off = 0u;   // set per geometry below
            }
            tabs[(size_t)t * (size_t)K + (size_t)k] = e;
        }
    }
    // the geometry decides where the row copies sit in LDS: stamp the chosen one's offset into the fused tables
    b->mode = b->geom_ok[0] ? 0 : 1;
    {
        // many particles: the wave form (one prologue per particle instead of one per workgroup of idle waves).  Measured
        // on 204 x 4096 x 6 fits (profiles/r05/batch_fits_small_k.txt): the workgroup form costs ~9 us per fit and
        // generation whatever K (two fits: 20.1 us), the wave form 25 us per generation up to a wave per SIMD and ~9 us
        // per further wave per SIMD (three fits: 25.4, eight: 32.1) -- they cross between two and three fits.  Longer
        // grids have longer waves: there the wave form waits until every SIMD has one.  NMRFIT_BATCH_WAVE_MIN overrides.
        int64_t wave_min = (b->n_chunks <= 16) ? 512 : 1024;
        if (const char *e = getenv("NMRFIT_BATCH_WAVE_MIN")) wave_min = atoll(e);
        if (b->geom_ok[1] && b->Ssum >= wave_min) b->mode = 1;
    }
    for (int t = 0; t < 8; ++t)
        for (int32_t k = 0; k < K; ++k) tabs[(size_t)t * (size_t)K + (size_t)k].upd.xrow_off = xrow_offset(b, b->mode);
    b->h_fits.assign(tabs.begin(), tabs.begin() + K);
    BATCH_HIP(hipMemcpyAsync(b->d_tables, tabs.data(), tabs.size() * sizeof(BatchFit), hipMemcpyHostToDevice, b->stream));
    BATCH_HIP(hipMemcpyAsync(b->d_bestx + b->Dsum, b->boff.data(), (size_t)K * sizeof(int64_t), hipMemcpyHostToDevice, b->stream));
    // ---- spectra: four uploads, one scatter kernel, one chunk-table kernel
    const size_t plane = (size_t)Nsum * sizeof(double);
    const double *host_arrays[] = {w, u, v, weights};
    for (int a = 0; a < 4; ++a)
        BATCH_HIP(hipMemcpyAsync(reinterpret_cast<unsigned char *>(d_raw) + (size_t)a * plane, host_arrays[a], plane,
                                 hipMemcpyHostToDevice, b->stream));
    BATCH_HIP(hipMemcpyAsync(base + o_lball, lower, (size_t)b->Dsum * sizeof(double), hipMemcpyHostToDevice, b->stream));
    BATCH_HIP(hipMemcpyAsync(base + o_uball, upper, (size_t)b->Dsum * sizeof(double), hipMemcpyHostToDevice, b->stream));
    {
        PrepareArgs a{};
        a.plane = Nsum;
        a.raw = d_raw;
        a.fits = b->d_tables;
        hipLaunchKernelGGL(batch_prepare_kernel, dim3((unsigned)((b->Nmax + 255) / 256), (unsigned)K), dim3(256), 0, b->stream, a);
        BATCH_HIP(hipGetLastError());
        hipLaunchKernelGGL(batch_chunk_minmax_kernel, dim3((unsigned)((b->n_chunks + 3) / 4), (unsigned)K), dim3(kWave * 4), 0, b->stream, a);
        BATCH_HIP(hipGetLastError());
    }
    BATCH_HIP(hipStreamSynchronize(b->stream));   // (the host vectors go out of scope)
#undef BATCH_HIP
    *out = b;
    return NMRFIT_OK;
}

static int part_destroy(BatchPart *b)
{
    if (!b) return NMRFIT_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->d_result) (void)hipFree(b->d_result);
    if (b->d_block) (void)hipFree(b->d_block);
    if (b->stream) give_stream(b->device, b->stream);
    delete b;
    return NMRFIT_OK;
}

// generation 0: positions, velocities, evaluation, personal bests, (g, fg) <- the best of them
static int batch_init(BatchPart *b)
{
    b->xp = b->b = 0;
    b->fold_pending = false;
    const int64_t Dmax = 4 + 3 * (int64_t)b->Pmax;
    const int64_t n = b->Smax * Dmax;
    hipLaunchKernelGGL(batch_init_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)b->K), dim3(256), 0, b->stream, table(b, 0), Dmax);
    NMRFIT_HIP(hipGetLastError());
    BatchLaunch g = b->geom[b->mode];
    g.fits = table(b, 8);
    int rc = launch_generation(g);
    if (rc != NMRFIT_OK) return rc;
    if ((rc = launch_tail(b, kBatchPbest | kBatchArgmin | kBatchApply, 1)) != NMRFIT_OK) return rc;
    b->initialized = true;
    return NMRFIT_OK;
}

// one generation of every swarm that has not stopped: ONE launch
static int batch_generation(BatchPart *b)
{
    BatchLaunch g = b->geom[b->mode];
    g.fits = table(b, b->xp + 2 * b->b + (b->fold_pending ? 4 : 0));
    const int rc = launch_generation(g);
    if (rc != NMRFIT_OK) return rc;
    b->xp ^= 1;                          // x / v and (p, fp): the buffers the launch has just written
    if (b->fold_pending) b->b ^= 1;      // it folded the generation before: particle 0 wrote the other state block
    b->fold_pending = true;              // its own generation waits for the next launch (or flush_fold)
    ++b->launches;
    return NMRFIT_OK;
}

static int part_step(BatchPart *b)
{
    int rc = bind_batch(b);
    if (rc != NMRFIT_OK) return rc;
    if (!b->initialized) return batch_init(b);
    return batch_generation(b);
}

static int read_summary(BatchPart *b, std::vector<double> &s)
{
    int rc = flush_fold(b);
    if (rc != NMRFIT_OK) return rc;
    s.resize((size_t)b->K * 4);
    NMRFIT_HIP(hipMemcpyAsync(s.data(), b->d_summary, s.size() * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    NMRFIT_HIP(hipStreamSynchronize(b->stream));
    return NMRFIT_OK;
}

static int part_status(BatchPart *b, int64_t *iteration, int32_t *stop_code, double *fg)
{
    int rc = bind_batch(b);
    if (rc != NMRFIT_OK) return rc;
    if (!b->initialized) {
        set_error("nmrfit_batch_status before the first generation");
        return NMRFIT_E_STATE;
    }
    std::vector<double> s;
    if ((rc = read_summary(b, s)) != NMRFIT_OK) return rc;
    for (int32_t k = 0; k < b->K; ++k) {
        if (iteration) iteration[k] = (int64_t)s[(size_t)k * 4];
        if (stop_code) stop_code[k] = (int32_t)s[(size_t)k * 4 + 1];
        if (fg) fg[k] = s[(size_t)k * 4 + 2];
    }
    return NMRFIT_OK;
}

static int part_best(BatchPart *b, double *x_best, double *f_best)
{
    int rc = bind_batch(b);
    if (rc != NMRFIT_OK) return rc;
    if (!b->initialized) {
        set_error("nmrfit_batch_best before the first generation");
        return NMRFIT_E_STATE;
    }
    std::vector<double> s;
    if ((rc = read_summary(b, s)) != NMRFIT_OK) return rc;
    if (f_best)
        for (int32_t k = 0; k < b->K; ++k) f_best[k] = s[(size_t)k * 4 + 3];
    if (x_best) {
        NMRFIT_HIP(hipMemcpyAsync(x_best, b->d_bestx, (size_t)b->Dsum * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        NMRFIT_HIP(hipStreamSynchronize(b->stream));
    }
    return NMRFIT_OK;
}


// FitUtility.generate_result (nmrfit/utils.py:226-295) for every fit of the part at its best position: ONE launch of the
// reconstruction kernel (result.hip) over the part's resident grids and best rows, enqueued on the part's stream;
// part_contributions_finish brings the arrays to the host (staged_d2h).  The host pointers are this part's shares.
static int part_contributions_enqueue(BatchPart *b, const int64_t *Nout, const double *w_out, double *real_out, double *imag_out,
                                      double *fit_out, double *data_out)
{
    int rc = bind_batch(b);
    if (rc != NMRFIT_OK) return rc;
    if (!b->initialized) {
        set_error("nmrfit_batch_contributions before the first generation");
        return NMRFIT_E_STATE;
    }
    if (b->d_result) {
        set_error("nmrfit_batch_contributions: a reconstruction of this batch is still in flight");
        return NMRFIT_E_STATE;
    }
    if ((rc = flush_fold(b)) != NMRFIT_OK) return rc;   // (the tail launch leaves every fit's best row in d_bestx)
    const int32_t K = b->K;
    // per fit: output length n_k (its own grid's, or Nout[k]); rows of contributions P_k x n_k; 4 x n_k; 2 x N_k
    int64_t n_contrib = 0, n_fit = 0, n_w = 0, n_max = 0;
    for (int32_t k = 0; k < K; ++k) {
        const int64_t nk = w_out ? Nout[k] : b->Nk[(size_t)k];
        if (nk < 0) {
            set_error("nmrfit_batch_contributions: negative output length");
            return NMRFIT_E_INVALID;
        }
        n_contrib += (int64_t)b->P[(size_t)k] * nk;
        n_fit += 4 * nk;
        n_w += w_out ? nk : 0;
        n_max = std::max(n_max, nk);
    }
    if (!real_out) n_contrib = 0;
    if (!fit_out) n_fit = 0;
    const int64_t n_data = data_out ? 2 * b->noff[(size_t)K] : 0;
    if (2 * n_contrib + n_fit + n_data == 0) return NMRFIT_OK;
    const size_t jobs_bytes = ((size_t)K * sizeof(ResultJob) + 255) & ~(size_t)255;
    NMRFIT_HIP(hipMalloc(&b->d_result, jobs_bytes + (size_t)(n_w + 2 * n_contrib + n_fit + n_data) * sizeof(double)));
    unsigned char *base = reinterpret_cast<unsigned char *>(b->d_result);
    double *d_w = reinterpret_cast<double *>(base + jobs_bytes);
    double *d_real = d_w + n_w, *d_imag = d_real + n_contrib, *d_fit = d_imag + n_contrib, *d_data = d_fit + n_fit;
    std::vector<ResultJob> jobs((size_t)K);
    int64_t at_contrib = 0, at_fit = 0, at_w = 0;
    for (int32_t k = 0; k < K; ++k) {
        const BatchFit &f = b->h_fits[(size_t)k];
        const int64_t nk = w_out ? Nout[k] : f.N;
        ResultJob &j = jobs[(size_t)k];
        j = ResultJob{};
        j.wc = f.wc;
        j.w_plain = w_out ? d_w + at_w : nullptr;
        j.x = b->d_bestx + b->boff[(size_t)k];
        j.u = f.u;
        j.v = f.v;
        j.w0 = f.w0;
        j.wspan = f.wspan;
        j.Nout = nk;
        j.N = f.N;
        j.P = f.P;
        j.real = real_out ? d_real + at_contrib : nullptr;
        j.imag = real_out ? d_imag + at_contrib : nullptr;
        j.fit = fit_out ? d_fit + at_fit : nullptr;
        j.data = data_out ? d_data + 2 * b->noff[(size_t)k] : nullptr;
        at_contrib += (int64_t)f.P * nk;
        at_fit += 4 * nk;
        at_w += w_out ? nk : 0;
    }
    hipStream_t st = b->stream;
    // (pageable host memory: the copy has left `jobs` when hipMemcpyAsync returns)
    NMRFIT_HIP(hipMemcpyAsync(base, jobs.data(), (size_t)K * sizeof(ResultJob), hipMemcpyHostToDevice, st));
    if (n_w) NMRFIT_HIP(hipMemcpyAsync(d_w, w_out, (size_t)n_w * sizeof(double), hipMemcpyHostToDevice, st));
    if ((rc = launch_result_jobs(st, reinterpret_cast<const ResultJob *>(base), K, std::max(n_max, data_out ? b->Nmax : 0), b->Pmax)) != NMRFIT_OK)
        return rc;
    // what goes where on the host, for part_contributions_finish (the copies are staged and synchronous: they would
    // serialise the parts' launches if they were made here)
    b->result_copies.clear();
    if (n_contrib) {
        b->result_copies.push_back({real_out, d_real, (size_t)n_contrib * sizeof(double)});
        b->result_copies.push_back({imag_out, d_imag, (size_t)n_contrib * sizeof(double)});
    }
    if (n_fit) b->result_copies.push_back({fit_out, d_fit, (size_t)n_fit * sizeof(double)});
    if (n_data) b->result_copies.push_back({data_out, d_data, (size_t)n_data * sizeof(double)});
    return NMRFIT_OK;
}

static int part_contributions_finish(BatchPart *b)
{
    if (!b->d_result) return NMRFIT_OK;
    (void)hipSetDevice(b->device);
    int rc = NMRFIT_OK;
    for (const BatchPart::ResultCopy &c : b->result_copies)
        if (rc == NMRFIT_OK) rc = staged_d2h(b->device, b->stream, c.host, c.dev, c.bytes);
    b->result_copies.clear();
    const hipError_t e = hipStreamSynchronize(b->stream);
    (void)hipFree(b->d_result);
    b->d_result = nullptr;
    if (rc == NMRFIT_OK && e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize(reconstruction)", __FILE__, __LINE__);
    return rc;
}


static int part_set_geometry(BatchPart *b, int mode)
{
    int rc = bind_batch(b);
    if (rc != NMRFIT_OK) return rc;
    if (mode < 0 || mode > 1 || !b->geom_ok[mode]) {
        set_error("nmrfit_batch_set_geometry: 0 (workgroup = particle) or 1 (wave = particle), where the shape allows it");
        return NMRFIT_E_UNSUPPORTED;
    }
    if (mode == b->mode) return NMRFIT_OK;
    if ((rc = flush_fold(b)) != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipStreamSynchronize(b->stream));
    // the row copies sit elsewhere in LDS: re-stamp the offset in the eight fused tables
    std::vector<BatchFit> tabs((size_t)8 * (size_t)b->K);
    NMRFIT_HIP(hipMemcpy(tabs.data(), b->d_tables, tabs.size() * sizeof(BatchFit), hipMemcpyDeviceToHost));
    for (BatchFit &f : tabs) f.upd.xrow_off = xrow_offset(b, mode);
    NMRFIT_HIP(hipMemcpy(b->d_tables, tabs.data(), tabs.size() * sizeof(BatchFit), hipMemcpyHostToDevice));
    b->mode = mode;
    return NMRFIT_OK;
}

static int part_geometry(const BatchPart *b, int32_t *mode, int32_t *waves_per_workgroup, int32_t *segments, int64_t *workgroups)
{
    if (!b) {
        set_error("null batch handle");
        return NMRFIT_E_INVALID;
    }
    const BatchLaunch &g = b->geom[b->mode];
    if (mode) *mode = b->mode;
    if (waves_per_workgroup) *waves_per_workgroup = g.wpb;
    if (segments) *segments = g.nseg;
    if (workgroups) *workgroups = g.blocks_per_fit * b->K;
    return NMRFIT_OK;
}

static int part_synchronize(BatchPart *b)
{
    int rc = bind_batch(b);
    if (rc != NMRFIT_OK) return rc;
    NMRFIT_HIP(hipStreamSynchronize(b->stream));
    return NMRFIT_OK;
}

// swarm state of fit k (any pointer may be NULL): x, v, p are S x D_k; fx, fp are S
static int part_get_state(BatchPart *b, int32_t k, double *x, double *v, double *p, double *fx, double *fp)
{
    int rc = bind_batch(b);
    if (rc != NMRFIT_OK) return rc;
    if (k < 0 || k >= b->K || !b->initialized) {
        set_error("nmrfit_batch_get_state: fit index out of range, or before the first generation");
        return NMRFIT_E_INVALID;
    }
    if ((rc = flush_fold(b)) != NMRFIT_OK) return rc;
    BatchFit f;
    NMRFIT_HIP(hipMemcpy(&f, table(b, b->xp + 2 * b->b) + k, sizeof f, hipMemcpyDeviceToHost));
    const int64_t S = b->Sk[(size_t)k];
    const size_t sd = (size_t)(S * b->D[(size_t)k]) * sizeof(double), s1 = (size_t)S * sizeof(double);
    hipStream_t st = b->stream;
    if (x) NMRFIT_HIP(hipMemcpyAsync(x, f.upd.x_in, sd, hipMemcpyDeviceToHost, st));
    if (v) NMRFIT_HIP(hipMemcpyAsync(v, f.upd.v_in, sd, hipMemcpyDeviceToHost, st));
    if (p) NMRFIT_HIP(hipMemcpyAsync(p, f.upd.p, sd, hipMemcpyDeviceToHost, st));
    if (fx) NMRFIT_HIP(hipMemcpyAsync(fx, f.fx, s1, hipMemcpyDeviceToHost, st));
    if (fp) NMRFIT_HIP(hipMemcpyAsync(fp, f.upd.p + S * b->D[(size_t)k], s1, hipMemcpyDeviceToHost, st));
    NMRFIT_HIP(hipStreamSynchronize(st));
    return NMRFIT_OK;
}


// ---- the batch: its fits divided over one or two parts, each with a stream of its own ---------------------------------
// A generation of a part is a few lock-step rounds of short waves: its last round drains with the SIMDs
// half empty, its first starts with every wave in the latency-bound prologue.  Two parts on two streams fill each
// other's gaps -- the launches of one generation of part A and part B are independent -- for 5-11 % more fits per
// second (K = 40: 216 -> 241, K = 200: 238 -> 251; profiles/r05/batch_two_streams.txt; K = 6 ... 12: +3-8 %).  From 6
// fits on (each part then still has the particles for the wave = particle geometry); NMRFIT_BATCH_STREAMS=1 turns it
// off (A/B knob).
struct nmrfit_batch {
    std::vector<BatchPart *> parts;
    std::vector<int32_t> first;      // first fit of each part (+ K at the end)
    std::vector<int64_t> boff;       // offset of each fit's row in the concatenated bounds / best arrays (+ total)
    std::vector<int64_t> noff;       // offset of each fit's first grid point in the concatenated spectra (+ total)
    std::vector<int64_t> prow;       // peaks before each fit (+ total)
    int32_t K = 0;
    int device = -1;
};

namespace {

int check_batch_handle(const nmrfit_batch *b)
{
    if (!b || b->parts.empty()) {
        set_error("null batch handle");
        return NMRFIT_E_INVALID;
    }
    NMRFIT_HIP(hipSetDevice(b->device));
    return NMRFIT_OK;
}

static int part_of(const nmrfit_batch *b, int32_t k)
{
    int p = 0;
    while (p + 1 < (int)b->parts.size() && k >= b->first[(size_t)p + 1]) ++p;
    return p;
}

}  // namespace

#pragma GCC visibility push(default)   // the C-ABI: the only symbols the library exports (build.sh: -fvisibility=hidden)
extern "C" {

int nmrfit_batch_create_ragged(int device, int32_t K, const int64_t *N, const double *w, const double *u, const double *v,
                               const double *weights, const int32_t *P, const double *lower, const double *upper,
                               const int64_t *swarmsize, const nmrfit_pso_params *params, int variant, int fit_im, nmrfit_batch **out)
{
    if (!out) {
        set_error("null out pointer");
        return NMRFIT_E_INVALID;
    }
    *out = nullptr;
    if (K <= 0 || !N || !swarmsize || !w || !u || !v || !weights || !P || !lower || !upper || !params) {
        set_error("nmrfit_batch_create: K, N, swarmsize must be > 0 and every array non-null");
        return NMRFIT_E_INVALID;
    }
    nmrfit_batch *b = new (std::nothrow) nmrfit_batch();
    if (!b) {
        set_error("out of host memory");
        return NMRFIT_E_INVALID;
    }
    b->K = K;
    b->device = device;
    b->boff.resize((size_t)K + 1);
    b->noff.resize((size_t)K + 1);
    b->prow.resize((size_t)K + 1);
    b->boff[0] = b->noff[0] = b->prow[0] = 0;
    for (int32_t k = 0; k < K; ++k) {
        b->boff[(size_t)k + 1] = b->boff[(size_t)k] + 4 + 3 * (int64_t)std::max(P[k], 0);
        b->noff[(size_t)k + 1] = b->noff[(size_t)k] + std::max<int64_t>(N[k], 0);
        b->prow[(size_t)k + 1] = b->prow[(size_t)k] + std::max(P[k], 0);
    }
    int nparts = (K >= 6) ? 2 : 1;
    if (const char *e = getenv("NMRFIT_BATCH_STREAMS")) nparts = std::max(1, std::min(atoi(e), (int)std::min<int32_t>(K, 8)));
    for (int p = 0; p <= nparts; ++p) b->first.push_back((int32_t)((int64_t)K * p / nparts));
    for (int p = 0; p < nparts; ++p) {
        const int32_t f0 = b->first[(size_t)p], f1 = b->first[(size_t)p + 1];
        const int64_t n0 = b->noff[(size_t)f0];
        BatchPart *part = nullptr;
        const int rc = part_create(device, f1 - f0, N + f0, w + n0, u + n0, v + n0, weights + n0, P + f0,
                                   lower + b->boff[(size_t)f0], upper + b->boff[(size_t)f0], swarmsize + f0, params + f0, variant,
                                   fit_im, &part);
        if (rc != NMRFIT_OK) {
            nmrfit_batch_destroy(b);
            return rc;
        }
        b->parts.push_back(part);
    }
    *out = b;
    return NMRFIT_OK;
}

int nmrfit_batch_create(int device, int32_t K, int64_t N, const double *w, const double *u, const double *v,
                        const double *weights, const int32_t *P, const double *lower, const double *upper,
                        int64_t swarmsize, const nmrfit_pso_params *params, int variant, int fit_im, nmrfit_batch **out)
{
    if (K <= 0 || N <= 0 || swarmsize <= 0) {
        if (out) *out = nullptr;
        set_error("nmrfit_batch_create: K, N, swarmsize must be > 0 and every array non-null");
        return NMRFIT_E_INVALID;
    }
    const std::vector<int64_t> lengths((size_t)K, N), swarms((size_t)K, swarmsize);
    return nmrfit_batch_create_ragged(device, K, lengths.data(), w, u, v, weights, P, lower, upper, swarms.data(), params, variant,
                                      fit_im, out);
}

int nmrfit_batch_destroy(nmrfit_batch *b)
{
    if (!b) return NMRFIT_OK;
    for (BatchPart *p : b->parts) (void)part_destroy(p);
    delete b;
    return NMRFIT_OK;
}

int nmrfit_batch_step(nmrfit_batch *b)
{
    int rc = check_batch_handle(b);
    for (size_t p = 0; rc == NMRFIT_OK && p < b->parts.size(); ++p) rc = part_step(b->parts[p]);
    return rc;
}

int nmrfit_batch_run(nmrfit_batch *b, int64_t maxiter, int32_t check_every)
{
    int rc = check_batch_handle(b);
    if (rc != NMRFIT_OK) return rc;
    if (maxiter < 0 || check_every < 1) {
        set_error("nmrfit_batch_run: maxiter must be >= 0 and check_every >= 1");
        return NMRFIT_E_INVALID;
    }
    for (BatchPart *p : b->parts)
        if (!p->initialized && (rc = batch_init(p)) != NMRFIT_OK) return rc;
    // every fit runs the generations a lone nmrfit_pso_run would: a stopped swarm's waves return at once, so the others'
    // generations do not touch it; a part leaves the loop when every one of its swarms has stopped (polled every
    // check_every).  The parts' launches are interleaved on their own streams.
    std::vector<char> done(b->parts.size(), 0);
    std::vector<double> s;
    for (int64_t it = 1; it <= maxiter; ++it) {
        bool any = false;
        for (size_t p = 0; p < b->parts.size(); ++p) {
            if (done[p]) continue;
            any = true;
            if ((rc = batch_generation(b->parts[p])) != NMRFIT_OK) return rc;
        }
        if (!any) break;
        if (it % check_every == 0 || it == maxiter) {
            for (size_t p = 0; p < b->parts.size(); ++p) {
                if (done[p]) continue;
                if ((rc = read_summary(b->parts[p], s)) != NMRFIT_OK) return rc;
                bool all = true;
                for (int32_t k = 0; k < b->parts[p]->K; ++k) all = all && s[(size_t)k * 4 + 1] != 0.0;
                done[p] = all ? 1 : 0;
            }
        }
    }
    return NMRFIT_OK;
}

int nmrfit_batch_status(nmrfit_batch *b, int64_t *iteration, int32_t *stop_code, double *fg)
{
    int rc = check_batch_handle(b);
    for (size_t p = 0; rc == NMRFIT_OK && p < b->parts.size(); ++p) {
        const int32_t f0 = b->first[p];
        rc = part_status(b->parts[p], iteration ? iteration + f0 : nullptr, stop_code ? stop_code + f0 : nullptr,
                         fg ? fg + f0 : nullptr);
    }
    return rc;
}

int nmrfit_batch_best(nmrfit_batch *b, double *x_best, double *f_best)
{
    int rc = check_batch_handle(b);
    for (size_t p = 0; rc == NMRFIT_OK && p < b->parts.size(); ++p) {
        const int32_t f0 = b->first[p];
        rc = part_best(b->parts[p], x_best ? x_best + b->boff[(size_t)f0] : nullptr, f_best ? f_best + f0 : nullptr);
    }
    return rc;
}

int nmrfit_batch_contributions(nmrfit_batch *b, const int64_t *Nout, const double *w_out, double *real_out, double *imag_out,
                               double *fit_out, double *data_out)
{
    int rc = check_batch_handle(b);
    if (rc != NMRFIT_OK) return rc;
    if ((w_out != nullptr) != (Nout != nullptr) || (!real_out != !imag_out)) {
        set_error("nmrfit_batch_contributions: Nout and w_out together or not at all; real_out and imag_out likewise");
        return NMRFIT_E_INVALID;
    }
    // every part enqueues its launch on its own stream, then the copies back are made part after part.  A part's share of
    // each output starts where the fits before it end.
    int64_t at_contrib = 0, at_fit = 0, at_w = 0;
    for (size_t p = 0; p < b->parts.size() && rc == NMRFIT_OK; ++p) {
        BatchPart *q = b->parts[p];
        const int32_t f0 = b->first[p], f1 = b->first[p + 1];
        rc = part_contributions_enqueue(q, Nout ? Nout + f0 : nullptr, w_out ? w_out + at_w : nullptr,
                                        real_out ? real_out + at_contrib : nullptr, imag_out ? imag_out + at_contrib : nullptr,
                                        fit_out ? fit_out + at_fit : nullptr, data_out ? data_out + 2 * b->noff[(size_t)f0] : nullptr);
        for (int32_t k = f0; k < f1; ++k) {
            const int64_t nk = Nout ? std::max<int64_t>(Nout[k], 0) : b->noff[(size_t)k + 1] - b->noff[(size_t)k];
            at_contrib += (b->prow[(size_t)k + 1] - b->prow[(size_t)k]) * nk;
            at_fit += 4 * nk;
            at_w += Nout ? nk : 0;
        }
    }
    for (BatchPart *q : b->parts) {
        const int rc2 = part_contributions_finish(q);
        if (rc == NMRFIT_OK) rc = rc2;
    }
    return rc;
}

/* ---- diagnostics (include/nmrfit_amd_diag.h) ---- */

int nmrfit_batch_set_geometry(nmrfit_batch *b, int mode)
{
    int rc = check_batch_handle(b);
    for (size_t p = 0; rc == NMRFIT_OK && p < b->parts.size(); ++p) rc = part_set_geometry(b->parts[p], mode);
    return rc;
}

int nmrfit_batch_geometry(const nmrfit_batch *b, int32_t *mode, int32_t *waves_per_workgroup, int32_t *segments, int64_t *workgroups)
{
    if (!b || b->parts.empty()) {
        set_error("null batch handle");
        return NMRFIT_E_INVALID;
    }
    int64_t total = 0;
    for (size_t p = 0; p < b->parts.size(); ++p) {   // (mode, waves and segments of the first part; workgroups of all)
        int64_t n = 0;
        const int rc = part_geometry(b->parts[p], p == 0 ? mode : nullptr, p == 0 ? waves_per_workgroup : nullptr,
                                     p == 0 ? segments : nullptr, &n);
        if (rc != NMRFIT_OK) return rc;
        total += n;
    }
    if (workgroups) *workgroups = total;
    return NMRFIT_OK;
}

int nmrfit_batch_synchronize(nmrfit_batch *b)
{
    int rc = check_batch_handle(b);
    for (size_t p = 0; rc == NMRFIT_OK && p < b->parts.size(); ++p) rc = part_synchronize(b->parts[p]);
    return rc;
}

int nmrfit_batch_get_state(nmrfit_batch *b, int32_t k, double *x, double *v, double *p, double *fx, double *fp)
{
    int rc = check_batch_handle(b);
    if (rc != NMRFIT_OK) return rc;
    if (k < 0 || k >= b->K) {
        set_error("nmrfit_batch_get_state: fit index out of range");
        return NMRFIT_E_INVALID;
    }
    const int q = part_of(b, k);
    return part_get_state(b->parts[(size_t)q], k - b->first[(size_t)q], x, v, p, fx, fp);
}

}  // extern "C"
#pragma GCC visibility pop
