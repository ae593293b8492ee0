// Internal declarations shared by the .hip translation units of libnmrfit_amd.so.
// Nothing here is part of the ABI (that is include/nmrfit_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "nmrfit_amd.h"

namespace nmrfit {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kWavesPerBlock = 4;    // 256-thread workgroups: one wave per SIMD of a CU
constexpr int kBlock = kWave * kWavesPerBlock;
#ifndef NMRFIT_POINTS
#define NMRFIT_POINTS 8   // tuning knob (tools/ab.py)
#endif
constexpr int kPointsPerLane = NMRFIT_POINTS;    // grid points register-blocked per lane per chunk
constexpr int kChunk = kWave * kPointsPerLane;   // 512 grid points per wave per chunk
constexpr int kMaxBlocks = 16;       // blocks per grid: the unit of the canonical summation order and of the phase
                                     // re-seeding (blk_chunks = ceil(n_chunks / 16)); objective.hip, pso_update.h
constexpr int kMaxPeaks = 960;       // LDS: one copy of the per-peak records per wave when the waves of a workgroup hold different
                                     // particles -- 4 x P x (32 B PeakLor + 8 B PeakWin [+ 16 B recurrence + 32 B PeakFast, dropped
                                     // above P ~ 450]) -- + block seeds, per-wave lane phase seeds, the Dawson table <= 160 KiB

// Physical order of the four grid arrays (centred w, u, v, weights) in device memory.  A lane's q-th point of a
// chunk is the grid point at offset lane + 64*q (lanes stride the grid); in memory each chunk is stored
// PAIR-INTERLEAVED, that point at offset 128*(q/2) + 2*lane + (q%2), so that a lane's points 2m and 2m+1 are one
// aligned 16-byte pair and a wave fetches them with ONE global_load_dwordx4 (1 KiB per wave-instruction) instead of
// two global_load_dwordx2: the vector-memory pipeline works through a wave's addresses at the same pace whatever
// the width, and at 32 eight-byte loads per chunk it -- not the ALUs -- paced the far-field kernel (round 4; see
// DESIGN.md).  Which lane evaluates which point, and in which order everything is summed, does not change: values
// are bit-identical to the plain layout.  The arrays are padded to whole chunks (zeros: weight 0).
__host__ __device__ inline int64_t grid_slot(int64_t j)
{
    const int64_t o = j & (int64_t)(kChunk - 1);
    const int64_t l = o & (int64_t)(kWave - 1), q = o >> 6;
    return (j - o) + (q >> 1) * (2 * kWave) + 2 * l + (q & 1);
}
static_assert(kWave == 64 && kPointsPerLane % 2 == 0, "grid_slot assumes wave64 and an even number of points per lane");

// Per-(particle, peak) constants staged in LDS: see objective.hip.
struct PeakLor {
    double ihw;   // 2/width (|t| capped at 1e18)
    double c;     // -(loc - w0) * ihw        so that t = (w_j - w0)*ihw + c
    double al;    // area*r*(2/(pi*width))    Lorentzian amplitude
    double ag2;   // 2*area*(1-r)*(2/width)*sqrt(ln2/pi)   Gaussian amplitude, factor 2 folds exp2(1)
};
struct PeakFast {   // the same Lorentzian as a plain reciprocal: al/(1+t^2) = 1/(ia + t'^2)
    double ihs;   // ihw / sqrt(al)
    double cs;    // c / sqrt(al)             so that t' = (w_j - w0)*ihs + cs = t / sqrt(al)
    double ia;    // 1 / al
    double pad;   // (32 B records: one ds_read_b128 + one ds_read_b64 per peak, like PeakLor)
};
struct PeakWin {
    float lo;     // (loc - w0) - G*width     the Gaussian is < 2^-64 of its amplitude outside [lo, hi]
    float hi;     // (f32, rounded outwards: 8 B per peak keep three workgroups per CU at P = 24)
};

void set_error(const std::string &msg);
void analyse_grid(const double *w, int64_t N, double *w0, double *wspan, double *lane_step, double *grid_dev);   // cabi.hip
struct DeviceInfo {
    bool known = false;
    int cus = 0;
    char arch[64] = {0};
};
int device_info_cached(int device, DeviceInfo *out);      // cabi.hip: compute units and arch name, once per process
hipError_t take_stream(int device, hipStream_t *out);     // cabi.hip: a recycled (or new) non-blocking stream
void give_stream(int device, hipStream_t s);              // ... handed back idle (synchronised)
int hip_fail(hipError_t e, const char *what, const char *file, int line);
// cabi.hip: device -> pageable host memory through the process's pinned double buffer (synchronous; stream-ordered after
// what is queued on `st`)
int staged_d2h(int device, hipStream_t st, void *dst_host, const void *src_dev, size_t bytes);

#define NMRFIT_HIP(call)                                                        \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return ::nmrfit::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

struct LaunchGeom {
    int64_t waves = 0;
    int32_t nseg = 1;
    int64_t seg_len = 0;
    int32_t waves_per_workgroup = 4;
};

}  // namespace nmrfit

struct nmrfit_ctx {
    int device = -1;
    int compute_units = 0;
    hipStream_t stream = nullptr;      // the stream launches go to (own_stream unless overridden)
    hipStream_t own_stream = nullptr;
    int64_t N = 0;
    double w0 = 0.0;             // centring offset: d_wc[j] = w[j] - w0
    double wspan = 0.0;          // max_j |w[j] - w0|
    double lane_step = 0.0;      // 64 * grid spacing when the grid is uniformly spaced, else 0 (Gaussian recurrence)
    double grid_dev = 0.0;       // bound on |(w[j+k] - w[j]) - k*spacing| over the grid (Gaussian recurrence)
    int64_t target_waves = 0;    // launch-geometry override (0 = heuristic)
    bool wide_workgroups = true; // eight-wave workgroups for particles cut into eight segments (NMRFIT_NO_WIDE_WORKGROUPS: A/B knob)
    void *d_block = nullptr;     // the one allocation behind d_wc, d_u, d_v, d_wt, d_chunk, d_stage
    double *d_wc = nullptr;      // centred grid
    double *d_u = nullptr, *d_v = nullptr, *d_wt = nullptr;
    double2 *d_chunk = nullptr;  // per 512-point chunk: (min, max) of the centred grid
    double *d_stage = nullptr;   // N doubles: plain-order landing buffer of an upload, before it is scattered into grid_slot order
    int64_t n_chunks = 0;
    // grow-on-demand workspace for the host-pointer entry points
    double *d_X = nullptr;
    int64_t cap_X = 0;           // doubles
    double *d_f = nullptr;
    int64_t cap_f = 0;
    double *d_partial = nullptr;
    int64_t cap_partial = 0;
    double *d_R = nullptr;
    int64_t cap_R = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int variant = NMRFIT_VARIANT_DEFAULT;
    int fit_im = 0;              // 0 real only; 1 reference-compatible fit_im=True; 2 all-peak imaginary model
    nmrfit::LaunchGeom last;
    // in-run timing (nmrfit_prof_*): event pairs around the objective kernel, step marks
    int64_t prof_cap = 0;
    std::vector<hipEvent_t> prof_k0, prof_k1, prof_marks;
    int64_t prof_nk = 0, prof_nm = 0;
    unsigned long long *d_clk = nullptr;   // [4]: (s_memtime, s_memrealtime) at the start / end of workgroup 0
};

namespace nmrfit {
// When the objective launch leaves per-block sums of squares instead of f (several segments
// per particle), a caller may take over the final sum (finalize_kernel's job) in its own kernel.
struct ObjectiveDeferred {
    bool needed = false;
    const double *partial = nullptr;   // [S * n_blocks] (x2 with fit_im)
    int64_t n_blocks = 0;
    int fit_im = 0;
    bool pbest_done = false;           // the launch also updated the personal bests (PsoFused::pbest)
    bool tail_done = false;            // ... and finished the generation: candidate record and fold (PsoFused::tail)
    bool need_flush = false;           // NOTHING was launched: a pending fold (PsoFused::pending) cannot ride in this launch's geometry;
                                       // the caller folds in its own launch and calls again without it
};
// Enqueue the objective (R_out == nullptr) or residual launch on ctx->stream.
// `fused` (swarm generations, pso.hip): advance every particle by the swarm's update rule in the
// kernel's prologue and evaluate the NEW positions (dX is then unused).
struct PsoFused;
constexpr int64_t kFusedMaxD = 400;   // 4 waves x D doubles of LDS for the updated rows (12.5 KiB at the limit)
int launch_objective(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *dX, double *df, double *dR,
                     ObjectiveDeferred *defer = nullptr, const PsoFused *fused = nullptr);
int ensure(nmrfit_ctx *ctx, double **buf, int64_t *cap, int64_t need);
// centred grid + per-chunk (min,max) table from the raw device copy of w
int prepare_grid(nmrfit_ctx *ctx, const double *d_w_raw);
// d_dst[grid_slot(j)] = d_src[j], j < N (d_dst: n_chunks * kChunk doubles, padding untouched)
int scatter_grid(nmrfit_ctx *ctx, const double *d_src, double *d_dst);
// gather every rank's n-double record over the communicator (comm.hip), on `stream` (the swarm's context's)
int comm_all_gather(nmrfit_comm *c, hipStream_t stream, const double *d_send, int64_t n, const double **d_all);
nmrfit_ctx *comm_ctx(const nmrfit_comm *c);       // the context a communicator was created on
bool comm_attach(nmrfit_comm *c);                 // claim it for ONE swarm (false: another swarm holds it); destroy order guard
void comm_detach(nmrfit_comm *c);
}  // namespace nmrfit
