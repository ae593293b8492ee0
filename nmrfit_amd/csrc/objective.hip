// objective.hip -- host side of the hot path: launch geometry (segments per particle, waves per workgroup), the LDS
// budget that picks the kernel form, and the launch itself (launch_objective); plus the small kernels around it
// (block-sum finalisation, grid preparation; the post-fit reconstruction is result.hip).  The objective kernel is
// objective_kernel.h; its instantiations live in objective_{default,farfield,norec}.hip.
#include "objective_launch.h"
#include "objective_math.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace nmrfit {
namespace {

// f[i] = sqrt( (sum of the particle's per-block sums, in grid order) / N ); with the imaginary
// part: the mean of the real and imaginary RMSE (two sums per block)
__global__ void finalize_kernel(const double *__restrict__ partial, int64_t S, int64_t n_chunks, int64_t N,
                                int fit_im, double *__restrict__ f)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    f[i] = finalize_value(partial + i * n_chunks * (fit_im ? 2 : 1), n_chunks, N, fit_im);
}

// per-chunk (min, max) of the centred grid: one wave per chunk
__global__ void chunk_minmax_kernel(const double *__restrict__ wc, int64_t N, int64_t n_chunks,
                                    double2 *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t c = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    double lo = INFINITY, hi = -INFINITY;
    for (int q = 0; q < kPointsPerLane; ++q) {
        const int64_t j = c * kChunk + q * kWave + lane;
        if (j < N) {
            const double x = wc[grid_slot(j)];
            lo = fmin(lo, x);
            hi = fmax(hi, x);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off, kWave));
        hi = fmax(hi, __shfl_down(hi, off, kWave));
    }
    if (lane == 0) out[c] = make_double2(lo, hi);
}

__global__ void centre_kernel(const double *__restrict__ w, int64_t N, double w0, double *__restrict__ wc)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) wc[grid_slot(j)] = w[j] - w0;
}

__global__ void scatter_grid_kernel(const double *__restrict__ src, int64_t N, double *__restrict__ dst)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) dst[grid_slot(j)] = src[j];
}

}  // namespace

// Builds the derived device arrays of a context (centred grid, chunk table).
int prepare_grid(nmrfit_ctx *ctx, const double *d_w_raw)
{
    const int64_t N = ctx->N;
    hipLaunchKernelGGL(centre_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_w_raw, N,
                       ctx->w0, ctx->d_wc);
    NMRFIT_HIP(hipGetLastError());
    const int waves_per_block = 4;
    hipLaunchKernelGGL(chunk_minmax_kernel, dim3((unsigned)((ctx->n_chunks + waves_per_block - 1) / waves_per_block)),
                       dim3(kWave * waves_per_block), 0, ctx->stream, ctx->d_wc, N, ctx->n_chunks, ctx->d_chunk);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

int scatter_grid(nmrfit_ctx *ctx, const double *d_src, double *d_dst)
{
    const int64_t N = ctx->N;
    hipLaunchKernelGGL(scatter_grid_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_src, N, d_dst);
    NMRFIT_HIP(hipGetLastError());
    return NMRFIT_OK;
}

// The kernel variant a launch actually runs (the requested one may not fit in LDS, or may not
// implement the imaginary part) and the dynamic LDS its per-wave records need.
constexpr size_t kStaticLds = kObjectiveStaticLds;   // objective_kernel's own __shared__ (wsums) + alignment slack
static_assert(kObjectiveStaticLds == (size_t)kWsumsCount * sizeof(double) + 64, "objective_launch.h and objective_math.h disagree");
// `slices`: copies of the per-peak records in a workgroup (1 when its waves are segments of one particle, else wpb);
// `rows`: copies of the updated row kept for a fused swarm generation (0: none)
size_t objective_lds(int variant, int32_t P, bool residual, int fit_im, int *variant_out, unsigned *aux_off, int wpb,
                     int slices, int rows)
{
    // the mixed-precision far-field kernel exists for objective launches without the imaginary channel; everything else
    // of a context set to it runs the fp64 far-field kernel (same records, same LDS)
    const bool want32 = variant == NMRFIT_VARIANT_FARFIELD32;
    if (want32) variant = NMRFIT_VARIANT_FARFIELD;
    const size_t np = (size_t)std::max(P, 1);
    // what a workgroup may take of a CU's 160 KiB for the records sized here: everything but the kernel's static
    // LDS and (swarm generations) the per-wave copies of the updated rows that launch_objective appends
    const size_t room = 160 * 1024 - kStaticLds -
                        (rows ? (size_t)rows * (size_t)(4 + 3 * (int64_t)P) * sizeof(double) + 16 : 0);
    const size_t lds_recs = (((size_t)slices * np * (sizeof(PeakLor) + sizeof(PeakWin)) + 15) & ~(size_t)15) +
                            (size_t)wpb * kMaxBlocks * sizeof(double2) + kSharedPrologueBytes +
                            (slices == 1 ? 0 : (size_t)wpb * 2 * kWave * sizeof(double));   // per-wave lane phase seeds
    const size_t lds_stage = (size_t)wpb * 3 * kChunk * sizeof(double);
    const size_t lds_far = (size_t)wpb * far_stride(fit_im) * sizeof(double);
    // Gaussian recurrence constants (d, C) per peak: objective launches of DEFAULT / FARFIELD
    const size_t lds_rec = residual ? 0 : (size_t)slices * np * sizeof(double2);
    const size_t lds_fast = (size_t)slices * np * sizeof(PeakFast);   // scaled records of the two-operation pair form (DEFAULT)
    // the all-peak imaginary model sums far peaks through the far-field scratch and evaluates Dawson's
    // integral from a table in LDS
    const size_t lds_im = (fit_im == 2) ? lds_far : 0;
    // (fit_im == 1 reads the same table: gathered intervals near the last peak, the asymptotic series elsewhere)
    const size_t lds_tab = (fit_im != 0) ? (size_t)kDawTabCount * sizeof(double) + 16 : 0;
    // the imaginary channel exists in DEFAULT, NOREC and FARFIELD; the A/B variants fail in launch_variant
    if (variant == NMRFIT_VARIANT_STAGED && fit_im != 0) variant = NMRFIT_VARIANT_DEFAULT;
    // STAGED needs three workgroups to still fit in a CU's 160 KiB (P <= 27); beyond that it
    // runs the unstaged kernel.
    if (variant == NMRFIT_VARIANT_STAGED && 3 * (lds_recs + lds_stage + kStaticLds) > 160 * 1024) variant = NMRFIT_VARIANT_DEFAULT;
    if (variant == NMRFIT_VARIANT_FARFIELD && lds_recs + lds_far + lds_rec + lds_tab + 16 > room)
        variant = NMRFIT_VARIANT_DEFAULT;   // P > ~600
    if (variant == NMRFIT_VARIANT_DEFAULT && lds_recs + lds_im + lds_rec + lds_fast + lds_tab + 16 > room)
        variant = NMRFIT_VARIANT_NOREC;     // P > ~450: no room for the recurrence / scaled records
    size_t lds = lds_recs + (variant == NMRFIT_VARIANT_STAGED ? lds_stage : 0) +
                 (variant == NMRFIT_VARIANT_FARFIELD ? lds_far : lds_im) +
                 ((variant == NMRFIT_VARIANT_FARFIELD || variant == NMRFIT_VARIANT_DEFAULT) ? lds_rec : 0) +
                 (variant == NMRFIT_VARIANT_DEFAULT ? lds_fast : 0);
    *aux_off = 0;
    if (lds_tab) {
        lds = (lds + 15) & ~(size_t)15;
        *aux_off = (unsigned)lds;
        lds += lds_tab;
    }
    if (want32 && variant == NMRFIT_VARIANT_FARFIELD && !residual && fit_im == 0) variant = NMRFIT_VARIANT_FARFIELD32;
    *variant_out = variant;
    return lds;
}

int launch_objective(nmrfit_ctx *ctx, int64_t S, int32_t P, const double *dX, double *df, double *dR,
                     ObjectiveDeferred *defer, const PsoFused *fused)
{
    if (defer) *defer = ObjectiveDeferred{};
    const int fit_im = dR ? 0 : ctx->fit_im;
    if (S == 0) return NMRFIT_OK;
    const int64_t N = ctx->N;
    // Segmenting: a wave is one (particle, segment) task; a segment is a whole number of blocks.  Swarms that already
    // supply enough waves get nseg = 1 and the wave writes f directly.  (Rounds 1-3 aimed at ~16 tasks per SIMD so that
    // the hardware dispatcher load-balances -- C3: 4096 one-per-particle waves 1.81 ms, 16384 waves 1.74 ms; the
    // round-4 rule below replaces it unless NMRFIT_SEG_RULE=3 or an explicit target is set.)
    int64_t target_waves = (int64_t)ctx->compute_units * 4 * 16;
    if (ctx->target_waves > 0) target_waves = ctx->target_waves;
    // Blocks: the unit of the canonical summation / phase re-seeding, a function of N only
    // (at most 16 per grid), so that results do not depend on S or on sharding.
    const int64_t n_chunks = (N + kChunk - 1) / kChunk;
    const int blk_chunks = (int)((n_chunks + kMaxBlocks - 1) / kMaxBlocks);
    const int64_t n_blocks = (n_chunks + blk_chunks - 1) / blk_chunks;
    const int64_t blk_len = (int64_t)blk_chunks * kChunk;
    int64_t nseg = std::max<int64_t>(1, std::min<int64_t>(n_blocks, (target_waves + S - 1) / S));
    static const bool seg_rule_r4 = [] {
        const char *e = getenv("NMRFIT_SEG_RULE");   // A/B knob: 3 = the round-3 rule (~16 tasks per SIMD)
        return !(e && atoi(e) == 3);
    }();
    if (ctx->target_waves == 0 && seg_rule_r4) {
        // Round 4 (the selectable kernels now run four waves per SIMD): as few segments as fill the chip ONCE -- every
        // further segment repeats a wave's prologue and cuts its chunk loop shorter (1024 x 16384 x 12: 16 segments
        // 69 us, 4 segments 60 us; 4096 x 4096 x 6: 4 segments 56 us, 1 segment 49 us) -- but four where a wave then
        // still has 16 chunks or more: the prologue no longer counts there, and four segments make the workgroup
        // the particle (f and the personal best finished in the launch)
        const int64_t slots = (int64_t)ctx->compute_units * 4 * 4;
        nseg = std::max<int64_t>(1, std::min<int64_t>(n_blocks, (slots + S - 1) / S));
        if (n_chunks / 4 >= 16) nseg = std::max<int64_t>(nseg, std::min<int64_t>(4, n_blocks));
    }
    // Short grids: a wave's own prologue (block seeds, pointers, the barrier of the shared part) still
    // costs a good fraction of a chunk, so prefer >= 2 chunks per wave as long as two waves per SIMD
    // remain (measured on C2, S=1024 N=4096, with the per-workgroup prologue: 8 segments 25.0 us,
    // 4 segments 22.7 us, 2 segments 23.3 us; round 1, prologue per wave: 8 -> 26.2, 2 -> 22.2).
    if (ctx->target_waves == 0) {
        const int64_t simds = (int64_t)ctx->compute_units * 4;
        while (nseg > 1 && n_chunks / nseg < 2 && S * ((nseg + 1) / 2) >= 2 * simds) nseg = (nseg + 1) / 2;
        // a small swarm whose waves just miss fitting on the chip at once (three per SIMD) while half
        // as many would leave it well filled: take the half (204 x 16384 x 12: 16 segments 30.4 us per
        // generation, 8 segments 28.5; tools/archive/nseg_generation_ab.py)
        if (nseg > 1 && S * nseg > 3 * simds && S * (nseg / 2) < 2 * simds && n_chunks / nseg <= 2) nseg /= 2;
    }
    int64_t seg_len = ((n_blocks + nseg - 1) / nseg) * blk_len;
    nseg = (N + seg_len - 1) / seg_len;
    const int64_t waves = S * nseg;
    int variant = NMRFIT_VARIANT_DEFAULT;
    unsigned aux_off = 0;
    const bool fused_rows = fused && fused->x_in;
    // LDS copies: per-peak records once per workgroup when its waves are segments of ONE particle (nseg a multiple of
    // the waves per workgroup), else once per wave; the updated row of a fused swarm generation likewise, plus two
    // more rows (g, the winner's row) when the workgroup may finish the generation (objective.hip, personal_best)
    // swarms the two whole-generation forms of a fused launch take (PsoFused::tail), by waves per workgroup
    auto tail_fits = [&](unsigned tail, int w) {
        return tail != 0u && S <= (int64_t)kDeferredPerLane * kWave * w;
    };
    auto copies = [&](int w, int *slices, int *rows) {
        const bool one_particle = (nseg % w == 0);   // (objective_body: `shared`)
        *slices = one_particle ? 1 : w;
        *rows = !fused_rows ? 0 : !one_particle ? w : (nseg == w && tail_fits(fused->tail, w)) ? 3 : 1;
    };
    int wpb = kWavesPerBlock, slices = 0, rows = 0;
    copies(wpb, &slices, &rows);
    size_t lds = objective_lds(ctx->variant, P, dR != nullptr, fit_im, &variant, &aux_off, wpb, slices, rows);
    // Eight segments per particle (small swarms on short grids -- the reference's default 204 x 4096): an EIGHT-wave
    // workgroup is the particle, as the four-wave workgroup is for four segments: one prologue per particle, block
    // sums through LDS, f (and, in a swarm generation, the personal best) finished in this launch.
    const size_t xrow_bytes = fused_rows ? (size_t)(4 + 3 * (int64_t)P) * sizeof(double) : 0;
    if (nseg == kWideWaves && !dR && fit_im == 0 && has_eight_wave_form(variant) && ctx->wide_workgroups) {
        int v8 = variant, s8 = 0, r8 = 0;
        unsigned aux8 = 0;
        copies(kWideWaves, &s8, &r8);
        const size_t lds8 = objective_lds(ctx->variant, P, false, fit_im, &v8, &aux8, kWideWaves, s8, r8);
        if (v8 == variant && lds8 + kStaticLds + 16 + (size_t)r8 * xrow_bytes <= 160 * 1024) {
            wpb = kWideWaves;
            slices = s8;
            rows = r8;
            lds = lds8;
            aux_off = aux8;
        }
    }
    const int64_t blocks = (waves + wpb - 1) / wpb;
    if (blocks > 0x7fffffffLL) {
        set_error("swarm too large for one launch");
        return NMRFIT_E_INVALID;
    }
    if (lds + kStaticLds + 16 + (size_t)rows * xrow_bytes > 160 * 1024) {
        set_error("too many peaks for the kernel's LDS records (with the imaginary model / the fused swarm update)");
        return NMRFIT_E_UNSUPPORTED;
    }
    // fused swarm update: one copy of the particle's updated row per wave, after everything else
    PsoFused upd{};
    if (fused && fused->x_in) {
        if (dR || (4 + 3 * (int64_t)P) > kFusedMaxD) {
            set_error("fused swarm update: objective launches with D <= kFusedMaxD only");
            return NMRFIT_E_INVALID;
        }
        upd = *fused;
        lds = (lds + 15) & ~(size_t)15;
        upd.xrow_off = (unsigned)lds;
        lds += (size_t)rows * (size_t)(4 + 3 * (int64_t)P) * sizeof(double);
    }
    // nseg == 1: the wave writes f; nseg == 4 (or 8, wide form): the waves of a workgroup are the particle's
    // segments and the workgroup writes f (block sums through LDS); otherwise per-block sums go to a
    // buffer and finalize_kernel (or the swarm's select kernel) adds them.  Same summation order in all.
    const bool direct_f = (nseg == 1 || nseg == wpb);
    if (nseg != wpb) upd.pbest = 0u;   // (only when ONE workgroup holds the whole particle; else the caller's kernel)
    if (upd.pbest == 0u || !tail_fits(upd.tail, wpb)) upd.tail = 0u;   // (the whole generation in this launch: only on top of the personal bests)
    if (fused && fused->tail != 0u && fused->pending != 0u && upd.tail == 0u) {
        // a generation waits to be folded and this launch cannot do it (the variant or fit_im changed between two
        // generations): nothing is launched, the caller folds in a launch of its own and comes back
        if (!defer) {
            set_error("launch_objective: pending fold without a deferred-launch record");
            return NMRFIT_E_STATE;
        }
        defer->need_flush = true;
        return NMRFIT_OK;
    }
    double *out = df;
    if (!direct_f) {
        int rc = ensure(ctx, &ctx->d_partial, &ctx->cap_partial, S * n_blocks * (fit_im ? 2 : 1));
        if (rc != NMRFIT_OK) return rc;
        out = ctx->d_partial;
    }
    ObjectiveLaunch la{};
    la.ctx = ctx;
    la.S = S;
    la.P = P;
    la.dX = dX;
    la.out = out;
    la.dR = dR;
    la.nseg = (int)nseg;
    la.seg_len = seg_len;
    la.blk_chunks = blk_chunks;
    la.seg_blocks = (int)(seg_len / blk_len);
    la.n_blocks = (int)n_blocks;
    la.blocks = blocks;
    la.lds = lds;
    la.fit_im = fit_im;
    la.upd = upd;
    la.aux_off = aux_off;
    la.wpb = wpb;
    int rc;
    switch (variant) {   // one translation unit per selectable variant (they compile in parallel)
        case NMRFIT_VARIANT_DEFAULT: rc = launch_objective_default(la); break;
        case NMRFIT_VARIANT_FARFIELD: rc = launch_objective_farfield(la); break;
        case NMRFIT_VARIANT_FARFIELD32: rc = launch_objective_farfield32(la); break;
        case NMRFIT_VARIANT_NOREC: rc = launch_objective_norec(la); break;
        default:
#ifdef NMRFIT_AB_BUILD
            rc = launch_objective_ab(variant, la);
#else
            set_error("this kernel variant exists in A/B builds of the library only (-DNMRFIT_AB_BUILD)");
            rc = NMRFIT_E_UNSUPPORTED;
#endif
            break;
    }
    if (rc != NMRFIT_OK) return rc;
    if (defer) defer->pbest_done = upd.x_in != nullptr && upd.pbest != 0u;
    if (defer) defer->tail_done = defer->pbest_done && upd.tail != 0u;
    if (!direct_f && defer) {   // the caller's own kernel adds the per-block sums (pso_tail_kernel)
        defer->needed = true;
        defer->partial = ctx->d_partial;
        defer->n_blocks = n_blocks;
        defer->fit_im = fit_im;
    } else if (!direct_f) {
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->d_partial, S, n_blocks, N, fit_im, df);
        NMRFIT_HIP(hipGetLastError());
    }
    ctx->last.waves = waves;
    ctx->last.waves_per_workgroup = wpb;
    ctx->last.nseg = (int32_t)nseg;
    ctx->last.seg_len = seg_len;
    return NMRFIT_OK;
}

}  // namespace nmrfit
