// swarm_prologue.h -- a swarm generation's position update, fused into the prologue of the objective kernel that
// evaluates the new positions (pso_update.h: PsoFused), and -- single rank -- the deferred fold of the generation
// before it.  Replaces the loop body of pyswarm.pso as nmrfit calls it (nmrfit/utils.py:176-182).
#pragma once
#include "objective_math.h"

namespace nmrfit {
namespace {

// Workgroup form: the waves of a workgroup are the segments of ONE particle (`shared`), or every wave holds a particle
// of its own and the fold is not deferred.  xrow: this slice's copy of the updated row in LDS (rows 1 and 2 behind it
// when the fold is deferred).  Returns true when the kernel is done (the swarm has stopped: rows carried over).
template <int WPB>
__device__ __forceinline__ bool swarm_prologue(const PsoFused &upd, double *const xrow, const int64_t D, const int64_t S,
                                               const int64_t particle, const int seg, const bool active, const bool shared,
                                               const int wave, const int lane, double *wsums, unsigned long long *clk)
{
    // (see vector_ptr: the swarm's pointers and constants live in vector registers for the length of this block)
    const auto gx_in = vector_ptr(upd.x_in), gv_in = vector_ptr(upd.v_in), gp = vector_ptr(upd.p);
    const auto gbest = vector_ptr(upd.best), glb = vector_ptr(upd.lb), gub = vector_ptr(upd.ub);
    const auto gflags = vector_ptr(upd.flags);
    const auto gx_out = vector_ptr_rw(upd.x_out), gv_out = vector_ptr_rw(upd.v_out), gp_rw = vector_ptr_rw(upd.p);
    const auto gbest_rw = vector_ptr_rw(upd.best), gcand = vector_ptr_rw(upd.cand);
    const auto gflags_rw = vector_ptr_rw(upd.flags);
    const double q_omega = vector_f64(upd.omega), q_phip = vector_f64(upd.phip), q_phig = vector_f64(upd.phig);
    const double q_minstep = vector_f64(upd.minstep), q_minfunc = vector_f64(upd.minfunc);
    // Swarm generation: the velocity / position update of this particle happens HERE, in the
    // prologue of the kernel that evaluates it (one launch fewer per generation).  Every wave of
    // the particle computes the same new row into its own LDS slice; the wave of segment 0 also
    // writes it (and the velocity) to the swarm's other state buffer -- never the one being
    // read, so the segments of a particle cannot race.  After a stop every launch is a no-op:
    // the row is carried over unchanged and the kernel returns.
    const bool deferred = upd.tail != 0u;   // (the host asks for it only when the workgroup is the particle: `shared`)
    const bool updater = !shared || wave == 0;   // the wave that moves the particle
    bool stopped = false;
    long long gen_done = 0, stop_code = 0;
    double *const grow = xrow + D, *const crow = xrow + 2 * D;   // (tail != 0: rows 1 and 2 of the row area)
    // Everything whose address is known goes out NOW, in one round trip: the flags and, into registers, the first 64
    // entries (all of them up to 20 peaks) of the particle's state, of the bounds and of g.
    const bool have0 = updater && lane < D;
    const int64_t idx0 = particle * D + lane;
    double x0 = 0.0, v0 = 0.0, pold0 = 0.0, lo0 = 0.0, hi0 = 0.0, g0 = 0.0, fg = 0.0;
    if (updater || !deferred) {   // (deferred form: wave 0 tells the workgroup what the fold said, through LDS)
        gen_done = gflags[0];
        stop_code = gflags[1];
    }
    if (updater) {
        if (deferred) fg = gbest[0];
        if (have0) {
            x0 = gx_in[idx0];
            v0 = gv_in[idx0];
            pold0 = gp[idx0];
            lo0 = glb[lane];
            hi0 = gub[lane];
            g0 = gbest[2 + lane];
        }
    }
    double rp0 = 0.0, rg0 = 0.0;   // the first entry's uniforms (deferred form: drawn while the winner's row is on its way)
    bool drawn0 = false;
    if (deferred) {
        // ---- deferred fold (PsoFused): the previous launch left the personal bests of its generation; before this
        // particle moves, its workgroup works out what the swarm's best is NOW -- as every other workgroup does,
        // from the same memory with the same operations (pso_update.h apply_wave, pso.hip argmin_block).
        if (upd.pending != 0u) {   // every wave: first index of the minimum over its share of fp
            const auto fpb = gp + S * D;
            const int kper = (int)((S + WPB * kWave - 1) / (WPB * kWave));   // <= kDeferredPerLane (launch_objective)
            const int64_t base = (int64_t)wave * kper * kWave + lane;
            double vv[kDeferredPerLane];
#pragma unroll
            for (int k = 0; k < kDeferredPerLane; ++k) {   // all loads of a lane in flight together
                const int64_t i = base + (int64_t)k * kWave;
                vv[k] = (k < kper && i < S) ? fpb[i] : INFINITY;
            }
            double best = INFINITY;
            long long bi = 0x7fffffffffffffffLL;
#pragma unroll
            for (int k = 0; k < kDeferredPerLane; ++k)
                if (vv[k] < best) {
                    best = vv[k];
                    bi = base + (long long)k * kWave;
                }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_down(best, off, kWave);
                const long long oi = __shfl_down(bi, off, kWave);
                if (lex_less(ob, oi, best, bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (lane == 0) {   // (the block-sum slots are free until the chunk loop ends)
                wsums[wave] = best;
                wsums[kMaxBlocks + wave] = __longlong_as_double(bi);
            }
        }
        __syncthreads();
        phase_stamp(clk, 10);   // every wave's share of the argmin over fp is in LDS
        if (wave == 0) {
            if (upd.pending != 0u) {
                double fc = wsums[0];
                long long bi = __double_as_longlong(wsums[kMaxBlocks]);
#pragma unroll
                for (int w2 = 1; w2 < WPB; ++w2) {
                    const double ob = wsums[w2];
                    const long long oi = __double_as_longlong(wsums[kMaxBlocks + w2]);
                    if (lex_less(ob, oi, fc, bi)) {
                        fc = ob;
                        bi = oi;
                    }
                }
                if (bi >= S) bi = 0;   // every fp is +inf: np.argmin -> 0, and the row is x[0] (pso.hip, argmin_block)
                const auto src = (fc < INFINITY) ? gp + bi * D : gx_in;
                const double c0 = have0 ? src[lane] : 0.0;   // the second (and last) round trip of the prologue
                if (stop_code == 0) gen_done += 1;   // (after a stop nothing folds and nothing counts: pso_apply_kernel)
                if (have0 && stop_code == 0) {       // meanwhile: this generation's uniforms of entry `lane`
                    uniform2(upd.seed, (uint32_t)(gen_done + 1), (uint32_t)lane, (uint64_t)(upd.offset + particle), &rp0, &rg0);
                    drawn0 = true;
                }
                int code = 0;   // 0: not better, 1: stop minfunc, 2: stop minstep, 3: accept
                {
#pragma clang fp contract(off)
                    double acc = 0.0;
                    if (have0) {
                        crow[lane] = c0;
                        grow[lane] = g0;
                        const double df = g0 - c0;
                        acc += df * df;
                    }
                    for (int64_t d = lane + kWave; d < D; d += kWave) {
                        const double c = src[d], gd = gbest[2 + d];
                        crow[d] = c;
                        grow[d] = gd;
                        const double df = gd - c;
                        acc += df * df;
                    }
                    if (stop_code == 0 && fc < fg) {
                        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
                        acc = __shfl(acc, 0, 64);
                        const double stepsize = sqrt(acc);
                        if (fabs(fg - fc) <= q_minfunc)
                            code = 1;
                        else if (stepsize <= q_minstep)
                            code = 2;
                        else
                            code = 3;
                    }
                    if (code == 1 || code == 2) stop_code = code;
                }
                wave_lds_fence();
                if (particle == 0) {   // ONE writer of the other state block (nobody reads it in this launch)
                    auto bo = gbest_rw + upd.flip;
                    auto fo = gflags_rw + upd.flip;
                    for (int64_t d = lane; d < D; d += kWave) {
                        const double c = crow[d];
                        bo[2 + d] = (code == 3) ? c : grow[d];
                        bo[2 + D + d] = (code != 0) ? c : gbest[2 + D + d];
                        gcand[1 + d] = c;
                    }
                    if (lane == 0) {
                        bo[0] = (code == 3) ? fc : fg;
                        bo[1] = (code != 0) ? fc : gbest[1];
                        fo[0] = gen_done;
                        fo[1] = stop_code;
                        gcand[0] = fc;
                    }
                }
                if (code == 3) {
                    g0 = c0;
                    for (int64_t d = lane + kWave; d < D; d += kWave) grow[d] = crow[d];
                }
                wave_lds_fence();
            } else {
                for (int64_t d = lane + kWave; d < D; d += kWave) grow[d] = gbest[2 + d];
                wave_lds_fence();
            }
            if (lane == 0) wsums[2 * kMaxBlocks + 6] = (stop_code != 0) ? 1.0 : 0.0;
            phase_stamp(clk, 11);   // folded
        }
    }
    stopped = stop_code != 0;
    const uint32_t gen = (uint32_t)(gen_done + 1);

    // every entry of the row: draw, move, clip (pso_update.h); the new row goes to LDS (what this launch evaluates)
    // and, by the wave of segment 0, with the velocity to the swarm's other state buffer.  The first entry of a
    // lane comes from the registers loaded above.
    if (updater)
    for (int64_t d = lane; d < D; d += kWave) {
        const int64_t idx = particle * D + d;
        const bool first = d < kWave;
        double xn = first ? x0 : gx_in[idx], vn = first ? v0 : gv_in[idx];
        const double pold = first ? pold0 : gp[idx];
        // (two loads and a select of VALUES: a select between an LDS and a global address crashes this compiler)
        double g_lds = 0.0, g_mem = 0.0;
        if (!first && deferred) g_lds = grow[d];
        if (!first && !deferred) g_mem = gbest[2 + d];
        const double gd = first ? g0 : deferred ? g_lds : g_mem;
        const double lo = first ? lo0 : glb[d], hi = first ? hi0 : gub[d];
        if (deferred) {   // the personal best as it stands: for the kernel's end (row 2 is free again), or carried over now
            crow[d] = pold;
            if (stopped) gp_rw[upd.pflip + idx] = pold;
        }
        if (!stopped) {
            double rp = rp0, rg = rg0;
            if (!(first && drawn0)) uniform2(upd.seed, gen, (uint32_t)d, (uint64_t)(upd.offset + particle), &rp, &rg);
            xn = update_value(xn, vn, pold, gd, lo, hi, rp, rg, q_omega, q_phip, q_phig, &vn);
        }
        xrow[d] = xn;
        if (active && seg == 0) {
            gx_out[idx] = xn;
            gv_out[idx] = vn;
        }
    }
    if (deferred && stopped && wave == 0 && lane == 0)   // (after a stop: the value is carried over like the rows)
        gp_rw[upd.pflip + S * D + particle] = gp[S * D + particle];
    if (!deferred && stopped) return true;   // the same for every wave of the grid
    phase_stamp(clk, 1);   // position update done
    if (shared) __syncthreads();   // wave 0's row is every wave's input
    if (deferred && wsums[2 * kMaxBlocks + 6] != 0.0) return true;   // (wave 0 told the workgroup: the same in every workgroup of the grid)
    wave_lds_fence();   // same-wave LDS write -> read
    return false;
}

// Wave form (device-batched fits, objective_batch.hip: K independent swarms in one launch, a wave per particle): the wave
// does for its particle what the workgroup form spreads over a workgroup -- the deferred fold of the generation before
// (argmin over its OWN swarm's fp, winner's row, pyswarm's acceptance / stopping rule; every wave of a swarm works out
// the same answer from the same memory, particle 0's wave writes it to the other state block), then the update -- with
// the same operations in the same order, so that a swarm's trajectory is bit-identical to the one a lone fit takes.
// xrow: this wave's three rows in LDS (new position | g | the personal best as it stands) + one double (its value).
// `upd` lives in global memory (the swarm's descriptor): fields are fetched where they are used.
// Returns true when the swarm has stopped (state carried over; the waves of a workgroup belong to one swarm).
__device__ __forceinline__ bool swarm_prologue_wave(const PsoFused &upd, double *const xrow, const int64_t D, const int64_t S,
                                                    const int64_t particle, const bool active, const int lane)
{
    const auto gx_in = vector_ptr(upd.x_in), gv_in = vector_ptr(upd.v_in), gp = vector_ptr(upd.p);
    const auto gbest = vector_ptr(upd.best), glb = vector_ptr(upd.lb), gub = vector_ptr(upd.ub);
    const auto gflags = vector_ptr(upd.flags);
    const auto gx_out = vector_ptr_rw(upd.x_out), gv_out = vector_ptr_rw(upd.v_out), gp_rw = vector_ptr_rw(upd.p);
    const double q_omega = vector_f64(upd.omega), q_phip = vector_f64(upd.phip), q_phig = vector_f64(upd.phig);
    double *const grow = xrow + D, *const crow = xrow + 2 * D;
    // one round trip: flags, fg, this particle's personal-best value and the first 64 entries of its state, of the
    // bounds and of g
    long long gen_done = gflags[0], stop_code = gflags[1];
    const double fg = gbest[0];
    const double fp_old = gp[S * D + particle];
    const bool have0 = lane < D;
    const int64_t idx0 = particle * D + lane;
    double x0 = 0.0, v0 = 0.0, pold0 = 0.0, lo0 = 0.0, hi0 = 0.0, g0 = 0.0;
    if (have0) {
        x0 = gx_in[idx0];
        v0 = gv_in[idx0];
        pold0 = gp[idx0];
        lo0 = glb[lane];
        hi0 = gub[lane];
        g0 = gbest[2 + lane];
    }
    double rp0 = 0.0, rg0 = 0.0;
    bool drawn0 = false;
    if (upd.pending != 0u) {
        // ---- deferred fold: first index of the minimum over this swarm's fp (np.argmin), four loads in flight per lane
        const auto fpb = gp + S * D;
        // (indices as 32-bit integers -- S is a launch-checked int -- and the partners below lane 32 through ds_swizzle:
        // lane l reads lane l ^ off, which for the lanes that still count, l < off, is lane l + off)
        double best = INFINITY;
        int bi32 = 0x7fffffff;
        for (int64_t b0 = 0; b0 < S; b0 += 4 * kWave) {
            double vv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = b0 + (int64_t)k * kWave + lane;
                vv[k] = (i < S) ? fpb[i] : INFINITY;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (vv[k] < best) {   // strict: the lowest index wins ties within a lane (indices ascend)
                    best = vv[k];
                    bi32 = (int)b0 + k * kWave + lane;
                }
        }
        auto take = [&](const double ob, const int oi) {
            if (ob < best || (ob == best && oi < bi32)) {
                best = ob;
                bi32 = oi;
            }
        };
        take(__shfl_down(best, 32, kWave), __shfl_down(bi32, 32, kWave));
        take(swizzle_xor<16>(best), __builtin_amdgcn_ds_swizzle(bi32, (16 << 10) | 0x1f));
        take(swizzle_xor<8>(best), __builtin_amdgcn_ds_swizzle(bi32, (8 << 10) | 0x1f));
        take(swizzle_xor<4>(best), __builtin_amdgcn_ds_swizzle(bi32, (4 << 10) | 0x1f));
        take(swizzle_xor<2>(best), __builtin_amdgcn_ds_swizzle(bi32, (2 << 10) | 0x1f));
        take(swizzle_xor<1>(best), __builtin_amdgcn_ds_swizzle(bi32, (1 << 10) | 0x1f));
        const double fc = wave_uniform(best);
        long long bi = (long long)__builtin_amdgcn_readfirstlane(bi32);
        if (bi >= S || bi < 0) bi = 0;   // every fp is +inf: np.argmin -> 0, and the row is x[0] (pso.hip, argmin_block)
        const auto src = (fc < INFINITY) ? gp + bi * D : gx_in;
        const double c0 = have0 ? src[lane] : 0.0;   // the second (and last) round trip of the prologue
        if (stop_code == 0) gen_done += 1;   // (after a stop nothing folds and nothing counts: pso_apply_kernel)
        if (have0 && stop_code == 0) {       // meanwhile: this generation's uniforms of entry `lane`
            uniform2(upd.seed, (uint32_t)(gen_done + 1), (uint32_t)lane, (uint64_t)(upd.offset + particle), &rp0, &rg0);
            drawn0 = true;
        }
        int code = 0;   // 0: not better, 1: stop minfunc, 2: stop minstep, 3: accept
        {
#pragma clang fp contract(off)
            double acc = 0.0;
            if (have0) {
                crow[lane] = c0;
                grow[lane] = g0;
                const double df = g0 - c0;
                acc += df * df;
            }
            for (int64_t d = lane + kWave; d < D; d += kWave) {
                const double c = src[d], gd = gbest[2 + d];
                crow[d] = c;
                grow[d] = gd;
                const double df = gd - c;
                acc += df * df;
            }
            if (stop_code == 0 && fc < fg) {
                acc = wave_uniform(wave_sum(acc));   // (the same tree as pso.hip's shuffle loop: x[l] += x[l + off])
                const double stepsize = sqrt(acc);
                if (fabs(fg - fc) <= upd.minfunc)
                    code = 1;
                else if (stepsize <= upd.minstep)
                    code = 2;
                else
                    code = 3;
            }
            if (code == 1 || code == 2) stop_code = code;
        }
        wave_lds_fence();
        if (particle == 0 && active) {   // ONE writer of the other state block (nobody reads it in this launch)
            const auto bo = vector_ptr_rw(upd.best) + upd.flip;
            const auto fo = vector_ptr_rw(upd.flags) + upd.flip;
            const auto gcand = vector_ptr_rw(upd.cand);
            for (int64_t d = lane; d < D; d += kWave) {
                const double c = crow[d];
                bo[2 + d] = (code == 3) ? c : grow[d];
                bo[2 + D + d] = (code != 0) ? c : gbest[2 + D + d];
                gcand[1 + d] = c;
            }
            if (lane == 0) {
                bo[0] = (code == 3) ? fc : fg;
                bo[1] = (code != 0) ? fc : gbest[1];
                fo[0] = gen_done;
                fo[1] = stop_code;
                gcand[0] = fc;
            }
        }
        if (code == 3) {
            g0 = c0;
            for (int64_t d = lane + kWave; d < D; d += kWave) grow[d] = crow[d];
        }
        wave_lds_fence();
    } else {
        for (int64_t d = lane + kWave; d < D; d += kWave) grow[d] = gbest[2 + d];
        wave_lds_fence();
    }
    const bool stopped = stop_code != 0;
    const uint32_t gen = (uint32_t)(gen_done + 1);
    // every entry of the row: draw, move, clip (pso_update.h); the new row goes to LDS (what this wave evaluates) and, with
    // the velocity, to the swarm's other state buffer; the personal best as it stands to row 2, for the kernel's end
    for (int64_t d = lane; d < D; d += kWave) {
        const int64_t idx = particle * D + d;
        const bool first = d < kWave;
        double xn = first ? x0 : gx_in[idx], vn = first ? v0 : gv_in[idx];
        const double pold = first ? pold0 : gp[idx];
        double g_lds = 0.0;
        if (!first) g_lds = grow[d];
        const double gd = first ? g0 : g_lds;
        const double lo = first ? lo0 : glb[d], hi = first ? hi0 : gub[d];
        crow[d] = pold;
        if (stopped && active) gp_rw[upd.pflip + idx] = pold;   // (after a stop: carried over unchanged)
        if (!stopped) {
            double rp = rp0, rg = rg0;
            if (!(first && drawn0)) uniform2(upd.seed, gen, (uint32_t)d, (uint64_t)(upd.offset + particle), &rp, &rg);
            xn = update_value(xn, vn, pold, gd, lo, hi, rp, rg, q_omega, q_phip, q_phig, &vn);
        }
        xrow[d] = xn;
        if (active) {
            gx_out[idx] = xn;
            gv_out[idx] = vn;
        }
    }
    if (lane == 0) {
        xrow[3 * D] = fp_old;
        if (stopped && active) gp_rw[upd.pflip + S * D + particle] = fp_old;
    }
    wave_lds_fence();   // same-wave LDS write -> read
    return stopped;
}

// ... and its end: pyswarm's `i_update = fx < fp; p[i_update] = x[i_update]; fp[i_update] = fx[i_update]`, written to the
// OTHER (p, fp) buffer whether the particle improved or not (PsoFused::pflip: a wave that starts late must still
// find every particle's personal best of the generation before this launch).  f: the same in every lane.
__device__ __forceinline__ void personal_best_wave(const PsoFused &upd, const double *xrow, const int64_t D, const int64_t S,
                                                   const int64_t particle, const int lane, const double f)
{
    const auto pb = vector_ptr_rw(upd.p) + upd.pflip;
    const double fp_old = xrow[3 * D];
    const bool better = f < fp_old;
    const double *keep = xrow + 2 * D;
    for (int64_t d = lane; d < D; d += kWave) pb[particle * D + d] = better ? xrow[d] : keep[d];
    if (lane == 0) pb[S * D + particle] = better ? f : fp_old;
}

}  // namespace
}  // namespace nmrfit
