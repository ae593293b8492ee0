// pso_update.h -- the swarm's random numbers and its velocity / position rule, shared by the
// stand-alone swarm kernels (pso.hip) and the objective kernel's fused prologue (objective.hip).
//
// pyswarm (github.com/tisimst/pyswarm, pso.py; called from nmrfit/utils.py:176-182):
//     v = omega*v + phip*rp*(p - x) + phig*rg*(g - x);   x = x + v;   clip to [lb, ub]
// with rp, rg ~ U[0,1).  Here the uniforms are Philox4x32-10 keyed by the seed with counter
// (generation, dimension, GLOBAL particle index), and the arithmetic is written without fused
// multiply-add so that it is bit-identical to the numpy mirror in nmrfit_amd/pso.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmrfit_internal.h"   // kMaxBlocks

namespace nmrfit {

struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// two uniforms in [0,1) with 53 random bits each
__device__ __forceinline__ void uniform2(uint64_t seed, uint32_t gen, uint32_t dim, uint64_t particle, double *a,
                                         double *b)
{
    U4 c{gen, dim, (uint32_t)particle, (uint32_t)(particle >> 32)};
    const U4 o = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t ua = ((uint64_t)o.y << 32) | o.x;
    const uint64_t ub = ((uint64_t)o.w << 32) | o.z;
    *a = (double)(ua >> 11) * 0x1.0p-53;
    *b = (double)(ub >> 11) * 0x1.0p-53;
}

// one element of the update: returns the new position, *vn the new velocity
__device__ __forceinline__ double update_value(double xo, double vo, double po, double g, double lo, double hi,
                                               double rp, double rg, double omega, double phip, double phig,
                                               double *vn_out)
{
#pragma clang fp contract(off)
    const double a = omega * vo;
    const double b = (phip * rp) * (po - xo);
    const double c = (phig * rg) * (g - xo);
    const double vn = (a + b) + c;
    double xn = xo + vn;
    if (xn < lo) xn = lo;
    if (xn > hi) xn = hi;
    *vn_out = vn;
    return xn;
}

// What the objective kernel needs to advance a particle before evaluating it (the position
// update fused into its prologue: one launch fewer per generation).  x_in == nullptr: not fused.
struct PsoFused {
    const double *x_in = nullptr, *v_in = nullptr;   // state before the update   [S x D]
    double *x_out = nullptr, *v_out = nullptr;        // state after it (ping-pong: never the same buffers)
    const double *p = nullptr;                        // personal bests            [S x D]
    const double *best = nullptr;                     // [0] fg, [1] best_f, [2..2+D) g
    const double *lb = nullptr, *ub = nullptr;
    const long long *flags = nullptr;                 // [0] completed generations, [1] stop code
    uint64_t seed = 0;
    int64_t offset = 0;                               // global index of this shard's first particle
    double omega = 0.0, phip = 0.0, phig = 0.0;
    unsigned xrow_off = 0;                            // byte offset of the per-wave x rows in dynamic LDS
    // Personal bests in the same launch (pyswarm: where fx < fp: p = x, fp = fx), for the launch geometries
    // in which ONE workgroup holds the whole particle (four segments = its four waves) and therefore knows f
    // at its end: the particle's step -- update, evaluate, personal best -- is then
    // complete inside this kernel and what is left for the next launch is the argmin over fp.  No new
    // pointers (the headline kernel has no scalar register to spare): the swarm keeps fp[S] right behind
    // p[S x D] in one allocation, so fp = p + S*D.
    unsigned pbest = 0;                               // 1: do it (sits in the padding after xrow_off)
    // ... and, single rank, the REST of the generation too (round 4), as a DEFERRED fold (swarms of up to kDeferredPerLane
    // x 64 x waves-per-workgroup particles): a launch ends with the personal bests and nothing else; the NEXT launch's
    // prologue -- every workgroup for itself, redundantly, from the same memory -- takes the argmin over fp, reads
    // the winner's row and folds it with pyswarm's rule before it moves its particle: the generation is ONE launch.
    // No hand-over inside a launch (the kernel boundary orders everything), and the two memory round trips of
    // argmin -> row sit at the START of a kernel, next to the loads of the particle's own state, instead of at its
    // end behind the slowest workgroup.  (g, fg, best, flags) are double-buffered: every workgroup reads the current
    // block, workgroup 0 writes the other one (`flip` 8-byte words away) and the host swaps them after the launch.
    // Whoever reads the swarm's state from outside first folds the last generation in a launch of its own (pso.hip,
    // flush_fold).  (First built with a ticket: every workgroup drew one once its personal best was complete in
    // memory and the last finished the generation -- 204 x 4096 x 6: 13.4 us per generation against 11.7 this way,
    // 256: 14.4 / 11.7, and beyond 256 particles the serialised tickets cost more than the launch they saved.)
    unsigned tail = 0;                                // 1: this launch's generation is folded by the next launch (or flush_fold)
    unsigned pending = 0;                             // a generation's personal bests are waiting to be folded
    int flip = 0;                                     // (other block) - (current block) of best / flags, in 8-byte words
    // The personal bests are double-buffered too: a workgroup that starts late (a grid of several rounds, a CU busy
    // with something else) must still see every particle's fp and p of the generation BEFORE this launch, so this
    // launch reads (p, fp) and writes every particle's row and value -- improved or carried over -- to the other
    // buffer, `pflip` 8-byte words away; the host swaps after the launch.
    int pflip = 0;
    double *cand = nullptr;                           // (D+1): candidate record
    double minstep = 0.0, minfunc = 0.0;
};
// fp entries a lane of the deferred fold's argmin looks at, at most: swarms of up to 4 x 64 x waves-per-workgroup
// particles (1024 with four-wave workgroups: the grids that are resident at once).  Beyond, every workgroup of
// every round pays the prologue's two memory round trips and the gain is gone (4096 x 65536 x 24 with 16 per
// lane: 1180 us per generation against 1177 with the separate launch)
constexpr int kDeferredPerLane = 4;

// Wait until every global store this wave has issued has completed.  For the agent-scope (sc1,
// write-through) atomic stores used to hand data to other workgroups of a running launch that means:
// visible to every XCD.  NOTE: neither __syncthreads() nor a workgroup-scope fence does this on
// gfx950 -- in the default (non-tgsplit) mode they wait for LDS traffic only (s_waitcnt lgkmcnt(0)),
// which is enough inside a workgroup but not for a hand-over through memory.
__device__ __forceinline__ void global_stores_done()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ bool lex_less(double v, long long i, double bv, long long bi)
{
    return v < bv || (v == bv && i < bi);
}

// f = sqrt(mean of squares) from the objective's per-block sums, added in grid order (the canonical
// summation order: see objective.hip).  The <= 16 (x2 with the imaginary channel) values are fetched
// by independent loads issued together -- a loop of load-then-add is one memory round trip per block,
// 8 to 16 of them on the critical path of kernels that have nothing else to do -- and then added
// strictly one after another, so the value is bit-identical to the sequential loop.
__device__ __forceinline__ double finalize_value(const double *partial, int64_t n_blocks, int64_t N, int fit_im)
{
    constexpr int kMax = kMaxBlocks;
    if (fit_im == 0) {
        double v[kMax];
#pragma unroll
        for (int c = 0; c < kMax; ++c) v[c] = (c < n_blocks) ? partial[c] : 0.0;
        double ss = 0.0;
#pragma unroll
        for (int c = 0; c < kMax; ++c)
            if (c < n_blocks) ss += v[c];
        return sqrt(ss / (double)N);
    }
    double v[2 * kMax];
#pragma unroll
    for (int c = 0; c < 2 * kMax; ++c) v[c] = (c < 2 * n_blocks) ? partial[c] : 0.0;
    double ss = 0.0, si = 0.0;
#pragma unroll
    for (int c = 0; c < kMax; ++c)
        if (c < n_blocks) {
            ss += v[2 * c];
            si += v[2 * c + 1];
        }
    return 0.5 * (sqrt(ss / (double)N) + sqrt(si / (double)N));
}

// one wave: fold the candidate records, apply pyswarm's acceptance / stopping rule
__device__ __forceinline__ void apply_wave(int lane, int64_t D, int nranks, int is_init, double minstep,
                                           double minfunc, const double *cands, long long *flags, double *best)
{
#pragma clang fp contract(off)
    // lowest value wins, lowest rank wins ties (every rank sees the same records)
    int win = 0;
    double fc = cands[0];
    for (int r = 1; r < nranks; ++r) {
        const double f = cands[(int64_t)r * (D + 1)];
        if (f < fc) {
            fc = f;
            win = r;
        }
    }
    const double *pc = cands + (int64_t)win * (D + 1) + 1;
    double *g = best + 2, *bx = best + 2 + D;
    const double fg = best[0];
    if (is_init) {
        for (int64_t d = lane; d < D; d += 64) {
            g[d] = pc[d];
            bx[d] = pc[d];
        }
        if (lane == 0) {
            best[0] = fc;
            best[1] = fc;
            flags[0] = 0;
        }
        return;
    }
    int code = 0;   // 0: not better, 1: stop minfunc, 2: stop minstep, 3: accept
    if (fc < fg) {
        double acc = 0.0;
        for (int64_t d = lane; d < D; d += 64) {
            const double df = g[d] - pc[d];
            acc += df * df;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        acc = __shfl(acc, 0, 64);
        const double stepsize = sqrt(acc);
        if (fabs(fg - fc) <= minfunc)
            code = 1;
        else if (stepsize <= minstep)
            code = 2;
        else
            code = 3;
    }
    if (code == 1 || code == 2) {
        for (int64_t d = lane; d < D; d += 64) bx[d] = pc[d];
        if (lane == 0) {
            best[1] = fc;
            flags[1] = code;
        }
    } else if (code == 3) {
        for (int64_t d = lane; d < D; d += 64) {
            g[d] = pc[d];
            bx[d] = pc[d];
        }
        if (lane == 0) {
            best[0] = fc;
            best[1] = fc;
        }
    }
    if (lane == 0) flags[0] = flags[0] + 1;
}

// personal best of particle i by one wave (pyswarm: i_update = fx < fp)
__device__ __forceinline__ void pbest_particle(int64_t i, int lane, int64_t D, const double *x, const double *fx,
                                               double *p, double *fp)
{
    const double f = fx[i];
    if (!(f < fp[i])) return;
    for (int64_t d = lane; d < D; d += kWave) p[i * D + d] = x[i * D + d];
    if (lane == 0) fp[i] = f;
}

// block-wide first index of the minimum of fp (np.argmin) -> candidate record (fused tail, empty shards).
// Must be called by every thread of the block (contains a barrier).
// While no particle has a finite objective yet (every fp still +inf: argmin 0) the record carries x[0] instead
// of p[0] -- pyswarm seeds g with x[0, :] in that case (its `else` branch after the first evaluation), and the
// fold's lowest-rank tie-break makes it GLOBAL particle 0's position; tests/test_pso_cpu.py pins this
// against the restated pyswarm loop.  Later folds ignore a record whose value is +inf.
__device__ __forceinline__ void argmin_block(int64_t S, int64_t D, const double *fp, const double *p, const double *x,
                                             double *cand, double *s_val, long long *s_idx)
{
    double best = INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    for (int64_t i = threadIdx.x; i < S; i += blockDim.x) {
        const double f = fp[i];
        if (f < best) {   // strict: the first (lowest) index wins ties within a thread's stride
            best = f;
            bi = i;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off, kWave);
        const long long oi = __shfl_down(bi, off, kWave);
        if (ob < best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_val[wave] = best;
        s_idx[wave] = bi;
    }
    __syncthreads();
    if (wave == 0) {
        const int nw = blockDim.x / kWave;
        best = (lane < nw) ? s_val[lane] : INFINITY;
        bi = (lane < nw) ? s_idx[lane] : 0x7fffffffffffffffLL;
        for (int off = 8; off > 0; off >>= 1) {
            const double ob = __shfl_down(best, off, kWave);
            const long long oi = __shfl_down(bi, off, kWave);
            if (ob < best || (ob == best && oi < bi)) {
                best = ob;
                bi = oi;
            }
        }
        bi = __shfl(bi, 0, kWave);
        if (bi >= S) bi = 0;   // every fp is +inf: np.argmin -> 0
        const double fbest = (S > 0) ? fp[bi] : INFINITY;
        const double *row = (fbest < INFINITY) ? p : x;
        if (lane == 0) cand[0] = fbest;
        for (int64_t d = lane; d < D; d += kWave) cand[1 + d] = (S > 0) ? row[bi * D + d] : 0.0;
    }
}

}  // namespace nmrfit
