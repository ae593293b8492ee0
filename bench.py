#!/usr/bin/env python3
"""
bench.py -- objective evaluations per second of the MI355X swarm generation.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / configs[3]): 24 peaks, 65536-point grid, 4096 particles
PER GPU (weak scaling: N GPUs evaluate a 4096*N swarm; N=8 is config C4).  One "step" is one
swarm generation on device-resident state: velocity/position update -> batched objective
(the hot path, one launch) -> personal-best update -> local argmin -> [N>1: one RCCL
all-gather of the (D+1)-double candidate] -> global-best fold.  Inputs are resident in HBM
before the timed region; stopping tests are disabled so every timed generation does full work.

metric  = particle*gridpoint*peak evaluations per second, whole job.
roofline: SURVEY.md 8(d)(i) streaming-operand byte model (32/P bytes per unit) against
          8 TB/s, from the objective kernel's own average duration measured with HIP events on
          its stream (a second pass of K objective-only launches on the final swarm);
          `valu` carries the honest binding figure (fp64 vector-ALU issue) -- see DESIGN.md.
cpu_baseline: the oracle (numpy restatement of the reference, 1 core = the reference's
          default processes=1) on a bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X spec: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", help="C3 (default, the metric's config) or C2")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--swarm-per-gpu", type=int, default=0, help="override the workload's swarm size per GPU")
    ap.add_argument("--cpu-pool", type=int, default=0, metavar="PROCS",
                    help="also time the reference's multiprocessing mode (Pool.map of the numpy oracle over PROCS "
                         "spawned workers); off by default, never use under rocprofv3 (workers inherit its preload)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the far-field / host-pointer extras (PMC passes)")
    ap.add_argument("--other-configs", action="store_true",
                    help="also time the C2 and C5 shapes (kernel only).  Off by default so that every "
                         "objective_kernel<0,false,0> launch of the default command has the C3 shape and the "
                         "rocprofv3 --stats average of that kernel is the number in roofline.kernel_ms")
    return ap.parse_args()


def cpu_baseline(spec, lower, upper, P, budget_s, pool_procs=0):
    """Reference-plumbing baseline: the numpy oracle, one particle per call, 1 core."""
    from oracle import nmrfit_oracle as onp
    from nmrfit_amd import synth
    N = spec["w"].size
    X = synth.make_swarm(lower, upper, 4096, seed=2, x_true=spec["x_true"])
    t0 = time.perf_counter()
    n = 0
    while n < X.shape[0]:
        onp.objective(X[n], spec["w"], spec["u"], spec["v"], spec["weights"])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": n * N * P / dt, "unit": "particle*gridpoint*peak/s", "cores": 1, "kind": "port",
           "sample": "%d particles of the same workload (N=%d, P=%d), numpy oracle one call per particle, %.1f s"
                     % (n, N, P, dt)}
    # strong-CPU line: plain-C oracle, OpenMP over particles, all host cores
    try:
        from oracle import c_oracle
        th = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        th = min(th, 16)     # a one-GPU box's CPU share
        per = min(X.shape[0], 4 * th)
        c_oracle.objective_batch(X[:per], spec["w"], spec["u"], spec["v"], spec["weights"], threads=th)   # warm the team
        m, t0 = 0, time.perf_counter()
        while m + per <= X.shape[0] and time.perf_counter() - t0 < 2.0:
            c_oracle.objective_batch(X[m:m + per], spec["w"], spec["u"], spec["v"], spec["weights"], threads=th)
            m += per
        dt = time.perf_counter() - t0
        out["c_openmp"] = {"value": m * N * P / dt, "cores": th, "sample": "%d particles, %.2f s" % (m, dt)}
    except Exception as e:  # the C oracle is optional for the baseline
        out["c_openmp"] = {"error": str(e)}
    if pool_procs > 0:
        # the reference's only parallel mode (utils.py:176-182, processes=n): Pool.map over particles
        from oracle import pool_baseline
        n, dt, _ = pool_baseline.timed_map(X, spec["w"], spec["u"], spec["v"], spec["weights"], pool_procs,
                                           max(2.0, budget_s / 3))
        out["numpy_pool"] = {"value": n * N * P / dt, "cores": pool_procs, "sample": "%d particles, %.2f s" % (n, dt)}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
    from nmrfit_amd import synth, _cabi
    from nmrfit_amd.equations import Evaluator
    from nmrfit_amd.pso import DeviceSwarm, TorchExchange

    cfg = synth.CONFIGS[args.workload]
    S_local, N, P = cfg.S, cfg.N, cfg.P
    if args.swarm_per_gpu > 0:
        S_local = args.swarm_per_gpu
    D = 4 + 3 * P
    spec = synth.make_spectrum(N, P, seed=1)

    # N > 1: one rank per GPU, torch.distributed over RCCL ("nccl") for the candidate exchange.
    # Rehearsal knobs (not used by the driver): NMRFIT_BENCH_FORCE_DIST=1 takes the RCCL path
    # with a single rank; NMRFIT_BENCH_BACKEND=gloo runs several ranks on one GPU with the
    # exchange staged through the host.
    dist = torch = None
    backend = os.environ.get("NMRFIT_BENCH_BACKEND", "nccl")
    use_dist = world > 1 or os.environ.get("NMRFIT_BENCH_FORCE_DIST") == "1"
    device = local_rank
    if use_dist:
        import torch
        import torch.distributed as dist
        if backend == "gloo":
            device = local_rank % max(1, _cabi.device_count())
        torch.cuda.set_device(device)
        kw = {}
        if "MASTER_ADDR" not in os.environ:
            kw = dict(init_method="tcp://127.0.0.1:29531", rank=rank, world_size=world)
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", device)
        dist.init_process_group(backend, **kw)

    ev = Evaluator(spec["w"], spec["u"], spec["v"], spec["weights"], device=device)
    ev.set_variant(args.variant)
    sw = DeviceSwarm(ev, spec["lower"], spec["upper"], swarmsize=S_local * world, offset=rank * S_local,
                     S_local=S_local, seed=1234, minstep=-1.0, minfunc=-1.0)   # never stop while timing

    if use_dist and backend == "nccl":
        # swarm kernels and the RCCL all-gather on ONE explicit stream, no host synchronisation
        # inside a generation: the same object nmrfit_amd.fit() uses for multi-GPU fits
        from nmrfit_amd.pso import RcclGeneration
        gen = RcclGeneration(sw, TorchExchange())
        fold = gen.fold

        def sync():
            torch.cuda.synchronize()
    elif use_dist:
        ex = TorchExchange()

        def fold():
            sw.apply_global(ex.gather_host(sw.candidate()))

        def sync():
            ev.synchronize()
    else:
        cand = sw.candidate_dev()

        def fold():
            sw.apply_global_dev(cand, 1)

        def sync():
            ev.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    sw.init()
    fold()
    for _ in range(args.warmup):
        sw.step_local()
        fold()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sw.step_local()
        fold()
    sync()
    barrier()
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # objective kernel alone, HIP events on its stream, same swarm positions
    st = sw.status()
    d_x = ev.dev_alloc(S_local * D * 8)
    d_f = ev.dev_alloc(S_local * 8)
    ev.upload(d_x, sw.state()["x"])
    for _ in range(2):
        ev.objective_batch_dev(S_local, P, d_x, d_f)
    ev.synchronize()
    ev.timer_begin()
    for _ in range(args.steps):
        ev.objective_batch_dev(S_local, P, d_x, d_f)
    t_kernel_ms = ev.timer_end() / args.steps
    geom = ev.last_launch()
    # opt-in far-field variant: same inputs, objective-only launches
    farfield = None
    if rank == 0 and args.variant == 0 and not args.no_extras:
        f_def = ev.download(d_f, (S_local,))
        ev.set_variant(_cabi.VARIANT_FARFIELD)
        for _ in range(2):
            ev.objective_batch_dev(S_local, P, d_x, d_f)
        ev.synchronize()
        ev.timer_begin()
        for _ in range(args.steps):
            ev.objective_batch_dev(S_local, P, d_x, d_f)
        ff_ms = ev.timer_end() / args.steps
        f_ff = ev.download(d_f, (S_local,))
        ev.set_variant(args.variant)
        farfield = {"kernel_ms": ff_ms, "units_per_s": float(S_local) * N * P / (ff_ms * 1e-3),
                    "max_rel_diff_vs_default": float(np.max(np.abs(f_ff - f_def) / np.maximum(np.abs(f_def), 1e-6))),
                    "note": "NMRFIT_VARIANT_FARFIELD: Lorentzian tails of distant peaks through one shared Taylor "
                            "expansion per 512-point chunk (fp64, truncation <= 1e-16 per term); opt-in, not the "
                            "configuration `value` is measured on"}
    # the other single-GPU configs of BASELINE.json, kernel-only (HIP events), for reference
    others = None
    if rank == 0 and world == 1 and args.workload == "C3" and args.other_configs:
        others = {}
        for name in ("C2", "C5"):
            c = synth.CONFIGS[name]
            sp2 = synth.make_spectrum(c.N, c.P, seed=1)
            if name == "C5":      # D+1 rows of a forward-difference Jacobian, residual vectors out
                X2, _ = synth.jacobian_rows(synth.make_swarm(sp2["lower"], sp2["upper"], 2, seed=4)[1])
            else:
                X2 = synth.make_swarm(sp2["lower"], sp2["upper"], c.S, seed=2, x_true=sp2["x_true"])
            with Evaluator(sp2["w"], sp2["u"], sp2["v"], sp2["weights"], device=device) as ev2:
                B, D2 = X2.shape
                dX2 = ev2.dev_alloc(X2.nbytes)
                df2 = ev2.dev_alloc(B * 8)
                dR2 = ev2.dev_alloc(B * c.N * 8) if name == "C5" else None
                ev2.upload(dX2, X2)
                run = ((lambda: ev2.residual_batch_dev(B, c.P, dX2, dR2, df2)) if name == "C5"
                       else (lambda: ev2.objective_batch_dev(B, c.P, dX2, df2)))
                for _ in range(5):
                    run()
                ev2.synchronize()
                ev2.timer_begin()
                for _ in range(50):
                    run()
                ms2 = ev2.timer_end() / 50
                others[name] = {"shape": {"rows": B, "grid": c.N, "peaks": c.P}, "kernel_ms": ms2,
                                "units_per_s": float(B) * c.N * c.P / (ms2 * 1e-3),
                                "kind": "residual_batch (R rows written)" if name == "C5" else "objective_batch"}
                ev2.dev_free(dX2)
                ev2.dev_free(df2)
                if dR2 is not None:
                    ev2.dev_free(dR2)
    # the host-pointer entry point (X uploaded, f downloaded every call): the PCIe-inclusive
    # rate, reported beside the resident one -- never as `value`
    host_ms = None
    if rank == 0 and world == 1 and not args.no_extras:
        Xh = sw.state()["x"]
        ev.objective_batch(Xh)
        t1 = time.perf_counter()
        for _ in range(5):
            ev.objective_batch(Xh)
        host_ms = (time.perf_counter() - t1) / 5 * 1e3

    units_step = float(S_local) * world * N * P
    value = units_step * args.steps / dt
    if rank == 0:
        units_launch = float(S_local) * N * P
        bytes_launch = S_local * (4 * N * 8) + S_local * D * 8 + S_local * 8
        ach = bytes_launch / (t_kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # VALU counters of the same command under rocprofv3 (separate --pmc pass), when committed
        valu_pmc = {}
        summ = os.path.join(ROOT, "profiles", "r01", "bench_c3_pmc_summary.json")
        if args.workload == "C3" and args.variant == 0 and os.path.exists(summ):
            try:
                sm = json.load(open(summ))
                insts = sm["objective_kernel_pmc"]["SQ_INSTS_VALU"]["mean"]
                valu_pmc = {"valu_busy_frac_pmc": sm.get("valu_busy_frac"),
                            "valu_instructions_per_unit_pmc": insts * 64.0 / (4096.0 * 65536.0 * 24.0),
                            "valu_cycles_per_instruction_pmc": sm.get("valu_cycles_per_inst"),
                            "pmc_source": "profiles/r01/bench_c3_pmc_summary.json"}
            except Exception:
                valu_pmc = {}
        line = {
            "metric": "objective evals/sec (swarm x grid x peaks)",
            "value": value, "unit": "particle*gridpoint*peak/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d peaks, %d-pt grid, swarm %d per GPU (%d total), one PSO generation per step"
                                   % (cfg.name, P, N, S_local, S_local * world),
                       "peaks": P, "grid": N, "swarm_per_gpu": S_local, "swarm_total": S_local * world,
                       "exchange": "rccl all_gather of %d doubles per generation" % (D + 1) if world > 1 else "none",
                       "variant": args.variant, "generations_done": st["iteration"], "swarm_best_f": st["fg"]},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "model": "streaming-operand bytes S*(4*N*8)+S*D*8+S*8 per launch (SURVEY 8(d)(i)); "
                                  "w/u/v/weights are shared by all particles and L2-resident, so this is an "
                                  "effective rate, not physical HBM traffic",
                         "kernel": "objective_kernel", "kernel_ms": t_kernel_ms,
                         "kernel_ms_note": "HIP events around K objective_batch_dev calls; each call is the "
                                           "objective kernel plus its ~5 us finalize launch when the grid is segmented",
                         "bytes_per_launch": bytes_launch, "units_per_launch": units_launch,
                         "launch": geom},
            "valu": {"units_per_s_kernel": units_launch / (t_kernel_ms * 1e-3),
                     "fp64_lane_ops_peak_per_s": FP64_VALU_PEAK_TFLOPS * 1e12 / 2,
                     "note": "binding resource is fp64 vector-ALU issue; see DESIGN.md for the per-unit "
                             "instruction count and the measured per-instruction costs", **valu_pmc},
        }
        if farfield is not None:
            line["farfield_variant"] = farfield
        if others:
            line["other_configs"] = others
        if host_ms is not None:
            line["host_pointer_call"] = {"ms": host_ms, "units_per_s": units_launch / (host_ms * 1e-3),
                                         "note": "nmrfit_objective_batch with host X/f (H2D + kernel + D2H per call)"}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(spec, spec["lower"], spec["upper"], P, args.cpu_seconds, args.cpu_pool)
        print(json.dumps(line))
        sys.stdout.flush()
    ev.dev_free(d_x)
    ev.dev_free(d_f)
    sw.close()
    ev.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
