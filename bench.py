#!/usr/bin/env python3
"""
bench.py -- objective evaluations per second of the MI355X swarm generation.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launches its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Either form runs one process per GPU.  With WORLD_SIZE unset and --gpus N > 1 this script is the
launcher: before touching any GPU it starts N rank processes of itself (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), relays rank 0's JSON line and exits non-zero if any
rank fails.  No PyTorch in any of it: the ranks find each other over standard-library sockets
(nmrfit_amd/rendezvous.py) and exchange through RCCL inside libnmrfit_amd.so.

Workload (BASELINE.json configs[2] / configs[3]): 24 peaks, 65536-point grid, 4096 particles
PER GPU (weak scaling: N GPUs evaluate a 4096*N swarm; N=8 is config C4).  One "step" is one
swarm generation on device-resident state, ONE C call (nmrfit_pso_step): velocity/position
update -> batched objective (the hot path, one launch) -> personal-best update -> local argmin
-> [N>1: one ncclAllGather of the (D+1)-double candidate] -> global-best fold.  Inputs are
resident in HBM before the timed region; stopping tests are disabled so every timed generation
does full work.

Order of a run: CPU baseline legs first (rank 0, N=1 only; nothing has touched the GPU yet, so
the Pool.map workers are not children of a GPU process) -> swarm init + W warm-up generations
-> >= 0.5 s of objective launches to bring the clocks to their loaded state -> barrier ->
EXACTLY K timed generations, each objective kernel bracketed by HIP events on its own stream
and each step by a mark (nmrfit_prof_*) -> barrier -> max over ranks.

metric  = particle*gridpoint*peak evaluations per second, whole job.
roofline: SURVEY.md 8(d)(i) streaming-operand byte model (32/P bytes per unit) against 8 TB/s,
          from the objective kernel's own durations INSIDE the timed loop (kernel <= step is
          asserted).  `roofline_valu` is the binding resource (fp64 vector-ALU issue).
          Fields that come from committed rocprofv3 --pmc passes rather than from this run are
          marked "from_committed_profile".
cpu_baseline: the oracle (numpy restatement of the reference, 1 core = the reference's
          default processes=1), the same through multiprocessing.Pool.map (the reference's
          only parallel mode, utils.py:182) and the plain-C OpenMP oracle, on a bounded sample
          of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SIMDS = 1024                   # 256 CUs x 4 SIMDs
PEAK_CLOCK_MHZ = 2400.0        # MI355X_MICROARCH.md
FP64_ISSUE_CYCLES = 4.0        # one wave64 fp64 VALU instruction occupies a SIMD's issue port for 4 cycles
PMC_SUMMARY = os.path.join("profiles", "r02", "bench_c3_pmc_summary.json")
PMC_SUMMARY_FALLBACK = os.path.join("profiles", "r01", "bench_c3_pmc_summary.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", help="C3 (default, the metric's config) or C2")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--swarm-per-gpu", type=int, default=0, help="override the workload's swarm size per GPU")
    ap.add_argument("--cpu-pool", type=int, default=-1, metavar="PROCS",
                    help="workers of the Pool.map CPU line (the reference's multiprocessing mode); default "
                         "min(16, usable cores); 0 disables.  Skipped automatically under rocprofv3 (workers "
                         "would inherit its preload)")
    ap.add_argument("--preheat-seconds", type=float, default=0.5)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the every-unit variants / far-field / host-pointer extras (PMC passes)")
    ap.add_argument("--other-configs", action="store_true",
                    help="also time the C2 and C5 shapes (kernel only).  Off by default so that every "
                         "objective_kernel<0,false,0> launch of the default command has the C3 shape and the "
                         "rocprofv3 --stats average of that kernel is the number in roofline.kernel_ms")
    return ap.parse_args()


# ---- launcher (N > 1 without a launcher's environment) -------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """Start n rank processes of this script, one per GPU.  The parent never touches the GPU (it
    loads neither the library nor HIP): it only waits, relays rank 0's output, and makes sure a
    failed rank takes the others down instead of leaving them blocked in a collective."""
    port = _free_port()
    token = "bench%d_%d" % (os.getpid(), int(time.time() * 1e3) & 0xFFFFFFF)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NMRFIT_RDZV_TOKEN=token)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    rc = 0
    out0 = b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0:
                    try:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                elif p.poll() is None:
                    continue
                pending.discard(r)
                if p.returncode != 0:
                    rc = rc or p.returncode or 1
                    sys.stderr.write("bench.py: rank %d exited with code %s\n" % (r, p.returncode))
            if rc:
                break
            time.sleep(0.05)
    finally:
        for p in procs:          # a failed run: end exactly the processes started here
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    return rc


# ---- CPU baseline legs (before any GPU call) -------------------------------------------------------
def under_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre.lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))     # a one-GPU box's CPU share


def cpu_baseline(spec, P, budget_s, pool_procs):
    """Reference-plumbing baseline: the numpy oracle, one particle per call, 1 core; then the
    reference's parallel mode (Pool.map over particles) and the plain-C OpenMP oracle."""
    from oracle import nmrfit_oracle as onp
    from nmrfit_amd import synth
    N = spec["w"].size
    X = synth.make_swarm(spec["lower"], spec["upper"], 4096, seed=2, x_true=spec["x_true"])
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
    if pool_procs > 0:
        # the reference's only parallel mode (utils.py:176-182, processes=n): Pool.map over particles
        try:
            from oracle import pool_baseline
            m, dtp, _ = pool_baseline.timed_map(X, spec["w"], spec["u"], spec["v"], spec["weights"], pool_procs,
                                                max(2.0, budget_s / 2))
            out["numpy_pool"] = {"value": m * N * P / dtp, "cores": pool_procs,
                                 "sample": "%d particles through multiprocessing.Pool(%d).map, %.2f s"
                                           % (m, pool_procs, dtp)}
        except Exception as e:
            out["numpy_pool"] = {"error": repr(e)}
    else:
        out["numpy_pool"] = {"skipped": "profiler preload detected" if under_profiler() else "disabled"}
    # strong-CPU line: plain-C oracle, OpenMP over particles, all host cores
    try:
        from oracle import c_oracle
        th = usable_cores()
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
    return out


def stats(a):
    a = np.asarray(a, dtype=np.float64)
    if a.size == 0:
        return None
    return {"min": float(a.min()), "median": float(np.median(a)), "mean": float(a.mean()), "max": float(a.max()),
            "n": int(a.size)}


def time_objective(ev, S, P, d_x, d_f, reps, heat_s=0.25):
    """Kernel-only durations (HIP events around each objective kernel, nmrfit_prof_*) of `reps`
    objective launches, after `heat_s` seconds of the same launches: every variant is timed in
    the loaded clock state, whatever the host did just before."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < heat_s:
        for _ in range(4):
            ev.objective_batch_dev(S, P, d_x, d_f)
        ev.synchronize()
    ev.prof_enable(reps)
    for _ in range(reps):
        ev.objective_batch_dev(S, P, d_x, d_f)
    k, _, _ = ev.prof_read()
    ev.prof_enable(0)
    return float(np.mean(k))


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))

    from nmrfit_amd import synth
    cfg = synth.CONFIGS[args.workload]
    S_local, N, P = cfg.S, cfg.N, cfg.P
    if args.swarm_per_gpu > 0:
        S_local = args.swarm_per_gpu
    D = 4 + 3 * P
    spec = synth.make_spectrum(N, P, seed=1)

    # ---- CPU legs first: nothing below has loaded the HIP library or touched the GPU yet ------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        pool = args.cpu_pool
        if pool < 0:
            pool = 0 if under_profiler() else usable_cores()
        cpu = cpu_baseline(spec, P, args.cpu_seconds, pool)

    from nmrfit_amd import _cabi, pso
    from nmrfit_amd.equations import Evaluator

    # N > 1: one rank per GPU, RCCL through the C-ABI.  Rehearsal knobs (not used by the
    # driver): NMRFIT_BENCH_FORCE_DIST=1 takes the RCCL path with a single rank;
    # NMRFIT_BENCH_BACKEND=host (alias: gloo) runs several ranks on ONE GPU with the record staged
    # through the host over sockets (RCCL refuses two ranks on one device).
    backend = os.environ.get("NMRFIT_BENCH_BACKEND", "rccl").lower()
    backend = {"nccl": "rccl", "gloo": "host"}.get(backend, backend)
    use_dist = world > 1 or os.environ.get("NMRFIT_BENCH_FORCE_DIST") == "1"
    device = local_rank
    if use_dist and (backend == "host" or os.environ.get("NMRFIT_BENCH_SHARE_GPU") == "1"):
        device = local_rank % max(1, _cabi.device_count())      # rehearsals: several ranks on one card

    ev = Evaluator(spec["w"], spec["u"], spec["v"], spec["weights"], device=device)
    ev.set_variant(args.variant)
    sw = pso.DeviceSwarm(ev, spec["lower"], spec["upper"], swarmsize=S_local * world, offset=rank * S_local,
                         S_local=S_local, seed=1234, minstep=-1.0, minfunc=-1.0)   # never stop while timing
    ex = None
    exchange_desc = "none"
    rccl_failure = None
    channel = None
    if use_dist:
        from nmrfit_amd import rendezvous
        channel = rendezvous.Channel()        # the star of sockets the ranks bootstrap over
    if use_dist and backend == "rccl":
        try:
            ex = pso.RcclExchange(ev, channel=channel)
        except _cabi.NmrfitError as e:
            # RCCL unavailable on EVERY rank alike (library missing, no unique id): say so loudly and
            # still measure, with the candidate record staged through the host.  (A failure inside
            # ncclCommInitRank on some ranks only cannot be recovered from: the run then fails.)
            rccl_failure = str(e)
            sys.stderr.write("bench.py rank %d: RCCL exchange unavailable (%s)\n" % (rank, e))
        # every rank takes the same path: RCCL only if every rank has a communicator
        oks = channel.all_gather(b"\x01" if ex is not None else b"\x00")
        if not all(o == b"\x01" for o in oks):
            if ex is not None:
                ex.close()
                ex = None
            rccl_failure = rccl_failure or "RCCL failed on rank(s) %s" % [i for i, o in enumerate(oks) if o != b"\x01"]
            backend = "host"
    if use_dist and backend == "rccl":
        sw.set_comm(ex)                       # the all-gather now happens inside nmrfit_pso_step
        exchange_desc = "ncclAllGather of %d doubles per generation inside nmrfit_pso_step (RCCL %s)" % (
            D + 1, ex.info()["rccl_version"])

        def step():
            sw.step()
    elif use_dist:
        ex = pso.SocketExchange(channel=channel)
        exchange_desc = "host-staged all-gather of %d doubles per generation (sockets%s)" % (
            D + 1, "; RCCL FAILED: " + rccl_failure if rccl_failure else "; rehearsal")
        state = {"first": True}

        def step():
            if not state["first"]:
                sw.step_local()
            state["first"] = False
            sw.apply_global(ex.gather_host(sw.candidate()))
    else:
        def step():
            sw.step()

    def barrier():
        ev.synchronize()
        if ex is not None:
            ex.barrier()

    sw.init()
    step()                                   # folds generation 0
    for _ in range(args.warmup):
        step()
    ev.synchronize()
    # bring the clocks to their loaded state: >= preheat_seconds of the same kernel on the same
    # positions (objective-only launches: the swarm does not advance, so the trajectory -- and
    # generations_done -- do not depend on how long this takes)
    d_x = ev.dev_alloc(S_local * D * 8)
    d_f = ev.dev_alloc(S_local * 8)
    ev.upload(d_x, sw.state()["x"])
    t0 = time.perf_counter()
    heat_launches = 0
    while time.perf_counter() - t0 < args.preheat_seconds:
        for _ in range(8):
            ev.objective_batch_dev(S_local, P, d_x, d_f)
        ev.synchronize()
        heat_launches += 8
    ev.prof_enable(args.steps)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.prof_mark()
        step()
    ev.prof_mark()
    barrier()
    dt = time.perf_counter() - t0
    k_ms, s_ms, clock_mhz = ev.prof_read()
    ev.prof_enable(0)
    ranks = None
    if ex is not None:
        dt_local = dt
        dt = float(ex.all_reduce([dt], "max")[0])
        km = float(np.mean(k_ms)) if len(k_ms) else 0.0
        lo = ex.all_reduce([km, dt_local, clock_mhz], "min")
        hi = ex.all_reduce([km, dt_local, clock_mhz], "max")
        ranks = {"kernel_ms_mean": {"min": float(lo[0]), "max": float(hi[0])},
                 "timed_region_s": {"min": float(lo[1]), "max": float(hi[1])},
                 "clock_mhz_in_run": {"min": float(lo[2]), "max": float(hi[2])},
                 "note": "spread over the ranks (every rank evaluates the same amount of work; `value` uses the slowest)"}
    ms_per_step = dt / args.steps * 1e3
    st = sw.status()
    geom = ev.last_launch()
    ev.upload(d_x, sw.state()["x"])

    # ---- extras on the final swarm positions (rank 0, after the timed region) ---------------
    variants = farfield = host_ms = others = None
    if rank == 0 and args.variant == 0 and not args.no_extras:
        f_def = None
        variants = {}
        reps = max(5, min(args.steps, 20))
        for name, vid in (("default", _cabi.VARIANT_DEFAULT), ("noskip", _cabi.VARIANT_NOSKIP),
                          ("baseline", _cabi.VARIANT_BASELINE), ("farfield", _cabi.VARIANT_FARFIELD)):
            ev.set_variant(vid)
            ms = time_objective(ev, S_local, P, d_x, d_f, reps)
            f = ev.download(d_f, (S_local,))
            if f_def is None:
                f_def = f
            variants[name + "_ms"] = ms
            variants[name + "_max_rel_diff_vs_default"] = float(np.max(np.abs(f - f_def) / np.maximum(np.abs(f_def), 1e-6)))
        ev.set_variant(args.variant)
        variants["note"] = ("objective kernel alone on the final swarm positions (mean of %d HIP-event pairs after "
                            "0.25 s of the same launches).  noskip / baseline evaluate every (particle, point, peak) "
                            "unit -- DEFAULT skips out-of-window Gaussians (exact to fp64 rounding); baseline is "
                            "IEEE divide + libdevice exp2 per unit; farfield is opt-in and never the configuration "
                            "`value` is measured on" % reps)
        farfield = {"kernel_ms": variants["farfield_ms"],
                    "units_per_s": float(S_local) * N * P / (variants["farfield_ms"] * 1e-3),
                    "max_rel_diff_vs_default": variants["farfield_max_rel_diff_vs_default"]}
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
                B = X2.shape[0]
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
    if rank == 0 and world == 1 and not args.no_extras:
        Xh = sw.state()["x"]
        ev.objective_batch(Xh)
        t1 = time.perf_counter()
        for _ in range(5):
            ev.objective_batch(Xh)
        host_ms = (time.perf_counter() - t1) / 5 * 1e3

    units_step = float(S_local) * world * N * P
    value = units_step * args.steps / dt
    rc = 0
    if rank == 0:
        units_launch = float(S_local) * N * P
        bytes_launch = S_local * (4 * N * 8) + S_local * D * 8 + S_local * 8
        kst, sst = stats(k_ms), stats(s_ms)
        t_kernel_ms = kst["mean"] if kst else float("nan")
        ach = bytes_launch / (t_kernel_ms * 1e-3) / 1e9
        # physical HBM traffic and VALU counters: separate rocprofv3 --pmc passes of this command,
        # committed under profiles/ -- NOT measured by this run
        traffic = None
        pmcf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmcf):
            try:
                traffic = json.load(open(pmcf)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        valu = {"bound": "fp64_valu_issue", "unit": "fraction of fp64 VALU issue slots (1024 SIMDs x 2.4 GHz / 4 cycles "
                                                     "per wave64 instruction)"}
        summ = next((p for p in (PMC_SUMMARY, PMC_SUMMARY_FALLBACK) if os.path.exists(os.path.join(ROOT, p))), None)
        if args.workload == "C3" and args.variant == 0 and summ:
            try:
                sm = json.load(open(os.path.join(ROOT, summ)))
                insts = sm["objective_kernel_pmc"]["SQ_INSTS_VALU"]["mean"]      # wave-instructions per launch
                ipu = insts * 64.0 / (4096.0 * 65536.0 * 24.0)
                peak = SIMDS * PEAK_CLOCK_MHZ * 1e6 / FP64_ISSUE_CYCLES          # wave-instructions / s
                achieved = ipu * units_launch / 64.0 / (t_kernel_ms * 1e-3)
                valu.update({"achieved": achieved, "peak": peak, "frac": achieved / peak,
                             "achieved_unit": "fp64 wave64 VALU instructions/s",
                             "valu_instructions_per_unit": ipu, "valu_busy_frac": sm.get("valu_busy_frac"),
                             "valu_cycles_per_instruction": sm.get("valu_cycles_per_inst"),
                             "from_committed_profile": True, "pmc_source": summ,
                             "note": "instructions per unit, busy fraction and cycles per instruction come from the "
                                     "committed SQ counter pass; the kernel time and clock are this run's"})
            except Exception as e:
                valu["error"] = repr(e)
        if clock_mhz > 0:
            valu["clock_mhz_in_run"] = clock_mhz
            valu["clock_note"] = ("s_memtime / s_memrealtime ticks of the first workgroup of the last timed objective "
                                  "launch (shader clock while the chip is loaded)")
        line = {
            "metric": "objective evals/sec (swarm x grid x peaks)",
            "value": value, "unit": "particle*gridpoint*peak/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d peaks, %d-pt grid, swarm %d per GPU (%d total), one PSO generation per step"
                                   % (cfg.name, P, N, S_local, S_local * world),
                       "peaks": P, "grid": N, "swarm_per_gpu": S_local, "swarm_total": S_local * world,
                       "exchange": exchange_desc, "variant": args.variant, "generations_done": st["iteration"],
                       "swarm_best_f": st["fg"], "preheat_launches": heat_launches},
            "step_ms": sst, "kernel_ms": kst,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_from_committed_profile": traffic is not None,
                         "model": "streaming-operand bytes S*(4*N*8)+S*D*8+S*8 per launch (SURVEY 8(d)(i)); "
                                  "w/u/v/weights are shared by all particles and L2-resident, so this is an "
                                  "effective rate, not physical HBM traffic (`traffic`); the binding resource "
                                  "is fp64 VALU issue, see roofline_valu",
                         "kernel": "objective_kernel", "kernel_ms": t_kernel_ms,
                         "kernel_ms_note": "mean of the HIP-event durations of the objective kernel alone, one pair "
                                           "per timed step, on the stream it is launched on",
                         "bytes_per_launch": bytes_launch, "units_per_launch": units_launch,
                         "launch": geom},
            "roofline_valu": valu,
        }
        if kst and not (t_kernel_ms <= ms_per_step * 1.0005):
            line["error"] = "objective kernel (%.4f ms) longer than the step that contains it (%.4f ms)" % (
                t_kernel_ms, ms_per_step)
            rc = 3
        if ranks is not None:
            line["ranks"] = ranks
        if variants is not None:
            line["variants"] = variants
            line["farfield_variant"] = farfield
        if others:
            line["other_configs"] = others
        if host_ms is not None:
            line["host_pointer_call"] = {"ms": host_ms, "units_per_s": units_launch / (host_ms * 1e-3),
                                         "note": "nmrfit_objective_batch with host X/f (H2D + kernel + D2H per call)"}
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
        sys.stdout.flush()
    ev.dev_free(d_x)
    ev.dev_free(d_f)
    if ex is not None:
        ex.barrier()
        if isinstance(ex, pso.RcclExchange):
            sw.set_comm(None)
        ex.close()
    if channel is not None:
        channel.close()
    sw.close()
    ev.close()
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    main()
